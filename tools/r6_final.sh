# Round-6 final collection on the GPU box: bench lines + rocprofv3 stats (tools/collect_profiles.sh), PMC of the convolution kernels, in-graph
# stamps of the replayed step, the reference loop, the data-parallel rehearsals and projections
cd $GRAFT_REPO_ROOT
export ROUND=r6
bash tools/collect_profiles.sh > gpurun_out/r6_collect.log 2>&1; tail -8 gpurun_out/r6_collect.log
bash tools/gpu_run.sh pmc > gpurun_out/r6_pmc.log 2>&1; cp gpurun_out/pmc_conv_raw.txt gpurun_out/r6_pmc_conv_raw.txt
CP_ALIGNER_DETAIL=1 TAIL=60 bash tools/gpu_run.sh critical-path > gpurun_out/r6_critical_path_fused.txt 2>&1; tail -45 gpurun_out/r6_critical_path_fused.txt
python bench.py --loop reference --no-cpu-baseline --no-exact-compare --no-kernel-timer --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r6_reference_loop_bench.json; cut -c1-200 gpurun_out/r6_reference_loop_bench.json
bash tools/ddp_rehearsal.sh 2>&1 | tail -30
