cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 20 > $O/bias1.out 2> $O/bias1.err; echo "rc $?"; tail -c 600 $O/bias1.out; grep -v "^  File\|^Extension" $O/bias1.err | tail -15
ICL_BIAS_REDUCE_EARLY=0 CP_ALIGNER_DETAIL=0 TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_early.txt | tail -14
ICL_WGRAD_REDUCE_EARLY=0 CP_ALIGNER_DETAIL=0 TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_off.txt | tail -14
