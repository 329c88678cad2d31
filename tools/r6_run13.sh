cd $GRAFT_REPO_ROOT
O=gpurun_out/r6n; mkdir -p $O
CP_ALIGNER_DETAIL=1 TAIL=60 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path.txt | tail -48
bash tools/gpu_run.sh timeline 2>&1 | tail -3
cp gpurun_out/step_trace.txt gpurun_out/queue_tails.txt $O/
