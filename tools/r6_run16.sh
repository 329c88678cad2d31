cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
CP_ALIGNER_DETAIL=1 TAIL=70 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path.txt | tail -70
bash tools/gpu_run.sh timeline > /dev/null 2>&1
cp gpurun_out/step_trace.txt gpurun_out/timeline.txt gpurun_out/queue_tails.txt $O/
