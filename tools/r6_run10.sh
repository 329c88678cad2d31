cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
bash tools/gpu_run.sh ab ICL_DEEP_EXTRA 0 1 2>&1 | tee $O/extra_ab.txt
bash tools/gpu_run.sh ab ICL_DEEP_EXTRA 0 2 2>&1 | tee -a $O/extra_ab.txt
ICL_UPDATE_WGS=160 bash tools/gpu_run.sh ab ICL_DEEP_EXTRA 0 1 2>&1 | tee -a $O/extra_ab.txt
