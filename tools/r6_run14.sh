cd $GRAFT_REPO_ROOT
O=gpurun_out/r6o; mkdir -p $O
bash tools/gpu_run.sh ab ICL_MAP_CHAINS_LAST 0 1 2>&1 | tee $O/map_chains_ab.txt
CP_ALIGNER_DETAIL=1 TAIL=60 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path.txt | tail -36
