cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -3 | tee $O/suite.txt
bash tools/gpu_run.sh ab ICL_LOSS_MULTI 0 1 2>&1 | tee $O/loss_multi_ab.txt
bash tools/gpu_run.sh stats r6l_unet 2>&1 | tee $O/stats.txt | head -4
