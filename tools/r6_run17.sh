cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r; mkdir -p $O
ICL_BIAS_REDUCE_EARLY=0 bash tools/gpu_run.sh bench 2>&1 | tail -2 | tee $O/first.txt
bash tools/gpu_run.sh bench 2>&1 | tail -2 | tee -a $O/first.txt
bash tools/gpu_run.sh ab ICL_WGRAD_REDUCE_EARLY 0 1 2>&1 | tee $O/reduce_early_ab.txt
bash tools/gpu_run.sh ab ICL_OPT_BRANCHES 1 4 2>&1 | tee $O/opt_branches_ab.txt
bash tools/gpu_run.sh ab ICL_BIAS_REDUCE_EARLY 0 1 2>&1 | tee $O/bias_early_ab.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x 2>&1 | tail -4 | tee $O/tests.txt
