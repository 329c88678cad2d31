R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
