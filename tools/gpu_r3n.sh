R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py -q -x -k "first_convolution or side_stream" > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error\|^E ' $O/r3n_tests.log | head -20
