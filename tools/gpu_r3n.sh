R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
for V in 1 0 1 0; do echo CIN1=$V; ICL_CONV_CIN1=$V python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
