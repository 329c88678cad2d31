R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
for V in 60 24 60 24; do echo V=$V; ICL_CONV_SPLIT_V=$V python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
for V in 60 24; do echo swin V=$V; ICL_CONV_SPLIT_V=$V python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --no-kernel-timer --steps 10 2>&1 | tail -1 | cut -c140-170; done
