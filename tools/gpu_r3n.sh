R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
for a in "" "--num-classes 16" ; do python3 bench.py $a --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --no-kernel-timer --steps 10 2>&1 | tail -1 | cut -c140-170
ICL_CONV_SPLIT_MIN=110592 python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --no-kernel-timer --steps 10 2>&1 | tail -1 | cut -c140-170
