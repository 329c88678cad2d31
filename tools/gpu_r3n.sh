R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -q -x -m gpu > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
for i in 1 2; do python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --no-kernel-timer --steps 10 2>&1 | tail -1 | cut -c140-170; done
