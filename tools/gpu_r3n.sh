R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -q -x -m gpu > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -5
python3 bench.py 2>/dev/null | tail -1 | cut -c1-250
