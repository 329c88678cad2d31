R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -q -x -m gpu > $O/r3n_tests.log 2>&1
grep -n 'passed\|failed\|Error\|error' $O/r3n_tests.log | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
