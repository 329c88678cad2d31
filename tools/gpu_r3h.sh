cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3h_tests.log 2>&1; echo "tests rc=$?" >> $O/r3h_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3h_tests.log | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh
