cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3g_tests.log 2>&1; echo "tests rc=$?" >> $O/r3g_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3g_tests.log | tail -8
python3 tools/conv_ab.py --rounds 5 --shapes "16,16,96,wgrad;48,16,96,wgrad;32,32,48,wgrad;96,32,48,wgrad;16,32,48,wgrad;64,64,24,wgrad;192,64,24,wgrad" --var ICL_WGRAD_TR_MERGE=0 --var ICL_WGRAD_TR_MERGE=1 > $O/r3g_ab_wgrad.log 2>&1
cat $O/r3g_ab_wgrad.log
