cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3d_tests.log 2>&1; echo "tests rc=$?" >> $O/r3d_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3d_tests.log | tail -20
python3 tests/diag/three_steps.py > $O/r3d_three.log 2>&1; grep -v Warning $O/r3d_three.log | tail -24
python3 tools/conv_ab.py --rounds 5 --shapes "16,16,96,fwd;48,16,96,fwd;32,32,48,fwd;96,32,48,fwd;16,48,96,fwd" --var ICL_CONV_SPLIT_V=0 --var ICL_CONV_SPLIT_V=4 --var ICL_CONV_SPLIT_V=8 --var ICL_CONV_SPLIT_V=12 > $O/r3d_ab_fwd.log 2>&1
cat $O/r3d_ab_fwd.log
python3 tools/conv_ab.py --rounds 5 --shapes "16,16,96,wgrad;48,16,96,wgrad;32,32,48,wgrad;96,32,48,wgrad;64,64,24,wgrad" --var ICL_WGRAD_TR=0 --var ICL_WGRAD_TR=2 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=1 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=3 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=5 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=6 > $O/r3d_ab_wgrad.log 2>&1
cat $O/r3d_ab_wgrad.log
