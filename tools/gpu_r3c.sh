cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3c_tests.log 2>&1; echo "tests rc=$?" >> $O/r3c_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3c_tests.log | tail -30
python3 tests/diag/mlp2_grad_sensitivity.py > $O/r3c_sens.log 2>&1; tail -5 $O/r3c_sens.log
python3 tools/conv_ab.py --rounds 4 --shapes "16,16,96,wgrad;96,32,48,wgrad" --var ICL_WGRAD_TR=2 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=1 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=2 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=3 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=4 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=5 --var ICL_WGRAD_TR=2,ICL_WGRAD_TR_DBG=6 > $O/r3c_ab_dbg.log 2>&1
cat $O/r3c_ab_dbg.log
