cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 tools/conv_ab.py --rounds 5 --shapes "16,16,96,wgrad;48,16,96,wgrad;32,16,48,wgrad" --var ICL_WGRAD_TR_PC=0 --var ICL_WGRAD_TR_PC=1 --var ICL_WGRAD_TR_PC=2 > $O/r3k_ab_pc.log 2>&1
cat $O/r3k_ab_pc.log
