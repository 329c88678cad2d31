cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 tools/conv_ab.py --rounds 4 --reps 5 --shapes "48,48,96,wgrad;96,48,96,wgrad;16,48,96,wgrad;48,96,48,wgrad;192,96,48,wgrad" --var ICL_WGRAD_TR_NCB3=0 --var ICL_WGRAD_TR_NCB3=1 --var ICL_WGRAD_TR=0 > $O/r3i_ab_ncb3.log 2>&1
cat $O/r3i_ab_ncb3.log
python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare > $O/r3i_swin.json 2>/dev/null; cut -c1-200 $O/r3i_swin.json
