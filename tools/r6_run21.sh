cd $GRAFT_REPO_ROOT
O=gpurun_out/r6v; mkdir -p $O
for f6 in 0 1 0 1; do echo "== FLAT6=$f6"; ICL_CONV_SPLIT_FLAT6=$f6 CP_ALIGNER_DETAIL=0 TAIL=40 bash tools/gpu_run.sh critical-path 2>&1 | grep -E "replayed|encoder done|up3 done|deep backward starts|pool4 ready|pool1 ready|backward done|step end"; done | tee $O/cp_f6.txt
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_FLAT6 0 1 2>&1 | tee $O/f6_ab.txt
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -5) 2>&1 | tee $O/suite.txt
