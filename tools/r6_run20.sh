cd $GRAFT_REPO_ROOT
O=gpurun_out/r6u; mkdir -p $O
for f6 in 0 1 0 1; do echo "== FLAT6=$f6"; ICL_CONV_SPLIT_FLAT6=$f6 CP_ALIGNER_DETAIL=0 TAIL=40 bash tools/gpu_run.sh critical-path 2>&1 | grep -E "replayed|encoder done|up3 done|joined|backward starts|deep backward starts|pool4 ready|c3 ready|pool1 ready|backward done|step end|update stream"; done | tee $O/cp_f6.txt
ICL_CONV_SPLIT_FLAT6=1 bash tools/gpu_run.sh stats r6u_f6 2>&1 | tail -5
