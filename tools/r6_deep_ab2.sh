cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
bash tools/gpu_run.sh ab ICL_QCHAIN 0 1 2>&1 | tee $O/qchain_ab_deep.txt
for w in 112 144 176; do
  bash tools/gpu_run.sh ab ICL_UPDATE_WGS 128 $w 2>&1 | tee -a $O/wgs_ab.txt
done
bash tools/gpu_run.sh ab ICL_UPDATE_PLACEMENT tail deep --num-classes 16 2>&1 | tee $O/deep_ab_nc16.txt
CP_ALIGNER_DETAIL=1 TAIL=60 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_detail.txt
