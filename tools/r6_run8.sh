cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
for w in 96 112; do
  bash tools/gpu_run.sh ab ICL_UPDATE_WGS 128 $w 2>&1 | tee -a $O/wgs_prefetch_ab.txt
done
python tests/diag/dense_wgrad_errors.py 2>&1 | grep -v amdgpu > $O/dense_errors.txt; cat $O/dense_errors.txt
python -m pytest tests -m gpu -q -k "compat_root or update_inside_backward or bit_reproducible" 2>&1 | tail -4 | tee $O/tests.txt
bash tools/gpu_run.sh timeline 2>&1 | tail -5
cp gpurun_out/step_trace.txt gpurun_out/timeline.txt gpurun_out/queue_tails.txt $O/
