cd $GRAFT_REPO_ROOT
O=gpurun_out/r6af; mkdir -p $O
bash tools/gpu_run.sh stats r6af 2>&1 | head -3
grep -E "reduce_unpack|colsum_multi|splitk_reduce" gpurun_out/r6af_kernel_stats.csv | cut -c1-60,100-200 | tee $O/reduce_stats.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "fixed_order or schedule or conv3d" 2>&1 | tail -2
