cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
SH="64,128,12,fwd;128,128,12,fwd;384,128,12,fwd;128,64,12,fwd;128,384,12,fwd;32,64,24,fwd;64,64,24,fwd;192,64,24,fwd;64,192,24,fwd;64,32,24,fwd"
for ks in 0 -1 2 4 8; do
  echo "== ICL_CONV_SPLIT_KSPLIT=$ks"; ICL_CONV_SPLIT_KSPLIT=$ks python tools/conv_time.py --shapes "$SH" 2>&1 | grep -v "^#"
done | tee $O/conv_ksplit.txt
echo "== planned for 128 CUs"; ICL_CONV_SPLIT_KSPLIT_CUS=128 python tools/conv_time.py --shapes "$SH" 2>&1 | grep -v "^#" | tee -a $O/conv_ksplit.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "conv3d or split" 2>&1 | tail -3 | tee $O/conv_tests.txt
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_KSPLIT 0 -1 2>&1 | tee $O/ksplit_ab.txt
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_KSPLIT_CUS 256 128 2>&1 | tee $O/ksplit_cus_ab.txt
