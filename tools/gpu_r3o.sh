# forward split kernel ablations (compile-time BF3_DEBUG bits: 1 no MFMA, 2 no split/LDS store of x, 4 no output store, 8 no fragment reads (V = 24 only))
R=$GRAFT_REPO_ROOT
cd $R
for lib in icl_amd/libicl_hip.so gpurun_in/libicl_dbg6.so gpurun_in/libicl_dbg14.so; do
  for V in 8 24; do
  echo "== lib '$lib' V=$V"
  for shape in "16 16 96" "48 16 96" "32 32 48"; do
    ICL_CONV_SPLIT_V=$V ICL_HIP_LIB=$R/$lib python3 tools/conv_one.py $shape fwd 5 2 2>&1 | tail -1
  done
  done
done
