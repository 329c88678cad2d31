R=$GRAFT_REPO_ROOT
cd $R
for lib in gpurun_in/libicl_base.so icl_amd/libicl_hip.so gpurun_in/libicl_base.so icl_amd/libicl_hip.so gpurun_in/libicl_base.so icl_amd/libicl_hip.so; do
  echo "== $lib"
  for shape in "16 16 96" "48 16 96" "32 32 48" "48 48 96"; do
    ICL_HIP_LIB=$R/$lib python3 tools/conv_one.py $shape fwd 10 2 2>&1 | tail -1
  done
done
