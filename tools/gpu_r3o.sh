R=$GRAFT_REPO_ROOT
cd $R
ICL_HIP_LIB=$R/gpurun_in/libicl_wgtr.so python3 tools/wgtr_stamps.py 16 16 96 | head -8
for lib in gpurun_in/libicl_base.so icl_amd/libicl_hip.so gpurun_in/libicl_base.so icl_amd/libicl_hip.so; do
  echo "== $lib"
  for shape in "16 16 96" "48 16 96" "32 32 48" "96 32 48" "64 64 24" "48 48 96"; do
    ICL_HIP_LIB=$R/$lib python3 tools/conv_one.py $shape wgrad 10 2 2>&1 | tail -1
  done
done
