R=$GRAFT_REPO_ROOT
cd $R
for lib in icl_amd/libicl_hip.so gpurun_in/libicl_nt1.so gpurun_in/libicl_nt2.so gpurun_in/libicl_nt3.so icl_amd/libicl_hip.so; do
  echo "== $lib (ICL_NT_POLICY: 1 plain loads, 2 plain stores, 3 both)"
  ICL_HIP_LIB=$R/$lib python3 bench.py --no-cpu-baseline --no-exact-compare 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], 'dgrad_sgd', r['mlp2_dgrad_sgd_one_pass']['avg_launch_us'], r['mlp2_dgrad_sgd_one_pass']['frac'], 'stream fwd', r['mlp2_weight_stream']['linear_stream_fwd']['avg_launch_us'])"
done
