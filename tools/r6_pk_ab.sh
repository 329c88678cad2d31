set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
for rep in 1 2 3; do
  python tools/conv_time.py > $O/conv_default_$rep.txt 2>&1
  ICL_HIP_LIB=$PWD/icl_amd/libicl_hip_noslp.so python tools/conv_time.py > $O/conv_noslp_$rep.txt 2>&1
done
tail -n 14 $O/conv_default_3.txt $O/conv_noslp_3.txt
bash tools/gpu_run.sh ab ICL_HIP_LIB $PWD/icl_amd/libicl_hip.so $PWD/icl_amd/libicl_hip_noslp.so 2>&1 | tee $O/step_ab.txt
for lib in default noslp; do
  if [ $lib = noslp ]; then E="ICL_HIP_LIB=$PWD/icl_amd/libicl_hip_noslp.so"; else E="ICL_X=1"; fi
  bash tools/pmc_conv.sh r6_${lib}_f16 $E -- 16 16 96 fwd 5 2
  bash tools/pmc_conv.sh r6_${lib}_w16 $E -- 16 16 96 wgrad 5 2
  bash tools/pmc_conv.sh r6_${lib}_w32 $E -- 32 32 48 wgrad 5 2
done
python3 tools/pmc_summary.py gpurun_out/pmc_r6_*_1 gpurun_out/pmc_r6_*_2 > $O/pmc_pk_ab.txt 2>&1
cut -c1-300 $O/pmc_pk_ab.txt | head -40
TAIL=8 bash tools/gpu_run.sh suite 2>&1 | tee $O/suite.txt
