# GPU box: the fused query chain (ICL_QCHAIN) against the operator-by-operator aligner — suite, step A/B, stamps of the replayed step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
TAIL=12 bash tools/gpu_run.sh suite 2>&1 | tee $O/suite.txt
bash tools/gpu_run.sh ab ICL_QCHAIN 0 1 2>&1 | tee $O/qchain_ab.txt
bash tools/gpu_run.sh ab ICL_QCHAIN 0 1 --num-classes 16 2>&1 | tee $O/qchain_ab_nc16.txt
TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_fused.txt
ICL_QCHAIN=0 TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_unfused.txt
bash tools/gpu_run.sh stats r6b_unet 2>&1 | tee $O/stats.txt
python tools/conv_time.py --shapes "16,16,96,fwd;48,16,96,fwd;32,16,48,fwd" > $O/conv_ws_two_units.txt 2>&1; cat $O/conv_ws_two_units.txt
