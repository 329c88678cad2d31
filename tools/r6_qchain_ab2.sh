cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
python tools/qchain_probe.py 2 2 > $O/qchain_probe_nc2.txt 2>&1; cat $O/qchain_probe_nc2.txt
python tools/qchain_probe.py 16 2 > $O/qchain_probe_nc16.txt 2>&1; tail -40 $O/qchain_probe_nc16.txt
python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite.txt
ICL_QCHAIN=0 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "reference_loop_body_unet3d_icl_through_compat_root" 2>&1 | tail -5 | tee $O/dropin_unfused.txt
bash tools/gpu_run.sh ab ICL_QCHAIN 0 1 2>&1 | tee $O/qchain_ab.txt
TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_fused.txt
