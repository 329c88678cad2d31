cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y; mkdir -p $O
python tools/conv_time.py --shapes "128,256,6,fwd;256,256,6,fwd;256,128,6,fwd" 2>&1 | grep -v "^#\|amdgpu.ids" | tee $O/conv_f6_8w.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "conv3d or split" 2>&1 | tail -2 | tee $O/conv_tests.txt
