cd $GRAFT_REPO_ROOT
O=gpurun_out/r6t; mkdir -p $O
SH="128,256,6,fwd;256,256,6,fwd;256,128,6,fwd"
for f6 in 0 1; do echo "== ICL_CONV_SPLIT_FLAT6=$f6"; ICL_CONV_SPLIT_FLAT6=$f6 python tools/conv_time.py --shapes "$SH" 2>&1 | grep -v "^#\|amdgpu.ids"; done | tee $O/conv_f6.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "conv3d or split or skip_and_pool or maxpool" 2>&1 | tail -3 | tee $O/conv_tests.txt
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_FLAT6 0 1 2>&1 | tee $O/f6_ab.txt
bash tools/gpu_run.sh ab ICL_MAXPOOL_BWD_X2 0 1 2>&1 | tee $O/maxpool_x2_ab.txt
bash tools/gpu_run.sh ab ICL_OPT_BRANCHES 1 4 2>&1 | tee $O/opt_branches_ab.txt
