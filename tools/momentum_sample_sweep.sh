cd /root/repo
python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_LN_LONGROW=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_GEMM_TINY_SPLIT=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_LN_LONGROW=0 ICL_GEMM_TINY_SPLIT=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_CONV_SPLIT=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_CONV_SPLIT=0 ICL_GEMM_TINY_SPLIT=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_CONV_STATS=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
ICL_LINEAR_STREAM=0 python tests/diag/momentum_sample.py 2>&1 | tail -1
bash tools/gpu_run.sh ab ICL_GEMM_TINY_SPLIT 0 1
