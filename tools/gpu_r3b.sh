# round 3, second GPU pass: all parity tests (no -x), the compat-root test with exact-fp32 convolutions, A/B of the kernels, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3b_tests.log 2>&1; echo "tests rc=$?" >> $O/r3b_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3b_tests.log | tail -30
ICL_CONV_SPLIT=0 python3 -m pytest tests/test_gpu_dropin.py -m gpu -q -k compat_root > $O/r3b_compat_exact.log 2>&1; grep -E "passed|failed|assert 0" $O/r3b_compat_exact.log | tail -5
ICL_WGRAD_TR=0 python3 -m pytest tests/test_gpu_dropin.py -m gpu -q -k compat_root > $O/r3b_compat_tr0.log 2>&1; grep -E "passed|failed|assert 0" $O/r3b_compat_tr0.log | tail -5
python3 tools/conv_ab.py --shapes "16,16,96,fwd;48,16,96,fwd;32,32,48,fwd;96,32,48,fwd" --var ICL_CONV_SPLIT_V=0 --var ICL_CONV_SPLIT_V=1 --var ICL_CONV_SPLIT_V=3 > $O/r3b_ab_fwd.log 2>&1
cat $O/r3b_ab_fwd.log
python3 tools/conv_ab.py --shapes "16,16,96,wgrad;48,16,96,wgrad;16,48,96,wgrad;32,32,48,wgrad;96,32,48,wgrad;16,32,48,wgrad;64,64,24,wgrad;192,64,24,wgrad;64,128,12,wgrad" --var ICL_WGRAD_TR=0 --var ICL_WGRAD_TR=1 > $O/r3b_ab_wgrad.log 2>&1
cat $O/r3b_ab_wgrad.log
python3 bench.py --no-cpu-baseline --no-exact-compare > $O/r3b_bench.json 2> $O/r3b_bench.err
cut -c1-300 $O/r3b_bench.json
