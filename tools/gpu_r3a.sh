# round 3, first GPU pass: parity tests, A/B of the forward-kernel schedules, bench with both
cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -x -q > $O/r3a_tests.log 2>&1; echo "tests rc=$?" >> $O/r3a_tests.log
tail -15 $O/r3a_tests.log
python3 tools/conv_ab.py --shapes "16,16,96,fwd;48,16,96,fwd;16,48,96,fwd;32,32,48,fwd;96,32,48,fwd;16,32,48,fwd;32,96,48,fwd" --var ICL_CONV_SPLIT_V=0 --var ICL_CONV_SPLIT_V=1 > $O/r3a_ab.log 2>&1
cat $O/r3a_ab.log
python3 bench.py --no-cpu-baseline --no-exact-compare > $O/r3a_bench_v1.json 2> $O/r3a_bench_v1.err
ICL_CONV_SPLIT_V=0 python3 bench.py --no-cpu-baseline --no-exact-compare > $O/r3a_bench_v0.json 2> $O/r3a_bench_v0.err
cut -c1-400 $O/r3a_bench_v1.json; cut -c1-400 $O/r3a_bench_v0.json
