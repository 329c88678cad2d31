cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 tests/diag/three_steps.py > $O/r3e_three.log 2>&1; grep -v Warning $O/r3e_three.log | tail -24
python3 bench.py --no-cpu-baseline > $O/r3e_bench.json 2> $O/r3e_bench.err; cut -c1-250 $O/r3e_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r3e_prof -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-compare --launch graph > /dev/null 2>&1
rm -f $GRAFT_REPO_ROOT/$O/r3e_prof/*kernel_trace.csv; ls $GRAFT_REPO_ROOT/$O/r3e_prof
