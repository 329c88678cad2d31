cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
python -X faulthandler bench.py --steps 5 > $O/a.json 2> $O/a.err; echo "default rc=$?"; tail -25 $O/a.err
python -X faulthandler bench.py --steps 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; echo "no-cpu rc=$?"; tail -5 $O/b.err
python -X faulthandler bench.py --steps 5 --no-other-workloads > $O/c.json 2> $O/c.err; echo "no-other rc=$?"; tail -5 $O/c.err
