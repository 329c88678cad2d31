R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for i in 1 2 3; do python3 bench.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['config']['launch'])"; done
python3 bench.py --force-ddp --no-cpu-baseline --no-exact-compare 2>/dev/null | tail -1 | cut -c1-200
python3 bench.py --gpus 2 2>&1 | tail -2 | cut -c1-300; echo "rc=$?"
python3 bench.py --feed --no-cpu-baseline --no-exact-compare --no-kernel-timer 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('feed'))"
