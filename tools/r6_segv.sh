cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
for i in 1 2; do
  python -X faulthandler bench.py --steps 5 --no-cpu-baseline > $O/d$i.json 2> $O/d$i.err; echo "default-no-cpu run $i rc=$?"; tail -3 $O/d$i.err
done
(time python bench.py > $O/full.json 2> $O/full.err); echo "full default rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6k/full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print(json.dumps(d['config']['other_workloads'])[:900])
print(d['cpu_baseline'])
PY
python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/gpu_run.sh ab ICL_WGRAD_DEFER_REDUCE 0 1 2>&1 | tee $O/defer_ab.txt
