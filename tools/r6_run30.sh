cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ae; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['config'].get('launch'))" || echo "$* failed"; }
for rep in 1 2 3 4; do
  run ICL_OPT_BRANCHES=1
  run ICL_OPT_BRANCHES=2
  run ICL_OPT_BRANCHES=4
done 2>&1 | tee $O/opt_branches_ab.txt
