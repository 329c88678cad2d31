cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aa; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['config'].get('launch'))" || echo "$* failed"; }
for rep in 1 2; do
  run X=0
  run GPU_MAX_HW_QUEUES=3
  run GPU_MAX_HW_QUEUES=5
  run GPU_MAX_HW_QUEUES=6
  run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
  run DEBUG_HIP_FORCE_GRAPH_QUEUES=6
done 2>&1 | tee $O/hw_queues.txt
