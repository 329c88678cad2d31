cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ac; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['config'].get('launch'))" || echo "$* failed"; }
for rep in 1 2 3; do
  run ICL_WGRAD_LATE=0
  run ICL_WGRAD_LATE=3
done 2>&1 | tee $O/wgrad_late3_ab.txt
for v in 3; do echo "== ICL_WGRAD_LATE=$v"; ICL_WGRAD_LATE=$v CP_ALIGNER_DETAIL=0 TAIL=40 bash tools/gpu_run.sh critical-path 2>&1 | grep -E "replayed|deep backward starts|pool4 ready|update stream|pool1 ready|backward done|step end"; done | tee $O/cp_late3.txt
