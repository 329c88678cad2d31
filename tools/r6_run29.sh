cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ad; mkdir -p $O
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['config'].get('launch'))" || echo "$* failed"; }
for rep in 1 2 3; do
  run ICL_TAIL_HEADS=0
  run ICL_TAIL_HEADS=1
done 2>&1 | tee $O/tail_heads_ab.txt
for v in 0 1; do echo "== ICL_TAIL_HEADS=$v"; ICL_TAIL_HEADS=$v CP_ALIGNER_DETAIL=0 TAIL=40 bash tools/gpu_run.sh critical-path 2>&1 | grep -E "replayed|pool1 ready|backward done|step end"; done | tee $O/cp_heads.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -x 2>&1 | grep -E "passed|failed" | tee $O/tests.txt
