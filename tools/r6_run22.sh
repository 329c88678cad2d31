cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w; mkdir -p $O
for rep in 1 2; do for v in 96 112 128 144 160; do
  ICL_UPDATE_WGS=$v timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ICL_UPDATE_WGS=$v', d['ms_per_step'])"
done; done | tee $O/update_wgs.txt
for rep in 1 2; do for v in 256 192 128 96; do
  ICL_DGRAD_KSPLIT_CUS=$v timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ICL_DGRAD_KSPLIT_CUS=$v', d['ms_per_step'])"
done; done | tee $O/dgrad_cus.txt
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/suite.txt
