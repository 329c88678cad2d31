#!/bin/bash
# hardware queues: the step forks into 5 streams (6 with the update stream); ROCclr maps streams onto GPU_MAX_HW_QUEUES (default 4) queues
cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do
for q in 4 8 16; do
for pl in fused gated; do
  GPU_MAX_HW_QUEUES=$q ICL_UPDATE_PLACEMENT=$pl python bench.py --no-cpu-baseline --no-exact-compare --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('queues $q $pl', d['ms_per_step'], d['value'])"
done; done; done
echo "== critical path, gated, 8 queues"
GPU_MAX_HW_QUEUES=8 ICL_UPDATE_PLACEMENT=gated python tools/critical_path.py 2>&1 | tail -22
echo "== critical path, fused, 8 queues"
GPU_MAX_HW_QUEUES=8 python tools/critical_path.py 2>&1 | tail -22
mkdir -p gpurun_out/r4z_prof; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4z_prof -o b --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 24 --warmup 3 > /dev/null 2>&1
cd /root/repo
f=$(find gpurun_out/r4z_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 27
print('launches/step', sum(int(r['Calls']) for r in rows) / steps, 'kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6)
PY
