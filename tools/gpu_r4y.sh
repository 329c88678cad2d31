#!/bin/bash
# GPU suite, then the fused LayerNorm backward on/off, launches per step
cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -25
for rep in 1 2; do
for ln in 1 0; do
  ICL_LN_FUSED_WGRAD=$ln python bench.py --no-cpu-baseline --no-exact-compare --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('unet ln_fused=$ln', d['ms_per_step'], d['value'])"
done; done
python bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('swin', d['ms_per_step'], d['value'])"
mkdir -p gpurun_out/r4y_prof; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r4y_prof -o b -- python3 /root/repo/bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 24 --warmup 3 > /dev/null 2>&1
cd /root/repo
f=$(find gpurun_out/r4y_prof -name "*kernel_stats.csv" | head -1); ls gpurun_out/r4y_prof | head
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 27
print('launches/step', sum(int(r['Calls']) for r in rows) / steps, 'kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6)
PY
