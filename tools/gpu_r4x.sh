#!/bin/bash
# A/B of FusedSGD.update_placement inside the captured step: stream priority of the update stream
cd /root/repo
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
s=torch.cuda.Stream(priority=5); print('prio 5 ->', s.priority)
s=torch.cuda.Stream(priority=-5); print('prio -5 ->', s.priority)
"
for rep in 1 2; do
for pr in 1 2 -1; do
for pl in gated free; do
  ICL_UPDATE_STREAM_PRIORITY=$pr ICL_UPDATE_PLACEMENT=$pl python bench.py --no-cpu-baseline --no-exact-compare --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$pl prio $pr', d['ms_per_step'], d['value'])"
done; done; done
echo "== critical path, gated prio 2"
ICL_UPDATE_STREAM_PRIORITY=2 ICL_UPDATE_PLACEMENT=gated python tools/critical_path.py 2>&1 | tail -24
