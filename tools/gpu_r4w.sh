#!/bin/bash
# round 4, run w: GPU suite + U-Net bench + kernel-stats launch count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python3 -m pytest $R/tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do python3 $R/bench.py --no-cpu-baseline --no-exact-compare 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unet', d['ms_per_step'], d['value'])"; done
rocprofv3 --kernel-trace --stats -d $O/r4w_prof -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --launch graph > /dev/null 2>&1
rm -f $O/r4w_prof/b_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/r4w_prof/b_kernel_stats.csv')))
print('launches/step', sum(int(r['Calls']) for r in rows)/27, 'kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows)/27e6)
for r in rows:
    if 'at::' in r['Name'] or 'rocclr' in r['Name']: print(' ', r['Name'][:90], int(r['Calls'])/27)
PY
