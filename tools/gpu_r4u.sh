#!/bin/bash
# round 4, run u: GPU suite + U-Net bench (A/B: fused skip+pool backward is always on; two bench runs)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r4u_gpu_tests.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-exact-compare 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unet', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['all_conv']['ms_per_step'])"; done > gpurun_out/r4u_bench.txt 2>&1
tail -4 gpurun_out/r4u_gpu_tests.txt; cat gpurun_out/r4u_bench.txt
