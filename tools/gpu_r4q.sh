#!/bin/bash
# round 4, run q: GPU suite + bench with InstanceNorm statistics from the convolution epilogue (ICL_CONV_STATS=0: the separate pass)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4q_gpu_tests.txt
python bench.py --no-cpu-baseline --no-exact-compare > gpurun_out/r4q_bench.json 2> gpurun_out/r4q_bench.err
ICL_CONV_STATS=0 python bench.py --no-cpu-baseline --no-exact-compare > gpurun_out/r4q_bench_nostats.json 2>/dev/null
python bench.py --no-cpu-baseline --no-exact-compare > gpurun_out/r4q_bench2.json 2>/dev/null
tail -6 gpurun_out/r4q_gpu_tests.txt; for f in r4q_bench r4q_bench_nostats r4q_bench2; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"; done
