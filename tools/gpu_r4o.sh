#!/bin/bash
# round 4, run o: GPU test suite + bench with the loader-wave forward kernel as the default for one cout block
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r4o_gpu_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/r4o_bench.json 2> gpurun_out/r4o_bench.err
ICL_CONV_SPLIT_V=60 python bench.py --no-cpu-baseline --no-exact-compare > gpurun_out/r4o_bench_v60.json 2>/dev/null
tail -25 gpurun_out/r4o_gpu_tests.txt; cut -c1-600 gpurun_out/r4o_bench.json; cut -c1-300 gpurun_out/r4o_bench_v60.json
