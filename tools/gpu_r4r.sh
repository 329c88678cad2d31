#!/bin/bash
# round 4, run r: full GPU suite (incl. the SwinUNETR bit-reproducibility test) + SwinUNETR and nc=16 benches
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4r_gpu_tests.txt
python bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare > gpurun_out/r4r_swin_bench.json 2> gpurun_out/r4r_swin.err
python bench.py --num-classes 16 --no-cpu-baseline --no-exact-compare > gpurun_out/r4r_nc16_bench.json 2>/dev/null
tail -8 gpurun_out/r4r_gpu_tests.txt; cut -c1-250 gpurun_out/r4r_swin_bench.json; cut -c1-250 gpurun_out/r4r_nc16_bench.json
