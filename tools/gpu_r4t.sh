#!/bin/bash
# round 4, run t: rehearsal of the multi-rank bench path (two ranks on the one GPU over gloo) + SwinUNETR bench after the slab-sum fix
mkdir -p gpurun_out
ICL_BENCH_SHARE_GPU=1 ICL_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-exact-compare > gpurun_out/r4t_rehearsal.json 2> gpurun_out/r4t_rehearsal.err
echo "rehearsal rc=$?"; tail -5 gpurun_out/r4t_rehearsal.err | cut -c1-300; tail -1 gpurun_out/r4t_rehearsal.json | cut -c1-1500
python bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare 2>/dev/null | tail -1 | cut -c1-200
python -m pytest tests/test_gpu_parity.py -q -k "swin" 2>&1 | tail -3
