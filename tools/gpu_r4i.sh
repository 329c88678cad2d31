#!/bin/bash
# round 4, run i: halo-fetch microbenchmark by layout, then the kernel-trace timeline of the replayed U-Net step
mkdir -p gpurun_out
timeout 120 tools/probe/halofetch 96 > gpurun_out/r4i_halofetch.txt 2>&1
timeout 120 tools/probe/halofetch 48 >> gpurun_out/r4i_halofetch.txt 2>&1
cat gpurun_out/r4i_halofetch.txt
bash tools/gpu_r4h.sh
