#!/bin/bash
# round 4, run v: loader waves + several cout blocks per workgroup (weight-plane DMA ring) against the shipped <2|3, 8, 60> kernels
mkdir -p gpurun_out
{
for s in "32 32 48 5 4" "96 32 48 5 4" "16 32 48 5 4" "64 64 24 5 4" "192 64 24 5 4" "48 48 96 3 4" "16 48 96 3 4" "96 48 96 3 4" "32 32 96 3 4"; do
  timeout 180 tools/probe/planesprobe $s 2>&1
done
} > gpurun_out/r4v_wsr.txt 2>&1
cat gpurun_out/r4v_wsr.txt
