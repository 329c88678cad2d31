#!/bin/bash
# round 4, run b: planes kernel configurations 1 (two 4-wave workgroups per CU) and 2 (double-buffered halo) against the shipped kernel
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96 5 1" "48 16 96 5 1" "32 16 96 5 1" "16 16 48 5 1" "16 16 96 5 2" "48 16 96 5 2" "32 32 48 5 2" "96 32 48 5 2" "16 32 48 5 2" "64 32 48 5 2" "32 32 96 3 2"; do
  timeout 120 $P $s
done
} > gpurun_out/r4b_planes_probe.txt 2>&1
cat gpurun_out/r4b_planes_probe.txt
