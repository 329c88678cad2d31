#!/bin/bash
# round 4, run c: in-kernel stamps and ablations of the planes kernel configurations
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96 3 0" "16 16 96 3 1" "48 16 96 3 0" "32 32 48 3 0" "32 32 48 3 2" "96 32 48 3 2"; do
  timeout 120 $P $s
done
} > gpurun_out/r4c_planes_stamps.txt 2>&1
cat gpurun_out/r4c_planes_stamps.txt
