#!/bin/bash
# round 4, run a: planes + LDS-DMA forward kernel against the shipped one (tools/probe/conv_planes_probe.hip)
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96" "48 16 96" "32 32 48" "96 32 48" "16 32 48" "64 64 24" "192 64 24" "48 48 96" "32 16 96"; do
  timeout 120 $P $s 5
done
} > gpurun_out/r4a_planes_probe.txt 2>&1
cat gpurun_out/r4a_planes_probe.txt
