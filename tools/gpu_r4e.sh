#!/bin/bash
# round 4, run e: in-kernel stamps of the wave-specialised forward kernel
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96 3 3" "48 16 96 3 3"; do
  timeout 120 $P $s 2>&1
done
} > gpurun_out/r4e_ws_stamps.txt 2>&1
cat gpurun_out/r4e_ws_stamps.txt
