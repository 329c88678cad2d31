#!/bin/bash
# round 4, run j: wave-specialised forward kernel, all loads of the next tile issued at once
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96 5 3" "48 16 96 5 3" "32 16 96 5 3" "16 16 48 5 3" "32 32 48 5 3" "96 32 48 5 3" "64 64 24 5 3" "16 48 96 3 3"; do
  timeout 120 $P $s 2>&1 | grep -v 'item 3\|item 4'
done
} > gpurun_out/r4j_ws2_probe.txt 2>&1
cat gpurun_out/r4j_ws2_probe.txt
