#!/bin/bash
# round 4, run d: wave-specialised forward kernel (8 consumer + 4 loader waves, fp32 input) against the shipped kernel
mkdir -p gpurun_out
P=tools/probe/planesprobe
{
for s in "16 16 96 5 3" "48 16 96 5 3" "32 16 96 5 3" "16 16 48 5 3" "32 32 48 5 3" "96 32 48 5 3" "16 48 96 3 3"; do
  timeout 120 $P $s 2>&1 | grep -v 'stamps\|ablation'
done
} > gpurun_out/r4d_ws_probe.txt 2>&1
cat gpurun_out/r4d_ws_probe.txt
