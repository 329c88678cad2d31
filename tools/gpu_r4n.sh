#!/bin/bash
# round 4, run n: weight gradient with 8 / 4 producer waves at priority 0 / 2
mkdir -p gpurun_out
{
for v in 8_0 8_2 4_2; do
  for s in "16 16 96 5" "48 16 96 3" "32 32 48 5 1" "96 32 48 3 1"; do
    timeout 120 tools/probe/wgradwsprobe_$v $s 2>&1
  done
done
} > gpurun_out/r4n_wgrad_ws2.txt 2>&1
cat gpurun_out/r4n_wgrad_ws2.txt
