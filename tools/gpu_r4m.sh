#!/bin/bash
# round 4, run m: weight gradient with producer waves against the shipped transposing-read kernel
mkdir -p gpurun_out
{
for s in "16 16 96 5" "48 16 96 5" "32 16 96 3" "32 32 48 5" "32 32 48 5 1" "96 32 48 5" "96 32 48 5 1" "16 32 48 5 1" "64 64 24 5 1"; do
  timeout 120 tools/probe/wgradwsprobe $s 2>&1
done
} > gpurun_out/r4m_wgrad_ws.txt 2>&1
cat gpurun_out/r4m_wgrad_ws.txt
