#!/bin/bash
# round 4, run l: wave-specialised kernel v3 (late waves' epilogue beside the deposit, loader priority 0)
mkdir -p gpurun_out
{
for s in "16 16 96 7 3" "48 16 96 5 3" "32 16 96 5 3" "16 16 48 5 3" "32 32 48 5 3" "96 32 48 5 3"; do
  timeout 120 tools/probe/planesprobe $s 2>&1 | grep -v 'item 3\|item 4'
done
} > gpurun_out/r4l_ws3.txt 2>&1
cat gpurun_out/r4l_ws3.txt
