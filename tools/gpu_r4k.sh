#!/bin/bash
# round 4, run k: wave-specialised kernel v2: loader priority 0 / 1 / 3, consumer weight loads in front of barrier A
mkdir -p gpurun_out
{
for pr in 0 1 3; do
  echo "=== loader priority $pr"
  for s in "16 16 96 5 3" "48 16 96 5 3" "32 16 96 3 3"; do
    timeout 120 tools/probe/planesprobe_p$pr $s 2>&1 | grep -v 'item 3\|item 4'
  done
done
} > gpurun_out/r4k_ws2_prio.txt 2>&1
cat gpurun_out/r4k_ws2_prio.txt
