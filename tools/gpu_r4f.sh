#!/bin/bash
# round 4, run f: loader-wave issue priority (s_setprio 0 / 3) of the wave-specialised forward kernel
mkdir -p gpurun_out
{
for pr in 0 3; do
  echo "=== loader priority $pr"
  for s in "16 16 96 5 3" "48 16 96 3 3" "16 16 48 3 3"; do
    timeout 120 tools/probe/planesprobe_p$pr $s 2>&1
  done
done
} > gpurun_out/r4f_ws_prio.txt 2>&1
cat gpurun_out/r4f_ws_prio.txt
