#!/bin/bash
# round 4, run g: the shipped forward kernel with y-slowest tile order per XCD (flag 1) and non-temporal output stores (flag 2)
mkdir -p gpurun_out
{
for f in 11 12 13; do
  echo "=== flags $((f-10))"
  for s in "16 16 96 7" "48 16 96 5" "32 32 48 7" "96 32 48 5" "64 64 24 7" "16 48 96 3"; do
    timeout 120 tools/probe/planesprobe $s $f 2>&1
  done
done
} > gpurun_out/r4g_tile_order.txt 2>&1
cat gpurun_out/r4g_tile_order.txt
