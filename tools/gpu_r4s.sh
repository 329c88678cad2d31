#!/bin/bash
# round 4, run s: SwinUNETR step with / without the statistics from the convolution epilogue, same box; kernel stats of the swin step
mkdir -p gpurun_out
for i in 1 2; do
  for st in 1 0; do
    ICL_CONV_STATS=$st python bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin stats=$st', d['ms_per_step'])"
  done
done > gpurun_out/r4s_swin_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4s_prof_swin -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-compare --launch graph --model swinunetr_icl > /dev/null 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/r4s_prof_swin/b_kernel_trace.csv
cd $GRAFT_REPO_ROOT; cat gpurun_out/r4s_swin_ab.txt; head -25 gpurun_out/r4s_prof_swin/b_kernel_stats.csv | cut -c1-150
