#!/bin/bash
# round 4, run h: kernel trace of the replayed U-Net step -> per-queue timeline (tools/timeline.py) for profiles/r4_timeline.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/r4_tl -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --launch graph --steps 8 --warmup 2 > $O/r4h_bench.json 2> $O/r4h_bench.err
T=$(ls $O/r4_tl/*kernel_trace.csv $O/r4_tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/timeline.py $T 3 0.25 > $O/r4h_timeline.txt 2>&1
python3 $R/tools/timeline.py $T 3 0.25 0 13 20 > $O/r4h_timeline_big.txt 2>&1
rm -f $O/r4_tl/*kernel_trace.csv $O/r4_tl/*/*kernel_trace.csv
cut -c1-400 $O/r4h_bench.json; head -60 $O/r4h_timeline.txt
