#!/usr/bin/env python3
"""Order of the kernels of one replayed step per hardware queue, from a rocprofv3 kernel trace: prints, for every queue, the last N kernels before
the step's optimiser kernels (the tails of the backward chains).   python tools/queue_tail.py <kernel_trace.csv> [N]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last launch of the first-convolution forward kernel on
start = max(i for i, r in enumerate(rows) if "conv_cin1_fwd_kernel" in r["Kernel_Name"])
step = rows[start:]
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
t0 = int(step[0]["Start_Timestamp"])
for q, ks in sorted(byq.items()):
    print(f"== queue {q}: {len(ks)} kernels, {sum(int(k['End_Timestamp']) - int(k['Start_Timestamp']) for k in ks) / 1e3:.0f} us of kernel time; last {n_tail}:")
    for k in ks[-n_tail:]:
        d = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
        wg = int(k["Grid_Size_X"]) * int(k["Grid_Size_Y"]) * int(k["Grid_Size_Z"]) // max(1, int(k["Workgroup_Size_X"]) * int(k["Workgroup_Size_Y"]))
        print(f"  +{(int(k['Start_Timestamp']) - t0) / 1e3:9.1f} us {d:7.1f} us  wgs {wg:6d}  {k['Kernel_Name'][:100]}")
