#!/usr/bin/env python3
"""Every kernel of one replayed step from a rocprofv3 kernel trace: start offset, duration, queue, name (one line per kernel).
usage: tools/step_trace.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "pack_weights_multi" in e[2]]
# replayed steps follow each other within a few hundred microseconds: take a pair of consecutive starts whose distance is the smallest typical one
d = sorted(ev[starts[k + 1]][0] - ev[starts[k]][0] for k in range(len(starts) - 1))
typ = d[len(d) // 4]
cands = [k for k in range(len(starts) - 1) if ev[starts[k + 1]][0] - ev[starts[k]][0] < 1.15 * typ]
k = cands[-back] if len(cands) >= back else cands[-1]
step = ev[starts[k]:starts[k + 1]]
t0 = step[0][0]
print(f"# step of {len(step)} kernels, {(ev[starts[k + 1]][0] - t0) / 1e6:.3f} ms to the next step's first kernel")
for s, e, n, q in step:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {n.replace('void ', '').replace('icl::', '').split('(')[0][:90]}")
