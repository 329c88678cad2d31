#!/usr/bin/env python3
"""Timeline summary of one replayed step from a rocprofv3 kernel trace (…_kernel_trace.csv): wall time of the step, union of
kernel-busy intervals, time with two or more kernels in flight, idle gaps, and the longest kernels that run alone.
usage: tools/timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
# a step starts with pack_weights_multi_kernel (the first launch of ICLTrainer._forward_backward)
starts = [i for i, e in enumerate(ev) if "pack_weights_multi" in e[2]]
i0, i1 = starts[-back - 1], starts[-back]
step = ev[i0:i1]
t0, t1 = step[0][0], max(e[1] for e in step)
pts = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
busy = multi = 0
depth, last = 0, t0
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d
    last = t
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, >= 2 kernels in flight {multi / 1e6:.3f} ms, "
      f"idle {((t1 - t0) - busy) / 1e6:.3f} ms, sum of kernel durations {sum(e - s for s, e, _, _ in step) / 1e6:.3f} ms")
queues = {}
for s, e, n, q in step:
    queues.setdefault(q, [0, 0]); queues[q][0] += 1; queues[q][1] += e - s
print("queues:", {q: (c, round(d / 1e6, 3)) for q, (c, d) in queues.items()})
gaps = []
depth, last = 0, t0
for t, d in pts:
    if depth == 0 and t > last: gaps.append(t - last)
    depth += d
    last = t
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps)}, top 10 (us): {[round(g / 1e3, 1) for g in gaps[:10]]}, gaps > 1 us: {sum(1 for g in gaps if g > 1000)}, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us")
# per-bucket view: which queue is busy with what (dominant kernel by time in the bucket), to read the critical path
if len(sys.argv) > 3:
    width = float(sys.argv[3]) * 1e6
    qs = sorted(queues)
    nb = int((t1 - t0) / width) + 1
    table = [[{} for _ in qs] for _ in range(nb)]
    for s, e, n, q in step:
        short = n.replace("void ", "").replace("icl::", "").split("(")[0][:34]
        b0, b1 = int((s - t0) / width), int((e - t0) / width)
        for b in range(b0, b1 + 1):
            lo, hi = max(s, t0 + b * width), min(e, t0 + (b + 1) * width)
            if hi > lo:
                d = table[b][qs.index(q)]
                d[short] = d.get(short, 0) + hi - lo
    for b in range(nb):
        cells = []
        for d in table[b]:
            tot = sum(d.values())
            top = max(d, key=d.get) if d else ""
            cells.append(f"{100 * tot / width:3.0f}% {top:34s}")
        print(f"{b * width / 1e6:5.1f} ms | " + " | ".join(cells))
# kernel sequence of a window [a, b] ms of the step: name, queue, start offset, duration (kernels >= min_us)
if len(sys.argv) > 6:
    a, b, min_us = float(sys.argv[4]) * 1e6, float(sys.argv[5]) * 1e6, float(sys.argv[6])
    small, names = 0, {}
    for s, e, n, q in step:
        if a <= s - t0 <= b:
            if (e - s) / 1e3 < min_us:
                k = n.replace('void ', '').replace('icl::', '').split('(')[0][:70]
                names.setdefault(k, [0, 0]); names[k][0] += 1; names[k][1] += e - s
            if (e - s) / 1e3 >= min_us:
                print(f"  +{(s - t0) / 1e6:7.3f} ms  q{q}  {(e - s) / 1e3:7.1f} us  {n.replace('void ', '').replace('icl::', '')[:110]}")
            else:
                small += e - s
    print(f"  (kernels under {min_us} us in the window: {small / 1e6:.3f} ms)")
    for k, (c, d) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print(f"     {c:3d} x {d / c / 1e3:5.1f} us  {k}")
