#!/usr/bin/env python3
"""Copy a rocprofv3 *_kernel_stats.csv into profiles/ with kernel names cut to 140 characters and a header comment.
usage: tools/trim_stats.py <in.csv> <out.csv> "<command that was profiled>" ["extra note"]"""
import csv
import sys

src, dst, cmd = sys.argv[1:4]
note = sys.argv[4] if len(sys.argv) > 4 else ""
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}\n")
    if note:
        f.write(f"# {note}\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        name = r["Name"][:140].replace(",", ";")
        f.write(f"{name},{r['Calls']},{r['TotalDurationNs']},{r['AverageNs']},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
print(dst, len(rows), "kernels")
