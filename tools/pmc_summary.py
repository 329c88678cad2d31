"""Summarise rocprofv3 --pmc output of tools/pmc_conv.sh / pmc_gemm.sh: per kernel, counters averaged per launch."""
import collections
import csv
import sys

for d in sys.argv[1:]:
    rows = list(csv.DictReader(open(d + "/p_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"][:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = len(cnt[k])
        if n < 5:
            continue
        print(d.split("/")[-1], k, "launches", n)
        print("   ", {c: round(x / n / 1e6, 3) for c, x in v.items()}, "(millions per launch)")
