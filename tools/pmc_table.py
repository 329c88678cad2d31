#!/usr/bin/env python3
"""Markdown table of the convolution PMC collection (tools/gpu_run.sh pmc): per shape tag the dominant kernel's counters per launch.
    python tools/pmc_table.py gpurun_out f16 f48 ...      (reads gpurun_out/pmc_<tag>_1|_2, hbm_<tag>_f|_w)"""
import collections
import csv
import sys

root, tags = sys.argv[1], sys.argv[2:]


def per_kernel(d):
    try:
        rows = list(csv.DictReader(open(f"{root}/{d}/p_counter_collection.csv")))
    except OSError:
        return {}
    agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: {c: x / len(cnt[k]) for c, x in v.items()} for k, v in agg.items() if len(cnt[k]) >= 5}


print("| tag | kernel | MFMA_BUSY (M) | kernel cycles (M) | **matrix pipe busy** | INSTS_VALU | INSTS_LDS | LDS_BANK_CONFLICT | LDS_IDX_ACTIVE | WAIT_INST_ANY | WAIT_ANY | FETCH_SIZE MB | WRITE_SIZE MB |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for t in tags:
    a, b = per_kernel(f"pmc_{t}_1"), per_kernel(f"pmc_{t}_2")
    f, w = per_kernel(f"hbm_{t}_f"), per_kernel(f"hbm_{t}_w")
    main = max((k for k in a if "split_weights" not in k and "reduce" not in k), key=lambda k: a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), default=None)
    if main is None:
        continue
    for k in a:
        if k != main and not ("splitk_reduce" in k):
            continue
        c1, c2 = a[k], b.get(k, {})
        cyc = c1.get("SQ_BUSY_CYCLES", 0) / 32
        pipe = c1.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc if cyc else 0
        m = lambda v: f"{v / 1e6:.2f}"      # noqa: E731
        name = k.split("(")[0].replace("void icl::", "").replace("icl::", "")
        fs = f.get(k, {}).get("FETCH_SIZE")
        wsz = w.get(k, {}).get("WRITE_SIZE")
        print(f"| {t} | `{name}` | {m(c1.get('SQ_VALU_MFMA_BUSY_CYCLES', 0))} | {cyc / 1e6:.4f} | **{100 * pipe:.0f} %** | {m(c1.get('SQ_INSTS_VALU', 0))} | "
              f"{m(c1.get('SQ_INSTS_LDS', 0))} | {m(c1.get('SQ_LDS_BANK_CONFLICT', 0))} | {m(c2.get('SQ_LDS_IDX_ACTIVE', 0))} | {m(c2.get('SQ_WAIT_INST_ANY', 0))} | "
              f"{m(c2.get('SQ_WAIT_ANY', 0))} | {'' if fs is None else round(fs * 1024 / 1e6, 1)} | {'' if wsz is None else round(wsz * 1024 / 1e6, 1)} |")
