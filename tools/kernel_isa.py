#!/usr/bin/env python3
"""Instruction census of one kernel in a device assembly file (hipcc --cuda-device-only -S): registers, LDS, scratch, and counts of
the instruction classes that matter for the matrix-bound kernels.   tools/kernel_isa.py /tmp/icl.s conv3d_bf16x3_fwd_kernel [filter]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
# simpler: split on kernel symbol labels and read metadata from the '.amdhsa_' block that follows
text = open(path).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
    sym, blk = m.group(1), m.group(2)
    if pat not in sym:
        continue
    demangled = sym
    code = blk.split(".amdhsa_kernel")[0]
    meta = blk.split(".amdhsa_kernel")[1] if ".amdhsa_kernel" in blk else ""
    ops = collections.Counter()
    for ln in code.splitlines():
        ln = ln.strip()
        if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"):
            continue
        ops[ln.split()[0]] += 1
    def grab(k):
        mm = re.search(rf"\.amdhsa_{k}\s+(\S+)", meta)
        return mm.group(1) if mm else "?"
    cls = collections.Counter()
    for op, n in ops.items():
        if op.startswith("v_mfma"): cls["mfma"] += n
        elif op.startswith("ds_read") or op.startswith("ds_load"): cls["ds_read"] += n
        elif op.startswith("ds_write") or op.startswith("ds_store"): cls["ds_write"] += n
        elif op.startswith(("global_load", "buffer_load")): cls["vmem_load"] += n
        elif op.startswith(("global_store", "buffer_store")): cls["vmem_store"] += n
        elif op.startswith("scratch_"): cls["scratch"] += n
        elif op.startswith("v_"): cls["valu"] += n
        elif op.startswith("s_waitcnt"): cls["waitcnt"] += n
        elif op.startswith("s_barrier"): cls["barrier"] += n
        elif op.startswith("s_"): cls["salu"] += n
    print(f"{sym}\n  vgpr {grab('next_free_vgpr')} accum_offset {grab('accum_offset')} sgpr {grab('next_free_sgpr')} lds {grab('group_segment_fixed_size')} "
          f"scratch {grab('private_segment_fixed_size')}\n  {dict(cls)}")
    if len(sys.argv) > 3:
        for op, n in sorted(ops.items(), key=lambda kv: -kv[1])[:int(sys.argv[3])]:
            print(f"    {op:32s} {n}")
