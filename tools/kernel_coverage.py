#!/usr/bin/env python3
"""Kernel symbols of icl_amd/libicl_hip.so against the kernel names in rocprofv3 kernel-stats CSVs (tools/kernel_coverage.sh)."""
import csv, glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "icl_amd", "libicl_hip.so")
# kernel names: the host-side launch stubs of the library (one per __global__ instantiation)
syms = set()
out = subprocess.run(["nm", "--demangle", lib], capture_output=True, text=True).stdout
for ln in out.splitlines():
    m = re.match(r"\S+\s+\S\s+(?:void )?(.*__device_stub__.*)$", ln.strip())
    if m:
        syms.add(m.group(1).replace("__device_stub__", ""))
def norm(n):
    n = re.sub(r"^void ", "", n.strip())
    n = re.sub(r"\(.*$", "", n)          # drop the argument list
    return n.replace(" ", "")
launched = {}
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*"))):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            launched.setdefault(norm(r["Name"]), set()).add(os.path.basename(d))
have = {norm(s): s for s in syms}
icl_launched = {k: v for k, v in launched.items() if k.startswith("icl::")}
print(f"kernels in libicl_hip.so: {len(have)}; distinct icl kernels launched: {len(icl_launched)}; other kernels launched (ATen / RCCL): {len(launched) - len(icl_launched)}")
never = sorted(k for k in have if k not in launched)
print(f"\nin the library, launched by none of (GPU suite, bench unet / unet16 / swin / reference loop): {len(never)}")
for k in never:
    print("   ", k)
only_suite = sorted(k for k, v in icl_launched.items() if v == {"suite"})
print(f"\nlaunched by the GPU suite only (no bench configuration uses them): {len(only_suite)}")
for k in only_suite:
    print("   ", k)
unknown = sorted(k for k in icl_launched if k not in have)
if unknown:
    print(f"\nlaunched but not matched to a library symbol (name normalisation): {len(unknown)}")
    for k in unknown[:20]:
        print("   ", k)
