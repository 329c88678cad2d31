"""Run one product of csrc/kernels/gemm.h a few times (for rocprofv3 --pmc / --kernel-trace):
   python tools/gemm_one.py <rows> <in> <out> <fwd|dgrad|wgrad> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

rows, i, o = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda")
x = torch.randn(rows, i, device=dev)
w = torch.randn(o, i, device=dev) * 0.05
b = torch.randn(o, device=dev)
g = torch.randn(rows, o, device=dev)
for _ in range(reps):
    if mode == "fwd":
        ops.linear_forward_raw(x, w, b)
    elif mode == "dgrad":
        ops.linear_dgrad_raw(g, w)
    else:
        ops._tall_atb(g, x, False)
torch.cuda.synchronize()
