"""One 96^3-level call of every HBM-bound glue kernel (run under rocprofv3 --kernel-trace --stats for kernel-only durations):
InstanceNorm+ReLU fwd / bwd, max-pool fwd / bwd, dropout fwd / bwd, 2x up-sampling + concat fwd / bwd on [2, 16, 96^3] (+ [2, 32, 48^3])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

dev = torch.device("cuda")
for _ in range(5):
    x = torch.randn(2, 16, 96, 96, 96, device=dev, requires_grad=True)
    g = torch.randn(2, 16, 96, 96, 96, device=dev)
    ops.instance_norm_relu(x).backward(g)
    x.grad = None
    y = ops.max_pool3d_2(x)
    y.backward(torch.randn_like(y))
    x.grad = None
    ops.dropout(x, 0.3, seed=7).backward(g)
    x.grad = None
    deep = torch.randn(2, 32, 48, 48, 48, device=dev, requires_grad=True)
    ops.upsample2x_concat(x, deep).backward(torch.randn(2, 48, 96, 96, 96, device=dev))
torch.cuda.synchronize()
print("done: [2,16,96^3] = 113 MB per tensor")
