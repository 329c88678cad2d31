"""HBM-bound glue kernels of the U-Net levels against their algorithmic bytes: InstanceNorm+ReLU forward (12 B/element: stats pass
+ apply pass) and backward (20 B/element), max-pool, upsample+concat, dropout.  `python tools/norm_probe.py` (GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

dev = torch.device("cuda")


def t(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for (n, c, r) in ((2, 16, 96), (2, 32, 48), (2, 64, 24), (2, 128, 12), (2, 48, 96)):
    x = torch.randn(n, c, r, r, r, device=dev, requires_grad=True)
    g = torch.randn(n, c, r, r, r, device=dev)
    el = x.numel()
    y = ops.instance_norm_relu(x)
    f = t(lambda: ops.instance_norm_relu(x))
    fb = t(lambda: ops.instance_norm_relu(x).backward(g))
    b = fb - f
    xd = x.detach()
    with torch.no_grad():
        cp = t(lambda: xd.clone())
        pool = t(lambda: ops.max_pool3d_2(xd))
        dr = t(lambda: ops.dropout(xd, 0.3, seed=5))
    print(f"[{n},{c},{r}^3] {el * 4 / 1e6:7.1f} MB  norm fwd {f:7.1f} us ({12 * el / f / 1e6:5.2f} TB/s)  bwd {b:7.1f} us ({20 * el / b / 1e6:5.2f} TB/s)"
          f"  clone {cp:6.1f} us ({8 * el / cp / 1e6:5.2f})  pool {pool:6.1f} us ({4.5 * el / pool / 1e6:5.2f})  dropout {dr:6.1f} us ({8 * el / dr / 1e6:5.2f})", flush=True)
