"""1x1x1 convolution on a big volume: the MFMA tile kernel vs a batched library GEMM W @ x[b] (both HBM-bound)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

dev = torch.device("cuda")


def t(fn, it=10):
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


for cin, cout, s in ((96, 48, 96), (48, 96, 96), (96, 48, 48), (1, 48, 96), (48, 2, 96), (192, 96, 24)):
    n = 2
    x = torch.randn(n, cin, s, s, s, device=dev)
    w = torch.randn(cout, cin, 1, 1, 1, device=dev)
    gb = 4 * n * s ** 3 * (cin + cout) / 1e9
    with torch.no_grad():
        a = t(lambda: ops.conv3d(x, w, None))
        b = t(lambda: torch.bmm(w.view(1, cout, cin).expand(n, cout, cin), x.view(n, cin, -1)))
    print(f"{cin}->{cout} @{s}^3: tile kernel {a:7.1f} us ({gb / a * 1e3:5.2f} TB/s)   batched GEMM {b:7.1f} us ({gb / b * 1e3:5.2f} TB/s)", flush=True)
