"""Weight gradient of a 1x1x1 convolution on a big volume (conv1x1_wgrad_kernel + its slab sum), alone on the chip: us per call and the
rate of the x + dY stream.   python tools/conv1x1_wgrad_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
L = _lib.lib()
for cin, cout, s in ((16, 2, 96), (96, 48, 96), (48, 2, 96), (1, 48, 96), (96, 48, 48), (192, 96, 24), (27, 16, 96)):
    n, S = 2, s ** 3
    x = torch.randn(n, cin, s, s, s, device=dev)
    gy = torch.randn(n, cout, s, s, s, device=dev)
    gw = torch.empty(cout, cin, device=dev)
    ws = torch.empty(max(L.icl_conv1x1_wgrad_ws_bytes(n, S, cin, cout) // 4, 1), device=dev)

    def run():
        _lib.check(L.icl_conv1x1_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), n, cin, cout, S, cin * S, cout * S,
                                       ops._stream(x)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ref = torch.einsum("bovs,bivs->oi", gy.view(n, cout, 1, S).double(), x.view(n, cin, 1, S).double())
    err = float((gw.double() - ref).abs().max() / ref.abs().max())
    print(f"{cin:4d}->{cout:<3d}@{s:<3d} {us:8.1f} us  {4.0 * n * S * (cin + cout) / us / 1e6:5.2f} TB/s   rel err {err:.1e}", flush=True)
