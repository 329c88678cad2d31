#!/usr/bin/env python3
"""First convolution of the backbones (1 -> 16 channels, 96^3, batch 2): forward and weight gradient, dedicated kernels
(csrc/kernels/conv_cin1.h) against the implicit-GEMM / shifted-planes path (ICL_CONV_CIN1=0).  python tools/first_conv_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
x = torch.randn(2, 1, 96, 96, 96, device=dev)
w = (torch.randn(16, 1, 3, 3, 3, device=dev) * 0.2).requires_grad_()
b = torch.zeros(16, device=dev, requires_grad=True)
g = torch.randn(2, 16, 96, 96, 96, device=dev)


def timed(fn, it=20):
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


for mode in ("1", "0", "1", "0"):
    os.environ["ICL_CONV_CIN1"] = mode
    with torch.no_grad():
        f = timed(lambda: ops.conv3d(x, w, b, zero_bias_grad=True))

    def fb():
        w.grad = None
        ops.conv3d(x, w, b, zero_bias_grad=True).backward(g)
    t = timed(fb)
    y = ops.conv3d(x, w, b, zero_bias_grad=True)
    w.grad = None
    y.backward(g)
    wr = w.detach().clone().requires_grad_()
    yr = F.conv3d(x, wr, None, padding=1)
    yr.backward(g)
    ey = float((y - yr).abs().max() / yr.abs().max())
    ew = float((w.grad - wr.grad).abs().max() / wr.grad.abs().max())
    print(f"ICL_CONV_CIN1={mode}: forward {f:6.1f} us, forward + weight gradient {t:6.1f} us (difference {t - f:6.1f});  rel err y {ey:.1e}  dW {ew:.1e}")
