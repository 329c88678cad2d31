"""Decoder up-sampling (UnetUp3_CT: 2x trilinear of the deep map + concat with the skip) forward / backward times at the U-Net levels."""
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


for (cs, cd, r) in ((16, 32, 96), (32, 64, 48), (64, 128, 24), (128, 256, 12)):
    skip = torch.randn(2, cs, r, r, r, device=dev, requires_grad=True)
    deep = torch.randn(2, cd, r // 2, r // 2, r // 2, device=dev, requires_grad=True)
    g = torch.randn(2, cs + cd, r, r, r, device=dev)
    with ops.KernelTimer() as kt:
        for _ in range(5):
            y = ops.upsample2x_concat(skip, deep)
            y.backward(g)
    f = t(lambda: ops.upsample2x_concat(skip.detach(), deep.detach()))
    fb = t(lambda: ops.upsample2x_concat(skip, deep).backward(g))
    out_b, in_b = 2 * cd * r ** 3 * 4, 2 * cd * (r // 2) ** 3 * 4
    print(f"skip {cs} + deep {cd} @{r}^3: forward {f:7.1f} us, forward+backward {fb:7.1f} us; deep part: {out_b / 1e6:.0f} MB fine, {in_b / 1e6:.0f} MB coarse"
          f"  -> backward ~{fb - f:6.1f} us = {(out_b + in_b) / (fb - f) / 1e6:5.2f} TB/s of fine-read + coarse-write", flush=True)
