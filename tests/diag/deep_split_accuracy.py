"""Distance to the fp64 convolution, per output channel (max |a - b| / max |b| within the channel, max over channels), of the deep-level
layer shapes on three paths: split products with the launcher's channel-chunk split, split products unsplit (ICL_CONV_SPLIT_KSPLIT=0;
rows of 12 / 6 voxels then run the fp32 kernels), exact-fp32 MFMA (ICL_CONV_SPLIT=0).  Forward / input gradient.
    python tests/diag/deep_split_accuracy.py        (MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd import ops  # noqa: E402
from icl_amd.utils.hashfill import synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)


def chan_err(a, b):
    d = (a.cpu().double() - b).abs().amax(dim=(0, 2, 3, 4))
    return float((d / b.abs().amax(dim=(0, 2, 3, 4)).clamp_min(1e-300)).max())


for cin, cout, r in ((32, 32, 48), (64, 64, 24), (192, 64, 24), (128, 128, 12), (384, 128, 12), (128, 256, 6), (256, 256, 6)):
    x = synthetic_volume((1, cin, r, r, r), 301)
    w = synthetic_volume((cout, cin, 3, 3, 3), 302) * 0.1
    gy = synthetic_volume((1, cout, r, r, r), 303)
    xr = x.double().requires_grad_()
    yr = torch.nn.functional.conv3d(xr, w.double(), None, padding=1)
    yr.backward(gy.double())
    row = []
    for name, env in (("split + chunk split", {"ICL_CONV_SPLIT": "1", "ICL_CONV_SPLIT_KSPLIT": "-1"}),
                      ("split, unsplit", {"ICL_CONV_SPLIT": "1", "ICL_CONV_SPLIT_KSPLIT": "0"}),
                      ("exact fp32", {"ICL_CONV_SPLIT": "0", "ICL_CONV_SPLIT_KSPLIT": "-1"})):
        os.environ.update(env)
        xg = x.to(dev).requires_grad_()
        y = ops.conv3d(xg, w.to(dev), None)
        y.backward(gy.to(dev))
        row.append(f"{name}: {chan_err(y.detach(), yr.detach()):.2e} / {chan_err(xg.grad, xr.grad):.2e}")
    print(f"{cin:4d}->{cout:<4d}@{r:<3d} K = {27 * cin:5d}   " + "   ".join(row))
