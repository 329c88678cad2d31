"""Print every (rows, in, out) the Linear products of one ICL step see, with launch counts — `python tools/linear_shapes.py
[unet_3D_icl|swinunetr_icl] [num_classes]` (GPU)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "unet_3D_icl"
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda")
if name == "swinunetr_icl":
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
else:
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
model.train()
seen = collections.Counter()
for fn in ("linear_forward_raw", "linear_dgrad_raw"):
    orig = getattr(ops, fn)

    def wrap(a, w, *rest, _orig=orig, _fn=fn, **kw):
        seen[(_fn, a.shape[0], w.shape[1], w.shape[0])] += 1
        return _orig(a, w, *rest, **kw)
    setattr(ops, fn, wrap)
tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1, max_iterations=10, update_in_backward=False))
tr.step(synthetic_volume((2, 1, 96, 96, 96), 1).to(dev), synthetic_labels((1, 96, 96, 96), 2, nc).to(dev))
for (fn, rows, i, o), n in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] - kv[0][2] * kv[0][3]):
    print(f"{fn:20s} rows {rows:7d}  in {i:6d}  out {o:6d}  x{n}")
