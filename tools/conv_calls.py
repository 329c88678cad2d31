#!/usr/bin/env python3
"""List every conv launch of one ICL step with its shape-derived FLOPs/bytes and HIP-event time (eager mode)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
dev = torch.device("cuda", 0)
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2 and sys.argv[2] == "swinunetr_icl":
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
else:
    model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1))
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev); lab = synthetic_labels((1, 96, 96, 96), 4242, nc, device=dev)
for _ in range(3): tr.step(vol, lab)
with ops.KernelTimer() as kt:
    tr.step(vol, lab)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, e0, e1, fl, by in kt.records:
    key = (name, round(fl / 1e6), round(by / 1e6, 2))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0
for (name, mf, mb), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f"{us:9.1f} us  x{n:<3d} {mf:9d} MFLOP {mb:9.2f} MB  {mf * n / us:7.1f} TF  {name}")
print("total conv us", round(tot))
