#!/usr/bin/env python3
"""Which Python lines of one eager ICL step still launch ATen kernels (fills, adds, copies)?  torch.profiler with stacks; every ATen op that
launched a device kernel is attributed to the innermost frame inside icl_amd/ (or 'autograd engine' when no such frame exists).
    python tools/aten_sites.py [unet_3D_icl|swinunetr_icl]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1337)
name = sys.argv[1] if len(sys.argv) > 1 else "unet_3D_icl"
if name == "swinunetr_icl":
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device=dev)
else:
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=0.01, w_pse=1.0), None)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2, device=dev)
for _ in range(3):
    tr.step(vol, lab)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(vol, lab)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not getattr(ev, "kernels", None):
        continue
    if not getattr(ev, "kernels", None):
        continue
    frame = "autograd engine / no icl_amd frame"
    for fr in ev.stack or []:
        if "icl_amd/" in fr:
            frame = fr.split("icl_amd/")[-1]
            break
    if frame.startswith("autograd"):      # name the autograd node that issued it
        par = ev.cpu_parent
        while par is not None and par.cpu_parent is not None and not par.name.endswith(("Backward0", "Backward1", "Backward")) and "AccumulateGrad" not in par.name:
            par = par.cpu_parent
        if par is not None:
            frame = f"autograd node: {par.name}"
    sites[(ev.name, frame, len(ev.kernels))] += 1
tot = 0
for (op, frame, nk), n in sorted(sites.items(), key=lambda kv: -kv[1] * kv[0][2]):
    print(f"{n * nk:4d} kernels  {op:28s} {frame}")
    tot += n * nk
print("total ATen kernels in the step:", tot)
