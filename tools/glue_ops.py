#!/usr/bin/env python3
"""Which Python lines issue the small torch kernels (add / fill / copy / sum) of one eager trainer step?
Prints (aten op, first icl_amd source frame) -> launch count, from a torch.profiler trace with stacks."""
import collections, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume

which = sys.argv[1] if len(sys.argv) > 1 else "unet_3D_icl"
dev = torch.device("cuda", 0)
if which == "unet_3D_icl":
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
else:
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device=dev)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1))
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2, device=dev)
for _ in range(2):
    tr.step(vol, lab)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(vol, lab)
    torch.cuda.synchronize()

WATCH = ("aten::add", "aten::add_", "aten::sum", "aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone",
         "aten::contiguous", "aten::mul", "aten::mean", "aten::zeros_like", "aten::full", "aten::ones", "aten::new_zeros",
         "aten::uniform_", "aten::rand", "aten::rand_like", "aten::randn_like", "aten::normal_", "aten::cat", "aten::stack",
         "aten::sub", "aten::div", "aten::neg", "aten::index_select", "aten::to", "aten::_to_copy")
counts = collections.Counter()
times = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if ev.name not in WATCH or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    # only ops that actually launch something
    def n_kernels(e):
        return len(e.kernels) + sum(n_kernels(c) for c in e.cpu_children)
    p = ev.cpu_parent
    if p is not None and p.name in WATCH:
        continue   # counted at the outermost watched op
    nk = n_kernels(ev)
    if nk == 0:
        continue
    frame = "<autograd engine>"
    e2 = ev
    while e2 is not None and frame == "<autograd engine>":
        for s in e2.stack or []:
            if "icl_amd" in s and "site-packages" not in s:
                frame = s.replace(root + "/", "")
                break
        if frame == "<autograd engine>" and e2.cpu_parent is not None and e2.cpu_parent.name not in WATCH and not (e2.stack):
            frame = "<in " + e2.cpu_parent.name + ">"
        e2 = e2.cpu_parent
    shape = ""
    counts[(ev.name, frame)] += nk
    def dev_time(e):
        return sum(k.duration for k in e.kernels) + sum(dev_time(c) for c in e.cpu_children)
    times[(ev.name, frame)] += dev_time(ev)
tot = 0
for (name, frame), n in counts.most_common(70):
    tot += n
    print(f"{n:4d}  {times[(name, frame)]:9.1f} us  {name:18s} {frame}")
print("sample stack:", next((e.stack for e in prof.events() if e.stack), None))
print("total launches from watched ops:", sum(counts.values()))
