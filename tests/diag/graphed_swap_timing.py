"""Diagnostic (GPU): where an iteration of the unchanged loop with FusedSGD(graph=True) spends its time (host clock per section,
GPU idle visible as the difference between the host-side total and the sum of the GPU-side sections)."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "compat"))
sys.path.insert(0, ROOT)
from networks.net_factory_3d import net_factory_3d  # noqa: E402
from utils import losses  # noqa: E402
from icl_amd.optim import FusedSGD  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402
from torch.nn.modules.loss import CrossEntropyLoss  # noqa: E402

graph = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
model = net_factory_3d(net_type="unet_3D_icl", in_chns=1, class_num=2)
dev = next(model.parameters()).device
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((2, 96, 96, 96), 4242, 2).to(dev)
opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, graph=graph)
ce, dice, aux, pse = CrossEntropyLoss(), losses.DiceLoss(2), losses.AuxLoss3D(2), losses.PseudoSoftLoss3D(2)
names = ["forward", "losses", "zero_grad", "backward", "step", "items"]
acc = {k: [0.0, 0.0] for k in names}
N = 20
for it in range(8 + N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    hs = [time.perf_counter()]
    ev[0].record()
    o = model(vol[:1], vol[1:])
    ev[1].record(); hs.append(time.perf_counter())
    l_ce = ce(o[0], lab[:1]); l_d = dice(torch.softmax(o[0], 1), lab[:1].unsqueeze(1)); l_a = aux(o[2], lab[:1]); l_p = pse(o[3], o[1])
    l_c = losses.softmax_mse_loss(o[3], o[4]); loss = l_d + l_ce + l_a + l_p + 10 * l_c
    ev[2].record(); hs.append(time.perf_counter())
    opt.zero_grad()
    ev[3].record(); hs.append(time.perf_counter())
    loss.backward()
    ev[4].record(); hs.append(time.perf_counter())
    opt.step()
    ev[5].record(); hs.append(time.perf_counter())
    vals = (loss.item(), l_ce.item(), l_d.item(), l_a.item(), l_p.item(), l_c.item())
    ev[6].record(); hs.append(time.perf_counter())
    torch.cuda.synchronize()
    if it >= 8:
        for i, k in enumerate(names):
            acc[k][0] += (hs[i + 1] - hs[i]) * 1e3 / N
            acc[k][1] += ev[i].elapsed_time(ev[i + 1]) / N
print("graph" if graph else "eager swap", "graphed:", opt._graph_state is not None)
for k in names:
    print(f"  {k:10s} host {acc[k][0]:7.3f} ms   gpu-stream {acc[k][1]:7.3f} ms")
print("  total host", round(sum(v[0] for v in acc.values()), 3), " total gpu-stream", round(sum(v[1] for v in acc.values()), 3))
