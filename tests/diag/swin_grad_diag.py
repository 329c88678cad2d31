"""Per-parameter gradient comparison of the HIP SwinUNETR-ICL step against the CPU oracle at 96^3 (debugging aid, GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd.networks.swinunetr_icl import SwinUNETR_icl  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402
from oracle import icl_oracle as O, swin_oracle as S  # noqa: E402

torch.set_num_threads(min(os.cpu_count(), 32))
dev = torch.device("cuda")
nc = 2
p = S.make_swin_params(nc, requires_grad=True)
m = SwinUNETR_icl((96, 96, 96), 1, nc, feature_size=48, device=dev)
for mod in m.modules():
    if isinstance(mod, DropPath):
        mod.drop_prob = 0.0
with torch.no_grad():
    for k, t in m.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            t.copy_(p[k])
vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
lab = synthetic_labels((1, 96, 96, 96), 4242, nc)
t0 = time.time()
outs = S.swinunetr_icl_forward(p, vol[:1], vol[1:], training=True)
total, _ = O.icl_losses(outs, lab, nc)
total.backward(retain_graph=True)
print("oracle step", time.time() - t0, flush=True)
tr = ICLTrainer(m, ICLConfig(num_classes=nc, labeled_bs=1))
m.train()
o2 = m(vol[:1].to(dev), vol[1:].to(dev))
loss, _ = tr.compute_loss(o2, lab.to(dev))
def flat(o):
    return [o[0], o[1]] + list(o[2]) + list(o[3]) + list(o[4])


names = ["logits_lab", "logits_unlab"] + [f"{n}{i}" for n in ("maps_lab", "maps_unlab", "maps_con") for i in range(3)]
_, parts_o = O.icl_losses(outs, lab, nc)
_, parts_g = tr.compute_loss(o2, lab.to(dev))
for term in ("dice", "ce", "aux", "pse", "con"):
    go = torch.autograd.grad(parts_o[term], flat(outs), retain_graph=True, allow_unused=True)
    gg = torch.autograd.grad(parts_g[term], flat(o2), retain_graph=True, allow_unused=True)
    for n, a, b in zip(names, gg, go):
        if b is None:
            continue
        a, b = a.cpu().double(), b.double()
        print("loss-term %-5s d/d%-12s l2 rel %.3e  (|g| %.3e)" % (term, n, float((a - b).norm() / b.norm().clamp_min(1e-30)), float(b.norm())), flush=True)
for n, a, b in zip(names, flat(o2), flat(outs)):
    a, b = a.detach().cpu().double(), b.detach().double()
    print("forward %-12s l2 rel %.3e" % (n, float((a - b).norm() / b.norm())))
loss.backward()
print("loss", float(loss), float(total))
# float64 run of the same oracle = the reference point for both fp32 implementations
t0 = time.time()
p64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in p.items()}
outs64 = S.swinunetr_icl_forward(p64, vol[:1].double(), vol[1:].double(), training=True)
total64, _ = O.icl_losses(outs64, lab, nc)
total64.backward()
print("oracle f64 step", time.time() - t0, float(total64), flush=True)
rows = []
for k, t in m.named_parameters():
    r = p[k].grad
    if r is None or t.grad is None:
        continue
    a, b, c = t.grad.cpu().double(), r.double(), p64[k].grad
    n = c.norm().clamp_min(1e-30)
    rows.append((float((a - c).norm() / n), float((b - c).norm() / n), float((a - b).norm() / n), k))
rows.sort(reverse=True)
print("   hip-vs-f64   cpu32-vs-f64   hip-vs-cpu32")
for r in rows:
    print("%.3e  %.3e  %.3e  %s" % r)
