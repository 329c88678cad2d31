"""Where does the gradient of sspa scale 1 go wrong?  Captures the intermediates (and their gradients) of
sspa.attn_convs0[1] on the first (labeled) call in both the HIP model and the CPU oracle."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd.networks.swinunetr_icl import SwinUNETR_icl  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402
from oracle import icl_oracle as O, swin_oracle as S  # noqa: E402

torch.set_num_threads(min(os.cpu_count(), 32))
dev = torch.device("cuda")
nc = 2
SCALE = int(os.environ.get("SCALE", "1"))
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

cap_o, cap_g = {}, {}


def keep(store, name, t):
    if name in store:
        return
    store[name] = [t.detach().cpu(), None]
    t.register_hook(lambda g, n=name: store[n].__setitem__(1, g.detach().cpu()))


orig = O.separable_conv3d


def sep(pp, pre, x, training):
    if pre != f"sspa.attn_convs0.{SCALE}" or "x" in cap_o:
        return orig(pp, pre, x, training)
    c = x.shape[1]
    keep(cap_o, "x", x)
    y = O._convnd(x, pp[f"{pre}.block.depthwise.weight"], None, padding=1, groups=c)
    keep(cap_o, "dw", y)
    y = F.relu(O._bn_train(pp, f"{pre}.block.bn_depth", y, training))
    keep(cap_o, "bn1", y)
    y = O._convnd(y, pp[f"{pre}.block.pointwise.weight"], None)
    keep(cap_o, "pw", y)
    y = F.relu(O._bn_train(pp, f"{pre}.block.bn_point", y, training))
    keep(cap_o, "bn2", y)
    return y


O.separable_conv3d = sep
outs = S.swinunetr_icl_forward(p, vol[:1], vol[1:], training=True)
total, _ = O.icl_losses(outs, lab, nc)
total.backward()

blk = m.sspa.attn_convs0[SCALE].block
blk.register_forward_pre_hook(lambda mod, inp: keep(cap_g, "x", inp[0]))
for name, sub in (("dw", blk.depthwise), ("bn1", blk.bn_depth), ("pw", blk.pointwise), ("bn2", blk.bn_point)):
    sub.register_forward_hook(lambda mod, inp, out, n=name: keep(cap_g, n, out))
tr = ICLTrainer(m, ICLConfig(num_classes=nc, labeled_bs=1))
m.train()
o2 = m(vol[:1].to(dev), vol[1:].to(dev))
loss, _ = tr.compute_loss(o2, lab.to(dev))
loss.backward()


def l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


for k in ("x", "dw", "bn1", "pw", "bn2"):
    a, b = cap_g[k], cap_o[k]
    print(k, tuple(a[0].shape), "fwd %.2e" % l2(a[0], b[0]), "grad %.2e" % l2(a[1], b[1]),
          "|grad| %.3e" % float(b[1].norm()), "frac(|x|<1e-6) %.2e" % float((b[0].abs() < 1e-6).float().mean()), flush=True)
x, g = cap_o["dw"]
print("dw stats per channel mean", x.mean(dim=(0, 2, 3, 4))[:4], "std", x.std(dim=(0, 2, 3, 4))[:4])
from icl_amd import ops  # noqa: E402
pre = f"sspa.attn_convs0.{SCALE}.block.bn_depth"
ga, be = p[pre + ".weight"].detach(), p[pre + ".bias"].detach()
print("gamma", ga[:4], "beta", be[:4])
xd, gyd = cap_o["dw"][0], cap_o["bn1"][1]
xg = xd.to(dev).requires_grad_()
h = xd.shape[1]
y = ops.batch_norm_relu(xg, ga.to(dev), be.to(dev), torch.zeros(h, device=dev), torch.ones(h, device=dev), True)
y.backward(gyd.to(dev))
print("isolated HIP BN bwd vs oracle-captured grad: %.2e   fwd %.2e" % (l2(xg.grad.cpu(), cap_o["dw"][1]), l2(y.detach().cpu(), cap_o["bn1"][0])))
x64 = xd.double().requires_grad_()
y64 = F.relu(F.batch_norm(x64, None, None, ga.double(), be.double(), True, 0.1, 1e-5))
y64.backward(gyd.double())
print("f64 torch vs oracle-captured: %.2e ; HIP vs f64: %.2e" % (l2(cap_o["dw"][1], x64.grad), l2(xg.grad.cpu(), x64.grad)))
# second: feed the HIP-captured tensors
xg2 = cap_g["dw"][0].to(dev).requires_grad_()
y2 = ops.batch_norm_relu(xg2, ga.to(dev), be.to(dev), torch.zeros(h, device=dev), torch.ones(h, device=dev), True)
y2.backward(cap_g["bn1"][1].to(dev))
print("isolated HIP BN bwd on HIP-captured inputs vs HIP-captured grad: %.2e" % l2(xg2.grad.cpu(), cap_g["dw"][1]))
print("mask agreement oracle-vs-hip fwd: ", float(((cap_o["bn1"][0] > 0) != (cap_g["bn1"][0] > 0)).float().mean()))
