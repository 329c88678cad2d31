"""Diagnostic (GPU): how sensitive is the sampled 13,824^2 mlp2.fc1 weight gradient of the ICL step (the [::432, ::432] sample that
tests/test_gpu_dropin.py compares with the reference golden) to rounding-level changes upstream?

Prints rel_err (max |a - b| / max |b|) of that sample, of final.weight's gradient and of the losses
  * against the reference golden, for the exact-fp32 convolutions (ICL_CONV_SPLIT=0) and the split-product ones (default);
  * between two exact-fp32 runs whose INPUT VOLUME differs by one part in 1e7 (about one fp32 ulp): what a rounding-sized
    perturbation does to the quantity on identical kernels.
If the last number is of the size of the first two, the distance to the golden is ReLU / max-pool decision noise (a handful of the
14 M activations change side), not an accuracy property of a kernel."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.networks.layers import Dropout3  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)
g = load_golden("model_unet3d_icl_nc2.npz")
BIG = "sspa.class_decoders.2.mlp2.fc1.weight"


def run(split, scale=1.0):
    os.environ["ICL_CONV_SPLIT"] = split
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0
    model.train()
    vol = (synthetic_volume((2, 1, 96, 96, 96), 1337) * scale).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, factored_mlp2_grads=False))
    outs = model(vol[:1], vol[1:])
    loss, parts = tr.compute_loss(outs, lab)
    loss.backward()
    named = dict(model.named_parameters())
    out = {"losses": [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(loss)],
           "big": named[BIG].grad[::432, ::432].cpu().numpy().copy(), "final": named["final.weight"].grad.cpu().numpy().copy(),
           "center": named["center.conv2.0.weight"].grad[::16, ::16].cpu().numpy().copy(),
           "maps0": outs[2][0].detach().cpu().numpy().copy()}
    del tr, model, outs, loss
    torch.cuda.empty_cache()
    return out


exact, split, pert = run("0"), run("1"), run("0", 1.0 + 1e-7)
for name, r in (("exact fp32 convolutions vs golden", exact), ("split-product convolutions vs golden", split)):
    print(f"{name:42s} mlp2 sample {rel_err(r['big'], g['grad.' + BIG + '_sub']):.2e}  final.weight {rel_err(r['final'], g['grad.final.weight']):.2e}  "
          f"center.conv2 {rel_err(r['center'], g['grad.center.conv2.0.weight_sub']):.2e}  maps_lab0 {rel_err(r['maps0'], g['maps_lab0']):.2e}  "
          f"losses {np.abs(np.array(r['losses']) - g['losses']).max():.1e}")
print(f"{'split vs exact':42s} mlp2 sample {rel_err(split['big'], exact['big']):.2e}  final.weight {rel_err(split['final'], exact['final']):.2e}  "
      f"center.conv2 {rel_err(split['center'], exact['center']):.2e}  maps_lab0 {rel_err(split['maps0'], exact['maps0']):.2e}")
print(f"{'exact, input x (1 + 1e-7) vs exact':42s} mlp2 sample {rel_err(pert['big'], exact['big']):.2e}  final.weight {rel_err(pert['final'], exact['final']):.2e}  "
      f"center.conv2 {rel_err(pert['center'], exact['center']):.2e}  maps_lab0 {rel_err(pert['maps0'], exact['maps0']):.2e}")
