"""Which tensors of one ICL step differ between two runs from the same state?  (gradients, updated weights, buffers)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402
from test_gpu_parity import _parity_mode  # noqa: E402

dev = torch.device("cuda")
vol = synthetic_volume((2, 1, 96, 96, 96), 77).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 78, 2).to(dev)
ops.SideStream.enabled = len(sys.argv) > 1 and sys.argv[1] == "side"


def run():
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    tr._forward_backward(vol, lab)
    out = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad " + k] = p.grad.detach().clone()
        for j, (g, x) in enumerate(getattr(p, "_icl_factors", None) or []):
            out[f"factor g{j} " + k] = g.detach().clone()
            out[f"factor x{j} " + k] = x.detach().clone()
    for k, b in model.named_buffers():
        out["buffer " + k] = b.detach().clone()
    return out


a, b = run(), run()
bad = 0
for k in a:
    d = float((a[k].double() - b[k].double()).abs().max())
    if d != 0.0:
        bad += 1
        print(f"{k:70s} max abs diff {d:.2e}  (max abs {float(b[k].abs().max()):.2e})")
print(f"{bad} of {len(a)} tensors differ")
