"""Diagnostic (GPU): the three reference trainer steps (tests/golden/model_unet3d_icl_nc2_steps.npz) with the split-product and the
exact-fp32 convolutions, update-inside-backward on and off: per-step |loss - golden| of the six loss terms, relative distance of the
parameter norms after step 3 and of the sampled 13,824^2 update.  Shows how much of the step-3 distance is the gradient noise of
steps 1-2 amplified by the update (lr 0.01, consistency weight 10); the third configuration repeats the exact-fp32 run with every
input volume scaled by (1 + 1e-7) and also prints its distance to the unperturbed run: what rounding-sized noise does to the
trajectory on identical kernels."""
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
g = load_golden("model_unet3d_icl_nc2_steps.npz")
BIG = "sspa.class_decoders.2.mlp2.fc1.weight"
vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(3)]
labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, 2).to(dev) for s in range(3)]
base = None
for split, fuse, factored, scale in (("1", True, True, 1.0), ("0", True, True, 1.0), ("0", True, True, 1.0 + 1e-7), ("1", False, False, 1.0)):
    os.environ["ICL_CONV_SPLIT"] = split
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=float(g["base_lr"]), max_iterations=int(g["max_iterations"]),
                                     update_in_backward=fuse, factored_mlp2_grads=factored))
    named = dict(model.named_parameters())
    w0 = named[BIG].detach()[::432, ::432].double().clone()
    print(f"ICL_CONV_SPLIT={split} update_in_backward={fuse} factored={factored} input scale 1 + {scale - 1.0:.0e}")
    mine = []
    for s in range(3):
        parts = tr.step(vols[s] * scale, labs[s])
        got = np.array([float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con", "loss")])
        mine.append(got)
        print(f"  step {s + 1} |loss - golden|", " ".join(f"{v:.1e}" for v in np.abs(got - g['losses'][s])),
              ("   |loss - unperturbed run| " + " ".join(f"{v:.1e}" for v in np.abs(got - base[s]))) if scale != 1.0 else "")
    if split == "0" and scale == 1.0:
        base = mine
    post = np.array([float(p.detach().double().norm()) for p in named.values()])
    print("  post-step-3 norms: max rel diff", float(np.max(np.abs(post - g["post_step3_norms"]) / g["post_step3_norms"])),
          " delta(mlp2 sample)", rel_err((named[BIG].detach()[::432, ::432].double() - w0).cpu().numpy(), g["delta_step3." + BIG + "_sub"]),
          " final.weight", rel_err(named["final.weight"].detach().cpu(), g["post_step3.final.weight"]))
    del tr, model
    torch.cuda.empty_cache()
