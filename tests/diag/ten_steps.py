"""Diagnostic (GPU): the TEN reference trainer steps (tests/golden/model_unet3d_icl_nc2_steps10.npz, round 6) with the split-product and
the exact-fp32 convolutions, and the exact-fp32 run repeated with every input volume scaled by (1 + 1e-7): per step |loss - golden| of
the six loss terms, the largest relative distance of a parameter norm, the distance of final.weight, of its momentum buffer and of the
sampled 13,824^2 update / momentum (max-norm and RMS) — for the perturbed run also against the unperturbed one.  The bands of
tests/test_gpu_parity.py::test_ten_trainer_steps_match_reference_golden are set from this output (profiles/r6_ten_steps_noise.txt)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err, rms_err  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.networks.layers import Dropout3  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)
g = load_golden("model_unet3d_icl_nc2_steps10.npz")
BIG = "sspa.class_decoders.2.mlp2.fc1.weight"
STEPS = len(g["losses"])
vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(STEPS)]
labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, 2).to(dev) for s in range(STEPS)]
base = None
for split, scale in (("1", 1.0), ("0", 1.0), ("0", 1.0 + 1e-7)):
    os.environ["ICL_CONV_SPLIT"] = split
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=float(g["base_lr"]), max_iterations=int(g["max_iterations"])))
    named = dict(model.named_parameters())
    w0 = named[BIG].detach()[::432, ::432].double().clone()
    print(f"ICL_CONV_SPLIT={split} input scale 1 + {scale - 1.0:.0e}")
    mine = []
    for s in range(STEPS):
        lr = tr.optimizer.param_groups[0]["lr"]
        parts = tr.step(vols[s] * scale, labs[s])
        got = np.array([float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con", "loss")])
        t = s + 1
        post = np.array([float(p.detach().double().norm()) for p in named.values()])
        mom = tr.optimizer.state[named[BIG]]["momentum_buffer"][::432, ::432].cpu()
        delta = (named[BIG].detach()[::432, ::432].double() - w0).cpu().numpy()
        rec = dict(loss=got, post=post, fw=named["final.weight"].detach().cpu().numpy().copy(), mom=mom.numpy().copy(), delta=delta,
                   fmom=tr.optimizer.state[named["final.weight"]]["momentum_buffer"].cpu().numpy().copy())
        mine.append(rec)
        line = (f"  step {t:2d} lr {lr:.6f} (golden {float(g['lr_used'][s]):.6f}) |loss - golden| " + " ".join(f"{v:.1e}" for v in np.abs(got - g['losses'][s]))
                + f"  norms {float(np.max(np.abs(post - g[f'post_step{t}_norms']) / g[f'post_step{t}_norms'])):.1e}"
                + f"  final.w {rel_err(rec['fw'], g[f'post_step{t}.final.weight']):.1e}  mom(final.w) {rel_err(rec['fmom'], g[f'momentum_step{t}.final.weight']):.1e}"
                + f"  delta(mlp2) max {rel_err(delta, g[f'delta_step{t}.{BIG}_sub']):.1e} rms {rms_err(delta, g[f'delta_step{t}.{BIG}_sub']):.1e}"
                + f"  mom(mlp2) max {rel_err(rec['mom'], g[f'momentum_step{t}.{BIG}_sub']):.1e} rms {rms_err(rec['mom'], g[f'momentum_step{t}.{BIG}_sub']):.1e}")
        print(line)
        if scale != 1.0:
            b = base[s]
            print("          vs the unperturbed run: |loss| " + " ".join(f"{v:.1e}" for v in np.abs(got - b["loss"]))
                  + f"  norms {float(np.max(np.abs(post - b['post']) / b['post'])):.1e}  final.w {rel_err(rec['fw'], b['fw']):.1e}"
                  + f"  mom(final.w) {rel_err(rec['fmom'], b['fmom']):.1e}  delta(mlp2) max {rel_err(delta, b['delta']):.1e} rms {rms_err(delta, b['delta']):.1e}"
                  + f"  mom(mlp2) max {rel_err(rec['mom'], b['mom']):.1e} rms {rms_err(rec['mom'], b['mom']):.1e}")
    if split == "0" and scale == 1.0:
        base = mine
    del tr, model
    torch.cuda.empty_cache()
