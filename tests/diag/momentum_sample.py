"""How stable is the three-step momentum sample of test_three_trainer_steps_match_reference_golden under changes that only move rounding?
Runs the three reference steps (eager) and prints, for the 32 x 32 sample of sspa.class_decoders.2.mlp2.fc1's momentum against the golden:
the max-norm relative error the test asserts on, the RMS relative error, and percentiles.   python tests/diag/momentum_sample.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from conftest import load_golden  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402
from test_gpu_parity import _parity_mode  # noqa: E402

dev = torch.device("cuda", 0)
g = load_golden("model_unet3d_icl_nc2_steps.npz")
vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(3)]
labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, 2).to(dev) for s in range(3)]
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
fill_like_reference_init(list(model.named_parameters()))
_parity_mode(model)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=float(g["base_lr"]), max_iterations=int(g["max_iterations"])))
for s in range(3):
    parts = tr.step(vols[s], labs[s])
big = dict(model.named_parameters())["sspa.class_decoders.2.mlp2.fc1.weight"]
a = tr.optimizer.state[big]["momentum_buffer"][::432, ::432].cpu().double().numpy()
b = np.asarray(g["momentum.sspa.class_decoders.2.mlp2.fc1.weight_sub"], dtype=np.float64)
d = np.abs(a - b)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ICL_"))
print(f"[{tag or 'defaults'}] loss3 {float(parts['loss']):.7f}  max|d|/max|ref| {d.max() / np.abs(b).max():.4f}  rms(d)/rms(ref) {np.sqrt((d ** 2).mean() / (b ** 2).mean()):.4f}  "
      f"median|d|/max|ref| {np.median(d) / np.abs(b).max():.4f}  p99 {np.percentile(d, 99) / np.abs(b).max():.4f}")
