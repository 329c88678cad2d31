"""Diagnostic (GPU): 200 trainer steps of the 3-D U-Net ICL from ONE seed with dropout / drop-path off — the split-product path, the
exact-fp32 path (ICL_CONV_SPLIT=0) and the exact-fp32 path again with every input volume scaled by (1 + 1e-7) (what rounding-sized noise
alone does over the same horizon).  Eight synthetic batches are cycled.  Prints the three loss trajectories at steps 1, 2, 5, 10, 20, 50,
100, 150, 200 and the RMS / max-norm distance of the sampled 13,824^2 `mlp2.fc1` momentum between the paths at steps 10 / 50 / 200
(profiles/r6_drift.txt; tests/test_gpu_parity.py::test_split_and_exact_paths_stay_together_over_200_steps asserts the bands)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err, rms_err  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.networks.layers import Dropout3  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)
BIG = "sspa.class_decoders.2.mlp2.fc1.weight"
STEPS = int(os.environ.get("DRIFT_STEPS", "200"))
NB = 8
vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(NB)]
labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, 2).to(dev) for s in range(NB)]
MARK = [s for s in (1, 2, 5, 10, 20, 50, 100, 150, 200) if s <= STEPS]
runs = {}
for tag, split, scale in (("split", "1", 1.0), ("exact", "0", 1.0), ("exact, inputs x (1 + 1e-7)", "0", 1.0 + 1e-7)):
    os.environ["ICL_CONV_SPLIT"] = split
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=0.01, max_iterations=30000))
    named = dict(model.named_parameters())
    losses, moms = [], {}
    for s in range(STEPS):
        parts = tr.step(vols[s % NB] * scale, labs[s % NB])
        losses.append(float(parts["loss"]))
        if s + 1 in (10, 50, 200):
            moms[s + 1] = tr.optimizer.state[named[BIG]]["momentum_buffer"][::432, ::432].cpu().numpy().copy()
    runs[tag] = (np.array(losses), moms)
    del tr, model
    torch.cuda.empty_cache()
print("step   " + "   ".join(f"{t:>28s}" for t in runs))
for s in MARK:
    print(f"{s:4d}   " + "   ".join(f"{runs[t][0][s - 1]:28.6f}" for t in runs))
ex = runs["exact"]
for tag in runs:
    if tag == "exact":
        continue
    d = np.abs(runs[tag][0] - ex[0])
    print(f"{tag} vs exact: max |loss difference| over steps 1-10 {d[:10].max():.2e}, 11-50 {d[10:50].max():.2e}, 51-{STEPS} {d[50:].max():.2e}; relative at step {STEPS}: {d[-1] / abs(ex[0][-1]):.2e}")
    for s, m in runs[tag][1].items():
        print(f"    momentum sample of {BIG} at step {s}: rms {rms_err(m, ex[1][s]):.3e} max {rel_err(m, ex[1][s]):.3e}")
