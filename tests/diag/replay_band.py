"""Diagnostic (GPU): how far do two runs of the same four ICL steps drift apart?  Prints the loss of steps 3 and 4 for two EAGER
runs and one run with two replayed steps (parity mode: dropout / drop-path off), plus the relative difference of two parameters.
Sources of run-to-run differences left in the step: fp32 atomics of the LayerNorm gamma/beta gradients (csrc/kernels/token.h);
the Cin-split convolutions and every weight gradient use fixed-order slab sums."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd import ops  # noqa: E402
from icl_amd.networks.aligner import DropPath  # noqa: E402
from icl_amd.networks.layers import Dropout3  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402

dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
runs = []
for mode in ("eager", "eager", "graph"):
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10))
    if mode == "graph":
        tr.capture(vol, lab, warmup=2)
        losses = [float(tr.step(vol, lab)["loss"]) for _ in range(2)]
    else:
        losses = [float(tr.step(vol, lab)["loss"]) for _ in range(4)][2:]
    runs.append((mode, losses, model.final.weight.detach().clone(), model.sspa.class_decoders[2].mlp2.fc1.bias.detach().clone()))
    del tr, model
    torch.cuda.empty_cache()
for mode, losses, _, _ in runs:
    print(mode, ["%.9f" % v for v in losses])
for a, b in ((0, 1), (0, 2)):
    la, lb = runs[a][1], runs[b][1]
    print(f"{runs[a][0]} vs {runs[b][0]}: loss rel diff", [abs(x - y) / abs(x) for x, y in zip(la, lb)],
          "final.weight", float((runs[a][2] - runs[b][2]).abs().max() / runs[a][2].abs().max()),
          "mlp2.fc1.bias", float((runs[a][3] - runs[b][3]).abs().max() / runs[a][3].abs().max()))
