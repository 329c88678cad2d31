"""How far do two single-rank runs of three ICL steps differ — update inside the backward pass against update in FusedSGD.step(),
and a run against its own repetition?  Prints, per step, the largest relative difference of the four 13,824^2 momentum buffers."""
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


def run(fuse, side):
    ops.StepRNG.tensor = None
    ops.SideStream.enabled = side
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=fuse))
    snaps = []
    for _ in range(3):
        loss = float(tr.step(vol, lab)["loss"])
        big = {k: p for k, p in model.named_parameters() if p.numel() >= 1 << 26}
        snaps.append((loss, {k: tr.optimizer.state[p]["momentum_buffer"].clone() for k, p in big.items()}))
    return snaps


def diff(a, b):
    for s, ((la, ma), (lb, mb)) in enumerate(zip(a, b)):
        print(f"  step {s}: loss {la:.7f} / {lb:.7f}   " +
              "  ".join(f"{k.split('.')[0]}.{k.split('.')[2]} {float((ma[k] - mb[k]).abs().max() / mb[k].abs().max()):.1e}" for k in ma), flush=True)


for side in (False, True):
    r0 = run(False, side)
    r1 = run(False, side)
    print(f"side stream {side}: step() against step()")
    diff(r0, r1)
    del r1
    r2 = run(True, side)
    print(f"side stream {side}: update in backward against step()")
    diff(r2, r0)
    del r0, r2
    torch.cuda.empty_cache()
