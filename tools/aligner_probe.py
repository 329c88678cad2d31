#!/usr/bin/env python3
"""Wall time of the two aligners alone (sspa on the labeled pair + uscl guided, forward + backward on fixed decoder features, 3D U-Net
ICL, nc = 2) as ONE pair of replayed hipGraphs, for the stream layouts of ops.SideStream:
    ICL_ALIGNER_STREAM=0 | ICL_ALIGNER_LANES=0 | default (three lanes)      python3 tools/aligner_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init  # noqa: E402

dev = torch.device("cuda", 0)
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
fill_like_reference_init(list(model.named_parameters()))
model.train()
feats = [torch.randn(2, c, r, r, r, device=dev, requires_grad=True) for c, r in ((256, 6), (128, 12), (64, 24))]


from icl_amd.networks.layers import BatchNormAct  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402

tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1))
tr.lr_dev = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
tr.optimizer.lr_dev = tr.lr_dev
ops.StepRNG.enable(dev)


def forward():
    # the step-scoped machinery of ICLTrainer._forward_backward (factored mlp2 gradients, SGD step inside the backward pass, deferred
    # bias gradients), without the backbone
    ops.StepRNG.begin_step()
    tr.packed.begin_step()
    tr.optimizer.zero_grad(set_to_none=True)
    for f in feats:
        f.grad = None
    BatchNormAct.defer_counters()
    ops.DeferredBiasGrads.begin()
    ops.FactoredGrads.world = 1
    ops.FactoredGrads.fused_optimizer = tr.optimizer
    ops.FactoredGrads.uses = {}
    ctx = ops.FactoredGrads(True)
    ctx.__enter__()
    with ops.SideStream(feats) as side:
        (maps_lab, qs_lab), (maps_con, _) = model.sspa.forward_labeled_pair(feats, 1)
        maps_un, _ = model.uscl([f[1:] for f in feats], qs_lab, "unlabeled")
    side.join(maps_lab + maps_un + maps_con)
    return ctx, sum((m * m).mean() for m in maps_lab + maps_un + maps_con)


def backward(ctx, loss):
    loss.backward()
    ctx.__exit__(None, None, None)
    ops.FactoredGrads.fused_optimizer = None
    ops.FactoredGrads.uses = None
    BatchNormAct.flush_counters()
    ops.DeferredBiasGrads.flush()
    tr.optimizer.step()
    tr.packed.end_step()
    ops.StepRNG.end_step()


s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    for _ in range(3):
        backward(*forward())
torch.cuda.current_stream(dev).wait_stream(s)
torch.cuda.synchronize()
pool = torch.cuda.graph_pool_handle()
gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gf, pool=pool):
    ctx, loss = forward()
with torch.cuda.graph(gb, pool=pool):
    backward(ctx, loss)
for _ in range(3):
    gf.replay(); gb.replay()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 20
tf = tb = 0.0
for _ in range(n):
    e[0].record(); gf.replay(); e[1].record(); gb.replay(); e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print(f"ICL_ALIGNER_STREAM={os.environ.get('ICL_ALIGNER_STREAM', '1')} ICL_ALIGNER_LANES={os.environ.get('ICL_ALIGNER_LANES', '3')}: "
      f"aligners alone forward {tf / n:.3f} ms, backward {tb / n:.3f} ms, both {(tf + tb) / n:.3f} ms")
