"""Diagnostic (GPU): FusedSGD(graph=True) capture under different stream topologies, each in its own process."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
CODE = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(r"%s", "compat")); sys.path.insert(0, r"%s")
from networks.net_factory_3d import net_factory_3d
from utils import losses
from icl_amd.optim import FusedSGD
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from torch.nn.modules.loss import CrossEntropyLoss
model = net_factory_3d(net_type="unet_3D_icl", in_chns=1, class_num=2)
dev = next(model.parameters()).device
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev); lab = synthetic_labels((2, 96, 96, 96), 4242, 2).to(dev)
opt = FusedSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, graph=True, graph_warmup=2)
ce, dice, aux, pse = CrossEntropyLoss(), losses.DiceLoss(2), losses.AuxLoss3D(2), losses.PseudoSoftLoss3D(2)
for it in range(5):
    o = model(vol[:1], vol[1:])
    loss = dice(torch.softmax(o[0], 1), lab[:1].unsqueeze(1)) + ce(o[0], lab[:1]) + aux(o[2], lab[:1]) + pse(o[3], o[1]) + 10 * losses.softmax_mse_loss(o[3], o[4])
    opt.zero_grad(); loss.backward(); opt.step()
    print(it, float(loss), "graphed", opt._graph_state is not None, opt._graph_failed, flush=True)
print("DONE", flush=True)
''' % (ROOT, ROOT)
for env in ({}, {"ICL_ALIGNER_STREAM": "0"}, {"ICL_WGRAD_LANE": "0"}, {"ICL_UPDATE_PLACEMENT": "fused"},
            {"ICL_ALIGNER_STREAM": "0", "ICL_WGRAD_LANE": "0", "ICL_UPDATE_PLACEMENT": "fused"}):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    tail = [l for l in r.stdout.splitlines() if l.strip()][-2:]
    print(env, "rc", r.returncode, tail, flush=True)
    if r.returncode:
        print(r.stderr[-2500:])
