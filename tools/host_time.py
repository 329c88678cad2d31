#!/usr/bin/env python3
"""Is the step host-bound?  Time the CPU-side issue of K steps without synchronising, then the drain."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
dev = torch.device("cuda", 0)
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev); model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1))
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev); lab = synthetic_labels((1, 96, 96, 96), 4242, 2, device=dev)
for _ in range(3): tr.step(vol, lab)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): tr.step(vol, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3*(t1-t0)/K:.2f} ms/step, total {1e3*(t2-t0)/K:.2f} ms/step, drain after issue {1e3*(t2-t1):.2f} ms")
