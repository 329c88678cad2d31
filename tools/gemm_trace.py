import os, sys, torch
sys.path.insert(0, '/root/repo')
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
dev = torch.device('cuda', 0)
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev); model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=0.01, w_pse=1.0), None)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev); lab = synthetic_labels((1, 96, 96, 96), 4242, 2, device=dev)
tr.step(vol, lab); torch.cuda.synchronize()
print("=== second step", file=sys.stderr)
tr.step(vol, lab); torch.cuda.synchronize()
