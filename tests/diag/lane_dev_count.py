"""How often the known lane deviation shows: one single-stream step, then N steps with three side-stream lanes; prints the
number of lane steps whose level-1 LayerNorm bias gradients differ from the single-stream ones.  Usage: lane_dev_count.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init, _parity_mode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
KEYS = ("uscl.norm_layers.1.bias", "uscl.class_decoders.1.norm1.bias")
res = []
for side, lanes in [(False, 0)] + [(True, 3)] * n:
    ops.SideStream.enabled, ops.SideStream.lanes = side, lanes
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    res.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    del tr, model
    torch.cuda.empty_cache()
bad, other = 0, 0
detail = []
for g in res[1:]:
    hit = False
    for k in g:
        d = (g[k] - res[0][k]).abs()
        if float(d.max()) > 0:
            if k in KEYS:
                hit = True
                detail.append(f"{k.split('.')[1]}{'.cd' if 'class' in k else ''}:{int((d > 0).sum())}el/{float(d.max() / res[0][k].abs().max()):.3f}")
            else:
                other += 1
    bad += hit
print(f"LANEDEV env={os.environ.get('ICL_LN_DBG', '-')}/{os.environ.get('AMD_OPT_FLUSH', '-')}/{os.environ.get('TAG', '-')}: {bad} of {n} lane steps deviate; other keys differing: {other}; {' '.join(detail)}")
