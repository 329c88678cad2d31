"""How often the known lane deviation shows: one single-stream step, then N steps with three side-stream lanes; prints the
number of lane steps whose level-1 LayerNorm bias gradients differ from the single-stream ones, and how many OTHER gradients differ (never
one so far).  Usage: lane_dev_count.py [N] [unet|unet16|swin]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init, _parity_mode
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "unet"
nc = 16 if which == "unet16" else 2
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, nc).to(dev)
KEYS = ("uscl.norm_layers.1.bias", "uscl.class_decoders.1.norm1.bias")
res = []
LANES = int(os.environ.get("LANES", "3"))
for side, lanes in [(False, 0)] + [(True, LANES)] * n:
    ops.SideStream.enabled, ops.SideStream.lanes = side, lanes
    ops.StepRNG.tensor = None
    if which == "swin":
        from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
        torch.manual_seed(20241003)
        model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
    else:
        model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1, max_iterations=10, update_in_backward=False))
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
print(f"LANEDEV {which} lanes={LANES} wgrad_lane={os.environ.get('ICL_WGRAD_LANE', '-')} lib={'packed' if 'packed' in os.environ.get('ICL_HIP_LIB', '') else 'shipped'} env={os.environ.get('ICL_LN_DBG', '-')}/{os.environ.get('AMD_OPT_FLUSH', '-')}/{os.environ.get('TAG', '-')}: {bad} of {n} lane steps deviate; other keys differing: {other}; {' '.join(detail)}")
