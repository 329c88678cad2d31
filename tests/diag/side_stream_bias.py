"""Diagnosis of test_aligner_side_stream_equals_single_stream failing in a fresh process: per configuration, the relative error of the
LayerNorm parameter gradients of the aligners against the first single-stream run."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import rel_err
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init, _parity_mode
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
runs = []
cfgs = [(False, 0), (False, 0), (True, 0), (True, 3), (True, 3)]
if len(sys.argv) > 1 and sys.argv[1] == "nodefer":
    ops.DeferredBiasGrads.begin = classmethod(lambda cls: None)
for side, lanes in cfgs:
    ops.SideStream.enabled, ops.SideStream.lanes = side, lanes
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    runs.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    del tr, model
    torch.cuda.empty_cache()
base = runs[0]
for (side, lanes), g in zip(cfgs[1:], runs[1:]):
    bad = sorted(((rel_err(g[k].cpu(), base[k].cpu()), k) for k in base), reverse=True)[:4]
    print(f"side={side} lanes={lanes}:", [(round(e, 6), k) for e, k in bad])
    for e, k in bad:
        if e > 0:
            d = (g[k] - base[k]).cpu()
            idx = d.nonzero().flatten().tolist()
            print("   ", k, "shape", tuple(g[k].shape), "differing elements", len(idx), "idx", idx[:12], "...", idx[-4:])
            print("      base", base[k].cpu()[idx[:6]].tolist(), "got", g[k].cpu()[idx[:6]].tolist())
