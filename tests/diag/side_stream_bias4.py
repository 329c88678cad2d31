"""dY of the level-1 LayerNorm layers and the bias gradients that come out, single stream against lanes.  (The `DeferredBiasGrads.dbg` hook it
reads was a temporary probe in ops.py and is gone; without it the script prints the gradient comparison only.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init, _parity_mode
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
log = []
orig = ops._LayerNorm.backward

def patched(ctx, gy):
    if gy.shape[-1] == 128 and gy.numel() // 128 == 1728:
        log.append(gy.detach().clone())
    return orig(ctx, gy)

ops._LayerNorm.backward = staticmethod(patched)
res = []
for side, lanes in ((False, 0), (True, 3), (True, 3), (True, 3)):
    ops.SideStream.enabled, ops.SideStream.lanes = side, lanes
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    log.clear()
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in model.named_parameters() if "norm_layers.1" in k or "class_decoders.1.norm1" in k}
    if getattr(ops.DeferredBiasGrads, "dbg", None):
        flat, copy, sums, params, outs = ops.DeferredBiasGrads.dbg
        names = {id(p): k for k, p in model.named_parameters()}
        print(f"  [{side},{lanes}] flat after the step vs its copy taken right behind the column sums: {int((flat != copy).sum())} elements differ")
        for p_, o, sm in zip(params, outs, sums):
            e = float((o.double() - sm).abs().max())
            if e > 1e-6 * float(sm.abs().max() + 1e-30) and "norm" in names.get(id(p_), ""):
                print(f"      {names.get(id(p_))}: column-sum kernel output vs torch sum of the same partials (both at flush): {e:.3e}")
        ops.DeferredBiasGrads.dbg = None
    res.append(([t.clone() for t in log], g))
    del tr, model
    torch.cuda.empty_cache()
base_gy, base_g = res[0]
for i, (gys, g) in enumerate(res[1:], 1):
    print(f"run {i}: dY tensors {len(gys)}; max |dY - dY_single_stream|:", [float((a - b).abs().max()) for a, b in zip(gys, base_gy)])
    for k in sorted(g):
        d = (g[k] - base_g[k]).abs()
        print(f"    {k}: max diff {float(d.max()):.3e} of max {float(base_g[k].abs().max()):.3e}; differing elements {int((d > 0).sum())}")
    for j, t in enumerate(gys):
        s = t.reshape(-1, 128).double().sum(0)
        for k in sorted(g):
            if k.endswith("bias") and k.startswith("uscl") and "query" not in k:
                print(f"      sum of dY[{j}] vs {k}: this run {float((s - g[k].double()).abs().max()):.3e}, single-stream value {float((s - base_g[k].double()).abs().max()):.3e}")
