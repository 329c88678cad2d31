"""Which of the two launches of the LayerNorm backward is wrong when the bias gradient of a level-1 LayerNorm differs between runs:
reference column sums of dY and of the kernel's chunk partials, queued on the same stream right behind the kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops, _lib
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
    x, weight, mean, rstd = ctx.saved_tensors
    c = x.shape[-1]
    L = _lib.lib()
    gyc = gy.contiguous()
    rows = x.numel() // c
    gx = torch.empty_like(x)
    ws = ops._ws(L.icl_layernorm_bwd_ws_bytes(rows, c), x)
    dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    _lib.check(L.icl_layernorm_bwd(ops._ptr(gyc), ops._ptr(x), ops._ptr(weight), ops._ptr(mean), ops._ptr(rstd), ops._ptr(gx), ops._ptr(dgb[0]),
                                   ops._ptr(dgb[1]), ops._ptr(ws), rows, c, ops._stream(x)), "layernorm_bwd")
    part = ws[:(ws.numel() // (2 * c)) * 2 * c].view(2, -1, c)
    if c == 128:
        log.append((rows, dgb[1].clone(), part[1].double().sum(0), gyc.reshape(-1, c).double().sum(0), str(torch.cuda.current_stream(x.device))))
    return gx, dgb[0], dgb[1], None

ops._LayerNorm.backward = staticmethod(patched)
ops.DeferredBiasGrads.begin = classmethod(lambda cls: None)
for rep in range(4):
    ops.SideStream.enabled, ops.SideStream.lanes = True, 3
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    log.clear()
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    for rows, db, chk, ref, st in log:
        e1 = float((db.double() - chk).abs().max() / ref.abs().max())
        e2 = float((chk - ref).abs().max() / ref.abs().max())
        flag = "  <-- " if max(e1, e2) > 1e-5 else ""
        print(f"rep {rep} rows {rows} {st[-24:]}: |kernel colsum - sum of partials| {e1:.2e}   |sum of partials - sum of dY| {e2:.2e}{flag}")
    del tr, model
    torch.cuda.empty_cache()
