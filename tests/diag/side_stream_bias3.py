"""Which chunk-partial row of the LayerNorm bias gradient is different at the end of the step, and what it holds (lanes on, deferred mode)."""
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
    n0 = len(ops.DeferredBiasGrads.pending) if ops.DeferredBiasGrads.pending is not None else -1
    out = orig(ctx, gy)
    p = ops.DeferredBiasGrads.pending
    if p is not None and len(p) == n0 + 2 and gy.shape[-1] == 128:
        part_b = p[-1][1]
        c = 128
        rows = gy.numel() // c
        chunks = part_b.shape[0]
        rpb = -(-rows // chunks)
        g2 = gy.contiguous().reshape(rows, c)
        pad = chunks * rpb - rows
        if pad:
            g2 = torch.cat([g2, g2.new_zeros(pad, c)], 0)
        ref = g2.reshape(chunks, rpb, c).double().sum(1)            # queued on the same stream, right behind the kernel
        early = part_b.clone()
        log.append((rows, chunks, part_b, ref, early, str(torch.cuda.current_stream(gy.device))[-16:]))
    return out

ops._LayerNorm.backward = staticmethod(patched)
for rep in range(5):
    ops.SideStream.enabled, ops.SideStream.lanes = True, 3
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    log.clear()
    # keep the deferred partials alive past the flush
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    for rows, chunks, part_b, ref, early, st in log:
        sc = float(ref.abs().max())
        d_late = (part_b.double() - ref).abs() / sc
        d_early = (early.double() - ref).abs() / sc
        bad = (d_late > 1e-5).nonzero()
        print(f"rep {rep} rows {rows} chunks {chunks} stream {st}: partials at the end of the step differ from the reference in {bad.shape[0]} elements"
              f" (copy taken right behind the kernel: {int((d_early > 1e-5).sum())})")
        if bad.shape[0]:
            r = sorted(set(bad[:, 0].tolist())); cc = sorted(set(bad[:, 1].tolist()))
            print("    rows", r, "cols", cc[:3], "..", cc[-3:], "ref", ref[r[0], cc[0]:cc[0] + 4].tolist(), "got", part_b[r[0], cc[0]:cc[0] + 4].tolist())
    del tr, model
    torch.cuda.empty_cache()
