"""Where the known lane deviation lands: which chunk row / columns of which LayerNorm call's dbeta partials differ from the
sums of that call's own dY, what the 64 bytes hold instead, and whether another live tensor of the step holds the same values."""
import gc, os, sys, torch
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
    c = gy.shape[-1]
    rows = gy.numel() // c
    keep = gy.detach().clone() if rows == 1728 and c == 128 else None
    out = orig(ctx, gy)
    pend = ops.DeferredBiasGrads.pending
    if keep is not None and pend is not None:
        log.append((keep, pend[-2][1], pend[-1][1], pend[-1][0], torch.cuda.current_stream().cuda_stream, ctx.saved_tensors[0].detach(),
                    ctx.saved_tensors[2], ctx.saved_tensors[3], out[0], ctx.saved_tensors[1].detach()))
    return out

ops._LayerNorm.backward = staticmethod(patched)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.SideStream.enabled, ops.SideStream.lanes = True, 3
    ops.StepRNG.tensor = None
    model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    names = {id(p): k for k, p in model.named_parameters()}
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    log.clear()
    tr._forward_backward(vol, lab)
    torch.cuda.synchronize()
    for i, (gy, pg, pb, bias, stream, x, mean, rstd, gx, gam) in enumerate(log):
        chunks = pb.shape[0]
        rpb = 1728 // chunks
        g2 = gy.reshape(chunks, rpb, 128).double()
        xh = ((x.reshape(1728, 128).double() - mean.double()[:, None]) * rstd.double()[:, None]).reshape(chunks, rpb, 128)
        for what, got, want in (("dbeta", pb, g2.sum(1)), ("dgamma", pg, (g2 * xh).sum(1))):
            d = (got.double() - want).abs()
            bad = d > 1e-4 * float(want.abs().max())
            if bool(bad.any()):
                rr, cc = bad.nonzero(as_tuple=True)
                print(f"WHERE step {step} call {i} ({names.get(id(bias))}, stream {stream:#x}) {what} partials [{chunks}x128] at {got.data_ptr():#x}: "
                      f"{int(bad.sum())} wrong; rows {sorted(set(rr.tolist()))} cols {min(cc.tolist())}..{max(cc.tolist())}; "
                      f"byte offset in block {(got.data_ptr() - pg.data_ptr()) + (rr[0].item() * 128 + cc.min().item()) * 4}")
                r = rr[0].item()
                c0 = cc.min().item()
                print("      holds:", [f"{v:.5g}" for v in got[r, c0:c0 + 16].tolist()])
                print("      wants:", [f"{v:.5g}" for v in want[r, c0:c0 + 16].tolist()])
                for r2 in range(chunks):      # the same columns of another chunk row?  (a 64-byte piece delivered to the wrong row)
                    if r2 != r and float((want[r2, c0:c0 + 16] - got[r, c0:c0 + 16].double()).abs().max()) < 1e-4 * float(want.abs().max()):
                        print(f"      = the expected content of row {r2}")
                if what == "dbeta":
                    for r in sorted(set(rr.tolist())):
                        diff = got[r, c0:c0 + 16].double() - want[r, c0:c0 + 16]
                        rows_ = g2[r, :, c0:c0 + 16]                      # [rpb, 16]: the chunk's rows of dY
                        cands = {}
                        for i in range(rpb):
                            cands[f"-row{i}"] = -rows_[i]
                            cands[f"+row{i}"] = rows_[i]
                        for w in range(4):
                            idx = [i for i in range(rpb) if (i // 4) % 4 == w]
                            cands[f"-wave{w}"] = -rows_[idx].sum(0)
                            for h in range(rpb // 16):
                                cands[f"-wave{w}.group{h}"] = -rows_[idx[4 * h:4 * h + 4]].sum(0)
                        # the same columns of a neighbouring 64-byte segment / of other chunks
                        for name, alt in (("cols-16", g2[r, :, c0 - 16:c0].sum(0) if c0 >= 16 else None), ("cols+16", g2[r, :, c0 + 16:c0 + 32].sum(0) if c0 + 32 <= 128 else None),
                                          ("cols+64", g2[r, :, c0 + 64:c0 + 80].sum(0) if c0 + 80 <= 128 else None)):
                            if alt is not None:
                                cands["holds=" + name] = alt - want[r, c0:c0 + 16]
                        best = sorted(((float((diff - v).norm() / diff.norm()), k) for k, v in cands.items()))[:3]
                        miss = int(best[0][1][4:]) if best[0][1].startswith("-row") else None
                        if miss is not None:
                            R = r * rpb + miss
                            xr = xh.reshape(1728, 128)[R]
                            gr = gy.reshape(1728, 128)[R].double()
                            gg = gr * gam.double()
                            want_gx = rstd[R].double() * (gg - gg.mean() - xr * (gg * xr).mean())
                            egx = (gx.reshape(1728, 128)[R].double() - want_gx).abs()
                            gg0 = gg.clone(); gg0[c0:c0 + 16] = 0
                            alt_gx = rstd[R].double() * (gg0 - gg0.mean() - xr * (gg0 * xr).mean())
                            print(f"      missing row {R}: |xhat| there {float(xr[c0:c0 + 16].abs().mean()):.3f}, |dY| there {float(gr[c0:c0 + 16].abs().mean()):.3e} (row mean {float(gr.abs().mean()):.3e}); "
                                  f"gx of that row vs formula: max err {float(egx.max()):.3e} of {float(want_gx.abs().max()):.3e}; vs formula with dY zeroed in those columns: {float((gx.reshape(1728, 128)[R].double() - alt_gx).abs().max()):.3e}; "
                                  f"dgamma partial err there {float((pg[r, c0:c0 + 16].double() - (g2[r] * xh[r]).sum(0)[c0:c0 + 16]).abs().max()):.3e} vs term {float((gr * xr)[c0:c0 + 16].abs().max()):.3e}")
                        print(f"      row {r}: |diff|/|want| {float(diff.norm() / want[r, c0:c0 + 16].norm()):.3f}; best explanations (residual): {best}")
                        # per-row-subset least squares is underdetermined; report the projection on every wave's own sum instead
                v0 = got[r, c0].item()
                for t in gc.get_objects():
                    try:
                        if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() >= 16 and t.data_ptr() != got.data_ptr():
                            hit = (t.detach().reshape(-1) == v0).nonzero()
                            if hit.numel():
                                print(f"      first value also found in a live tensor {tuple(t.shape)} at element {hit[0].item()} (ptr {t.data_ptr():#x})")
                    except Exception:
                        pass
    tot = {}
    for gy, pg, pb, bias, stream, x, mean, rstd, gx, gam in log:
        tot.setdefault(id(bias), [bias, 0, 0])
        tot[id(bias)][1] = tot[id(bias)][1] + gy.reshape(-1, 128).double().sum(0)
        tot[id(bias)][2] = tot[id(bias)][2] + pb.double().sum(0)
    for bias, want, fromparts in tot.values():
        d = (bias.grad.double() - want).abs()
        d2 = (fromparts - want).abs()
        bad = (d > 1e-4 * float(want.abs().max())).nonzero().flatten().tolist()
        print(f"WHERE step {step} {names.get(id(bias))}: grad vs sums of dY: {len(bad)} columns off {bad[:1]}..{bad[-1:]} (max {float(d.max()):.3e} of {float(want.abs().max()):.3e}); "
              f"sum of the partials as they are now vs sums of dY: max {float(d2.max()):.3e}; calls {sum(1 for l in log if l[3] is bias)}")
    del tr, model
    torch.cuda.empty_cache()
print("WHERE done")
