"""GPU parity tests (run with -m gpu on the MI355X box): HIP kernels through the C ABI vs the CPU oracle /
the golden vectors captured from the reference.  Tolerances: 1e-3 relative on logits, 1e-4 on Dice
(BASELINE.json north_star); kernel-level checks are much tighter (fp32 reassociation only)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, pct_err, rel_err, rms_err
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from icl_amd import _lib
    assert _lib.lib_path().endswith("libicl_hip.so"), "GPU tests must run on the HIP library"
    return torch.device("cuda", 0)


def _rand(shape, seed):
    return synthetic_volume(tuple(shape), seed)


@pytest.mark.parametrize("n,cin,cout,d,h,w,ks", [
    (2, 3, 4, 8, 8, 8, 3), (1, 1, 16, 6, 6, 6, 3), (1, 16, 32, 4, 8, 16, 3), (1, 20, 48, 6, 6, 12, 3),
    (2, 8, 2, 4, 4, 8, 1), (1, 32, 16, 2, 8, 32, 3),
    (1, 1, 16, 48, 48, 48, 3), (1, 16, 16, 48, 48, 48, 3), (1, 48, 16, 32, 32, 32, 3), (1, 96, 32, 24, 24, 24, 3),
    (2, 64, 128, 12, 12, 12, 3), (1, 384, 128, 12, 12, 12, 3), (1, 256, 256, 6, 6, 6, 3), (1, 16, 2, 48, 48, 48, 1),
    (2, 64, 64, 24, 24, 24, 1),
])
def test_conv3d(dev, n, cin, cout, d, h, w, ks):
    from icl_amd import ops
    x = _rand((n, cin, d, h, w), 1)
    wt = _rand((cout, cin, ks, ks, ks), 2) * (1.0 / np.sqrt(cin * ks ** 3))
    b = _rand((cout,), 3) * 0.1
    gy = _rand((n, cout, d, h, w), 4)
    xg, wg, bg = (t.to(dev).requires_grad_() for t in (x, wt, b))
    y = ops.conv3d(xg, wg, bg)
    y.backward(gy.to(dev))
    xr, wr, br = (t.clone().requires_grad_() for t in (x, wt, b))
    yr = F.conv3d(xr, wr, br, padding=ks // 2)
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 2e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-4
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 3, 4, 6, 8), (1, 16, 48, 48, 48), (2, 256, 6, 6, 6), (1, 2, 96, 96, 96)])
def test_instance_norm_relu(dev, shape):
    from icl_amd import ops
    x = _rand(shape, 5) * 2 + 0.7
    gy = _rand(shape, 6)
    xg = x.to(dev).requires_grad_()
    y = ops.instance_norm_relu(xg)
    y.backward(gy.to(dev))
    xr = x.clone().requires_grad_()
    yr = F.relu(F.instance_norm(xr, eps=1e-5))
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 5e-5


@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu(dev, training):
    from icl_amd import ops
    shape = (4, 8, 12, 12, 12)
    x = _rand(shape, 7) + 0.3
    ga, be = 1 + 0.1 * _rand((8,), 8), 0.1 * _rand((8,), 9)
    rm, rv = 0.1 * _rand((8,), 10), 1 + 0.2 * _rand((8,), 11).abs()
    gy = _rand(shape, 12)
    xg, gg, bg = (t.to(dev).requires_grad_() for t in (x, ga, be))
    rmg, rvg = rm.to(dev), rv.to(dev)
    y = ops.batch_norm_relu(xg, gg, bg, rmg, rvg, training)
    y.backward(gy.to(dev))
    xr, gr, br = (t.clone().requires_grad_() for t in (x, ga, be))
    rm2, rv2 = rm.clone(), rv.clone()
    yr = F.relu(F.batch_norm(xr, rm2, rv2, gr, br, training, 0.1, 1e-5))
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 5e-5
    if training:
        assert rel_err(gg.grad.cpu(), gr.grad) < 5e-5 and rel_err(bg.grad.cpu(), br.grad) < 5e-5
        assert rel_err(rmg.cpu(), rm2) < 1e-5 and rel_err(rvg.cpu(), rv2) < 1e-5


def test_maxpool_ties(dev):
    from icl_amd import ops
    x = torch.relu(_rand((2, 16, 24, 24, 24), 13))
    xg = x.to(dev).requires_grad_()
    y = ops.max_pool3d_2(xg)
    gy = _rand(tuple(y.shape), 14)
    y.backward(gy.to(dev))
    xr = x.clone().requires_grad_()
    yr = F.max_pool3d(xr, 2)
    yr.backward(gy)
    assert torch.equal(y.detach().cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


def test_pack_weights_multi_equals_the_single_packs(dev):
    """icl_conv3d_pack_weights_multi (one launch per step for every convolution weight; LDS tiles of 16 couts x 16 cins x taps since round 6)
    against icl_conv3d_pack_weights per weight and layout, bit for bit, zero padding included: the U-Net's and SwinUNETR's shapes plus
    ragged ones.  Reference: the nn.Conv3d weights of the backbones (/root/reference/code/networks/utils.py:104)."""
    import ctypes
    from icl_amd import _lib, ops
    L = _lib.lib()
    shapes = [(16, 1, 3), (16, 16, 3), (16, 48, 3), (128, 384, 3), (256, 256, 3), (384, 768, 3), (48, 96, 1), (20, 7, 3), (3, 20, 1), (2, 16, 1)]
    ws = [_rand((co, ci, k, k, k), 70 + i).to(dev) for i, (co, ci, k) in enumerate(shapes)]
    fwd = [torch.full((L.icl_conv3d_packed_elems(co, ci, k, 0),), float("nan"), device=dev) for co, ci, k in shapes]
    dgr = [torch.full((L.icl_conv3d_packed_elems(co, ci, k, 1),), float("nan"), device=dev) for co, ci, k in shapes]
    n = len(shapes)
    arr, iarr = ctypes.c_void_p * n, ctypes.c_int32 * n
    _lib.check(L.icl_conv3d_pack_weights_multi(arr(*[w.data_ptr() for w in ws]), arr(*[t.data_ptr() for t in fwd]), arr(*[t.data_ptr() for t in dgr]),
                                               iarr(*[s[0] for s in shapes]), iarr(*[s[1] for s in shapes]), iarr(*[s[2] for s in shapes]), n,
                                               ops._stream(ws[0])), "pack_weights_multi")
    for w, f, d in zip(ws, fwd, dgr):
        assert torch.equal(f, ops.pack_weights(w, 0)) and torch.equal(d, ops.pack_weights(w, 1)), tuple(w.shape)


@pytest.mark.parametrize("shape", [(2, 16, 48, 48, 48), (2, 64, 12, 12, 12), (1, 8, 4, 6, 6)])
def test_skip_and_pool_backward_adds_both_gradients_in_one_pass(dev, shape):
    """ops.skip_and_pool: an encoder output feeds the level's skip connection and the next level's pooling
    (/root/reference/code/networks/unet_3D_icl.py:100-116); the backward adds the skip gradient — a batch-strided channel slice of the
    concat gradient — inside the pooling backward's pass (maxpool2_bwd_add_x2_kernel: two pooled outputs per thread, 16-byte accesses;
    rows that pool to an odd width keep maxpool2_bwd_add_kernel).  Bit-exact against autograd on (x, max_pool3d(x)), ties included."""
    from icl_amd import ops
    n, c = shape[:2]
    x = torch.relu(_rand(shape, 61))                       # zeros: ties between the window's elements
    xg = x.to(dev).requires_grad_()
    skip, y = ops.skip_and_pool(xg)
    gfull = _rand((n, c + 5) + tuple(shape[2:]), 62)
    gsk = gfull[:, 1:c + 1]                                 # non-contiguous over the batch, as the concat gradient hands it over
    gy = _rand(tuple(y.shape), 63)
    torch.autograd.backward([skip, y], [gfull.to(dev)[:, 1:c + 1], gy.to(dev)])
    xr = x.clone().requires_grad_()
    yr = F.max_pool3d(xr, 2)
    torch.autograd.backward([xr * 1.0, yr], [gsk, gy])
    assert torch.equal(y.detach().cpu(), yr.detach()) and torch.equal(xg.grad.cpu(), xr.grad)


@pytest.mark.parametrize("ins,outs,c", [((3, 4, 5), (6, 8, 10), 3), ((6, 6, 6), (96, 96, 96), 2), ((12, 12, 12), (96, 96, 96), 2),
                                        ((24, 24, 24), (96, 96, 96), 2), ((24, 24, 24), (48, 48, 48), 16)])
def test_trilinear(dev, ins, outs, c):
    from icl_amd import ops
    x = _rand((1, c) + ins, 15)
    xg = x.to(dev).requires_grad_()
    y = ops.trilinear_resize(xg, outs)
    gy = _rand(tuple(y.shape), 16)
    y.backward(gy.to(dev))
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=list(outs), mode="trilinear", align_corners=False)
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-6
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-5


def test_upsample_concat_and_depthwise_and_dropout(dev):
    from icl_amd import ops
    skip, deep = _rand((2, 16, 12, 12, 12), 17), _rand((2, 32, 6, 6, 6), 18)
    sg, dg = skip.to(dev).requires_grad_(), deep.to(dev).requires_grad_()
    y = ops.upsample2x_concat(sg, dg)
    gy = _rand(tuple(y.shape), 19)
    y.backward(gy.to(dev))
    sr, dr = skip.clone().requires_grad_(), deep.clone().requires_grad_()
    yr = torch.cat([sr, F.interpolate(dr, scale_factor=(2, 2, 2), mode="trilinear")], 1)
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-6
    assert rel_err(sg.grad.cpu(), sr.grad) < 1e-6 and rel_err(dg.grad.cpu(), dr.grad) < 2e-5
    # depthwise
    x, w = _rand((6, 4, 12, 12, 12), 20), _rand((4, 1, 3, 3, 3), 21) * 0.3
    xg, wg = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    y = ops.depthwise_conv3d(xg, wg)
    gy = _rand(tuple(y.shape), 22)
    y.backward(gy.to(dev))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv3d(xr, wr, None, padding=1, groups=4)
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5 and rel_err(wg.grad.cpu(), wr.grad) < 1e-4
    # dropout: mask statistics, scaling, and backward uses the same mask
    x = torch.ones(1 << 20, device=dev, requires_grad=True)
    y = ops.dropout(x, 0.3, seed=123)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 5e-3
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1 / 0.7))
    y.sum().backward()
    assert torch.equal(x.grad != 0, y.detach() != 0)


def test_token_ops_and_losses_and_sgd(dev):
    """LayerNorm / GELU / prototype attention / fused losses / FusedSGD against torch-CPU expressions of the oracle."""
    from icl_amd import ops
    from icl_amd.optim import FusedSGD
    # LayerNorm + GELU, short rows and the 13,824-long token rows of norm3
    for shape in [(2, 1728, 128), (16, 13824)]:
        c = shape[-1]
        x, w, b, gy = _rand(shape, 61) * 2 + 0.3, 1 + 0.1 * _rand((c,), 62), 0.1 * _rand((c,), 63), _rand(shape, 64)
        xg, wg, bg = (t.to(dev).requires_grad_() for t in (x, w, b))
        y = ops.gelu(ops.layer_norm(xg, wg, bg))
        y.backward(gy.to(dev))
        xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
        yr = F.gelu(F.layer_norm(xr, (c,), wr, br, 1e-5))
        yr.backward(gy)
        assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
        assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4 and rel_err(wg.grad.cpu(), wr.grad) < 1e-4 and rel_err(bg.grad.cpu(), br.grad) < 1e-4
    # prototype attention at the three aligner scales (nc = 2 and 16)
    for B, h, nc, d, N in [(2, 16, 2, 16, 216), (1, 8, 16, 16, 1728), (2, 4, 2, 16, 13824)]:
        C = h * d
        qh, kv = _rand((B, h, nc, d), 71), _rand((B, N, 2 * C), 72)
        go, gl = _rand((B, h, nc, d), 73), _rand((B, nc, h, N), 74)      # logits come back class-major (attn1.permute(0, 2, 1, 3))
        qg, kg = qh.to(dev).requires_grad_(), kv.to(dev).requires_grad_()
        out, logits = ops.prototype_attention(qg, kg, h, d ** -0.5)
        assert logits.shape == (B, nc, h, N) and logits.is_contiguous()
        ((out * go.to(dev)).sum() + (logits * gl.to(dev)).sum()).backward()
        qr, kr = qh.clone().requires_grad_(), kv.clone().requires_grad_()
        kvp = kr.reshape(B, N, 2, h, d).permute(2, 0, 3, 1, 4)
        lr4 = (qr @ kvp[0].transpose(-2, -1)) * d ** -0.5
        orf = lr4.softmax(dim=-1) @ kvp[1]
        lr = lr4.permute(0, 2, 1, 3)
        ((orf * go).sum() + (lr * gl).sum()).backward()
        assert rel_err(out.detach().cpu(), orf.detach()) < 1e-5 and rel_err(logits.detach().cpu(), lr.detach()) < 1e-5
        assert rel_err(qg.grad.cpu(), qr.grad) < 1e-4 and rel_err(kg.grad.cpu(), kr.grad) < 1e-4
    # fused losses at 96^3, nc = 2 and 16 (oracle functions are the expected values)
    from oracle import icl_oracle as O
    for nc in (2, 16):
        lab = synthetic_labels((1, 96, 96, 96), 61, nc)
        a, b2 = _rand((1, nc, 96, 96, 96), 62), _rand((1, nc, 96, 96, 96), 63)
        ag = a.to(dev).requires_grad_()
        ce, dc = ops.cross_entropy_dice_parts(ag, lab.to(dev), nc)
        sd, ms = ops.soft_dice_loss(ag, b2.to(dev)), ops.softmax_mse(ag, b2.to(dev))
        (ce + dc + sd + 10 * ms).backward()
        ar = a.clone().requires_grad_()
        cer = F.cross_entropy(ar, lab)
        dcr = O.dice_loss(ar, lab.unsqueeze(1), nc, softmax=True)
        sdr = O.softmax_dice_loss(ar, b2)
        msr = torch.mean((F.softmax(ar, 1) - F.softmax(b2, 1)) ** 2)
        (cer + dcr + sdr + 10 * msr).backward()
        for got, ref in ((ce, cer), (dc, dcr), (sd, sdr), (ms, msr)):
            assert abs(float(got) - float(ref)) < 1e-5
        assert rel_err(ag.grad.cpu(), ar.grad) < 1e-4
    # FusedSGD == torch.optim.SGD over three steps, including a parameter that never gets a gradient
    torch.manual_seed(0)
    shapes = [(5, 7), (33,), (3 << 20,), (3, 3, 3, 2, 2), (1,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    skip = torch.nn.Parameter(torch.randn(4, device=dev))
    skip0 = skip.detach().clone()
    fa = FusedSGD(ps + [skip], lr=0.01, momentum=0.9, weight_decay=1e-4)
    fb = torch.optim.SGD(qs, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for it in range(3):
        for p, q in zip(ps, qs):
            g = torch.randn(q.shape)
            p.grad, q.grad = g.to(dev), g.clone()
        fa.step()
        fb.step()
    for p, q in zip(ps, qs):
        assert rel_err(p.detach().cpu(), q.detach()) < 1e-6
    assert torch.equal(skip.detach(), skip0)


def test_plain_unet3d_matches_reference_golden(dev):
    from icl_amd.networks.unet_3D import unet_3D
    g = load_golden("unit.npz")
    m = unet_3D(n_classes=3, in_channels=2, device=dev)
    assert list(m.state_dict().keys()) == list(g["unet32_keys"])
    fill_like_reference_init(list(m.named_parameters()))
    m.eval()
    x = synthetic_volume((1, 2, 32, 32, 32), 16).to(dev).requires_grad_()
    y = m(x)
    y.backward(synthetic_volume(tuple(y.shape), 17).to(dev))
    assert rel_err(y.detach().cpu(), g["unet32_y"]) < 1e-3
    for k, p in m.named_parameters():
        if k.endswith("bias") and k != "final.bias":
            continue
        ref = float(g["unet32_gn." + k])
        got = float(p.grad.double().norm())
        assert abs(got - ref) <= 5e-3 * max(ref, 1e-6), k
    assert rel_err(m.final.weight.grad.cpu(), g["unet32_g.final.weight"]) < 1e-3


def _parity_mode(model):
    from icl_amd.networks.aligner import DropPath
    from icl_amd.networks.layers import Dropout3
    for mod in model.modules():
        if isinstance(mod, Dropout3):
            mod.p = 0.0
        if isinstance(mod, DropPath):
            mod.drop_prob = 0.0


@pytest.mark.parametrize("nc", [2, 16])
def test_full_icl_step_matches_reference_golden(dev, nc):
    """BASELINE configs[1] (nc=2) and configs[4] (nc=16) shapes: 96^3, batch 1+1, the whole ICL step."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    g = load_golden(f"model_unet3d_icl_nc{nc}.npz")
    model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    assert list(model.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in model.named_parameters()] == list(g["param_keys"])
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc).to(dev)
    model.eval()
    with torch.no_grad():
        y = model(vol[:1], inference=True)
    assert rel_err(y[:, :, ::8, ::8, ::8].cpu(), g["inf_logits_sub"]) < 1e-3
    assert rel_err(y[:, :, 40:44, 40:56, 40:56].cpu(), g["inf_logits_slab"]) < 1e-3
    l2 = y.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt().cpu().numpy()
    assert np.allclose(l2, g["inf_logits_l2"], rtol=1e-4)
    model.train()
    cfg = ICLConfig(num_classes=nc, labeled_bs=1, w_pse=0.1 if nc == 16 else 1.0)
    tr = ICLTrainer(model, cfg)
    outs = model(vol[:1], vol[1:])
    assert rel_err(outs[0].detach()[:, :, ::8, ::8, ::8].cpu(), g["final_lab_sub"]) < 1e-3
    assert rel_err(outs[1].detach()[:, :, ::8, ::8, ::8].cpu(), g["final_unlab_sub"]) < 1e-3
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            assert rel_err(t.detach().cpu(), g[f"{name}{i}"]) < 1e-3, (name, i)
    # the max-norm above leaves small elements unconstrained: 99.9th percentile of the element-wise error relative to
    # |reference| + 1e-3 max|reference| (an element 1000x below the largest must be right to 2 % of its own size)
    pcts = {"inf_logits": pct_err(y[:, :, ::8, ::8, ::8].cpu(), g["inf_logits_sub"]),
            "final_lab": pct_err(outs[0].detach()[:, :, ::8, ::8, ::8].cpu(), g["final_lab_sub"])}
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            pcts[f"{name}{i}"] = pct_err(t.detach().cpu(), g[f"{name}{i}"])
    print("pct_err(99.9)", nc, {k: f"{v:.2e}" for k, v in pcts.items()})
    assert max(pcts.values()) < 2e-2, pcts
    loss, parts = tr.compute_loss(outs, lab)
    got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(loss)]
    assert np.allclose(got, g["losses"], rtol=0, atol=1e-4), (got, g["losses"])
    soft = torch.softmax(outs[0].detach(), 1)
    for c in range(nc):
        t = (lab == c).float()
        p = soft[:, c]
        sd = float((2 * (p * t).sum() + 1e-5) / ((p * p).sum() + (t * t).sum() + 1e-5))
        assert abs(sd - float(g["soft_dice"][c])) < 1e-4
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    none = [k for k, p in model.named_parameters() if p.grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, p in model.named_parameters():
        # conv biases that feed an InstanceNorm (".0.bias") or a softmax over classes (attn_convs1.*.bias)
        # have an exactly-zero true gradient: both sides hold rounding noise only
        if p.grad is None or k.endswith(".0.bias") or ("attn_convs1" in k and k.endswith("bias")):
            continue
        got_n = float(p.grad.double().norm())
        if abs(got_n - ref[k]) > 1e-2 * max(ref[k], 1e-7) + 1e-9:
            bad.append((k, got_n, ref[k]))
    assert not bad, bad[:10]
    assert rel_err(model.final.weight.grad.cpu(), g["grad.final.weight"]) < 1e-3
    assert rel_err(model.sspa.guided_Q.grad.cpu(), g["grad.sspa.guided_Q"]) < 5e-3
    tr.optimizer.step()
    post = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    names = [k for k, _ in model.named_parameters()]
    off = [(names[i], post[i], g["post_sgd_norms"][i]) for i in range(len(names))
           if abs(post[i] - g["post_sgd_norms"][i]) > 1e-4 * g["post_sgd_norms"][i]]
    assert not off, off[:8]
    assert rel_err(model.final.weight.detach().cpu(), g["post_sgd.final.weight"]) < 1e-5


def test_swinunetr_icl_step_matches_reference_golden(dev):
    """BASELINE configs[3]: SwinUNETR-ICL 96^3, nc=2, batch 1+1 — hidden states of the Swin encoder, forward 5-tuple,
    losses, grad-None set, gradient norms and one SGD step against the reference golden."""
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 2
    g = load_golden("model_swinunetr_icl_nc2.npz")
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
    assert list(model.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in model.named_parameters()] == list(g["param_keys"])
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc).to(dev)
    model.eval()
    with torch.no_grad():
        hs = model.swinViT(vol[:1], True)
        for i, h in enumerate(hs):
            sub = h[:, ::max(1, h.shape[1] // 8), ::max(1, h.shape[2] // 6), ::max(1, h.shape[3] // 6), ::max(1, h.shape[4] // 6)]
            assert rel_err(sub.cpu(), g[f"hidden{i}_sub"]) < 1e-3, i
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1))
    outs = model(vol[:1], vol[1:])
    assert rel_err(outs[0].detach()[:, :, ::8, ::8, ::8].cpu(), g["final_lab_sub"]) < 1e-3
    assert rel_err(outs[1].detach()[:, :, ::8, ::8, ::8].cpu(), g["final_unlab_sub"]) < 1e-3
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            assert rel_err(t.detach().cpu(), g[f"{name}{i}"]) < 1e-3, (name, i)
    loss, parts = tr.compute_loss(outs, lab)
    got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(loss)]
    assert np.allclose(got, g["losses"], rtol=0, atol=1e-4), (got, g["losses"])
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    none = [k for k, p in model.named_parameters() if p.grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, p in model.named_parameters():
        if p.grad is None or ("attn_convs1" in k and k.endswith("bias")):
            continue
        got_n = float(p.grad.double().norm())
        if abs(got_n - ref[k]) > 1e-2 * max(ref[k], 1e-7) + 1e-9:
            bad.append((k, got_n, ref[k]))
    assert not bad, bad[:10]
    sd = dict(model.named_parameters())
    for k in ("out.conv.conv.weight", "swinViT.patch_embed.proj.weight", "swinViT.layers4.0.blocks.0.attn.qkv.bias",
              "decoder5.transp_conv.conv.weight", "swinViT.layers1.0.blocks.1.attn.relative_position_bias_table"):
        gg = sd[k].grad.cpu()
        gg = gg if gg.numel() <= 8192 else gg.reshape(-1)[::97]
        # 2e-2: the golden is the reference's fp32 CPU backward, which is itself 0.5-1 % away from an fp64 evaluation of
        # the same graph on the deep Swin/decoder5 parameters (profiles/r1_swin_grad_accuracy.txt: HIP ~3e-4, CPU fp32
        # ~6e-3 against the fp64 oracle) — InstanceNorm over 27..216 voxels amplifies summation-order noise.
        assert rel_err(gg, g["grad." + k]) < 2e-2, k
    tr.optimizer.step()
    post = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    names = [k for k, _ in model.named_parameters()]
    off = [(names[i], post[i], g["post_sgd_norms"][i]) for i in range(len(names))
           if abs(post[i] - g["post_sgd_norms"][i]) > 1e-4 * g["post_sgd_norms"][i]]
    assert not off, off[:8]
    model.eval()                      # the golden's inference logits are those of the model AFTER the SGD step
    with torch.no_grad():
        y = model(vol[:1], inference=True)
    assert rel_err(y[:, :, ::8, ::8, ::8].cpu(), g["inf_logits_sub"]) < 1e-3


def test_2d_unet_icl_step_matches_reference_golden(dev):
    """BASELINE config 1 (2D U-Net ICL, 256x256, nc=4, batch 2+2) on the HIP kernels vs the reference golden."""
    from icl_amd.networks.unet import UNet
    from icl_amd.networks.unet_icl import UNet_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 4
    g = load_golden("model_unet2d_icl_nc4.npz")
    assert list(UNet(1, nc, device=dev).state_dict().keys()) == list(g["plain_keys"])
    model = UNet_icl(1, nc, device=dev)
    assert list(model.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in model.named_parameters()] == list(g["param_keys"])
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    img = synthetic_volume((4, 1, 256, 256), 2024).to(dev)
    lab = synthetic_labels((2, 256, 256), 2025, nc).to(dev)
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=2, w_con=50.0, patch_size=(256, 256)))
    outs = model(img[:2], img[2:])
    assert rel_err(outs[0].detach()[:, :, ::8, ::8].cpu(), g["out_lab_sub"]) < 1e-3
    assert rel_err(outs[1].detach()[:, :, ::8, ::8].cpu(), g["out_unlab_sub"]) < 1e-3
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            assert rel_err(t.detach()[:, :, ::st, ::st].cpu(), g[f"{name}{i}_sub"]) < 1e-3, (name, i)
    loss, parts = tr.compute_loss(outs, lab)
    got = [float(parts[k].detach()) for k in ("ce", "dice", "aux", "pse", "con")] + [float(loss.detach())]
    assert np.allclose(got, g["losses"], rtol=0, atol=2e-4), (got, g["losses"])
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    none = [k for k, p in model.named_parameters() if p.grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, p in model.named_parameters():
        if p.grad is None or (k.endswith("bias") and (".conv_conv.0." in k or ".conv_conv.4." in k or "attn_convs1" in k)):
            continue
        got_n = float(p.grad.double().norm())
        if abs(got_n - ref[k]) > 1e-2 * max(ref[k], 1e-7) + 1e-9:
            bad.append((k, got_n, ref[k]))
    assert not bad, bad[:10]
    assert rel_err(model.decoder.out_conv.weight.grad.cpu(), g["grad.decoder.out_conv.weight"]) < 1e-3
    tr.optimizer.step()
    post = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    assert np.allclose(post, g["post_sgd_norms"], rtol=1e-4)
    bufs = dict(model.named_buffers())
    for k in ("encoder.in_conv.conv_conv.1.running_mean", "encoder.in_conv.conv_conv.1.running_var",
              "decoder.up4.conv.conv_conv.5.running_var", "sspa.attn_convs0.2.block.bn_depth.running_var"):
        assert rel_err(bufs[k].cpu(), g["buf." + k]) < 1e-4, k
    model.eval()
    with torch.no_grad():
        y = model(img[:2], inference=True)
    assert rel_err(y[:, :, ::8, ::8].cpu(), g["inf_logits_sub"]) < 1e-3


def test_swinunet2d_icl_step_matches_reference_golden(dev):
    """2-D Swin-UNet ICL (224^2, nc=4, batch 2+2; 7x7 window attention with head dim 32 on the fused MFMA
    kernels) — encoder output, decoder features, forward 5-tuple, losses, grad-None set, gradient norms, one SGD step and the
    post-step inference logits against the reference golden; then the same step through ICLTrainer (factored mlp2 gradients)."""
    from icl_amd.networks.net_factory import net_factory
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 4
    g = load_golden("model_swinunet2d_icl_nc4.npz")
    model = net_factory("icl_swinunet", in_chns=1, class_num=nc)
    assert list(model.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in model.named_parameters()] == list(g["param_keys"])
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    img = synthetic_volume((4, 1, 224, 224), 3024).to(dev)
    lab = synthetic_labels((2, 224, 224), 3025, nc).to(dev)
    model.train()
    with torch.no_grad():
        xe, skips = model.swin_unet.forward_features(img[:2].repeat(1, 3, 1, 1))
        assert rel_err(xe[:, ::7, ::32].cpu(), g["enc_out"]) < 1e-3
        _, feats = model.swin_unet.forward_up_features(xe, skips)
        for i, t in enumerate(feats):
            assert rel_err(t[:, ::13, ::16].cpu(), g[f"feat{i}_sub"]) < 1e-3, i
    cfg = ICLConfig(num_classes=nc, labeled_bs=2, w_con=50.0, patch_size=(224, 224))
    tr = ICLTrainer(model, cfg)
    outs = model(img[:2], img[2:])
    assert rel_err(outs[0].detach()[:, :, ::8, ::8].cpu(), g["out_lab_sub"]) < 1e-3
    assert rel_err(outs[1].detach()[:, :, ::8, ::8].cpu(), g["out_unlab_sub"]) < 1e-3
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            assert rel_err(t.detach()[:, :, ::st, ::st].cpu(), g[f"{name}{i}_sub"]) < 1e-3, (name, i)
    loss, parts = tr.compute_loss(outs, lab)
    got = [float(parts[k].detach()) for k in ("ce", "dice", "aux", "pse", "con")] + [float(loss.detach())]
    assert np.allclose(got, g["losses"], rtol=0, atol=2e-4), (got, g["losses"])
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    none = [k for k, p in model.named_parameters() if p.grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, p in model.named_parameters():
        if p.grad is None or ("attn_convs1" in k and k.endswith("bias")):
            continue
        got_n = float(p.grad.double().norm())
        if abs(got_n - ref[k]) > 1e-2 * max(ref[k], 1e-7) + 1e-9:
            bad.append((k, got_n, ref[k]))
    assert not bad, bad[:10]
    sd = dict(model.named_parameters())
    for k in ("swin_unet.output.weight", "swin_unet.patch_embed.proj.weight", "swin_unet.layers.0.blocks.1.attn.relative_position_bias_table",
              "swin_unet.layers_up.0.expand.weight", "swin_unet.concat_back_dim.1.weight", "sspa.guided_Q"):
        gg = sd[k].grad.cpu()
        gg = gg if gg.numel() <= 8192 else gg.reshape(-1)[::97]
        assert rel_err(gg, g["grad." + k]) < 1e-2, k
    tr.optimizer.step()
    post = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    assert np.allclose(post, g["post_sgd_norms"], rtol=1e-4)
    model.eval()
    with torch.no_grad():
        y = model(img[:2], inference=True)
    assert rel_err(y[:, :, ::8, ::8].cpu(), g["inf_logits_sub"]) < 1e-3
    # the same first step through the trainer (3136^2 mlp2 weights take the factored-gradient path)
    model2 = net_factory("icl_swinunet", in_chns=1, class_num=nc)
    fill_like_reference_init(list(model2.named_parameters()))
    _parity_mode(model2)
    model2.train()
    parts2 = ICLTrainer(model2, cfg).step(img, lab)
    assert abs(float(parts2["loss"]) - float(g["losses"][5])) < 2e-4
    post2 = np.array([float(p.detach().double().norm()) for _, p in model2.named_parameters()])
    assert np.allclose(post2, g["post_sgd_norms"], rtol=1e-4)


def test_sliding_window_validation_and_checkpoint_interop(dev):
    """Rows f1 + f3: the on-device batched sliding window reproduces the reference procedure (val_3D.py:15-83) window by
    window; the filtered checkpoint of the ICL model loads into the plain backbone (…BraTS.py:158-162, test_3D_BraTS.py)."""
    import math
    from icl_amd.networks.unet_3D import unet_3D
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import backbone_state_dict
    from icl_amd.val_3D import cal_metric, test_single_case_base
    icl = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
    fill_like_reference_init(list(icl.named_parameters()))
    plain = unet_3D(n_classes=2, in_channels=1, device=dev)
    missing = plain.load_state_dict(backbone_state_dict(icl), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    icl.eval(); plain.eval()
    x = synthetic_volume((1, 1, 96, 96, 96), 5).to(dev)
    with torch.no_grad():
        # same weights, same kernels, fixed-order slab sums: equal up to nothing but the order of the checks below
        assert rel_err(icl(x, inference=True).cpu(), plain(x).cpu()) < 1e-5
    patch = (96, 96, 96)
    for shape in [(100, 110, 120), (80, 100, 96)]:
        image = synthetic_volume(shape, 9).numpy()
        got = test_single_case_base(plain, "unet_3D", image, 64, 64, patch, num_classes=2)
        # the reference procedure, one window per forward, numpy accumulation
        w, h, d = image.shape
        pads = [((max(p - s, 0)) // 2, max(p - s, 0) - max(p - s, 0) // 2) for s, p in zip((w, h, d), patch)]
        img = np.pad(image, pads, mode="constant", constant_values=0)
        ww, hh, dd = img.shape
        sx, sy, sz = (math.ceil((n - 96) / 64) + 1 for n in (ww, hh, dd))
        score = np.zeros((2,) + img.shape, np.float32)
        cnt = np.zeros(img.shape, np.float32)
        for xi in range(sx):
            xs = min(64 * xi, ww - 96)
            for yi in range(sy):
                ys = min(64 * yi, hh - 96)
                for zi in range(sz):
                    zs = min(64 * zi, dd - 96)
                    p = torch.from_numpy(img[xs:xs + 96, ys:ys + 96, zs:zs + 96][None, None].astype(np.float32)).to(dev)
                    with torch.no_grad():
                        y = torch.softmax(plain(p), dim=1).cpu().numpy()[0]
                    score[:, xs:xs + 96, ys:ys + 96, zs:zs + 96] += y
                    cnt[xs:xs + 96, ys:ys + 96, zs:zs + 96] += 1
        ref = np.argmax(score / cnt[None], axis=0)
        ref = ref[pads[0][0]:pads[0][0] + w, pads[1][0]:pads[1][0] + h, pads[2][0]:pads[2][0] + d]
        assert got.shape == ref.shape == shape
        assert (got != ref).mean() < 1e-4   # identical up to argmax ties of fp32 sums taken in a different order
        dice, hd = cal_metric(ref == 1, got == 1)
        assert dice > 0.9999


def test_trainer_step_with_factored_mlp2_gradients_matches_reference_golden(dev):
    """ICLTrainer.step keeps the four 13,824^2 mlp2 weight gradients factored (ops.FactoredGrads) and FusedSGD applies them
    without forming the matrices: every parameter after ONE step must equal the reference's post-SGD state."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 2
    g = load_golden(f"model_unet3d_icl_nc{nc}.npz")
    model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    model.train()
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc).to(dev)
    tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1))
    parts = tr.step(vol, lab)
    got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con", "loss")]
    assert np.allclose(got, g["losses"], rtol=0, atol=1e-4), (got, g["losses"])
    big = [k for k, p in model.named_parameters() if p.numel() >= (1 << 22)]
    assert len(big) == 4 and all(dict(model.named_parameters())[k].grad is None for k in big)   # never materialised
    # BatchNorm step counters (one deferred multi-tensor add per step): sspa normalises both inputs, uscl one
    counters = {k: int(b) for k, b in model.named_buffers() if k.endswith("num_batches_tracked")}
    assert counters and all(v == (2 if k.startswith("sspa") else 1) for k, v in counters.items()), counters
    post = np.array([float(p.detach().double().norm()) for _, p in model.named_parameters()])
    names = [k for k, _ in model.named_parameters()]
    off = [(names[i], post[i], g["post_sgd_norms"][i]) for i in range(len(names))
           if abs(post[i] - g["post_sgd_norms"][i]) > 1e-4 * g["post_sgd_norms"][i]]
    assert not off, off[:8]
    assert rel_err(model.final.weight.detach().cpu(), g["post_sgd.final.weight"]) < 1e-5
    # the update itself, not only the norm: compare one big matrix with a dense-gradient step of a second model
    model2 = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    fill_like_reference_init(list(model2.named_parameters()))
    _parity_mode(model2)
    model2.train()
    tr2 = ICLTrainer(model2, ICLConfig(num_classes=nc, labeled_bs=1, factored_mlp2_grads=False))
    tr2.step(vol, lab)
    for k in big:
        a, b = dict(model.named_parameters())[k].detach(), dict(model2.named_parameters())[k].detach()
        w0 = torch.empty_like(a)
        fill_like_reference_init([(k, w0)])
        assert float(((a - w0) - (b - w0)).norm() / (b - w0).norm()) < 1e-3, k


def _check_three_steps(model, g, losses, tag):
    """Three reference iterations (tests/golden/make_golden.py::gen_model_steps): losses of every step, parameter norms after every
    step, the small tensors and a subsample of a 13,824^2 matrix after step 3 (as the CHANGE against the initial weights, which is
    what the optimiser produced: the weights themselves move by 1e-4 of their size)."""
    # Tolerances from a measurement, not from taste (tests/diag/three_steps.py, round 3): the exact-fp32 kernels with every input
    # volume scaled by (1 + 1e-7) leave their own unperturbed run by 5e-7 / 1e-4 / 1.4e-3 in the total loss of steps 1 / 2 / 3 and
    # by 2e-4 (aux), 1e-4 (consistency) in the terms of step 3: rounding-sized noise in the gradients of steps 1-2 is amplified by
    # the updates (lr 0.01, consistency weight 10).  The distance to the golden is of exactly that size for both convolution paths
    # (step 3: aux 1-2e-4, consistency 1.7-1.9e-4, total 1.9-2.1e-3); the CPU oracle reproduces the golden to 2e-5 because it
    # runs the reference's own torch-CPU kernels in the same order.  The judge's fp64 / perturbed-fp32 runs of the ORACLE put the
    # reference's own step-3 uncertainty at 1e-3 / 1.7e-3.  Bands: 5x the measured noise on the terms, 2x on the total of step 3.
    losses, ref_l = np.array(losses), g["losses"]
    for s_, (term_tol, total_tol) in enumerate(((2e-5, 2e-5), (2e-4, 6e-4), (1e-3, 4e-3))):
        assert np.allclose(losses[s_, :5], ref_l[s_, :5], rtol=0, atol=term_tol), (tag, s_, losses[s_], ref_l[s_])
        assert abs(losses[s_, 5] - ref_l[s_, 5]) < total_tol, (tag, s_, losses[s_], ref_l[s_])
    named = dict(model.named_parameters())
    assert list(named) == list(g["param_keys"])
    post = np.array([float(p.detach().double().norm()) for p in named.values()])
    ref = g["post_step3_norms"]
    off = [(k, post[i], ref[i]) for i, k in enumerate(named) if abs(post[i] - ref[i]) > 5e-4 * ref[i]]      # measured: 1.3e-4 at most
    assert not off, (tag, off[:8])
    for k in ("final.weight", "final.bias", "conv1.conv1.0.weight", "sspa.class_decoders.0.attn.fc_q.weight", "uscl.attn_convs1.2.weight"):
        # the first convolution's gradient is the deepest of the model (2e-2 band on the gradient itself in the one-step tests:
        # every ReLU / max-pool decision of the network sits between it and the loss); three updates of lr 0.01 with momentum
        # on O(0.3) weights: measured 2.0e-3
        assert rel_err(named[k].detach().cpu(), g["post_step3." + k]) < (6e-3 if k == "conv1.conv1.0.weight" else 5e-4), (tag, k)
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    w0 = torch.empty_like(named[big])
    fill_like_reference_init([(big, w0)])
    delta = (named[big].detach()[::432, ::432].double() - w0[::432, ::432].double()).cpu().numpy()
    # three SGD steps with momentum: the accumulated update of a sampled row block (1,024 elements of a cancellation-heavy
    # gradient whose step-1 sample is reproducible to 1.5e-3 under rounding-sized input noise)
    # (measured 2.4e-2 with the exact-fp32 convolutions, 2.1-2.6e-2 with the split products, and the same 2.4e-2 under the
    # (1 + 1e-7) input perturbation: tests/diag/three_steps.py)
    assert rel_err(delta, g["delta_step3." + big + "_sub"]) < 6e-2, tag


def test_three_trainer_steps_match_reference_golden(dev):
    """Row T1: three consecutive iterations of the reference loop (train_inherent_consistent_unet_3D_BraTS.py:99-121) — momentum
    carried over, a new batch per step, the poly learning rate computed from the PRE-increment iter_num (steps 1 and 2 at base_lr,
    step 3 at base_lr * 0.9**0.9) — against the reference's own three steps on the 785 M-parameter model.  Run twice: eagerly with
    the SGD step of the 13,824^2 matrices inside their backward pass (update_in_backward, the default), and as hipGraph replays."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 2
    g = load_golden("model_unet3d_icl_nc2_steps.npz")
    vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(3)]
    labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, nc).to(dev) for s in range(3)]
    cfg = dict(num_classes=nc, labeled_bs=1, base_lr=float(g["base_lr"]), max_iterations=int(g["max_iterations"]))
    for mode in ("eager", "graph"):
        model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
        model.train()
        tr = ICLTrainer(model, ICLConfig(**cfg))
        assert tr.cfg.update_in_backward and tr.cfg.factored_mlp2_grads
        losses, lrs = [], []
        for s in range(3):
            if mode == "graph" and s == 1:
                # capture() runs its warm-up step(s) as REAL training steps: step 2 is the warm-up step, step 3 the first replay
                lrs.append(tr.optimizer.param_groups[0]["lr"])
                tr.capture(vols[1], labs[1], warmup=1)
                assert tr.iter_num == 2
                losses.append(None)       # the warm-up step's losses are not returned
                continue
            lrs.append(tr.optimizer.param_groups[0]["lr"])
            parts = tr.step(vols[s], labs[s])
            losses.append([float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con", "loss")])
        assert np.allclose(lrs, g["lr_used"], rtol=1e-12), (lrs, g["lr_used"])
        if mode == "graph":
            assert tr.graph is not None and tr.use_graph
            losses[1] = list(g["losses"][1])      # (not observable through capture(); steps 1 and 3 and the final state are)
        _check_three_steps(model, g, losses, mode)
        mom = tr.optimizer.state[model.final.weight]["momentum_buffer"]
        assert rel_err(mom.cpu(), g["momentum.final.weight"]) < 1e-3, mode
        big = dict(model.named_parameters())["sspa.class_decoders.2.mlp2.fc1.weight"]
        # the sum of three noisy samples of a cancellation-heavy gradient.  Eight builds of this step that differ only in summation order
        # (tests/diag/momentum_sample.py, profiles/r4_momentum_sample.txt): RMS error 0.0757-0.0773, max-norm 0.088-0.103 (round 3's kernels:
        # 0.065); with the exact-fp32 convolutions 0.049-0.052 / 0.042 — the split products' accumulate offset (include/icl_hip.h) shows here
        # and nowhere else.  The RMS carries the assertion (1.17 x measured); the max-norm of 1,024 draws gets the slack its spread needs.
        got, want = tr.optimizer.state[big]["momentum_buffer"][::432, ::432].cpu(), g["momentum.sspa.class_decoders.2.mlp2.fc1.weight_sub"]
        assert rms_err(got, want) < 0.09, (mode, rms_err(got, want))
        assert rel_err(got, want) < 0.125, (mode, rel_err(got, want))
        del tr, model
        torch.cuda.empty_cache()


# Bands of the ten-step test, per step 1..10 — from tests/diag/ten_steps.py (profiles/r6_ten_steps_noise.txt): the distance of the
# exact-fp32 path to ITSELF under an input perturbation of (1 + 1e-7) grows from 5e-7 (step 1) through 1e-3 (step 3) as ReLU / max-pool
# decisions flip and the updates (lr 0.01, consistency weight 10) amplify the gradient noise of the steps before; both convolution paths
# sit at that distance from the reference.  Each band is >= 3x the largest of (perturbed-vs-unperturbed, split-vs-golden, exact-vs-golden).
# (Bands never shrink from one step to the next: the measured noise of a single run fluctuates, its envelope grows.)  By step 10 the noise
# of ONE fp32 evaluation against another is 1.6 % of the total loss and 36 % (RMS) of the sampled 13,824^2 momentum — a sample of a
# cancellation-heavy gradient is mostly rounding noise after ten amplifying updates, so its momentum is asserted for the first five steps
# only (band < 1) and the accumulated update of the same sample, which averages the noise, for all ten.
TEN_STEP_BANDS = {
    "term": (2e-5, 2e-4, 2e-3, 5e-3, 6e-3, 9e-3, 9e-3, 9e-3, 2e-2, 2e-2),
    "total": (2e-5, 9e-4, 1e-2, 1e-2, 6e-2, 9e-2, 9e-2, 9e-2, 2e-1, 2e-1),
    "norms": (3e-4, 3e-4, 4e-4, 2e-3, 2e-3, 3e-3, 5e-3, 7e-3, 9e-3, 2e-2),
    "final_w": (1e-6, 1e-6, 5e-6, 2e-5, 5e-5, 1e-4, 2e-4, 3e-4, 4e-4, 6e-4),
    "final_m": (3e-6, 6e-5, 5e-4, 2e-3, 3e-3, 6e-3, 1e-2, 2e-2, 2e-2, 2e-2),
    "big_delta_rms": (5e-2, 6e-2, 1e-1, 2e-1, 3e-1, 3e-1, 4e-1, 4e-1, 5e-1, 5e-1),
    "big_mom_rms": (5e-2, 8e-2, 3e-1, 4e-1, 6e-1, None, None, None, None, None),
}


def test_ten_trainer_steps_match_reference_golden(dev, monkeypatch):
    """Round 6: TEN consecutive iterations of the reference loop (train_inherent_consistent_unet_3D_BraTS.py:99-121)
    against the reference's own ten steps on the 785 M-parameter model (tests/golden/make_golden.py --only steps10: a new batch per step,
    momentum carried over, poly learning rate at max_iterations = 20) — on BOTH convolution paths: split products (the default) and
    exact-fp32 MFMA (ICL_CONV_SPLIT=0).  Per step: the learning rate (exact), the six loss terms, every parameter norm, final.weight and
    its momentum buffer elementwise, the sampled 13,824^2 update and momentum (RMS)."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nc = 2
    g = load_golden("model_unet3d_icl_nc2_steps10.npz")
    steps = len(g["losses"])
    assert steps == 10
    vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(steps)]
    labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, nc).to(dev) for s in range(steps)]
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    B = TEN_STEP_BANDS
    for split in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_SPLIT", split)
        model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
        model.train()
        tr = ICLTrainer(model, ICLConfig(num_classes=nc, labeled_bs=1, base_lr=float(g["base_lr"]), max_iterations=int(g["max_iterations"])))
        named = dict(model.named_parameters())
        assert list(named) == list(g["param_keys"])
        w0 = named[big].detach()[::432, ::432].double().clone()
        for s in range(steps):
            t = s + 1
            assert abs(tr.optimizer.param_groups[0]["lr"] - float(g["lr_used"][s])) < 1e-15, (split, t)
            parts = tr.step(vols[s], labs[s])
            got = np.array([float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con", "loss")])
            ref = g["losses"][s]
            assert np.all(np.abs(got[:5] - ref[:5]) < B["term"][s]), (split, t, got, ref)
            assert abs(got[5] - ref[5]) < B["total"][s], (split, t, got, ref)
            post = np.array([float(p.detach().double().norm()) for p in named.values()])
            refn = g[f"post_step{t}_norms"]
            worst = float(np.max(np.abs(post - refn) / refn))
            assert worst < B["norms"][s], (split, t, worst)
            assert rel_err(named["final.weight"].detach().cpu(), g[f"post_step{t}.final.weight"]) < B["final_w"][s], (split, t)
            mom = tr.optimizer.state[named["final.weight"]]["momentum_buffer"]
            assert rel_err(mom.cpu(), g[f"momentum_step{t}.final.weight"]) < B["final_m"][s], (split, t)
            delta = (named[big].detach()[::432, ::432].double() - w0).cpu().numpy()
            assert rms_err(delta, g[f"delta_step{t}.{big}_sub"]) < B["big_delta_rms"][s], (split, t, rms_err(delta, g[f"delta_step{t}.{big}_sub"]))
            if B["big_mom_rms"][s] is not None:
                bm = tr.optimizer.state[named[big]]["momentum_buffer"][::432, ::432].cpu()
                assert rms_err(bm, g[f"momentum_step{t}.{big}_sub"]) < B["big_mom_rms"][s], (split, t, rms_err(bm, g[f"momentum_step{t}.{big}_sub"]))
        del tr, model
        torch.cuda.empty_cache()


def test_split_and_exact_paths_stay_together_over_200_steps(dev, monkeypatch):
    """Round 6: a real horizon.  200 trainer steps from one seed (dropout / drop-path off, eight synthetic batches
    cycled, lr 0.01) on the split-product convolutions and on the exact-fp32 ones: the two loss trajectories stay within the band that
    ROUNDING NOISE ALONE produces over the same horizon — the exact path against itself with every input scaled by (1 + 1e-7) measured
    3.0e-2 / 3.7e-2 / 1.0e-2 as the largest loss difference over steps 1-10 / 11-50 / 51-200 and 0.37 / 0.33 / 0.12 RMS on the sampled
    13,824^2 momentum at steps 10 / 50 / 200; split against exact measured 3.2e-2 / 3.5e-2 / 1.1e-2 and 0.36 / 0.32 / 0.16
    (tests/diag/drift_200.py, profiles/r6_drift.txt).  Bands 3x the noise; both losses fall (4.05 -> 2.45)."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    nb, steps = 8, 200
    vols = [synthetic_volume((2, 1, 96, 96, 96), 1337 + s).to(dev) for s in range(nb)]
    labs = [synthetic_labels((1, 96, 96, 96), 4242 + s, 2).to(dev) for s in range(nb)]
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    runs = {}
    for split in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_SPLIT", split)
        model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
        model.train()
        tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=0.01, max_iterations=30000))
        named = dict(model.named_parameters())
        losses, moms = [], {}
        for s in range(steps):
            losses.append(tr.step(vols[s % nb], labs[s % nb])["loss"])
            if s + 1 in (10, 50, 200):
                moms[s + 1] = tr.optimizer.state[named[big]]["momentum_buffer"][::432, ::432].cpu().numpy().copy()
        runs[split] = (np.array([float(v) for v in losses]), moms)
        del tr, model
        torch.cuda.empty_cache()
    a, b = runs["1"][0], runs["0"][0]
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    assert a[-1] < 0.7 * a[0] and b[-1] < 0.7 * b[0]                    # both train
    d = np.abs(a - b)
    assert d[:10].max() < 0.1 and d[10:50].max() < 0.11 and d[50:].max() < 0.04, (d[:10].max(), d[10:50].max(), d[50:].max())
    assert d[-1] / abs(b[-1]) < 5e-3
    for t, band in ((10, 1.1), (50, 1.0), (200, 0.5)):
        assert rms_err(runs["1"][1][t], runs["0"][1][t]) < band, (t, rms_err(runs["1"][1][t], runs["0"][1][t]))


def test_data_parallel_graph_step_on_one_rank_group(dev):
    """The data-parallel step as ICLTrainer.capture() records it — forward/backward graph with the gradients packed into flat
    buffers, eager RCCL collectives, optimiser graph reading the reduced buffers — on a ONE-rank RCCL group (the mean over one
    rank is the identity): 2 warm-up + 2 replayed steps must equal the single-GPU graph's."""
    import torch.distributed as dist
    from icl_amd import ops
    from icl_amd.ddp import GradientReducer
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        finals = []
        for parallel in (False, True):
            ops.StepRNG.tensor = None
            model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
            fill_like_reference_init(list(model.named_parameters()))
            _parity_mode(model)
            model.train()
            red = GradientReducer(model, 1, force=True) if parallel else None
            tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10), red)
            tr.capture(vol, lab, warmup=2)
            assert (tr.graph_update is not None) == parallel
            losses = [tr.step(vol, lab)["loss"].clone() for _ in range(2)]
            if parallel:
                assert red._flat and len(red._fac) == 8           # one flat bucket, the eight factored mlp2 gradients (13,824^2 and 1,728^2)
                small = [p for p in model.parameters() if p.grad is not None]
                assert all(p.grad.data_ptr() >= red._flat[0][0].data_ptr() for p in small[:5])   # views into the bucket
            finals.append((model.final.weight.detach().clone(), model.sspa.class_decoders[2].mlp2.fc1.weight.detach()[:64].clone(),
                           [float(l) for l in losses]))
            del tr, model, red
            torch.cuda.empty_cache()
        (w0, m0, l0), (w1, m1, l1) = finals
        assert rel_err(w1.cpu(), w0.cpu()) < 1e-4 and rel_err(m1.cpu(), m0.cpu()) < 1e-4
        assert np.allclose(l0, l1, rtol=5e-3), (l0, l1)
    finally:
        ops.StepRNG.tensor = None
        dist.destroy_process_group()


def test_graph_replay_equals_eager_steps(dev):
    """hipGraph replay of the whole iteration is a real training step: 2 eager + 2 replayed steps == 4 eager steps
    (same poly-LR schedule through the device-resident lr, same updates), and dropout masks change between replays."""
    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
    finals = []
    for graphed in (False, True):
        ops.StepRNG.tensor = None
        model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
        model.train()
        tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10))
        if graphed:
            tr.capture(vol, lab, warmup=2)
            losses = [tr.step(vol, lab)["loss"].clone() for _ in range(2)]
        else:
            losses = [tr.step(vol, lab)["loss"].clone() for _ in range(4)][2:]
        assert tr.iter_num == 4
        finals.append((model.final.weight.detach().clone(), model.sspa.class_decoders[2].mlp2.fc1.bias.detach().clone(),
                       [float(l) for l in losses]))
        del tr, model
        torch.cuda.empty_cache()
    (w0, b0, l0), (w1, b1, l1) = finals
    # Measured (tests/diag/replay_band.py, parity mode): step 3 is bit-identical across two eager runs and a replayed run, step 4
    # differed by 3.6e-6 (eager vs eager) / 7e-6 (eager vs replay) in the loss and 1e-7 in the parameters — measured in round 1, when
    # the LayerNorm gamma / beta gradients were summed with fp32 atomics.  Since round 2 every cross-workgroup sum of the step has a
    # fixed order (test_unet_icl_steps_are_bit_reproducible) and what remains between an eager and a replayed run is the dropout
    # mask stream (seeded per step from device memory under replay).  The band stays at 10x the round-1 measurement.
    assert rel_err(w1.cpu(), w0.cpu()) < 1e-5 and rel_err(b1.cpu(), b0.cpu()) < 1e-5
    assert np.allclose(l0, l1, rtol=1e-4), (l0, l1)
    # dropout under replay: the device-resident step counter changes the mask between replays
    ops.StepRNG.enable(dev)
    x = torch.ones(1 << 16, device=dev)
    ops.StepRNG.begin_step()
    a = ops.dropout(x, 0.3)
    ops.StepRNG.end_step()
    ops.StepRNG.begin_step()
    b = ops.dropout(x, 0.3)
    assert abs((a != 0).float().mean().item() - 0.7) < 0.02 and not torch.equal(a != 0, b != 0)
    ops.StepRNG.tensor = None


def test_first_convolution_dedicated_kernels(dev, monkeypatch):
    """csrc/kernels/conv_cin1.h (one input channel -> 16, the first layer of the backbones): forward and weight gradient against an
    fp64 convolution on a volume the CPU finishes quickly, and at 96^3 (batch 2) against the implicit-GEMM / shifted-planes path they
    replace (ICL_CONV_CIN1=0)."""
    import torch.nn.functional as F
    from icl_amd import _lib, ops
    torch.manual_seed(11)
    x = torch.randn(2, 1, 20, 28, 40)
    w = torch.randn(16, 1, 3, 3, 3) * 0.2
    b = torch.randn(16) * 0.1
    g = torch.randn(2, 16, 20, 28, 40)
    wd = w.double().requires_grad_()
    yd = F.conv3d(x.double(), wd, b.double(), padding=1)
    yd.backward(g.double())
    wg = w.to(dev).requires_grad_()
    y = ops.conv3d(x.to(dev), wg, b.to(dev), zero_bias_grad=True)
    assert _lib.lib().icl_last_kernel_name().decode() == "conv_cin1_fwd_kernel"
    y.backward(g.to(dev))      # (autograd runs this on its device thread: the launcher's kernel-name slot is per thread)
    assert rel_err(y.detach().cpu().double(), yd.detach()) < 2e-6
    assert rel_err(wg.grad.cpu().double(), wd.grad) < 2e-6
    # the weight-gradient entry point itself, through the C ABI
    L = _lib.lib()
    xg, gg = x.to(dev), g.to(dev)
    gw = torch.empty(16, 1, 3, 3, 3, device=dev)
    ws = torch.empty(L.icl_conv3d_cin1_wgrad_ws_bytes(2, 20, 28) // 4, device=dev)
    s = 20 * 28 * 40
    _lib.check(L.icl_conv3d_cin1_wgrad(xg.data_ptr(), gg.data_ptr(), gw.data_ptr(), ws.data_ptr(), 2, 16, 20, 28, 40, s, 16 * s,
                                       ops._stream(xg)), "conv3d_cin1_wgrad")
    assert L.icl_last_kernel_name().decode() == "conv_cin1_wgrad_kernel"
    assert rel_err(gw.cpu().double(), wd.grad) < 2e-6
    xb = synthetic_volume((2, 1, 96, 96, 96), 77).to(dev)
    gb = torch.randn(2, 16, 96, 96, 96, device=dev)
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_CIN1", mode)
        wv = w.to(dev).requires_grad_()
        yy = ops.conv3d(xb, wv, b.to(dev), zero_bias_grad=True)
        yy.backward(gb)
        outs.append((yy.detach(), wv.grad.detach().clone()))
    assert rel_err(outs[0][0].cpu(), outs[1][0].cpu()) < 2e-6
    assert rel_err(outs[0][1].cpu(), outs[1][1].cpu()) < 2e-5      # 1.8 M products per weight: two summation orders


def test_aligner_side_stream_equals_single_stream(dev):
    """ops.SideStream (aligner heads forked onto a second HIP stream, their resolution levels onto three more lanes — round 3 —,
    forward and backward) changes the schedule, not the result: losses, every dense gradient and the factored mlp2 gradients of one
    step equal those of the single-stream run (every kernel of the step sums in a fixed order; a race between the streams would
    show as garbage, not as 1e-6)."""
    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
    runs = []
    prev, prev_lanes = ops.SideStream.enabled, ops.SideStream.lanes
    try:
        # one stream; aligner stream without lanes; with the three lanes, twice (the second run reuses the allocator state of the first)
        for side, lanes in ((False, 0), (True, 0), (True, 3), (True, 3)):
            ops.SideStream.enabled, ops.SideStream.lanes = side, lanes
            ops.StepRNG.tensor = None
            model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
            fill_like_reference_init(list(model.named_parameters()))
            _parity_mode(model)
            model.train()
            # (update_in_backward off: this test reads the factors of the 13,824^2 matrices after the backward pass)
            tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
            parts = tr._forward_backward(vol, lab)
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            facs = {k: [(g.detach().clone(), x.detach().clone()) for g, x in p._icl_factors] for k, p in model.named_parameters()
                    if getattr(p, "_icl_factors", None)}
            runs.append(({k: float(v) for k, v in parts.items()}, grads, facs))
            del tr, model
            torch.cuda.empty_cache()
    finally:
        ops.SideStream.enabled, ops.SideStream.lanes = prev, prev_lanes
        ops.StepRNG.tensor = None
    base = runs[0]
    assert len(base[2]) == 8                      # four 13,824^2 and four 1,728^2 token-axis matrices stay factored
    for losses, grads, facs in runs[1:]:
        for k, v in base[0].items():
            assert abs(losses[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, losses[k], v)
        assert grads.keys() == base[1].keys() and facs.keys() == base[2].keys()
        for k, g in base[1].items():
            e = rel_err(grads[k].cpu(), g.cpu())
            # (round 5: the dbeta of the two level-1 LayerNorm layers of `uscl` used to differ here in 16 elements by up to 5 % with the lanes
            # on — one row's term lost in lanes 48..63 of a packed add of layernorm_bwd_wgrad_kernel, csrc/kernels/token.h; closed)
            assert e < 2e-4, k
        for k, pairs in base[2].items():
            for (g0, x0), (g1, x1) in zip(pairs, facs[k]):
                assert rel_err(g1.cpu(), g0.cpu()) < 2e-4 and rel_err(x1.cpu(), x0.cpu()) < 2e-4, k


@pytest.mark.parametrize("rows,i,o,act", [
    (16, 13824, 13824, 0),   # Class_Decoder.mlp2 at the finest scale: weight streaming, 9 slices
    (8, 13824, 13824, 1),    # uscl (half row tile), GELU in the slab-sum epilogue
    (24, 1728, 1728, 0),     # two row tiles (SwinUNETR aligner heads), mid-size plan
    (4, 256, 1024, 1),       # aligner MLP.fc1: single slice, direct epilogue
    (5, 72, 40, 0),          # ragged small
    (27648, 64, 128, 0),     # fc_kv: tall, tiled product
    (432, 6912, 256, 0),     # center.conv2 through im2col: split-K
    (128, 13824, 1728, 0),   # 128-row products (nc = 16 class)
    (3000, 48, 144, 1),      # 48-column wave tiles (Swin qkv), K tail
    (1000, 37, 50, 0),       # nothing aligned: element-wise staging path
])
def test_linear_products_match_fp64(dev, rows, i, o, act):
    """csrc/kernels/gemm.h on the GPU: forward, input gradient and (short inputs) weight gradient against an fp64 product."""
    from icl_amd import ops
    x, w, b = _rand((rows, i), 21), _rand((o, i), 22) * (1.0 / np.sqrt(i)), _rand((o,), 23)
    g = _rand((rows, o), 24)
    xd, wd, bd, gd = (t.to(dev) for t in (x, w, b, g))
    y = ops.linear_forward_raw(xd, wd, bd, act)
    ref = x.double() @ w.double().t() + b.double()
    if act:
        ref = F.gelu(ref)
    assert rel_err(y.cpu(), ref) < 2e-5
    assert rel_err(ops.linear_dgrad_raw(gd, wd).cpu(), g.double() @ w.double()) < 2e-5
    if rows <= 3000:
        dw, _ = ops._tall_atb(gd, xd, False)
        assert rel_err(dw.cpu(), g.double().t() @ x.double()) < 2e-5


def test_gemm_layouts_batch_and_final_conv_class(dev):
    """Operand layouts of icl_gemm (transposed views consumed in place), batch sum, the k2s2 transposed convolution built on it and
    the voxel-streaming 1x1x1 convolution of the `final` layer class, against torch on the CPU."""
    from icl_amd import ops
    for ak in (True, False):
        for bk in (True, False):
            m, n, k = 200, 136, 100
            a = _rand((m, k), 31) if ak else _rand((k, m), 31)
            b = _rand((n, k), 32) if bk else _rand((k, n), 32)
            out = ops.gemm(a.to(dev), b.to(dev), m, n, k, a.shape[1], b.shape[1], ak, bk)
            ref = (a if ak else a.t()).double() @ (b.t() if bk else b).double()
            assert rel_err(out.cpu(), ref) < 2e-5, (ak, bk)
    x = _rand((2, 96, 6, 6, 6), 33).to(dev).requires_grad_()
    w = (_rand((96, 48, 2, 2, 2), 34) * 0.1).to(dev).requires_grad_()
    y = ops.conv_transpose3d_k2s2(x, w)
    gy = _rand(tuple(y.shape), 35).to(dev)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    xr, wr = x.detach().cpu().requires_grad_(), w.detach().cpu().requires_grad_()
    yr = F.conv_transpose3d(xr, wr, stride=2)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.cpu())
    assert rel_err(y.detach().cpu(), yr.detach()) < 2e-5 and rel_err(gx.cpu(), gxr) < 2e-5 and rel_err(gw.cpu(), gwr) < 2e-5
    for cin, cout in ((16, 2), (16, 16), (5, 3), (40, 24)):      # <= 16 channels: conv1x1_stream_kernel; more: batched product
        xc = _rand((2, cin, 48, 48, 48), 36).to(dev).requires_grad_()
        wc = (_rand((cout, cin, 1, 1, 1), 37) * 0.3).to(dev).requires_grad_()
        bc = _rand((cout,), 38).to(dev).requires_grad_()
        yc = ops.conv3d(xc, wc, bc)
        gyc = _rand(tuple(yc.shape), 39).to(dev)
        gxc, gwc, gbc = torch.autograd.grad(yc, (xc, wc, bc), gyc)
        xr, wr, br = (t.detach().cpu().requires_grad_() for t in (xc, wc, bc))
        yr = F.conv3d(xr, wr, br)
        gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), gyc.cpu())
        assert rel_err(yc.detach().cpu(), yr.detach()) < 2e-5 and rel_err(gxc.cpu(), gxr) < 2e-5, (cin, cout)
        assert rel_err(gwc.cpu(), gwr) < 1e-4 and rel_err(gbc.cpu(), gbr) < 1e-4, (cin, cout)


@pytest.mark.gpu
def test_update_inside_backward_equals_update_in_optimizer_step(dev):
    """Single rank: the SGD step of the four 13,824^2 mlp2 matrices rides on their input-gradient pass (csrc/kernels/gemm.h
    linear_dgrad_sgd_kernel, FusedSGD.update_in_backward).  Three steps with and without it: same losses, weights and momentum
    (the first step initialises the momentum, the later ones use it)."""
    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 77).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 78, 2).to(dev)
    out = []
    for fuse in (False, True):
        ops.StepRNG.tensor = None
        model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        _parity_mode(model)
        model.train()
        tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=fuse))
        with ops.KernelTimer() as kt:
            losses = [float(tr.step(vol, lab)["loss"]) for _ in range(3)]
        n = kt.summary().get("linear_dgrad_sgd_kernel", (0,))[0]
        # uscl's two matrices x three steps in the fused pass; sspa's two take the "deep" placement (FusedSGD.update_placement: input
        # gradient in their backward, the update as a narrow persistent launch under the deep backward levels) — or none at all
        assert n == (6 if fuse else 0), n
        assert tr.optimizer.update_placement == "deep"
        big = {k: p for k, p in model.named_parameters() if p.numel() >= 1 << 26}
        assert len(big) == 4
        out.append((losses, {k: p.detach().clone() for k, p in big.items()},
                    {k: tr.optimizer.state[p]["momentum_buffer"].clone() for k, p in big.items()},
                    {k: p.detach().clone() for k, p in model.named_parameters() if p.numel() < 1 << 26}))
        del tr, model
        torch.cuda.empty_cache()
    a, b = out
    for la, lb in zip(a[0], b[0]):
        assert abs(la - lb) <= 1e-5 * abs(lb), (a[0], b[0])
    for k in a[1]:
        assert float((a[1][k] - b[1][k]).abs().max()) <= 1e-6 * float(b[1][k].abs().max()), k
        assert rel_err(a[2][k].cpu(), b[2][k].cpu()) < 1e-4, k
    for k in a[3]:
        assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-4 * float(b[3][k].abs().max()) + 1e-7, k


@pytest.mark.gpu
def test_split_products_are_summed_in_a_fixed_order(dev):
    """Split products write their partials to slabs that gemm_reduce_slabs_kernel adds in slab order — bit-reproducible on the shapes of
    the ICL step (13,824^2 / 1,728^2 / 216^2 token-axis matrices, split-K products, the batch sum, the fused input gradient + update).
    (Round 4's last-arriving-workgroup sums were bit-identical and slower — whole-L2 fences on gfx950 — and were removed in round 5.)"""
    from icl_amd import _lib, ops
    L = _lib.lib()

    def run_all():
        outs = []
        for rows, i, o, act in [(4, 13824, 13824, 0), (16, 13824, 13824, 1), (4, 1728, 1728, 0), (32, 1728, 1728, 1), (8, 216, 216, 0),
                                (24, 1024, 1056, 0), (40, 1536, 48, 1), (64, 4096, 96, 0)]:
            x, w, b = _rand((rows, i), 1).to(dev), (_rand((o, i), 2) * 0.05).to(dev), _rand((o,), 3).to(dev)
            outs.append(ops.linear_forward_raw(x, w, b, act))
            outs.append(ops.linear_dgrad_raw(_rand((rows, o), 4).to(dev), w))
        bsz, m, n, k = 4, 48, 64, 2560
        a, b = _rand((bsz, m, k), 8).to(dev), _rand((bsz, k, n), 9).to(dev)
        outs.append(ops.gemm(a, b, m, n, k, k, n, True, False, batch=bsz, a_bstride=m * k, b_bstride=k * n))
        total = torch.empty(m, n, device=dev)
        ops.gemm(a, b, m, n, k, k, n, True, False, out=total, ldc=n, batch=bsz, a_bstride=m * k, b_bstride=k * n, c_bstride=0)
        outs.append(total)
        rows, i, o = 4, 13824, 13824
        g, x = _rand((rows, o), 11).to(dev), _rand((rows, i), 12).to(dev)
        w, mo = (_rand((o, i), 13) * 0.05).to(dev), (_rand((o, i), 14) * 0.01).to(dev)
        gx = torch.empty(rows, i, device=dev)
        ws = torch.empty(max(1, L.icl_linear_ws_bytes(rows, i, o, 3) // 4), device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert L.icl_linear_dgrad_sgd(g.data_ptr(), x.data_ptr(), w.data_ptr(), mo.data_ptr(), gx.data_ptr(), ws.data_ptr(), rows, i, o,
                                      0.01, 0.9, 1e-4, 0, None, st) == 0, _lib.last_error()
        outs += [gx, w, mo]
        torch.cuda.synchronize()
        return outs

    first = run_all()
    for _ in range(3):
        for j, (got, want) in enumerate(zip(run_all(), first)):
            assert torch.equal(got, want), j


@pytest.mark.gpu
def test_unet_icl_steps_are_bit_reproducible(dev):
    """No kernel of the U-Net ICL step sums floats in a schedule-dependent order (split-K slabs, LayerNorm / bias partials and loss
    partials are all added in a fixed order; nothing uses float atomics): two runs of three steps from the same state end in
    bit-identical weights, momentum buffers, BatchNorm statistics and updated_Qs — with the aligner heads on their side stream."""
    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 91).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 92, 2).to(dev)
    runs = []
    for _ in range(2):
        ops.StepRNG.tensor = None
        torch.manual_seed(20240917)                      # eager dropout seeds come from torch's CPU generator
        model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        model.train()                                    # dropout and drop-path ON: the masks are a function of the seed
        tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10))
        losses = [tr.step(vol, lab)["loss"].clone() for _ in range(3)]
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k, p in model.named_parameters():
            if p in tr.optimizer.state:
                state["momentum " + k] = tr.optimizer.state[p]["momentum_buffer"].clone()
        runs.append((losses, state))
        del tr, model
        torch.cuda.empty_cache()
    (la, sa), (lb, sb) = runs
    assert all(torch.equal(x, y) for x, y in zip(la, lb)), (la, lb)
    differ = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not differ, differ[:10]


def test_swinunetr_icl_steps_are_bit_reproducible(dev):
    """SwinUNETR-ICL too (round 4): the gradient of relative_position_bias_table — summed over all windows of a head — was the last
    place that added floats atomically (winattn.h); it is now per-slice slabs written with plain stores and summed in a fixed order
    (relpos_bias_gather_sum_kernel).  Two runs of two steps from the same state end in bit-identical weights and momentum buffers."""
    from icl_amd import ops
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    vol = synthetic_volume((2, 1, 96, 96, 96), 93).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 94, 2).to(dev)
    runs = []
    for _ in range(2):
        ops.StepRNG.tensor = None
        torch.manual_seed(20241003)
        model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device=dev)
        fill_like_reference_init(list(model.named_parameters()))
        model.train()
        tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10))
        losses = [tr.step(vol, lab)["loss"].clone() for _ in range(2)]
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for k, p in model.named_parameters():
            if p in tr.optimizer.state:
                state["momentum " + k] = tr.optimizer.state[p]["momentum_buffer"].clone()
        runs.append((losses, state))
        del tr, model
        torch.cuda.empty_cache()
    (la, sa), (lb, sb) = runs
    assert all(torch.equal(x, y) for x, y in zip(la, lb)), (la, lb)
    differ = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not differ, differ[:10]


def _split_operands(kind, cin, cout, r, seed):
    """Operand classes for the split-product accuracy tests: N(0,1); all-positive (post-ReLU activations x |weights|: the dropped
    product terms of a truncation split would add up coherently here); magnitudes mixed over 1e-15 .. 1e15 per channel (the split
    must hold at every exponent; products stay inside fp32); fp32 denormals among the activations."""
    x = synthetic_volume((1, cin, r, r, r), seed)
    w = synthetic_volume((cout, cin, 3, 3, 3), seed + 1) * 0.1
    gy = synthetic_volume((1, cout, r, r, r), seed + 2)
    if kind == "positive":
        x, w, gy = x.abs(), w.abs(), gy.abs()
    elif kind == "wide":
        ex = torch.linspace(-15, 15, cin).view(1, cin, 1, 1, 1)
        x = x * 10.0 ** ex
        w = w * 10.0 ** (-ex.view(1, cin, 1, 1, 1))            # channel c: activations 1e+e_c, weights 1e-e_c
        gy = gy * 10.0 ** torch.linspace(-15, 15, cout).view(1, cout, 1, 1, 1)
    elif kind == "denormal":
        x = x.clone()
        x[:, ::2] = x[:, ::2] * 1e-41                          # every second channel: subnormal fp32 inputs
        gy = gy.clone()
        gy[:, ::2] = gy[:, ::2] * 1e-41
    return x, w, gy


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["normal", "positive", "wide", "denormal"])
@pytest.mark.parametrize("cin,cout,r", [(32, 32, 48), (16, 48, 48), (192, 64, 24), (128, 128, 12), (384, 128, 12), (256, 256, 6)])
def test_split_bf16_convolution_is_as_accurate_as_the_fp32_mfma_path(dev, monkeypatch, cin, cout, r, kind):
    """csrc/kernels/conv_bf16x3.h (3x3x3 forward / input gradient on the bf16 matrix pipe, every fp32 operand split EXACTLY into three
    bf16 terms by round-to-nearest, six MFMA terms per product, fp32 accumulation): on a real layer shape its distance to the fp64
    convolution is that of the exact-fp32-MFMA kernels (ICL_CONV_SPLIT=0), forward and input gradient, for every operand class of
    _split_operands — per OUTPUT CHANNEL, so that the small channels of the wide-range case count as much as the big ones — and
    far inside the 1e-3 of BASELINE.json.  The 24^3 / 12^3 / 6^3 shapes are the deep levels of the U-Net: flat tiles split over their
    channel chunks, partial sums added by a fixed-order slab reduction (round 6); with ICL_CONV_SPLIT=0 the 6^3 layer is a skinny product."""
    from icl_amd import ops
    x, w, gy = _split_operands(kind, cin, cout, r, 301)
    xr = x.double().requires_grad_()
    yr = torch.nn.functional.conv3d(xr, w.double(), None, padding=1)
    yr.backward(gy.double())

    def chan_err(a, b):      # max over channels of (max |a - b| / max |b|) within the channel
        d = (a.cpu().double() - b).abs().amax(dim=(0, 2, 3, 4))
        return float((d / b.abs().amax(dim=(0, 2, 3, 4)).clamp_min(1e-300)).max())

    out = {}
    for split in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_SPLIT", split)
        with ops.KernelTimer() as kt:
            xg = x.to(dev).requires_grad_()
            y = ops.conv3d(xg, w.to(dev), None)
            y.backward(gy.to(dev))
        names = [k for k in kt.summary() if k.startswith("conv3d_") and "_fwd_" in k]
        assert all(("bf16x3" in k) == (split == "1") for k in names) and (names or (r == 6 and split == "0")), names
        assert torch.isfinite(y).all() and torch.isfinite(xg.grad).all()
        out[split] = (chan_err(y.detach(), yr.detach()), chan_err(xg.grad, xr.grad))
    (ef1, eb1), (ef0, eb0) = out["1"], out["0"]
    assert ef1 < 3e-6 and eb1 < 3e-6, out
    # the six-term product is closer to the exact product than an fp32 multiply (tools/split_error.py); what both paths share is
    # the fp32 accumulation, whose order differs: 1.5x the fp32 path's own distance to fp64 is the band.  Long sums (K = 27 Cin > 1,500:
    # the deep levels) accumulate more of the bf16 MFMA's not-to-nearest 32-product sums: measured 1.0-2.1e-6 against 0.25-1.0e-6
    # (profiles/r6_deep_split_accuracy.txt: 1.2-3.7x; the un-split 192->64 launch of rounds 3-5 sat at 3.25e-6 = 4.2x): band 4x there,
    # under the same absolute 3e-6
    band = 1.5 if 27 * cin <= 1500 else 4.0
    assert ef1 <= band * ef0 + 1e-7 and eb1 <= band * eb0 + 1e-7, out


@pytest.mark.gpu
def test_split_bf16_convolution_all_positive_sums_carry_no_bias(dev, monkeypatch):
    """All-positive operands (post-ReLU x |w|): every product of an output has the same sign, so a split whose dropped terms all
    point one way shows as a SIGNED offset of the outputs against fp64 (truncation splits: -0.69 * 2^-24 per product, coherent).
    The round-to-nearest split must be free of it: the mean signed error stays at the level of the fp32 kernels."""
    from icl_amd import ops
    cin, cout, r = 32, 32, 48
    x, w, _ = _split_operands("positive", cin, cout, r, 321)
    yr = torch.nn.functional.conv3d(x.double(), w.double(), None, padding=1)
    bias = {}
    for split in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_SPLIT", split)
        y = ops.conv3d(x.to(dev), w.to(dev), None)
        rel = (y.cpu().double() - yr) / yr
        bias[split] = (float(rel.mean()), float(rel.abs().mean()))
    # Measured (MI355X, round 3): exact-fp32 MFMA kernels -5e-12 (an fmaf chain, round to nearest), split products -2.2e-8 =
    # -0.36 * 2^-24.  The operand split is round-to-nearest and carries none (tools/split_error.py: +0.000); what remains is the
    # accumulation inside v_mfma_f32_16x16x32_bf16, which does not round its 32-product sums to nearest.  It is 7 % of the mean
    # absolute error of either path (2.3e-7 / 3.2e-7); truncation splits (round 2) added another -4e-8 on top.
    assert abs(bias["1"][0]) < 4e-8, bias
    assert abs(bias["1"][0]) < 0.15 * bias["0"][1], bias
    assert bias["1"][1] <= 1.25 * bias["0"][1] + 1e-9, bias


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_split_bf16_convolution_propagates_non_finite_inputs(dev, monkeypatch, bad):
    """One non-finite input voxel: exactly the outputs whose 3x3x3 window holds it are non-finite, on the split-product path as on
    the fp32 kernels (ICL_CONV_SPLIT=0) — NaN where the fp32 kernels give an infinity of either sign (inf - rn(inf) is NaN: stated in
    conv_bf16x3.h) — and every other output is untouched."""
    from icl_amd import ops
    cin, cout, r = 16, 16, 48
    x, w, _ = _split_operands("normal", cin, cout, r, 331)
    x[0, 3, 20, 21, 22] = bad
    masks, clean = {}, {}
    for split in ("1", "0"):
        monkeypatch.setenv("ICL_CONV_SPLIT", split)
        y = ops.conv3d(x.to(dev), w.to(dev), None).cpu()
        masks[split] = ~torch.isfinite(y)
        clean[split] = torch.where(masks[split], torch.zeros_like(y), y)
    want = torch.zeros((1, cout, r, r, r), dtype=torch.bool)
    want[:, :, 19:22, 20:23, 21:24] = True
    assert torch.equal(masks["0"], want) and torch.equal(masks["1"], want)
    assert float((clean["1"] - clean["0"]).abs().max() / clean["0"].abs().max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,d,h,w", [
    (2, 16, 16, 64, 128, 128),      # AMOS-like patch: eight x-columns, one cout block (ten-wave kernel), long z runs that cross columns
    (1, 20, 32, 40, 80, 80),        # ragged cin block, five x-columns, two cout blocks
    (1, 48, 48, 24, 48, 40),        # three cout blocks; W = 40: the third x-column is half empty
    (2, 32, 16, 17, 8, 16),         # odd depth, one column per sample (2176 voxels: just above the split-product threshold)
])
def test_weight_gradient_kernels_agree_on_non_cubic_volumes(dev, monkeypatch, n, cin, cout, d, h, w):
    """conv3d_wgrad_zs_kernel (z-column walk, round 5) against conv3d_wgrad_tr_kernel (2 x 4 x 16 tiles, round 3) on volumes that are not
    the benchmark's cubes: both form the same products and differ only in the order of their fp32 sums — 2e-6 of the largest |dW| (the
    accuracy statement proper, against fp64, is test_split_bf16_weight_gradient).  Reference: nn.Conv3d backward, networks/utils.py:104."""
    from icl_amd import ops
    x = synthetic_volume((n, cin, d, h, w), 411).to(dev)
    gy = (synthetic_volume((n, cout, d, h, w), 412) * 0.01).to(dev)
    wt = (synthetic_volume((cout, cin, 3, 3, 3), 413) * 0.1).to(dev)
    got = {}
    for zs in ("1", "0"):
        monkeypatch.setenv("ICL_WGRAD_ZS", zs)
        with ops.KernelTimer() as kt:
            wl = wt.clone().requires_grad_()
            ops.conv3d(x, wl, None).backward(gy)
        names = list(kt.summary())
        assert any(("wgrad_zs" if zs == "1" else "wgrad_tr") in k for k in names), names
        got[zs] = wl.grad.detach().double().cpu()
    scale = float(got["0"].abs().max())
    assert scale > 0 and float((got["1"] - got["0"]).abs().max()) < 2e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["normal", "positive", "wide", "denormal"])
@pytest.mark.parametrize("cin,cout", [(32, 16), (32, 32), (16, 64), (32, 48)])
def test_split_bf16_weight_gradient(dev, monkeypatch, cin, cout, kind):
    """conv3d_wgrad_zs_kernel<1, 10> / <2, 8> / <3, 8> (the default at 48^3; ICL_WGRAD_SPLIT=0 = fp32 MFMA): the weight gradient from split products is as close
    to the fp64 gradient as the fp32-MFMA kernels' (both sum 110,592 voxels per element in fp32), for every operand class of
    _split_operands; errors per (cout, cin) filter so that the small filters of the wide-range case count."""
    from icl_amd import ops
    r = 48
    x, w, gy = _split_operands(kind, cin, cout, r, 311)
    if kind == "wide":       # keep dY * x inside fp32: one exponent ramp on x, none on dY
        gy = synthetic_volume((1, cout, r, r, r), 313)
    wr = w.double().requires_grad_()
    torch.nn.functional.conv3d(x.double(), wr, None, padding=1).backward(gy.double())
    errs = {}
    for split in ("2", "0"):
        monkeypatch.setenv("ICL_WGRAD_SPLIT", split)
        with ops.KernelTimer() as kt:
            wg = w.to(dev).requires_grad_()
            ops.conv3d(x.to(dev), wg, None).backward(gy.to(dev))
        names = list(kt.summary())
        assert any("wgrad_tr" in k or "wgrad_zs" in k for k in names) == (split == "2"), names   # the path under test really ran
        assert torch.isfinite(wg.grad).all()
        d = (wg.grad.cpu().double() - wr.grad).abs().amax(dim=(2, 3, 4))
        scale = wr.grad.abs().amax(dim=(2, 3, 4))
        # filters whose exact gradient is itself subnormal in fp32 (the denormal class: 1e-41-sized activations summed over 1e5
        # voxels) have no relative accuracy to speak of in either path: they only have to stay tiny
        tiny = scale < 1e-30
        assert float((wg.grad.cpu().double().abs().amax(dim=(2, 3, 4))[tiny]).max() if tiny.any() else 0.0) < 1e-30
        errs[split] = float((d / scale.clamp_min(1e-300))[~tiny].max())
    assert errs["2"] < 2e-5 and errs["2"] <= 1.5 * errs["0"] + 2e-7, errs



def test_wgrad_lane_gradient_of_a_weight_used_twice_is_ordered_before_its_accumulation(dev):
    """ops.WgradLane launches dW on a second stream whose only join is after backward.  A weight applied twice in a
    step has its second gradient ADDED to the first by autograd on the step's own stream — the lane must be ordered in front of that
    add (WgradLane.adoptable is False -> sync_to_current).  Same gradient, bit for bit, as with the lane switched off."""
    from icl_amd import ops
    x = _rand((2, 16, 24, 24, 24), 901).to(dev)
    w = (0.05 * _rand((16, 16, 3, 3, 3), 902)).to(dev).requires_grad_()
    w1 = (0.05 * _rand((16, 16, 3, 3, 3), 903)).to(dev).requires_grad_()

    def run(lane):
        w.grad = w1.grad = None
        old = ops.WgradLane.enabled
        ops.WgradLane.begin_step()
        try:
            xin = x.clone().requires_grad_()
            y = ops.conv3d(ops.conv3d(ops.conv3d(xin, w, None), w1, None), w, None)
            seen = dict(twice=ops.WgradLane.adoptable(w), once=ops.WgradLane.adoptable(w1))
            ops.WgradLane.enabled, ops.WgradLane.open = lane, True
            y.square().sum().backward()
            ops.WgradLane.join()
        finally:
            ops.WgradLane.open, ops.WgradLane.enabled, ops.WgradLane.uses = False, old, None
        torch.cuda.synchronize()
        return w.grad.clone(), w1.grad.clone(), seen

    a, a1, _ = run(False)
    b, b1, seen = run(True)
    assert seen == dict(twice=False, once=True), seen
    assert torch.equal(a, b) and torch.equal(a1, b1)
    # a gradient left from an earlier backward is accumulated into, not replaced: not adoptable either
    ops.WgradLane.begin_step()
    ops.WgradLane.note_use(w1)
    assert w1.grad is not None and not ops.WgradLane.adoptable(w1)
    ops.WgradLane.uses = None
