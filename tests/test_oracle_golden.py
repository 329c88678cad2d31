"""Pin the CPU oracle (oracle/icl_oracle.py) to golden vectors captured from the real reference.

Fixtures: tests/golden/unit.npz, model_unet3d_icl_nc{2,16}.npz, made by tests/golden/make_golden.py.
Tolerance: the oracle and the reference both run torch-CPU fp32 kernels, so they agree to
rounding; 2e-5 relative leaves room for thread-count dependent summation order.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from oracle import icl_oracle as O

TOL = 2e-5


def test_unetconv3_matches_reference(golden_unit):
    g = golden_unit
    p = O.make_params(O.unet_conv3_shapes("m.", 3, 4), requires_grad=True, strip="m.")
    x = synthetic_volume((2, 3, 8, 8, 8), 11).requires_grad_()
    y = O.unet_conv3(p, "m", x)
    y.backward(synthetic_volume(tuple(y.shape), 12))
    assert rel_err(y.detach(), g["conv3_y"]) < TOL
    assert rel_err(x.grad, g["conv3_gx"]) < TOL
    for k, t in p.items():
        gg = g["conv3_g." + k[2:]]
        if k.endswith("bias"):  # bias grads before an InstanceNorm are pure rounding noise
            assert np.abs(t.grad.numpy()).max() < 1e-4 and np.abs(gg).max() < 1e-4
        else:
            assert rel_err(t.grad, gg) < 1e-4, k


def test_up3ct_matches_reference(golden_unit):
    g = golden_unit
    p = O.make_params(O.unet_conv3_shapes("m.conv.", 12, 4), requires_grad=True, strip="m.")
    s = synthetic_volume((2, 4, 8, 8, 8), 13).requires_grad_()
    d = synthetic_volume((2, 8, 4, 4, 4), 14).requires_grad_()
    y = O.unet_up3_ct(p, "m", s, d)
    y.backward(synthetic_volume(tuple(y.shape), 15))
    assert rel_err(y.detach(), g["up_y"]) < TOL
    assert rel_err(s.grad, g["up_gskip"]) < 1e-4
    assert rel_err(d.grad, g["up_gdeep"]) < 1e-4


def test_plain_unet3d_32_matches_reference(golden_unit):
    g = golden_unit
    shapes = O.backbone_shapes(3, 2)
    assert [k for k, _ in shapes] == list(g["unet32_keys"])  # 38-key checkpoint contract
    assert len(shapes) == 38
    p = O.make_params(shapes, requires_grad=True)
    x = synthetic_volume((1, 2, 32, 32, 32), 16).requires_grad_()
    y, feats = O.backbone(p, x)
    y.backward(synthetic_volume(tuple(y.shape), 17))
    assert rel_err(y.detach(), g["unet32_y"]) < 1e-4
    # the input gradient has crossed every InstanceNorm of the net: torch's own CPU kernels move it by ~1e-3 with the number of
    # threads they sum over (0.7e-3 at 8 threads, 1.4e-3 at 4 against the golden run)
    assert rel_err(x.grad, g["unet32_gx"]) < 3e-3
    for k, t in p.items():
        ref = float(g["unet32_gn." + k])
        if k.endswith("bias") and k != "final.bias":
            continue
        got = float(t.grad.double().norm())
        assert abs(got - ref) <= 1e-3 * max(ref, 1e-6), k
    assert rel_err(p["final.weight"].grad, g["unet32_g.final.weight"]) < 1e-4


@pytest.mark.parametrize("tag,train", [("tr", True), ("ev", False)])
def test_aligner_matches_reference(golden_unit, tag, train):
    g = golden_unit
    chans, res, heads, nc = (32, 16, 8), (2, 4, 8), (4, 2, 1), 3
    p = O.make_params(O.aligner_shapes("a.", chans, res, nc, heads), requires_grad=True, strip="a.")
    p.update(O.aligner_buffers("a.", heads))
    feats = [synthetic_volume((2, 32, 2, 2, 2), 21).requires_grad_(),
             synthetic_volume((2, 16, 4, 4, 4), 22).requires_grad_(),
             synthetic_volume((2, 8, 8, 8, 8), 23).requires_grad_()]
    maps, qs = O.inherent_consistent(p, "a", feats, heads, None, "labeled", train)
    maps_u, qs_u = O.inherent_consistent(p, "a", feats, heads, qs, "unlabeled", train)
    loss = sum((m * synthetic_volume(tuple(m.shape), 30 + i)).sum() for i, m in enumerate(maps)) \
        + sum((m * synthetic_volume(tuple(m.shape), 40 + i)).sum() for i, m in enumerate(maps_u))
    loss.backward()
    for i in range(3):
        assert rel_err(maps[i].detach(), g[f"al_{tag}_map{i}"]) < 1e-4
        assert rel_err(qs[i].detach(), g[f"al_{tag}_q{i}"]) < 1e-4
        assert rel_err(maps_u[i].detach(), g[f"al_{tag}_mapu{i}"]) < 1e-4
        assert rel_err(qs_u[i].detach(), g[f"al_{tag}_qu{i}"]) < 1e-4
        assert rel_err(feats[i].grad, g[f"al_{tag}_gfeat{i}"]) < 1e-3
    none = set(g[f"al_{tag}_none"])
    for k, t in p.items():
        if not t.requires_grad:
            continue
        name = k[2:]
        if name in none:
            assert t.grad is None, name
            continue
        ref = float(g[f"al_{tag}_gn." + name])
        got = float(t.grad.double().norm())
        assert abs(got - ref) <= 2e-3 * max(ref, 1e-5), (name, got, ref)


def test_query_attention_reshape_quirk(golden_unit):
    g = golden_unit
    shapes = [("q.fc_q.weight", (8, 8)), ("q.fc_q.bias", (8,)), ("q.fc_kv.weight", (16, 8)),
              ("q.fc_kv.bias", (16,)), ("q.proj.weight", (8, 8)), ("q.proj.bias", (8,))]
    p = O.make_params(shapes, strip="q.")
    q = synthetic_volume((2, 3, 8), 51).requires_grad_()
    x = synthetic_volume((2, 8, 8), 52).requires_grad_()
    o, a = O.query_attention(p, "q", q, x, 2)
    (o.sum() + (a * synthetic_volume(tuple(a.shape), 53)).sum()).backward()
    assert rel_err(o.detach(), g["qa_o"]) < TOL
    assert rel_err(a.detach(), g["qa_a"]) < TOL
    assert rel_err(q.grad, g["qa_gq"]) < 1e-4
    assert rel_err(x.grad, g["qa_gx"]) < 1e-4


def test_losses_match_reference(golden_unit):
    g = golden_unit
    nc = 3
    lab = synthetic_labels((1, 96, 96, 96), 61, nc)
    logit = synthetic_volume((1, nc, 96, 96, 96), 62).requires_grad_()
    l1 = O.dice_loss(torch.softmax(logit, 1), lab.unsqueeze(1), nc)
    l2 = torch.nn.functional.cross_entropy(logit, lab)
    (l1 + l2).backward()
    assert abs(float(l1) - float(g["loss_dice"])) < 1e-6
    assert abs(float(l2) - float(g["loss_ce"])) < 1e-6
    assert rel_err(logit.grad[:, :, ::8, ::8, ::8], g["loss_dice_ce_glogit_sub"]) < 1e-4
    maps = [synthetic_volume((1, nc, r, r, r), 63 + i).requires_grad_() for i, r in enumerate((6, 12, 24))]
    la = O.aux_loss_3d(maps, lab, nc)
    la.backward()
    assert abs(float(la) - float(g["loss_aux"])) < 2e-6
    for i in range(3):
        assert rel_err(maps[i].grad, g[f"loss_aux_g{i}"]) < 1e-4
        maps[i].grad = None
    pred = synthetic_volume((1, nc, 96, 96, 96), 70).requires_grad_()
    lp = O.pseudo_soft_loss_3d(maps, pred)
    lp.backward()
    assert pred.grad is None  # detached target (losses.py:294)
    assert abs(float(lp) - float(g["loss_pse"])) < 2e-6
    for i in range(3):
        assert rel_err(maps[i].grad, g[f"loss_pse_g{i}"]) < 1e-4
        maps[i].grad = None
    maps_b = [synthetic_volume((1, nc, r, r, r), 73 + i).requires_grad_() for i, r in enumerate((6, 12, 24))]
    lc = O.softmax_mse_loss(maps, maps_b)
    lc.backward()
    assert maps_b[0].grad is None
    assert abs(float(lc) - float(g["loss_con"])) < 1e-7
    for i in range(3):
        assert rel_err(maps[i].grad, g[f"loss_con_g{i}"]) < 1e-4


@pytest.mark.slow
@pytest.mark.parametrize("nc", [2])
def test_full_model_step_matches_reference(nc):
    """SURVEY.md §8 T1 at BASELINE config 2 shape: forward 5-tuple, 5 losses, grad norms,
    grad-None set and one SGD step of the 785 M parameter model against the reference."""
    g = load_golden(f"model_unet3d_icl_nc{nc}.npz")
    shapes = O.unet_3d_icl_shapes(nc)
    assert [k for k, _ in shapes] == list(g["param_keys"])
    p = O.make_params(shapes, requires_grad=True)
    p.update(O.aligner_buffers("sspa.", O.UNET3D_HEADS))
    p.update(O.aligner_buffers("uscl.", O.UNET3D_HEADS))
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc)
    with torch.no_grad():
        y = O.unet_3d_icl_forward(p, vol[:1], inference=True)
    assert rel_err(y[:, :, ::8, ::8, ::8], g["inf_logits_sub"]) < 1e-4
    outs = O.unet_3d_icl_forward(p, vol[:1], vol[1:], training=True)
    assert rel_err(outs[0].detach()[:, :, ::8, ::8, ::8], g["final_lab_sub"]) < 1e-4
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            assert rel_err(t.detach(), g[f"{name}{i}"]) < 2e-4, (name, i)
    total, parts = O.icl_losses(outs, lab, nc)
    got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(total)]
    assert np.allclose(got, g["losses"], rtol=0, atol=1e-5), (got, g["losses"])
    total.backward()
    none = [k for k, _ in shapes if p[k].grad is None]
    assert none == list(g["grad_none"]) and len(none) == 33
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    for k, r in ref.items():
        if (k.endswith("bias") and ".0.bias" in k) or ("attn_convs1" in k and k.endswith("bias")):
            continue  # conv bias before InstanceNorm / before a class softmax: rounding noise
        got = float(p[k].grad.double().norm())
        assert abs(got - r) <= 2e-3 * max(r, 1e-7) + 1e-9, (k, got, r)
    assert rel_err(p["final.weight"].grad, g["grad.final.weight"]) < 1e-4
    bufs = {}
    O.sgd_step(p, {k: p[k].grad for k, _ in shapes}, bufs, lr=0.01)
    post = np.array([float(p[k].detach().double().norm()) for k, _ in shapes])
    assert np.allclose(post, g["post_sgd_norms"], rtol=1e-6)
    assert rel_err(p["final.weight"].detach(), g["post_sgd.final.weight"]) < 1e-6


@pytest.mark.slow
def test_three_trainer_steps_match_reference():
    """Row T1, three consecutive iterations of the reference loop (train_..._BraTS.py:99-121) on the 785 M-parameter model: the
    oracle's forward / losses / backward + sgd_step (momentum carried over) + poly_lr from the PRE-increment iter_num reproduce the
    reference's three loss vectors, the learning rates it used and its parameters after step 3."""
    g = load_golden("model_unet3d_icl_nc2_steps.npz")
    nc = 2
    shapes = O.unet_3d_icl_shapes(nc)
    names = [k for k, _ in shapes]
    assert names == list(g["param_keys"])
    p = O.make_params(shapes, requires_grad=True)
    p.update(O.aligner_buffers("sspa.", O.UNET3D_HEADS))
    p.update(O.aligner_buffers("uscl.", O.UNET3D_HEADS))
    base_lr, max_it = float(g["base_lr"]), int(g["max_iterations"])
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    w0 = p[big].detach()[::432, ::432].double().clone()
    bufs, lr, iter_num = {}, base_lr, 0
    for s in range(3):
        vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + s)
        lab = synthetic_labels((1, 96, 96, 96), 4242 + s, nc)
        for k in names:
            p[k].grad = None
        outs = O.unet_3d_icl_forward(p, vol[:1], vol[1:], training=True)
        total, parts = O.icl_losses(outs, lab, nc)
        got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(total)]
        assert np.allclose(got, g["losses"][s], rtol=0, atol=2e-5), (s, got, g["losses"][s])
        assert abs(lr - float(g["lr_used"][s])) < 1e-15
        total.backward()
        O.sgd_step(p, {k: p[k].grad for k in names}, bufs, lr=lr)
        lr = O.poly_lr(base_lr, iter_num, max_it)       # from the pre-increment iter_num, used from the next step
        iter_num += 1
        post = np.array([float(p[k].detach().double().norm()) for k in names])
        assert np.allclose(post, g[f"post_step{s + 1}_norms"], rtol=2e-6), s
    for k in ("final.weight", "final.bias", "conv1.conv1.0.weight", "sspa.class_decoders.0.attn.fc_q.weight", "uscl.attn_convs1.2.weight"):
        # element-wise after three updates: the weights moved by lr * (momentum sums of gradients that agree to ~1e-3), i.e. the
        # two fp32 CPU backward passes (reference modules / functional oracle, different summation orders) differ by ~5e-5 of
        # the weight scale on the first convolution, whose gradient sums 884,736 voxels per element
        assert rel_err(p[k].detach(), g["post_step3." + k]) < 2e-4, k
    delta = (p[big].detach()[::432, ::432].double() - w0).numpy()
    assert rel_err(delta, g["delta_step3." + big + "_sub"]) < 5e-3
    assert sorted(bufs) == sorted(g["momentum_keys"])
    assert rel_err(bufs["final.weight"], g["momentum.final.weight"]) < 1e-4


@pytest.mark.slow
def test_first_three_of_ten_trainer_steps_match_reference():
    """Round 6: the ten-step golden (tests/golden/make_golden.py --only steps10; max_iterations = 20, so the poly learning rate moves
    at every step from the third on).  The whole schedule `lr_used` is checked against the oracle's poly_lr; the oracle then runs the
    first THREE iterations (under a minute of CPU; the full ten are what the GPU test compares the HIP path with): losses, parameter norms
    after every step, the sampled 13,824^2 update and momentum after step 3."""
    g = load_golden("model_unet3d_icl_nc2_steps10.npz")
    nc = 2
    shapes = O.unet_3d_icl_shapes(nc)
    names = [k for k, _ in shapes]
    assert names == list(g["param_keys"])
    base_lr, max_it = float(g["base_lr"]), int(g["max_iterations"])
    want = [base_lr] + [O.poly_lr(base_lr, k, max_it) for k in range(len(g["lr_used"]) - 1)]
    assert np.allclose(want, g["lr_used"], rtol=1e-13, atol=0), (want, g["lr_used"])
    assert len(g["losses"]) == 10 and np.all(np.diff(g["losses"][:, 5]) < 0)      # the reference's total loss falls at every step
    p = O.make_params(shapes, requires_grad=True)
    p.update(O.aligner_buffers("sspa.", O.UNET3D_HEADS))
    p.update(O.aligner_buffers("uscl.", O.UNET3D_HEADS))
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    w0 = p[big].detach()[::432, ::432].double().clone()
    bufs, lr, iter_num = {}, base_lr, 0
    for s in range(3):
        vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + s)
        lab = synthetic_labels((1, 96, 96, 96), 4242 + s, nc)
        for k in names:
            p[k].grad = None
        outs = O.unet_3d_icl_forward(p, vol[:1], vol[1:], training=True)
        total, parts = O.icl_losses(outs, lab, nc)
        got = [float(parts[k]) for k in ("dice", "ce", "aux", "pse", "con")] + [float(total)]
        # (the oracle and the reference sum in different orders; the updates amplify that from step to step — measured at step 4: 2e-5 on
        # the consistency term, 2.4e-4 on the total, which weights it by 10)
        assert np.allclose(got[:5], g["losses"][s][:5], rtol=0, atol=(2e-5, 2e-5, 5e-5, 1e-4)[s]), (s, got, g["losses"][s])
        assert abs(got[5] - g["losses"][s][5]) < (2e-5, 5e-5, 3e-4, 1e-3)[s], (s, got, g["losses"][s])
        assert abs(lr - float(g["lr_used"][s])) < 1e-15
        total.backward()
        O.sgd_step(p, {k: p[k].grad for k in names}, bufs, lr=lr)
        lr = O.poly_lr(base_lr, iter_num, max_it)
        iter_num += 1
        post = np.array([float(p[k].detach().double().norm()) for k in names])
        ref = g[f"post_step{s + 1}_norms"]
        worst = float(np.max(np.abs(post - ref) / ref))
        assert worst < (2e-6, 2e-6, 4e-6, 5e-5)[s], (s, worst)      # (measured at step 4: 2.2e-5 — one small tensor whose update nearly cancels)
    assert rel_err(p["final.weight"].detach(), g["post_step3.final.weight"]) < 2e-4
    delta = (p[big].detach()[::432, ::432].double() - w0).numpy()
    assert rel_err(delta, g["delta_step3." + big + "_sub"]) < 1e-2
    assert rel_err(bufs["final.weight"], g["momentum_step3.final.weight"]) < 2e-4


@pytest.mark.slow
def test_2d_unet_icl_step_matches_reference():
    """BASELINE config 1: 2D U-Net ICL, 256x256, nc=4, batch 2+2 — the reference's own CPU-runnable case."""
    nc = 4
    g = load_golden("model_unet2d_icl_nc4.npz")
    shapes = O.unet_icl_2d_shapes(nc)
    assert [k for k, _ in shapes] == list(g["param_keys"])
    assert len([k for k in g["plain_keys"]]) == 136      # plain UNet checkpoint contract
    p = O.make_params(shapes, requires_grad=True)
    p.update(O.backbone2d_buffers())
    p.update(O.aligner_buffers("sspa.", O.UNET2D_HEADS))
    p.update(O.aligner_buffers("uscl.", O.UNET2D_HEADS))
    img = synthetic_volume((4, 1, 256, 256), 2024)
    lab = synthetic_labels((2, 256, 256), 2025, nc)
    outs = O.unet_icl_2d_forward(p, img[:2], img[2:], training=True)
    assert rel_err(outs[0].detach()[:, :, ::8, ::8], g["out_lab_sub"]) < 1e-4
    assert rel_err(outs[1].detach()[:, :, ::8, ::8], g["out_unlab_sub"]) < 1e-4
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            assert rel_err(t.detach()[:, :, ::st, ::st], g[f"{name}{i}_sub"]) < 2e-4, (name, i)
    total, parts = O.icl_losses_2d(outs, lab, nc)
    got = [float(parts[k].detach()) for k in ("ce", "dice", "aux", "pse", "con")] + [float(total.detach())]
    assert np.allclose(got, g["losses"], rtol=0, atol=2e-5), (got, g["losses"])
    total.backward()
    none = [k for k, _ in shapes if p[k].grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    for k, r in ref.items():
        if k.endswith("bias") and (".conv_conv.0." in k or ".conv_conv.4." in k or "attn_convs1" in k):
            continue  # conv bias in front of a batch-statistics BatchNorm / class softmax: rounding noise
        got_n = float(p[k].grad.double().norm())
        assert abs(got_n - r) <= 3e-3 * max(r, 1e-7) + 1e-9, (k, got_n, r)
    assert rel_err(p["decoder.out_conv.weight"].grad, g["grad.decoder.out_conv.weight"]) < 1e-4
    assert rel_err(p["encoder.in_conv.conv_conv.1.running_var"], g["buf.encoder.in_conv.conv_conv.1.running_var"]) < 1e-5


@pytest.mark.slow
def test_swinunetr_icl_step_matches_reference():
    """SwinUNETR-ICL, nc=2, on 64^3 volumes (the reference class with img_size 64: token grids 32^3 ..
    2^3 — padded + shifted and clipped windows —, aligner grids 4^3 / 8^3 / 16^3, 136 M parameters): hidden states of the vendored
    Swin encoder, forward 5-tuple, grad-None set, gradient norms, sampled gradients, one SGD step and the post-step inference
    logits.  (The 96^3 BASELINE shape of the same model is checked against its own reference golden on the GPU,
    tests/test_gpu_parity.py; a 96^3 oracle pass costs 1.5-5 minutes of CPU.)  AuxLoss3D / PseudoSoftLoss3D resize to a
    hard-coded 96^3, so the golden drives the aligner paths with plain quadratics instead (tests/golden/make_golden.py).
    The five MONAI blocks are a restatement on both sides: "parity unpinned" for row S5, see oracle/swin_oracle.py."""
    from oracle import swin_oracle as S
    nc, side = 2, 64
    g = load_golden("model_swinunetr_icl_64_nc2.npz")
    res = (side // 16, side // 8, side // 4)
    shapes = S.swinunetr_icl_shapes(nc, res=res)
    assert [k for k, _ in shapes] == list(g["param_keys"])
    assert [",".join(map(str, s)) for _, s in shapes] == list(g["param_shapes"])
    p = S.make_swin_params(nc, requires_grad=True, res=res)
    # parameters + relative_position_index + BatchNorm running stats
    assert set(p.keys()) == {k for k in g["keys"] if not k.endswith("num_batches_tracked")}
    vol = synthetic_volume((2, 1, side, side, side), 1337)
    lab = synthetic_labels((1, side, side, side), 4242, nc)
    with torch.no_grad():
        hs = S.swin_vit(p, vol[:1])
        for i, h in enumerate(hs):
            sub = h[:, ::max(1, h.shape[1] // 8), ::max(1, h.shape[2] // 6), ::max(1, h.shape[3] // 6), ::max(1, h.shape[4] // 6)]
            assert rel_err(sub, g[f"hidden{i}_sub"]) < 1e-4, i
    outs = S.swinunetr_icl_forward(p, vol[:1], vol[1:], training=True)
    assert rel_err(outs[0].detach()[:, :, ::8, ::8, ::8], g["final_lab_sub"]) < 1e-4
    assert rel_err(outs[1].detach()[:, :, ::8, ::8, ::8], g["final_unlab_sub"]) < 1e-4
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            assert rel_err(t.detach(), g[f"{name}{i}"]) < 2e-4, (name, i)
    soft = torch.softmax(outs[0], 1)
    l_ce = torch.nn.functional.cross_entropy(outs[0], lab)
    l_dice = O.dice_loss(soft, lab.unsqueeze(1), nc)
    l_aux = sum(t.pow(2).mean() for t in outs[2])
    l_pse = sum(t.pow(2).mean() for t in outs[3])
    l_con = O.softmax_mse_loss(outs[3], outs[4])
    total = l_dice + l_ce + l_aux + l_pse + 10 * l_con
    got = [float(v.detach()) for v in (l_dice, l_ce, l_aux, l_pse, l_con, total)]
    assert np.allclose(got, g["losses"], rtol=0, atol=2e-5), (got, g["losses"])
    total.backward()
    none = [k for k, _ in shapes if p[k].grad is None]
    assert none == list(g["grad_none"])
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, r in ref.items():
        if "attn_convs1" in k and k.endswith("bias"):
            continue
        got_n = float(p[k].grad.double().norm())
        if abs(got_n - r) > 2e-3 * max(r, 1e-7) + 1e-9:
            bad.append((k, got_n, r))
    assert not bad, bad[:10]
    for k in ("out.conv.conv.weight", "swinViT.patch_embed.proj.weight", "swinViT.layers4.0.blocks.0.attn.qkv.bias",
              "decoder5.transp_conv.conv.weight", "swinViT.layers1.0.blocks.1.attn.relative_position_bias_table"):
        gg = p[k].grad
        gg = gg if gg.numel() <= 8192 else gg.reshape(-1)[::97]
        assert rel_err(gg, g["grad." + k]) < 2e-4, k
    O.sgd_step(p, {k: p[k].grad for k, _ in shapes}, {}, lr=0.01)
    post = np.array([float(p[k].detach().double().norm()) for k, _ in shapes])
    assert np.allclose(post, g["post_sgd_norms"], rtol=1e-6)
    with torch.no_grad():             # the golden's inference logits are those of the model AFTER the SGD step
        y = S.swinunetr_icl_forward(p, vol[:1], inference=True)
    assert rel_err(y[:, :, ::8, ::8, ::8], g["inf_logits_sub"]) < 1e-4


@pytest.mark.slow
def test_swinunet2d_icl_step_matches_reference():
    """2-D Swin-UNet ICL (224^2, nc=4, batch 2+2) — encoder output, decoder features handed to the
    aligners, forward 5-tuple, losses, grad-None set (57: the aligners' unused proj/norm layers among them), gradient norms
    and one SGD step against the reference golden."""
    from oracle import swinunet2d_oracle as W
    nc = 4
    g = load_golden("model_swinunet2d_icl_nc4.npz")
    shapes = W.swinunet_icl_shapes(nc)
    assert [k for k, _ in shapes] == list(g["param_keys"])
    assert [",".join(map(str, s)) for _, s in shapes] == list(g["param_shapes"])
    p = W.make_params(nc, requires_grad=True)
    assert set(p.keys()) == {k for k in g["keys"] if not k.endswith("num_batches_tracked")}
    img = synthetic_volume((4, 1, 224, 224), 3024)
    lab = synthetic_labels((2, 224, 224), 3025, nc)
    with torch.no_grad():
        xe, skips = W.forward_features(p, img[:2].repeat(1, 3, 1, 1))
        assert rel_err(xe[:, ::7, ::32], g["enc_out"]) < 1e-4
        _, feats = W.forward_up_features(p, xe, skips)
        for i, t in enumerate(feats):
            assert rel_err(t[:, ::13, ::16], g[f"feat{i}_sub"]) < 1e-4, i
    outs = W.swinunet_icl_forward(p, img[:2], img[2:], training=True)
    assert rel_err(outs[0].detach()[:, :, ::8, ::8], g["out_lab_sub"]) < 1e-4
    assert rel_err(outs[1].detach()[:, :, ::8, ::8], g["out_unlab_sub"]) < 1e-4
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            assert rel_err(t.detach()[:, :, ::st, ::st], g[f"{name}{i}_sub"]) < 2e-4, (name, i)
    total, parts = W.icl_losses(outs, lab, nc)
    got = [float(parts[k].detach()) for k in ("ce", "dice", "aux", "pse", "con")] + [float(total.detach())]
    assert np.allclose(got, g["losses"], rtol=0, atol=2e-5), (got, g["losses"])
    total.backward()
    none = [k for k, _ in shapes if p[k].grad is None]
    assert none == list(g["grad_none"]) and len(none) == 57
    ref = dict(zip(g["grad_norm_keys"], g["grad_norms"]))
    bad = []
    for k, r in ref.items():
        if "attn_convs1" in k and k.endswith("bias"):
            continue
        got_n = float(p[k].grad.double().norm())
        if abs(got_n - r) > 3e-3 * max(r, 1e-7) + 1e-9:
            bad.append((k, got_n, r))
    assert not bad, bad[:10]
    O.sgd_step(p, {k: p[k].grad for k, _ in shapes}, {}, lr=0.01)
    post = np.array([float(p[k].detach().double().norm()) for k, _ in shapes])
    assert np.allclose(post, g["post_sgd_norms"], rtol=1e-6)
    with torch.no_grad():
        y = W.swinunet_icl_forward(p, img[:2], inference=True)
    assert rel_err(y[:, :, ::8, ::8], g["inf_logits_sub"]) < 1e-4
