"""CPU checks of the SwinUNETR host logic and its kernels through tests/hipemu.

The window/shift/padding/merging plumbing of icl_amd/networks/swinunetr.py is compared with the oracle restatement
(oracle/swin_oracle.py, itself pinned to the reference by tests/golden/model_swinunetr_icl_nc2.npz) on small volumes
where clipped windows (n < 343), padded windows and shifted windows all occur.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
from build_emu import build_emu  # noqa: E402

from conftest import rel_err  # noqa: E402
from icl_amd import _lib, ops  # noqa: E402
from icl_amd.networks import swinunetr as SW  # noqa: E402
from icl_amd.utils.hashfill import synthetic_volume  # noqa: E402
from oracle import icl_oracle as O  # noqa: E402
from oracle import swin_oracle as S  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emu_library():
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    yield
    _lib._use_library_for_tests(None)


def _rand(shape, seed, grad=False):
    t = synthetic_volume(tuple(shape), seed)
    return t.requires_grad_() if grad else t


def _load(module, p, strip=""):
    sd = module.state_dict()
    with torch.no_grad():
        for k, t in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            t.copy_(p[strip + k])


@pytest.mark.parametrize("act", [0, 2])
def test_instance_norm_residual_act(act):
    x, r = _rand((2, 3, 4, 6, 8), 11, True), _rand((2, 3, 4, 6, 8), 12, True)
    gy = _rand((2, 3, 4, 6, 8), 13)
    y = ops.instance_norm_add_act(x, r, act)
    y.backward(gy)
    xr, rr = x.detach().clone().requires_grad_(), r.detach().clone().requires_grad_()
    yr = F.instance_norm(xr, eps=1e-5) + rr
    yr = F.leaky_relu(yr, 0.01) if act == 2 else yr
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-4 and rel_err(r.grad, rr.grad) < 1e-6
    # odd row length: scalar path of the forward kernel
    x2, r2 = _rand((1, 2, 3, 3, 3), 14), _rand((1, 2, 3, 3, 3), 15)
    assert rel_err(ops.instance_norm_add_act(x2, r2, 2), F.leaky_relu(F.instance_norm(x2, eps=1e-5) + r2, 0.01)) < 1e-5
    y3 = ops.instance_norm_act(x2, 2)
    assert rel_err(y3, F.leaky_relu(F.instance_norm(x2, eps=1e-5), 0.01)) < 1e-5


def test_layernorm_without_affine():
    x = (_rand((2, 3, 3, 3, 24), 21) * 2 + 0.3).requires_grad_()
    gy = _rand((2, 3, 3, 3, 24), 22)
    y = ops.layer_norm(x, None, None)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    yr = F.layer_norm(xr, [24])
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5 and rel_err(x.grad, xr.grad) < 1e-4


def test_conv_transpose_k2s2():
    x, w = _rand((2, 5, 3, 4, 2), 31, True), _rand((5, 7, 2, 2, 2), 32, True)
    gy = _rand((2, 7, 6, 8, 4), 33)
    y = ops.conv_transpose3d_k2s2(x, w)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    yr = F.conv_transpose3d(xr, wr, stride=2)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-5 and rel_err(w.grad, wr.grad) < 1e-5


@pytest.mark.parametrize("dims,shifted", [((14, 14, 14), True), ((7, 14, 7), False), ((4, 4, 4), False)])
def test_window_attention_matches_oracle(dims, shifted):
    """ops.window_attention (region ids) == oracle window_attention (dense 0/-100 mask), incl. clipped windows."""
    heads, c, b = 3, 12, 2
    ws, ss = SW.get_window_size(dims, (7, 7, 7), (3, 3, 3))
    n = ws[0] * ws[1] * ws[2]
    nw = (dims[0] // ws[0]) * (dims[1] // ws[1]) * (dims[2] // ws[2])
    p = O.make_params([("a.qkv.weight", (3 * c, c)), ("a.qkv.bias", (3 * c,)), ("a.proj.weight", (c, c)), ("a.proj.bias", (c,)),
                       ("a.relative_position_bias_table", (2197, heads))])
    p["a.relative_position_index"] = S.relative_position_index()
    x = _rand((b * nw, n, c), 41)
    mask = S.compute_mask(dims, ws, ss) if shifted else None
    want = S.window_attention(p, "a", x, mask, heads)
    m = SW.WindowAttention(c, heads, (7, 7, 7))
    _load(m, p, "a.")
    regions = SW.window_regions(dims, ws, ss, "cpu") if shifted else None
    got = m(x, regions)
    assert rel_err(got.detach(), want) < 1e-5
    if shifted:
        dense = (regions.unsqueeze(1) != regions.unsqueeze(2)).float() * -100.0
        assert torch.equal(dense, mask)


@pytest.mark.parametrize("size", [32, (32, 64, 32)])
def test_swin_transformer_matches_oracle(size):
    """swinViT on 32^3 (stages 16/8/4/2: padded+shifted 7^3 windows, then clipped unshifted windows) and on a
    non-cubic volume, feature_size 12, forward hidden states and parameter gradients."""
    f = 12
    size = (size,) * 3 if isinstance(size, int) else size
    p = O.make_params(S.swin_vit_shapes("swinViT.", 1, f), requires_grad=True)
    p.update(S.swin_buffers())
    m = SW.SwinTransformer(1, f, (7, 7, 7), (2, 2, 2, 2), (3, 6, 12, 24))
    _load(m, p, "swinViT.")
    x = _rand((1, 1) + size, 51)
    want = S.swin_vit(p, x)
    got = m(x, True)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape
        assert rel_err(g.detach(), w.detach()) < 2e-4, i
    gs = [_rand(w.shape, 60 + i) for i, w in enumerate(want)]
    sum((g * t).sum() for g, t in zip(got, gs)).backward()
    sum((w * t).sum() for w, t in zip(want, gs)).backward()
    bad = []
    for k, t in m.named_parameters():
        r = p["swinViT." + k].grad
        if r is None:
            assert t.grad is None or float(t.grad.abs().max()) == 0.0, k
            continue
        if rel_err(t.grad, r) > 2e-3:
            bad.append((k, rel_err(t.grad, r)))
    assert not bad, bad


def test_state_dict_keys_match_reference():
    """Key-for-key the reference's SwinUNETR_icl state_dict (golden 'keys'), built on the meta device (844 M parameters)."""
    from conftest import load_golden
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    g = load_golden("model_swinunetr_icl_nc2.npz")
    m = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device="meta")
    assert list(m.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in m.named_parameters()] == list(g["param_keys"])
    assert [",".join(map(str, t.shape)) for _, t in m.named_parameters()] == list(g["param_shapes"])
