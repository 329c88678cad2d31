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


@pytest.mark.parametrize("rows,o,i,bias", [(2500, 144, 48, True), (2049, 40, 8, True), (4100, 96, 200, False)])
def test_linear_tall_weight_gradient(rows, o, i, bias):
    """ops.linear backward on the tall path (>= 2048 rows): dW / db from csrc/kernels/linear_wgrad.h; ragged last row group,
    output blocks that are not multiples of 48, several row slices."""
    x = _rand((2, rows // 2 + rows % 2, i), 81)[:, :, :].reshape(-1, i)[:rows].clone().requires_grad_()
    w = (_rand((o, i), 82) * 0.3).requires_grad_()
    b = _rand((o,), 83).requires_grad_() if bias else None
    gy = _rand((rows, o), 84)
    y = ops.linear(x, w, b)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    br = b.detach().clone().requires_grad_() if bias else None
    yr = F.linear(xr, wr, br)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5 and rel_err(x.grad, xr.grad) < 1e-5
    assert rel_err(w.grad, wr.grad) < 1e-5
    if bias:
        assert rel_err(b.grad, br.grad) < 1e-5


def test_conv_transpose_k2s2():
    x, w = _rand((2, 5, 13, 10, 8), 31, True), _rand((5, 7, 2, 2, 2), 32, True)     # 2080 rows: tall dW path
    gy = _rand((2, 7, 26, 20, 16), 33)
    y = ops.conv_transpose3d_k2s2(x, w)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    yr = F.conv_transpose3d(xr, wr, stride=2)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-5 and rel_err(w.grad, wr.grad) < 1e-5
    # fused with the concat of UnetrUpBlock: [up | skip]
    skip = _rand((2, 3, 26, 20, 16), 34, True)
    x2, w2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    gy2 = _rand((2, 10, 26, 20, 16), 35)
    z = ops.conv_transpose3d_k2s2(x2, w2, skip)
    z.backward(gy2)
    sr = skip.detach().clone().requires_grad_()
    xr2, wr2 = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    zr = torch.cat((F.conv_transpose3d(xr2, wr2, stride=2), sr), 1)
    zr.backward(gy2)
    assert rel_err(z.detach(), zr.detach()) < 1e-5 and torch.equal(skip.grad, sr.grad)
    assert rel_err(x2.grad, xr2.grad) < 1e-5 and rel_err(w2.grad, wr2.grad) < 1e-5


@pytest.mark.parametrize("dims,shifted", [((14, 14, 14), True), ((7, 14, 7), False), ((4, 4, 4), False)])
def test_window_attention_matches_oracle(dims, shifted):
    """ops.window_attention (region ids) == oracle window_attention (dense 0/-100 mask), incl. clipped windows."""
    heads, c, b = 3, 48, 2
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


def _window_attention_torch(qkv, table, index, regions, heads, scale):
    """WindowAttention.forward core written with torch ops (swinunetr_icl.py:728-747)."""
    b_, n, c3 = qkv.shape
    c = c3 // 3
    bias = table[index[:n, :n].reshape(-1)].reshape(n, n, heads).permute(2, 0, 1)
    q, k, v = qkv.view(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if regions is not None:
        nw = regions.shape[0]
        mask = (regions.unsqueeze(1) != regions.unsqueeze(2)).to(attn.dtype) * -100.0
        attn = (attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    return (attn.softmax(-1) @ v).transpose(1, 2).reshape(b_, n, c)


@pytest.mark.parametrize("dims,shifted,batch,heads", [((14, 14, 14), True, 1, 2), ((7, 7, 7), False, 3, 1), ((6, 6, 6), False, 2, 3),
                                                     ((4, 4, 4), False, 1, 2), ((7, 14, 7), True, 2, 1)])
@pytest.mark.parametrize("online", ["0", "2"])
def test_window_attention_kernel_fwd_bwd(monkeypatch, dims, shifted, batch, heads, online):
    """The fused MFMA kernels (csrc/kernels/winattn.h) against the torch formula: output, dqkv and the bias gradient summed
    over all windows; n = 343 (22 key blocks, 7 padded keys), 216 and 64, with and without the shift mask.  online: the forward with
    the whole score row block in registers (0) / with the online softmax over chunks of four key blocks (2 = forced: the launcher
    only picks it for >= 512 workgroups); the backward kernels consume the log-sum-exp of either."""
    monkeypatch.setenv("ICL_WINATTN_ONLINE", online)
    ws, ss = SW.get_window_size(dims, (7, 7, 7), (3, 3, 3))
    n = ws[0] * ws[1] * ws[2]
    nw = (dims[0] // ws[0]) * (dims[1] // ws[1]) * (dims[2] // ws[2])
    c = heads * 16
    qkv = (_rand((batch * nw, n, 3 * c), 71) * 1.5).requires_grad_()
    table = _rand((2197, heads), 72).requires_grad_()
    index = S.relative_position_index()
    regions = SW.window_regions(dims, ws, ss, "cpu") if shifted else None
    gy = _rand((batch * nw, n, c), 73)
    y = ops.window_attention(qkv, table, index, regions, heads, 0.25)
    y.backward(gy)
    qr, tr = qkv.detach().clone().requires_grad_(), table.detach().clone().requires_grad_()
    yr = _window_attention_torch(qr, tr, index, regions, heads, 0.25)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(qkv.grad, qr.grad) < 1e-4
    assert rel_err(table.grad, tr.grad) < 1e-4


@pytest.mark.parametrize("heads,batch,masked", [(2, 3, True), (1, 2, False)])
def test_window_attention_kernel_head_dim_32(heads, batch, masked):
    """2-D Swin-UNet windows (networks/swinunet_icl.py:120-155): 7 x 7 tokens (n = 49, 15 padded keys), head dim 32, 169-row
    bias table, with and without a shift mask (4 windows of a 14 x 14 image) — output, dqkv and table gradient."""
    from oracle import swinunet2d_oracle as W
    n, c, nw = 49, heads * 32, 4
    qkv = (_rand((batch * nw, n, 3 * c), 171) * 1.5).requires_grad_()
    table = _rand((169, heads), 172).requires_grad_()
    index = W.relative_position_index()
    mask = W.attn_mask(14, 7, 3)
    regions = None
    if masked:   # region ids whose pairwise (in)equality reproduces the reference's 0 / -100 mask
        img = torch.zeros((1, 14, 14, 1))
        cnt = 0
        for hs in (slice(0, -7), slice(-7, -3), slice(-3, None)):
            for ws in (slice(0, -7), slice(-7, -3), slice(-3, None)):
                img[:, hs, ws, :] = cnt
                cnt += 1
        regions = W.window_partition(img, 7).view(-1, 49).to(torch.int32).contiguous()
        assert torch.equal((regions.unsqueeze(1) != regions.unsqueeze(2)).float() * -100.0, mask)
    gy = _rand((batch * nw, n, c), 173)
    y = ops.window_attention(qkv, table, index, regions, heads, 32 ** -0.5)
    y.backward(gy)
    qr, tr = qkv.detach().clone().requires_grad_(), table.detach().clone().requires_grad_()
    yr = _window_attention_torch(qr, tr, index, regions, heads, 32 ** -0.5)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(qkv.grad, qr.grad) < 1e-4
    assert rel_err(table.grad, tr.grad) < 1e-4


@pytest.mark.parametrize("grid", [(8, 8, 8), (5, 9, 5), (2, 2, 2)])
def test_swin_stage_matches_oracle(grid):
    """One BasicLayer (two blocks + patch merging) on small token grids: 8^3 (padded to 14^3, shifted 7^3 windows with the
    region mask), 5x9x5 (windows clipped to 5x7x5, shift on one axis only, odd sizes -> the PatchMerging padding path; the
    reference's pad argument order only works when all three sizes have the same parity) and 2^3
    (one clipped window): forward and every parameter gradient against the oracle."""
    c, heads = 48, 3
    shapes = [(k, sh) for k, sh in S.swin_vit_shapes("swinViT.", 1, c) if k.startswith("swinViT.layers1.0.")]
    p = O.make_params(shapes, requires_grad=True)
    p.update({k: v for k, v in S.swin_buffers().items() if k.startswith("swinViT.layers1.0.")})
    m = SW.BasicLayer(c, 2, heads, (7, 7, 7))
    _load(m, p, "swinViT.layers1.0.")
    x = _rand((2, c) + grid, 51)
    if any(g % 2 for g in grid):
        # the oracle restates only the even-size path of PatchMerging; compare the blocks and check the merge separately
        xo = x.permute(0, 2, 3, 4, 1)
        mask = S.compute_mask([-(-g // w) * w for g, w in zip(grid, SW.get_window_size(grid, (7, 7, 7)))],
                              *SW.get_window_size(grid, (7, 7, 7), (3, 3, 3)))
        for i in range(2):
            xo = S.swin_block(p, f"swinViT.layers1.0.blocks.{i}", xo, mask, heads, (7, 7, 7), (0, 0, 0) if i == 0 else (3, 3, 3))
        xg = x.permute(0, 2, 3, 4, 1).contiguous()
        regions = SW.window_regions([-(-g // w) * w for g, w in zip(grid, SW.get_window_size(grid, (7, 7, 7)))],
                                    *SW.get_window_size(grid, (7, 7, 7), (3, 3, 3)), "cpu")
        for blk in m.blocks:
            xg = blk(xg, regions)
        assert rel_err(xg.detach(), xo.detach()) < 2e-5
        d, h, w = grid
        xp = F.pad(xo.detach(), (0, 0, 0, d % 2, 0, w % 2, 0, h % 2))      # swinunetr_icl.py:951, argument order as written
        cat = torch.cat([xp[:, i::2, j::2, k::2, :] for i, j, k in SW.PatchMerging.SLICES], -1)
        want = F.linear(F.layer_norm(cat, (8 * c,), p["swinViT.layers1.0.downsample.norm.weight"], p["swinViT.layers1.0.downsample.norm.bias"]),
                        p["swinViT.layers1.0.downsample.reduction.weight"])
        assert rel_err(m.downsample(xo.detach()).detach(), want.detach()) < 2e-5
        return
    want = S.basic_layer(p, "swinViT.layers1.0", x, heads)
    got = m(x.permute(0, 2, 3, 4, 1).contiguous()).permute(0, 4, 1, 2, 3)
    assert rel_err(got.detach(), want.detach()) < 2e-5
    gy = _rand(want.shape, 60)
    (got * gy).sum().backward()
    (want * gy).sum().backward()
    bad = []
    for k, t in m.named_parameters():
        r = p["swinViT.layers1.0." + k].grad
        if rel_err(t.grad, r) > 1e-3:
            bad.append((k, rel_err(t.grad, r)))
    assert not bad, bad


@pytest.mark.parametrize("grid", [(4, 6, 2), (8, 8, 8)])
def test_patch_merging_gather_matches_slices(grid):
    """PatchMerging's concatenation of eight strided slices (two repeated, two missing) as one row gather; its gradient as one
    gather-sum — against the slices themselves under autograd."""
    d, h, w = grid
    c = 12
    x = _rand((2, d, h, w, c), 71, True)
    idx, back = SW.merge_index(grid, SW.PatchMerging.SLICES, "cpu")
    y = ops.gather_rows_dup(x.reshape(2, d * h * w, c), idx, back).view(2, d // 2, h // 2, w // 2, 8 * c)
    xr = x.detach().clone().requires_grad_()
    yr = torch.cat([xr[:, i::2, j::2, k::2, :] for i, j, k in SW.PatchMerging.SLICES], -1)
    assert torch.equal(y.detach(), yr.detach())
    gy = _rand(tuple(yr.shape), 72)
    y.backward(gy)
    yr.backward(gy)
    assert rel_err(x.grad, xr.grad) < 1e-6


def test_patch_embed_and_hidden_state_norm():
    p = O.make_params([("proj.weight", (48, 1, 2, 2, 2)), ("proj.bias", (48,))])
    m = SW.PatchEmbed(1, 48)
    _load(m, p)
    x = _rand((2, 1, 8, 12, 4), 52)
    want = F.conv3d(x, p["proj.weight"], p["proj.bias"], stride=2)
    with torch.no_grad():
        got = m(x)
        assert rel_err(got.permute(0, 4, 1, 2, 3), want) < 1e-5
        assert rel_err(SW.SwinTransformer.proj_out(got, True), S.proj_out(want)) < 1e-5


def test_state_dict_keys_match_reference():
    """Key-for-key the reference's SwinUNETR_icl state_dict (golden 'keys'), built on the meta device (844 M parameters)."""
    from conftest import load_golden
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    g = load_golden("model_swinunetr_icl_nc2.npz")
    m = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device="meta")
    assert list(m.state_dict().keys()) == list(g["keys"])
    assert [k for k, _ in m.named_parameters()] == list(g["param_keys"])
    assert [",".join(map(str, t.shape)) for _, t in m.named_parameters()] == list(g["param_shapes"])


# ------------------------------------------------------------------------------------------ 2-D Swin-UNet (row f4)
def test_swinunet2d_state_dict_keys_match_reference():
    from conftest import load_golden
    from icl_amd.networks.vision_transformer import SwinUnet
    g = load_golden("model_swinunet2d_icl_nc4.npz")
    m = SwinUnet(None, 224, 4, device="meta")
    assert list(m.state_dict().keys()) == list(g["keys"])            # incl. attn_mask / relative_position_index buffers
    assert [k for k, _ in m.named_parameters()] == list(g["param_keys"])
    assert [",".join(map(str, t.shape)) for _, t in m.named_parameters()] == list(g["param_shapes"])
    plain = SwinUnet(None, 224, 4, device="meta", icl=False)
    assert all(k.startswith("swin_unet.") for k in plain.state_dict())


def test_swinunet2d_blocks_match_oracle():
    """Shifted 7x7-window block on a 14^2 token grid (head dim 32, region mask), patch embedding, merging and both patch
    expansions of the 2-D Swin-UNet against oracle/swinunet2d_oracle.py (the whole 224^2 model is checked on the GPU)."""
    from icl_amd.networks import swinunet_icl as M
    from oracle import swinunet2d_oracle as W
    c, heads, res = 64, 2, 14
    shapes = W._block_shapes("b.", c, heads)
    p = O.make_params(shapes, requires_grad=True)
    p["b.attn.relative_position_index"] = W.relative_position_index()
    p["b.attn_mask"] = W.attn_mask(res, 7, 3)
    blk = M.SwinTransformerBlock(c, (res, res), heads, 7, 3)
    assert torch.equal(blk.attn_mask, p["b.attn_mask"])
    _load(blk, p, "b.")
    x = _rand((2, res * res, c), 201)
    want = W.swin_block(p, "b", x, res, heads, 3)
    got = blk(x)
    assert rel_err(got.detach(), want.detach()) < 2e-5
    gy = _rand(want.shape, 202)
    (got * gy).sum().backward()
    (want * gy).sum().backward()
    for k, t in blk.named_parameters():
        assert rel_err(t.grad, p["b." + k].grad) < 1e-3, k
    # patch embed / merging / expansions
    pe = O.make_params([("patch_embed.proj.weight", (96, 3, 4, 4)), ("patch_embed.proj.bias", (96,)),
                        ("patch_embed.norm.weight", (96,)), ("patch_embed.norm.bias", (96,)),
                        ("m.reduction.weight", (32, 64)), ("m.norm.weight", (64,)), ("m.norm.bias", (64,)),
                        ("e.expand.weight", (64, 32)), ("e.norm.weight", (16,)), ("e.norm.bias", (16,)),
                        ("u.expand.weight", (256, 16)), ("u.norm.weight", (16,)), ("u.norm.bias", (16,))])
    with torch.no_grad():
        emb = M.PatchEmbed(224, 4, 3, 96)
        _load(emb, pe, "patch_embed.")
        img = _rand((1, 3, 224, 224), 203)
        ref = F.conv2d(img, pe["patch_embed.proj.weight"], pe["patch_embed.proj.bias"], stride=4).flatten(2).transpose(1, 2)
        ref = F.layer_norm(ref, (96,), pe["patch_embed.norm.weight"], pe["patch_embed.norm.bias"])
        assert rel_err(emb(img), ref) < 1e-5
        t = _rand((2, 64, 16), 204)
        mg = M.PatchMerging((8, 8), 16)
        _load(mg, pe, "m.")
        assert rel_err(mg(t), W.patch_merging(pe, "m", t, 8)) < 1e-5
        t2 = _rand((2, 16, 32), 205)
        ex = M.PatchExpand((4, 4), 32, 2)
        _load(ex, pe, "e.")
        assert rel_err(ex(t2), W.patch_expand(pe, "e", t2, 4)) < 1e-5
        t3 = _rand((2, 16, 16), 206)
        up = M.PatchExpand((4, 4), 16, 4)
        _load(up, pe, "u.")
        assert rel_err(up(t3), W.patch_expand(pe, "u", t3, 4, 4)) < 1e-5
