"""GPU parity of the kernel instantiations that the three benchmarked configurations never launch (round 5: `tools/kernel_coverage.sh`
listed 41 of 208 kernels that neither the GPU suite nor a bench configuration had run on hardware — the emulator was their only
check): head dimension 8 of the prototype attention (small `feature_scale`), LayerNorm rows of 4,097 .. 8,192 and > 16,384 columns,
window attention on windows that do not fill their key-block template, the bf16-split factored SGD update of many factor rows (the
data-parallel gathered factors), fp32-MFMA convolution tiles of volumes whose extents are not multiples of 8, and the non-vector /
transposed layouts of the tiled product.  Every check is against a plain torch (CPU, fp64 where it matters) expression."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from icl_amd.utils.hashfill import synthetic_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from icl_amd import _lib
    assert _lib.lib_path().endswith("libicl_hip.so"), "GPU tests must run on the HIP library"
    return torch.device("cuda", 0)


def _rand(shape, seed):
    return synthetic_volume(tuple(shape), seed)


@pytest.mark.parametrize("B,h,nc,d,N", [(2, 4, 2, 8, 216), (1, 2, 16, 8, 1728), (2, 8, 3, 8, 64)])
def test_prototype_attention_head_dimension_8(dev, B, h, nc, d, N):
    """unet_3D_icl with a feature_scale above 4 has 8-wide heads (unet_3D_icl.py:282-296): attn_logits / softmax_pv / bwd kernels <8>."""
    from icl_amd import ops
    C = h * d
    qh, kv = _rand((B, h, nc, d), 71), _rand((B, N, 2 * C), 72)
    go, gl = _rand((B, h, nc, d), 73), _rand((B, nc, h, N), 74)
    qg, kg = qh.to(dev).requires_grad_(), kv.to(dev).requires_grad_()
    out, logits = ops.prototype_attention(qg, kg, h, d ** -0.5)
    ((out * go.to(dev)).sum() + (logits * gl.to(dev)).sum()).backward()
    qr, kr = qh.clone().requires_grad_(), kv.clone().requires_grad_()
    kvp = kr.reshape(B, N, 2, h, d).permute(2, 0, 3, 1, 4)
    lr4 = (qr @ kvp[0].transpose(-2, -1)) * d ** -0.5
    orf = lr4.softmax(dim=-1) @ kvp[1]
    lr = lr4.permute(0, 2, 1, 3)
    ((orf * go).sum() + (lr * gl).sum()).backward()
    assert rel_err(out.detach().cpu(), orf.detach()) < 1e-5 and rel_err(logits.detach().cpu(), lr.detach()) < 1e-5
    assert rel_err(qg.grad.cpu(), qr.grad) < 1e-4 and rel_err(kg.grad.cpu(), kr.grad) < 1e-4


@pytest.mark.parametrize("shape", [(3, 6000), (2, 20000), (5, 1000), (2, 3000)])
def test_layer_norm_rows_between_the_model_sizes(dev, shape):
    """Rows of 4,097 .. 8,192 columns (layernorm_*_longrow_kernel<32>), of more than 16,384 (layernorm_*_kernel<true>), and two sizes
    between the aligner's 128 / 1,728 / 13,824."""
    from icl_amd import ops
    c = shape[-1]
    x, w, b, gy = _rand(shape, 61) * 2 + 0.3, 1 + 0.1 * _rand((c,), 62), 0.1 * _rand((c,), 63), _rand(shape, 64)
    xg, wg, bg = (t.to(dev).requires_grad_() for t in (x, w, b))
    y = ops.layer_norm(xg, wg, bg)
    y.backward(gy.to(dev))
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = F.layer_norm(xr, (c,), wr, br, 1e-5)
    yr.backward(gy.double())
    assert rel_err(y.detach().cpu().double(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu().double(), xr.grad) < 1e-4
    assert rel_err(wg.grad.cpu().double(), wr.grad) < 1e-4 and rel_err(bg.grad.cpu().double(), br.grad) < 1e-4


def _window_attention_torch(qkv, table, index, regions, heads, scale):
    b_, n, c3 = qkv.shape
    c = c3 // 3
    bias = table[index[:n, :n].reshape(-1)].view(n, n, heads).permute(2, 0, 1)
    q, k, v = qkv.view(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4).unbind(0)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.unsqueeze(0)
    if regions is not None:
        nw = regions.shape[0]
        mask = (regions.unsqueeze(1) != regions.unsqueeze(2)).to(attn.dtype) * -100.0
        attn = (attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    return (attn.softmax(-1) @ v).transpose(1, 2).reshape(b_, n, c)


@pytest.mark.parametrize("n,dh,heads,nw,batch,masked", [
    (64, 16, 2, 4, 2, True),      # 4 key blocks on the 6-block template            window_attn_*<6, 16, false>
    (96, 16, 1, 2, 3, False),     # exactly 6                                         <6, 16, true>
    (90, 32, 2, 2, 1, True),      # 6 blocks, 6 padded keys, head dimension 32        <6, 32, true>
    (200, 16, 3, 2, 2, True),     # 13 of 14                                          <14, 16, false>
    (300, 16, 2, 3, 1, False),    # 19 of 22                                          <22, 16, false>
])
def test_window_attention_on_windows_that_do_not_fill_their_template(dev, n, dh, heads, nw, batch, masked):
    """The window-attention kernels are instantiated for 6 / 14 / 22 key blocks of 16 and specialised for windows that fill them exactly
    (343 = 22 blocks with 9 padded keys, 216 = 14); clipped windows of other volumes take the general variants."""
    from icl_amd import ops
    c = heads * dh
    T = 500
    qkv = (_rand((batch * nw, n, 3 * c), 71) * 1.5)
    table = _rand((T, heads), 72)
    gen = torch.Generator().manual_seed(7)
    index = torch.randint(0, T, (n, n), generator=gen)
    regions = torch.randint(0, 3, (nw, n), generator=gen).to(torch.int32) if masked else None
    gy = _rand((batch * nw, n, c), 73)
    qg, tg = qkv.to(dev).requires_grad_(), table.to(dev).requires_grad_()
    y = ops.window_attention(qg, tg, index.to(dev), regions.to(dev) if masked else None, heads, dh ** -0.5)
    y.backward(gy.to(dev))
    qr, tr = qkv.double().requires_grad_(), table.double().requires_grad_()
    yr = _window_attention_torch(qr, tr, index, regions, heads, dh ** -0.5)
    yr.backward(gy.double())
    assert rel_err(y.detach().cpu().double(), yr.detach()) < 1e-5
    assert rel_err(qg.grad.cpu().double(), qr.grad) < 1e-4
    assert rel_err(tg.grad.cpu().double(), tr.grad) < 1e-4


@pytest.mark.parametrize("rows,n,k,first", [(256, 2048, 2048, 1), (192, 1728, 2432, 0)])
def test_factored_sgd_update_of_many_factor_rows_on_split_products(dev, rows, n, k, first):
    """icl_sgd_step_factored_split (optim.FusedSGD._step_factored from SPLIT_MIN_ROWS = 192 factor rows: the gathered factors of a
    data-parallel step): d = g^T x from exact three-way bf16 splits, then the SGD(momentum, weight decay) update of p and m in the
    same pass — against fp64."""
    from icl_amd import _lib
    L = _lib.lib()
    lr, mom, wd = 0.05, 0.9, 1e-3
    g, x = _rand((rows, n), 11) * 0.1, _rand((rows, k), 12)
    p0, m0 = _rand((n, k), 13), _rand((n, k), 14) * 0.01
    p, m = p0.to(dev), m0.to(dev)
    gd, xd = g.to(dev), x.to(dev)
    ws = torch.empty(max(1, L.icl_sgd_factored_split_ws_bytes(rows, n, k) // 4), dtype=torch.float32, device=dev)
    rc = L.icl_sgd_step_factored_split(p.data_ptr(), m.data_ptr(), gd.data_ptr(), xd.data_ptr(), ws.data_ptr(), rows, n, k, lr, mom, wd,
                                       first, None, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "sgd_step_factored_split")
    torch.cuda.synchronize()
    d = g.double().t() @ x.double() + wd * p0.double()
    mr = d if first else mom * m0.double() + d
    pr = p0.double() - lr * mr
    assert rel_err(m.cpu().double(), mr) < 2e-6
    assert rel_err(p.cpu().double(), pr) < 2e-6


@pytest.mark.parametrize("n,cin,cout,d,h,w", [
    (1, 16, 16, 8, 12, 16), (1, 16, 32, 8, 12, 16), (2, 8, 48, 4, 12, 16),      # H % 8 != 0: fp32 MFMA tiles of 4 x 4 x 16
    (1, 64, 32, 6, 6, 6), (1, 20, 24, 6, 6, 6),                                 # 6^3 with two cout blocks / ragged channels
    (1, 12, 20, 5, 7, 9),                                                       # nothing divides: the generic kernel
])
def test_conv3d_on_extents_the_models_do_not_have(dev, monkeypatch, n, cin, cout, d, h, w):
    from icl_amd import ops
    monkeypatch.setenv("ICL_CONV_SPLIT", "0")      # exact-fp32 MFMA kernels for every shape (the split-product kernels are covered elsewhere)
    x = _rand((n, cin, d, h, w), 1)
    wt = _rand((cout, cin, 3, 3, 3), 2) * (1.0 / np.sqrt(cin * 27))
    b = _rand((cout,), 3) * 0.1
    gy = _rand((n, cout, d, h, w), 4)
    xg, wg, bg = (t.to(dev).requires_grad_() for t in (x, wt, b))
    y = ops.conv3d(xg, wg, bg)
    y.backward(gy.to(dev))
    xr, wr, br = (t.clone().requires_grad_() for t in (x, wt, b))
    yr = F.conv3d(xr, wr, br, padding=1)
    yr.backward(gy)
    assert rel_err(y.detach().cpu(), yr.detach()) < 2e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 2e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-4
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-4


@pytest.mark.parametrize("rows,i,o", [(40, 70, 50), (200, 333, 129), (1000, 50, 70), (37, 1000, 24), (3000, 130, 10), (5, 4000, 4000),
                                      (700, 64, 256), (128, 257, 64)])
def test_linear_on_shapes_without_16_byte_rows(dev, rows, i, o):
    """Linear forward, input gradient and weight gradient on row lengths that are not multiples of 4 floats (the tiled product's
    scalar-load variants, k-strided operands included) and on tall / wide / skinny aspect ratios between the models' own."""
    from icl_amd import ops
    x, w, b, gy = _rand((rows, i), 21), _rand((o, i), 22) * (1.0 / np.sqrt(i)), _rand((o,), 23) * 0.1, _rand((rows, o), 24)
    xg, wg, bg = (t.to(dev).requires_grad_() for t in (x, w, b))
    y = ops.linear(xg, wg, bg)
    y.backward(gy.to(dev))
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    yr.backward(gy.double())
    assert rel_err(y.detach().cpu().double(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu().double(), xr.grad) < 1e-5
    assert rel_err(wg.grad.cpu().double(), wr.grad) < 1e-5
    assert rel_err(bg.grad.cpu().double(), br.grad) < 1e-5


@pytest.mark.parametrize("cout", [32, 48])
def test_conv3d_forced_4x8x16_tile_with_two_and_three_cout_blocks(dev, monkeypatch, cout):
    """ICL_CONV_FORCE_TILE=48 (a tests-only switch, also used on the emulator) puts 32- / 48-cout layers on the 4 x 8 x 16 fp32 tile that
    the launcher otherwise gives to 16-cout layers only."""
    monkeypatch.setenv("ICL_CONV_FORCE_TILE", "48")
    test_conv3d_on_extents_the_models_do_not_have(dev, monkeypatch, 1, 8, cout, 4, 16, 16)


@pytest.mark.parametrize("m,n,k", [(40, 600, 100), (200, 136, 100), (1000, 40, 64), (48, 1000, 37), (333, 77, 129), (2000, 24, 50)])
def test_gemm_operand_layouts_on_tall_wide_and_ragged_shapes(dev, m, n, k):
    """icl_gemm with every combination of k-contiguous / k-strided operands, on shapes that pick each of its three wave arrangements
    (1 x 4, 2 x 2, 4 x 1) with and without 16-byte rows."""
    from icl_amd import ops
    for ak in (True, False):
        for bk in (True, False):
            a = _rand((m, k), 31) if ak else _rand((k, m), 31)
            b = _rand((n, k), 32) if bk else _rand((k, n), 32)
            out = ops.gemm(a.to(dev), b.to(dev), m, n, k, a.shape[1], b.shape[1], ak, bk)
            ref = (a if ak else a.t()).double() @ (b.t() if bk else b).double()
            assert rel_err(out.cpu().double(), ref) < 2e-5, (ak, bk)


@pytest.mark.parametrize("i,o", [(48, 96), (96, 48), (144, 48), (48, 144), (48, 48), (96, 96)])
def test_linear_on_the_token_grids_of_other_swin_widths(dev, i, o):
    """>= 16,384 rows with 48 / 96 / 144 columns in or out: the register-blocked row kernels (linear_rows_kernel) of the Swin stage-0 / 1
    projections, here in the width pairs SwinUNETR-ICL's own layers do not use (forward and input gradient take the two weight layouts)."""
    test_linear_on_shapes_without_16_byte_rows(dev, 20000, i, o)
