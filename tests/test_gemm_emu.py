"""CPU checks (tests/hipemu) of the dense-product kernels of csrc/kernels/gemm.h against torch-CPU matmuls: the weight-streaming
Linear forward / input gradient for skinny inputs, the LDS-tiled general product in all four operand layouts (k-contiguous /
k-strided), split-K, batching, the batch sum and the autograd wrappers built on them."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
from build_emu import build_emu  # noqa: E402

from conftest import rel_err  # noqa: E402
from icl_amd import _lib, ops  # noqa: E402
from icl_amd.utils.hashfill import synthetic_volume  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emu_library():
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    yield
    _lib._use_library_for_tests(None)


def _rand(shape, seed):
    return synthetic_volume(tuple(shape), seed)


@pytest.mark.parametrize("rows,i,o,act", [
    (16, 1024, 1024, 0),     # one row tile, streaming (>= 2^20 weights)
    (8, 1088, 1040, 1),      # partial row tile, K not a multiple of the 64-float chunk, GELU epilogue
    (24, 1024, 1056, 0),     # two row tiles
    (5, 72, 40, 1),          # small weights: tiled product, ragged everything
    (300, 64, 136, 0),       # tall input, tiled product
    (40, 1536, 48, 0),       # split-K (few output tiles, deep K)
    (200, 48, 144, 1),       # 48-column wave tiles: 128 x 144 block of six waves (Swin qkv), K tail
    (300, 96, 96, 0),        # 128 x 96 block
    (150, 192, 48, 0),       # 256 x 48 block
])
def test_linear_forward_and_input_gradient(rows, i, o, act):
    x, w, b = _rand((rows, i), 1), _rand((o, i), 2) * 0.05, _rand((o,), 3)
    y = ops.linear_forward_raw(x, w, b, act)
    ref = F.linear(x, w, b)
    if act:
        ref = F.gelu(ref)
    assert rel_err(y, ref) < 2e-5
    g = _rand((rows, o), 4)
    gx = ops.linear_dgrad_raw(g, w)
    assert rel_err(gx, g @ w) < 2e-5


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("m,n,k", [(70, 150, 100), (200, 40, 36), (33, 260, 72), (68, 152, 100)])
def test_gemm_layouts(ak, bk, m, n, k):
    a = _rand((m, k), 5) if ak else _rand((k, m), 5)
    b = _rand((n, k), 6) if bk else _rand((k, n), 6)
    bias = _rand((n,), 7)
    out = ops.gemm(a, b, m, n, k, a.shape[1], b.shape[1], ak, bk, bias=bias)
    ref = (a if ak else a.t()) @ (b.t() if bk else b) + bias
    assert rel_err(out, ref) < 2e-5


def test_gemm_split_k_batched_and_batch_sum():
    bsz, m, n, k = 2, 48, 64, 640
    a, b = _rand((bsz, m, k), 8), _rand((bsz, k, n), 9)
    out = ops.gemm(a, b, m, n, k, k, n, True, False, batch=bsz, a_bstride=m * k, b_bstride=k * n)
    assert rel_err(out, a @ b) < 2e-5
    total = torch.empty(m, n)
    ops.gemm(a, b, m, n, k, k, n, True, False, out=total, ldc=n, batch=bsz, a_bstride=m * k, b_bstride=k * n, c_bstride=0)
    assert rel_err(total, (a @ b).sum(0)) < 2e-5


def test_split_products_are_summed_in_a_fixed_order():
    """Split products (weight-streaming slices, split-K, the batch sum, the fused input gradient + update) write their partials to slabs
    that gemm_reduce_slabs_kernel adds in slab order: the results of repeated runs agree bit for bit.  (Round 4 also summed them in the
    last-arriving workgroup — bit-identical, measured slower on MI355X, removed in round 5: profiles/r4_tickets_ab.txt.)"""
    L = _lib.lib()

    def run_all():
        outs = []
        for rows, i, o, act in [(16, 1024, 1024, 0), (8, 1088, 1040, 1), (24, 1024, 1056, 0)]:
            x, w, b = _rand((rows, i), 1), _rand((o, i), 2) * 0.05, _rand((o,), 3)
            outs.append(ops.linear_forward_raw(x, w, b, act))
            outs.append(ops.linear_dgrad_raw(_rand((rows, o), 4), w))
        outs.append(ops.linear_forward_raw(_rand((40, 1536), 5), _rand((48, 1536), 6) * 0.05, _rand((48,), 7), 1))      # split-K, GELU
        bsz, m, n, k = 2, 48, 64, 640
        a, b = _rand((bsz, m, k), 8), _rand((bsz, k, n), 9)
        outs.append(ops.gemm(a, b, m, n, k, k, n, True, False, batch=bsz, a_bstride=m * k, b_bstride=k * n))
        total = torch.empty(m, n)
        ops.gemm(a, b, m, n, k, k, n, True, False, out=total, ldc=n, batch=bsz, a_bstride=m * k, b_bstride=k * n, c_bstride=0)
        outs.append(total)
        rows, i, o = 12, 1088, 1040
        g, x = _rand((rows, o), 11), _rand((rows, i), 12)
        w, mo = _rand((o, i), 13) * 0.05, _rand((o, i), 14) * 0.01
        gx = torch.empty(rows, i)
        ws = torch.empty(max(1, L.icl_linear_ws_bytes(rows, i, o, 3) // 4))
        assert L.icl_linear_dgrad_sgd(g.data_ptr(), x.data_ptr(), w.data_ptr(), mo.data_ptr(), gx.data_ptr(), ws.data_ptr(), rows, i, o,
                                      0.01, 0.9, 1e-4, 0, None, None) == 0, _lib.last_error()
        outs += [gx, w, mo]
        return outs

    first = run_all()
    for _ in range(2):
        for got, want in zip(run_all(), first):
            assert torch.equal(got, want)


@pytest.mark.parametrize("rows,i,o", [(4, 64, 256), (8, 256, 1024), (16, 1024, 1024), (24, 1056, 1024), (32, 128, 64), (2, 132, 192)])
def test_small_linear_backward_in_one_launch_equals_the_two_launches(rows, i, o):
    """icl_linear_bwd_small: the weight-streaming input-gradient launch with extra workgroup columns for the outer-product weight gradient —
    bit for bit the results of icl_linear_dgrad and icl_linear_wgrad_small (one slice and several, one and two row tiles)."""
    L = _lib.lib()
    g, x, w = _rand((rows, o), 21), _rand((rows, i), 22), _rand((o, i), 23) * 0.05
    gx_ref = ops.linear_dgrad_raw(g, w)
    gw_ref = ops._tall_atb(g, x, False)[0]
    gx, gw = torch.empty(rows, i), torch.empty(o, i)
    need = L.icl_linear_ws_bytes(rows, i, o, 1)
    ws = torch.empty(max(1, need // 4))
    rc = L.icl_linear_bwd_small(g.data_ptr(), w.data_ptr(), x.data_ptr(), gx.data_ptr(), gw.data_ptr(), ws.data_ptr() if need else None,
                                rows, i, o, None)
    assert rc == 0, (rc, _lib.last_error())
    assert torch.equal(gx, gx_ref) and torch.equal(gw, gw_ref)


def test_linear_autograd_matches_torch():
    x = _rand((2, 6, 40), 10).requires_grad_()
    w = (_rand((24, 40), 11) * 0.2).requires_grad_()
    b = _rand((24,), 12).requires_grad_()
    y = ops.linear(x, w, b)
    gy = _rand(y.shape, 13)
    y.backward(gy)
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    F.linear(xr, wr, br).backward(gy)
    assert rel_err(y.detach(), F.linear(xr, wr, br).detach()) < 2e-5
    for got, ref in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        assert rel_err(got, ref) < 2e-5


def test_conv_transpose_k2s2_is_one_product_on_the_channel_major_input():
    x = _rand((2, 12, 2, 3, 4), 14).requires_grad_()
    w = (_rand((12, 5, 2, 2, 2), 15) * 0.3).requires_grad_()
    skip = _rand((2, 3, 4, 6, 8), 16).requires_grad_()
    y = ops.conv_transpose3d_k2s2(x, w, skip)
    gy = _rand(y.shape, 17)
    y.backward(gy)
    xr, wr, sr = (t.detach().clone().requires_grad_() for t in (x, w, skip))
    ref = torch.cat((F.conv_transpose3d(xr, wr, stride=2), sr), 1)
    ref.backward(gy)
    assert rel_err(y.detach(), ref.detach()) < 2e-5
    for got, want in ((x.grad, xr.grad), (w.grad, wr.grad), (skip.grad, sr.grad)):
        assert rel_err(got, want) < 2e-5


@pytest.mark.parametrize("rows,i,o,first", [
    (16, 1024, 1024, 1),     # one row tile, momentum initialised from the gradient
    (12, 1088, 1040, 0),     # ragged rows, strip and slice tails
    (24, 1024, 1056, 0),     # two row tiles
    (2, 192, 132, 0),        # tiny: one slice, partial tiles everywhere
])
def test_input_gradient_and_sgd_update_in_one_pass(rows, i, o, first):
    """icl_linear_dgrad_sgd == linear_dgrad (old weight) followed by torch.optim.SGD's momentum step on dW = g^T x."""
    L = _lib.lib()
    g, x = _rand((rows, o), 11), _rand((rows, i), 12)
    w0, m0 = _rand((o, i), 13) * 0.05, _rand((o, i), 14) * 0.01
    lr, mom, wd = 0.01, 0.9, 1e-4
    w, m = w0.clone(), (torch.full_like(m0, float("nan")) if first else m0.clone())
    gx = torch.empty(rows, i)
    ws = torch.empty(max(1, L.icl_linear_ws_bytes(rows, i, o, 3) // 4))
    lr_dev = torch.tensor([lr])
    rc = L.icl_linear_dgrad_sgd(g.data_ptr(), x.data_ptr(), w.data_ptr(), m.data_ptr(), gx.data_ptr(), ws.data_ptr(), rows, i, o,
                                123.0, mom, wd, first, lr_dev.data_ptr(), None)     # the device scalar overrides the host lr
    assert rc == 0, _lib.last_error()
    d = g.double().t() @ x.double() + wd * w0.double()
    m_ref = d if first else mom * m0.double() + d
    w_ref = w0.double() - lr * m_ref
    assert rel_err(gx, (g.double() @ w0.double()).float()) < 2e-5
    assert rel_err(m, m_ref.float()) < 2e-5
    assert float((w - w_ref.float()).abs().max()) < 2e-6 * float(w_ref.abs().max()) + 1e-7


def test_fused_update_matches_optimizer_step_through_autograd():
    """ops.linear with FactoredGrads.fused_optimizer set: same gx, weight and momentum as backward + FusedSGD.step(); a weight used
    twice in the step falls back to the factored update in step()."""
    from icl_amd.optim import FusedSGD
    torch.manual_seed(0)

    def run(fuse, twice):
        lin = torch.nn.Linear(1536, 1408)
        with torch.no_grad():
            lin.weight.copy_(_rand((1408, 1536), 21) * 0.05)
            lin.bias.copy_(_rand((1408,), 22))
        opt = FusedSGD(lin.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        opt.can_update_in_backward = lambda p, rows: p.dim() == 2     # the product path only takes 2^26-element matrices
        xs = []
        for step in range(2):
            x = _rand((6, 1536), 23 + step).requires_grad_()
            opt.zero_grad()
            ops.FactoredGrads.fused_optimizer = opt if fuse else None
            ops.FactoredGrads.uses = {} if fuse else None
            prev_min = ops.FactoredGrads.min_elems
            try:
                with ops.FactoredGrads(True):
                    y = ops.linear(x, lin.weight, lin.bias, lin)
                    if twice:
                        y = y + ops.linear(x * 0.5, lin.weight, lin.bias, lin)
                    (y * _rand(tuple(y.shape), 30 + step)).sum().backward()
            finally:
                ops.FactoredGrads.fused_optimizer = None
                ops.FactoredGrads.uses = None
                ops.FactoredGrads.min_elems = prev_min
            if fuse and not twice:
                assert lin.weight.grad is None and not getattr(lin.weight, "_icl_factors", None)
            opt.step()
            xs.append(x.grad.clone())
        return xs, lin.weight.detach().clone(), opt.state[lin.weight]["momentum_buffer"].clone(), lin.bias.detach().clone()

    for twice in (False, True):
        a, b = run(True, twice), run(False, twice)
        for ga, gb in zip(a[0], b[0]):
            assert rel_err(ga, gb) < 2e-5
        assert rel_err(a[1], b[1]) < 1e-6 and rel_err(a[2], b[2]) < 2e-5 and rel_err(a[3], b[3]) < 1e-6


def test_update_inside_backward_refuses_a_weight_with_another_use(monkeypatch):
    """A second use of the weight outside _LinearFactored.  (1) through ops.linear's dense path (more rows than FactoredGrads
    takes): counted, so nothing is fused and the step equals the unfused one.  (2) through torch's own F.linear, which ops.linear
    cannot count: the fused kernel bumps the weight's version counter, so autograd raises when F.linear's backward unpacks the
    weight it saved instead of differentiating through the already-updated matrix."""
    from icl_amd.optim import FusedSGD
    # (the thresholds of the factored path scaled down with the layer: the emulator needs minutes for a 600-row product against 2 M weights)
    monkeypatch.setattr(ops.FactoredGrads, "min_elems", 1 << 16)
    monkeypatch.setattr(ops.FactoredGrads, "max_rows", 32)

    def run(fuse, second):
        lin = torch.nn.Linear(512, 448)
        with torch.no_grad():
            lin.weight.copy_(_rand((448, 512), 41) * 0.05)
            lin.bias.copy_(_rand((448,), 42))
        opt = FusedSGD(lin.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        opt.can_update_in_backward = lambda p, rows: p.dim() == 2
        x = _rand((6, 512), 43).requires_grad_()
        xt = _rand((60, 512), 44)          # 60 rows > FactoredGrads.max_rows: ops.linear takes the dense _Linear path
        ops.FactoredGrads.fused_optimizer = opt if fuse else None
        ops.FactoredGrads.uses = {} if fuse else None
        try:
            with ops.FactoredGrads(True):
                y = ops.linear(x, lin.weight, lin.bias, lin).sum()
                if second == "dense":
                    y = y + ops.linear(xt, lin.weight, lin.bias, lin).sum() * 0.01
                elif second == "torch":
                    y = y + F.linear(xt[:4], lin.weight, lin.bias).sum() * 0.01
                y.backward()
        finally:
            ops.FactoredGrads.fused_optimizer = None
            ops.FactoredGrads.uses = None
        opt.step()
        return x.grad.clone(), lin.weight.detach().clone()

    a, b = run(True, "dense"), run(False, "dense")
    assert rel_err(a[0], b[0]) < 2e-5 and rel_err(a[1], b[1]) < 1e-6
    with pytest.raises(RuntimeError, match="modified by an inplace operation|updated inside its backward"):
        run(True, "torch")


@pytest.mark.parametrize("rows,i,o,act", [(16400, 48, 144, 0), (16390, 48, 192, 1), (16385, 192, 48, 0), (16384, 96, 96, 1)])
def test_tall_skinny_linear_with_weights_in_registers(rows, i, o, act):
    """linear_rows_kernel (>= 16,384 rows, the token layers of SwinUNETR stage 0): forward with bias / GELU and the input gradient
    (weight read as [K][N]), ragged last 16-row block."""
    x, w, b = _rand((rows, i), 31), _rand((o, i), 32) * 0.1, _rand((o,), 33)
    y = ops.linear_forward_raw(x, w, b, act)
    assert "linear_rows_kernel" in _lib.lib().icl_last_kernel_name().decode()
    ref = F.linear(x, w, b)
    if act:
        ref = F.gelu(ref)
    assert rel_err(y, ref) < 2e-5
    g = _rand((rows, o), 34)
    gx = ops.linear_dgrad_raw(g, w)
    assert "linear_rows_kernel" in _lib.lib().icl_last_kernel_name().decode()
    assert rel_err(gx, g @ w) < 2e-5
