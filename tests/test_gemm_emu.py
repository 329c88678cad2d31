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
