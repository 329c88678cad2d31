"""CPU checks of the HIP kernel SOURCES (icl_amd/csrc/kernels) through tests/hipemu.

The kernels are compiled unchanged against a fiber-based emulation of the device environment
(workgroups, LDS, barriers, wave shuffles, MFMA lane maps) and driven through the same C ABI and
autograd wrappers as on the GPU.  Expected values come from torch-CPU functional ops — the very
operators the oracle (oracle/icl_oracle.py) is built from.
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
from icl_amd.utils.hashfill import synthetic_volume  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def emu_library():
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    yield
    _lib._use_library_for_tests(None)


def _rand(shape, seed, grad=False):
    t = synthetic_volume(tuple(shape), seed)
    return t.requires_grad_() if grad else t


@pytest.mark.parametrize("n,cin,cout,d,h,w,ks", [
    (2, 3, 4, 8, 8, 8, 3),      # ragged channels (padding of packed weights)
    (1, 1, 16, 6, 6, 6, 3),     # first-layer shape class (Cin = 1), non-multiple-of-4 width
    (1, 16, 32, 4, 8, 16, 3),   # two cout groups per block
    (1, 20, 48, 6, 6, 12, 3),   # partial cin chunk + partial cout tile
    (2, 8, 2, 4, 4, 8, 1),      # 1x1x1, few channels: icl_conv1x1_small (forward and input gradient)
    (4, 16, 1, 6, 6, 6, 1),     # attn_convs1 class (h -> 1)
    (2, 4, 4, 3, 4, 5, 1),      # pointwise h -> h
    (1, 20, 3, 4, 4, 8, 1),     # 1x1x1 beyond 16 channels: implicit-GEMM kernel (final conv class)
    (1, 4, 4, 3, 3, 5, 1),      # voxel count not a multiple of 4: implicit-GEMM kernel
    (1, 32, 16, 2, 8, 32, 3),   # wide row: several x tiles
    (1, 8, 48, 5, 12, 12, 3),   # static 4x4x16 tile, masked x/z borders, NB = 48
    (2, 4, 16, 3, 16, 24, 3),   # static 2x8x16 tile, CinP = 4 chunks, partial x tile
    (1, 16, 32, 3, 8, 24, 3),   # flat 2x8x24 tile (NB 32), z border
    (1, 8, 96, 4, 8, 12, 3),    # flat 4x4x12 tile, NB 48
    (2, 16, 64, 6, 6, 6, 3),    # flat 6x6x6 tile, scalar staging, padded row groups
])
def test_conv3d_fwd_bwd(n, cin, cout, d, h, w, ks):
    _conv_check(n, cin, cout, d, h, w, ks)


@pytest.mark.parametrize("n,cin,cout,d,h,w", [
    (1, 16, 16, 4, 8, 16),      # one tile, one cout block
    (2, 32, 20, 5, 9, 20),      # two 16-channel chunks, ragged cout, partial tiles in z / y / x, two samples
    (1, 16, 48, 3, 4, 12),      # three cout blocks per workgroup, a volume smaller than one tile
    (1, 48, 40, 2, 8, 16),      # two cout blocks, the second one ragged; three chunks
    (2, 20, 12, 5, 9, 24),      # weight gradient on split products (one cout block): ragged cin block, partial tiles in z / y / x
    (1, 32, 16, 4, 8, 16),      # weight gradient: two cin blocks, forward and input gradient split as well
    (1, 20, 32, 3, 9, 24),      # weight gradient with two cout blocks per workgroup (2 x 8 x 16 tiles): ragged cin block, partial tiles
    (2, 16, 60, 2, 8, 16),      # ... two such workgroup columns, the last cout block ragged (60 -> 64), two samples
    (1, 16, 16, 3, 9, 24),      # rows of 24 voxels: flat 2 x 8 x 24 tiles (row blocks that wrap from one row to the next), one cout block,
    (2, 32, 48, 5, 10, 24),     # ... three cout blocks, two chunks, odd depth, partial y tile, two samples
    (1, 32, 32, 4, 24, 24),     # ... two cout blocks, three y tiles
])
def test_conv3d_split_bf16_products(monkeypatch, n, cin, cout, d, h, w):
    """conv_bf16x3.h / conv_bf16x3_ws.h / conv_wgrad_tr.h: forward, input gradient and weight gradient with each fp32 operand split
    exactly into three bf16 terms (six bf16 MFMA terms per product, fp32 accumulation) — same tolerance as the fp32-MFMA kernels, and the
    two paths agree to fp32 rounding.  One cout block on 4 x 8 x 16 tiles runs the loader-wave kernel, everything else the single-role
    kernel; rows of 24 voxels the flat tiles."""
    monkeypatch.setenv("ICL_CONV_SPLIT_MIN", "1")
    monkeypatch.setenv("ICL_CONV_SPLIT", "1")
    monkeypatch.setenv("ICL_WGRAD_SPLIT", "1")
    _conv_check(n, cin, cout, d, h, w, 3)
    x = _rand((n, cin, d, h, w), 11)
    wt = _rand((cout, cin, 3, 3, 3), 12) * 0.2
    with torch.no_grad():
        y1 = ops.conv3d(x, wt, None)
        monkeypatch.setenv("ICL_CONV_SPLIT", "0")
        y0 = ops.conv3d(x, wt, None)
    ref = F.conv3d(x.double(), wt.double(), None, padding=1)
    e1, e0 = float((y1.double() - ref).abs().max()), float((y0.double() - ref).abs().max())
    # about as close to the fp64 result as the fp32-MFMA path (a max over a few thousand outputs of two different summation orders:
    # the emulated bf16 MFMA rounds once per 32 products; the accuracy statement proper is the GPU test on real layer shapes)
    assert e1 <= 3.0 * e0 + 2e-7 * float(ref.abs().max()), (e1, e0)


@pytest.mark.parametrize("n,cin,cout,d,h,w,ksplit", [
    (2, 48, 64, 4, 8, 12, 2),       # rows of 12 voxels: flat 4 x 4 x 12 tiles on four waves, two cout blocks; three chunks in slices of 1 + 2
    (1, 32, 60, 8, 4, 12, 2),       # ... ragged last cout block (60 -> 64), two z tiles, one chunk per slice
    (1, 64, 32, 4, 4, 12, 3),       # ... four chunks in three slices (1 + 1 + 2)
    (1, 48, 32, 4, 8, 24, 3),       # rows of 24 voxels: the flat 2 x 8 x 24 tiles with one chunk per slice
    (2, 32, 48, 3, 10, 24, 2),      # ... three cout blocks, odd depth, partial y tile, two samples
    (1, 32, 16, 2, 8, 24, 5),       # ... one cout block (all three weight planes resident), more slices asked than chunks
    (2, 48, 32, 6, 6, 6, 3),        # a whole 6^3 volume as one flat tile of 16 row blocks (216 of 256 rows live), two samples
    (1, 32, 60, 6, 6, 6, 2),        # ... ragged last cout block
])
def test_conv3d_split_products_split_over_channel_chunks(monkeypatch, n, cin, cout, d, h, w, ksplit):
    """conv_bf16x3.h, round 6: the deep levels' launches split the channel chunks of a tile over workgroups (grid.z), every slice writes
    raw partial sums into a slab and splitk_reduce_kernel adds them (+ bias) in a fixed order; rows of 12 voxels run on flat 4 x 4 x 12
    tiles that exist for such launches only.  Forward and input gradient against torch, through icl_conv3d_fwd (slab behind the split
    weights) and through icl_conv3d_fwd_presplit_ws (the step's path); a split launch hands out no InstanceNorm statistics.
    Reference: nn.Conv3d of UnetConv3 at the conv3 / conv4 / up_concat4 / up_concat3 levels (networks/unet_3D.py:41-47,53-54)."""
    import ctypes
    monkeypatch.setenv("ICL_CONV_SPLIT_MIN", "1")
    monkeypatch.setenv("ICL_CONV_SPLIT", "1")
    monkeypatch.setenv("ICL_CONV_SPLIT_KSPLIT", str(ksplit))
    L = _lib.lib()
    s = d * h * w
    want_slices = min(ksplit, cin // 16)
    assert L.icl_conv3d_fwd_presplit_ws_bytes(n, cin, cout, d, h, w) == want_slices * n * cout * s * 4
    assert L.icl_conv3d_fwd_stats_slots(n, cin, cout, d, h, w) == 0
    _conv_check(n, cin, cout, d, h, w, 3)
    x = _rand((n, cin, d, h, w), 41)
    wt = _rand((cout, cin, 3, 3, 3), 42) * 0.2
    b = _rand((cout,), 43)
    wp = ops.pack_weights(wt, 0)
    wsplit = torch.empty(L.icl_conv3d_split_ws_bytes(cin, cout) // 4, dtype=torch.float32)
    arr, iarr = ctypes.c_void_p * 1, ctypes.c_int32 * 1
    _lib.check(L.icl_conv3d_split_weights_multi(arr(wp.data_ptr()), arr(wsplit.data_ptr()), iarr(cin), iarr(cout), 1, None), "split")
    y = torch.full((n, cout, d, h, w), float("nan"))
    assert ops.conv3d_forward_raw(x, wp, b, n, cin, cout, d, h, w, 3, cin * s, y, cout * s, wsplit=wsplit, want_stats=True) is None
    name = L.icl_last_kernel_name().decode()
    assert "bf16x3" in name and ("flat12" in name) == (w == 12) and ("flat6" in name) == (w == 6), name
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    assert float((y.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    # unsplit: the same sums in another order (rows of 12 fall back to the fp32 kernels) — close, not equal
    monkeypatch.setenv("ICL_CONV_SPLIT_KSPLIT", "0")
    assert L.icl_conv3d_fwd_presplit_ws_bytes(n, cin, cout, d, h, w) == 0
    y0 = torch.empty_like(y)
    ops.conv3d_forward_raw(x, wp, b, n, cin, cout, d, h, w, 3, cin * s, y0, cout * s, wsplit=wsplit)
    assert ("bf16x3" in L.icl_last_kernel_name().decode()) == (w == 24)
    assert float((y - y0).abs().max()) < 2e-5 * float(ref.abs().max()) and not torch.equal(y, y0)


@pytest.mark.parametrize("n,cin,cout,d,h,w,wgs", [
    (2, 16, 16, 10, 16, 16, 3),     # one cout block; 2 samples x 2 columns x 10 tiles on three workgroups: runs that start inside a column and cross into the next
    (1, 20, 32, 9, 8, 40, 2),       # two cout blocks, ragged cin block, three columns in x (the last one 8 of 16 wide), odd depth
    (1, 16, 48, 8, 8, 16, 1),       # three cout blocks (one tile of loads in flight), one column walked by one workgroup
    (2, 40, 16, 8, 16, 16, 0),      # three cin blocks (the last ragged); the launcher's own split: one tile per workgroup (prologue path only)
])
def test_conv3d_weight_gradient_walking_z_columns(monkeypatch, n, cin, cout, d, h, w, wgs):
    """conv_wgrad_zs.h: 1 x 8 x 16 tiles walked down z with a ring of four halo planes in LDS — a phase stages only the plane the next
    tile adds; a column's first tile stages its other two planes behind the barrier.  Against torch's weight gradient in fp64, and
    against the 2 x 4 x 16-tile kernel (conv_wgrad_tr.h) to fp32 rounding.  Reference: nn.Conv3d backward, networks/utils.py:104."""
    monkeypatch.setenv("ICL_CONV_SPLIT_MIN", "1")
    monkeypatch.setenv("ICL_CONV_SPLIT", "1")
    monkeypatch.setenv("ICL_WGRAD_SPLIT", "1")
    monkeypatch.setenv("ICL_WGRAD_ZS", "1")
    if wgs:
        monkeypatch.setenv("ICL_WGRAD_ZS_MAX_WGS", str(wgs))
    x = _rand((n, cin, d, h, w), 31)
    gy = _rand((n, cout, d, h, w), 32)
    wt = (_rand((cout, cin, 3, 3, 3), 33) * 0.2)

    def wgrad():
        wl = wt.clone().requires_grad_()
        ops.conv3d(x, wl, None).backward(gy)
        return wl.grad

    g1 = wgrad()
    monkeypatch.setenv("ICL_WGRAD_ZS", "0")
    g0 = wgrad()
    wr = wt.double().requires_grad_()
    F.conv3d(x.double(), wr, None, padding=1).backward(gy.double())
    scale = float(wr.grad.abs().max())
    e1, e0 = float((g1.double() - wr.grad).abs().max()), float((g0.double() - wr.grad).abs().max())
    assert e1 < 2e-6 * scale + 3.0 * e0, (e1, e0, scale)
    assert not torch.equal(g1, g0) or n * d * h * w < 2048      # the two kernels sum in different orders: equal bits would mean the switch did nothing


@pytest.mark.parametrize("n,cin,cout,d,h,w", [
    (2, 16, 16, 8, 8, 16),      # one cout block: the loader-wave kernel; two samples, one XCD share each... (16 tiles: 2 per XCD)
    (1, 32, 40, 5, 9, 20),      # three cout blocks (ragged), partial tiles in z / y / x: values outside the volume must not be counted
    (2, 16, 32, 4, 8, 24),      # rows of 24 voxels: the flat tile, two cout blocks
    (3, 16, 16, 12, 8, 16),     # 3 samples x 3 tiles on eight XCD shares of 2 tiles: a workgroup would span two samples -> refused
])
def test_conv3d_hands_instance_norm_its_statistics(monkeypatch, n, cin, cout, d, h, w):
    """conv_bf16x3.h bf3_stats_*: the forward kernels write one (count, mean, M2) summary per (sample, channel, workgroup) from the
    epilogue's registers; icl_norm_fwd_given_stats merges them instead of re-reading the output.  Reference: Conv3d -> InstanceNorm3d ->
    ReLU (networks/utils.py:104-106)."""
    import ctypes
    monkeypatch.delenv("ICL_CONV_SPLIT_V", raising=False)
    monkeypatch.setenv("ICL_CONV_SPLIT_MIN", "1")
    monkeypatch.setenv("ICL_CONV_SPLIT", "1")
    L = _lib.lib()
    x = _rand((n, cin, d, h, w), 21) + 0.5                      # a non-zero mean: the summaries must carry it
    wt = _rand((cout, cin, 3, 3, 3), 22) * 0.2
    b = _rand((cout,), 23)
    wp = ops.pack_weights(wt, 0)
    wsplit = torch.empty(L.icl_conv3d_split_ws_bytes(cin, cout) // 4, dtype=torch.float32)
    arr, iarr = ctypes.c_void_p * 1, ctypes.c_int32 * 1
    _lib.check(L.icl_conv3d_split_weights_multi(arr(wp.data_ptr()), arr(wsplit.data_ptr()), iarr(cin), iarr(cout), 1, None), "split")
    s = d * h * w
    y = torch.empty((n, cout, d, h, w))
    stats = ops.conv3d_forward_raw(x, wp, b, n, cin, cout, d, h, w, 3, cin * s, y, cout * s, wsplit=wsplit, want_stats=True)
    ref = F.conv3d(x.double(), wt.double(), b.double(), padding=1)
    assert float((y.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    if n == 3:
        assert stats is None and L.icl_conv3d_fwd_stats_slots(n, cin, cout, d, h, w) == 0
        return
    assert stats is not None and stats.shape[0] == n * cout and stats.shape[2] == 3
    cnt = stats[:, :, 0].double().sum(1)
    assert torch.equal(cnt, torch.full_like(cnt, float(s)))     # every voxel counted once, nothing outside the volume
    mean = (stats[:, :, 0].double() * stats[:, :, 1].double()).sum(1) / s
    yd = y.double().reshape(n * cout, s)
    assert torch.allclose(mean, yd.mean(1), rtol=0, atol=1e-6)
    m2 = (stats[:, :, 2].double() + stats[:, :, 0].double() * (stats[:, :, 1].double() - mean[:, None]) ** 2).sum(1)
    assert torch.allclose(m2 / s, yd.var(1, unbiased=False), rtol=1e-5, atol=1e-8)
    # the normalisation from the summaries == the normalisation with its own statistics pass == torch
    out = ops._NormAct.apply(y, None, None, None, None, 0, True, 1, 1e-5, 0.0, stats)
    own = ops.instance_norm_relu(y)
    want = F.relu(F.instance_norm(ref.float()))
    assert float((out - own).abs().max()) < 2e-6 and float((out - want).abs().max()) < 2e-5


@pytest.mark.parametrize("n,cin,cout,s3", [(2, 16, 2, (32, 32, 32)), (1, 12, 5, (8, 96, 96)), (1, 16, 16, (4, 128, 128))])
def test_dropout_folded_into_the_1x1_convolution_equals_the_two_ops(n, cin, cout, s3):
    """ops.dropout_conv1x1 (`final(dropout2(up1))`): the mask applied on the convolution's input loads (forward), on the input gradient's stores
    and on the weight gradient's x operand — the results of dropout followed by the convolution, bit for bit, with the same seed."""
    x = _rand((n, cin) + s3, 101, True)
    w = (_rand((cout, cin, 1, 1, 1), 102) * 0.3).requires_grad_()
    b = _rand((cout,), 103, True)
    gy = _rand((n, cout) + s3, 104)
    y = ops.dropout_conv1x1(x, w, b, 0.3, seed=1234)
    assert y.grad_fn.__class__.__name__ == "_DropoutConv1x1Backward"
    y.backward(gy)
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    yr = ops.conv3d(ops.dropout(xr, 0.3, seed=1234), wr, br)
    yr.backward(gy)
    dropped = float((ops.dropout(xr.detach(), 0.3, seed=1234) == 0).float().mean())
    assert 0.25 < dropped < 0.35
    assert torch.equal(y.detach(), yr.detach()) and torch.equal(x.grad, xr.grad)
    assert torch.equal(w.grad, wr.grad) and torch.equal(b.grad, br.grad)


@pytest.mark.parametrize("use", ["both", "first", "second"])
def test_split_batch_gradient_is_one_concatenation(use):
    """x[:k], x[k:] through ops.split_batch: the gradient is one launch (icl_concat2), a half that received no gradient comes back as zeros."""
    x = _rand((3, 2, 4, 2, 6), 91).requires_grad_()
    a, b = ops.split_batch(x, 1)
    assert torch.equal(a, x[:1]) and torch.equal(b, x[1:])
    ga, gb = _rand(a.shape, 92), _rand(b.shape, 93)
    loss = (a * ga).sum() * (use != "second") + (b * gb).sum() * (use != "first")
    if use == "first":
        loss = (a * ga).sum()
    elif use == "second":
        loss = (b * gb).sum()
    loss.backward()
    want = torch.cat([ga if use != "second" else torch.zeros_like(ga), gb if use != "first" else torch.zeros_like(gb)], 0)
    assert torch.equal(x.grad, want)


@pytest.mark.parametrize("w,x2", [(8, "1"), (8, "0"), (12, "1"), (6, "1")])
def test_skip_and_pool_adds_the_two_gradients_in_one_pass(monkeypatch, w, x2):
    """ops.skip_and_pool (maxpool2_bwd_add_kernel / maxpool2_bwd_add_x2_kernel: one or two pooled outputs per thread; rows of 6 pool to an
    odd width and keep the first): the gradient of an encoder output = its skip gradient (a batch-strided channel slice of the concat
    gradient, as _UpCat.backward hands it over) + the pooling backward; reference: autograd on x -> (x, max_pool3d(x))."""
    monkeypatch.setenv("ICL_MAXPOOL_BWD_X2", x2)
    torch.manual_seed(0)
    x = _rand((2, 3, 4, 6, w), 31, True)
    skip, y = ops.skip_and_pool(x)
    gsk = _rand((2, 5, 4, 6, w), 32)[:, :3]          # non-contiguous over the batch
    gy = _rand(tuple(y.shape), 33)
    torch.autograd.backward([skip, y], [gsk, gy])
    xr = x.detach().clone().requires_grad_()
    yr = F.max_pool3d(xr, 2)
    torch.autograd.backward([xr * 1.0, yr], [gsk, gy])
    assert torch.equal(y, yr) and torch.equal(x.grad, xr.grad)
    # only one of the two consumers has a gradient
    x2 = _rand((1, 2, 2, 4, 4), 34, True)
    s2, y2 = ops.skip_and_pool(x2)
    y2.sum().backward()
    assert torch.equal(x2.grad, torch.autograd.grad(F.max_pool3d(x2, 2).sum(), x2)[0])


def test_pack_weights_multi_equals_the_single_packs():
    """icl_conv3d_pack_weights_multi (LDS tiles of 16 couts x 16 cins x taps, both layouts in one launch, round 6) writes exactly what
    icl_conv3d_pack_weights writes per weight and layout — zero padding included — for ragged channel counts and both kernel sizes.
    Reference: the nn.Conv3d weights of the backbones ([Cout][Cin][k][k][k], networks/utils.py:104)."""
    import ctypes
    L = _lib.lib()
    shapes = [(16, 1, 3), (20, 7, 3), (48, 33, 3), (3, 20, 1), (32, 64, 3), (17, 16, 1)]
    ws = [_rand((co, ci, k, k, k), 70 + i) for i, (co, ci, k) in enumerate(shapes)]
    fwd = [torch.full((L.icl_conv3d_packed_elems(co, ci, k, 0),), float("nan")) for co, ci, k in shapes]
    dgr = [torch.full((L.icl_conv3d_packed_elems(co, ci, k, 1),), float("nan")) for co, ci, k in shapes]
    n = len(shapes)
    arr, iarr = ctypes.c_void_p * n, ctypes.c_int32 * n
    _lib.check(L.icl_conv3d_pack_weights_multi(arr(*[w.data_ptr() for w in ws]), arr(*[t.data_ptr() for t in fwd]), arr(*[t.data_ptr() for t in dgr]),
                                               iarr(*[s[0] for s in shapes]), iarr(*[s[1] for s in shapes]), iarr(*[s[2] for s in shapes]), n, None),
               "pack_weights_multi")
    for w, f, d in zip(ws, fwd, dgr):
        assert torch.equal(f, ops.pack_weights(w, 0)) and torch.equal(d, ops.pack_weights(w, 1)), tuple(w.shape)


def test_conv3d_forced_big_tile(monkeypatch):
    monkeypatch.setenv("ICL_CONV_FORCE_TILE", "48")
    _conv_check(1, 16, 16, 6, 8, 16, 3)
    _conv_check(1, 8, 32, 4, 16, 16, 3)


def test_conv3d_wgrad_tall_slab_reduction():
    """Weight gradient whose partial slabs are added by reduce_unpack_wgrad_tall_kernel (16 slab lanes per packed element; on the GPU
    the 16-channel layers at 96^3 with 512+ slabs) — run in a subprocess: the threshold is read once per process."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_kernels_emu as t; from icl_amd import _lib;"
            "_lib._use_library_for_tests(t.build_emu(), host_pointers=True);"
            "t._conv_check(2, 16, 16, 8, 8, 32, 3); t._conv_check(1, 5, 40, 4, 8, 16, 3)") % (os.path.dirname(here), here)
    env = dict(os.environ, ICL_WGRAD_TALL_MIN="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def _conv_check(n, cin, cout, d, h, w, ks):
    x = _rand((n, cin, d, h, w), 1, True)
    wt = (_rand((cout, cin, ks, ks, ks), 2) * 0.2).requires_grad_()
    b = (_rand((cout,), 3) * 0.1).requires_grad_()
    gy = _rand((n, cout, d, h, w), 4)
    y = ops.conv3d(x, wt, b)
    y.backward(gy)
    xr, wr, br = x.detach().clone().requires_grad_(), wt.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    yr = F.conv3d(xr, wr, br, padding=ks // 2)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-5
    assert rel_err(wt.grad, wr.grad) < 1e-5
    assert rel_err(b.grad, br.grad) < 1e-5


@pytest.mark.parametrize("shape", [(2, 3, 4, 6, 8), (1, 2, 16, 24, 24), (2, 5, 3, 3, 5)])
def test_instance_norm_relu(shape):
    x = (_rand(shape, 5) * 2 + 0.7).requires_grad_()
    gy = _rand(shape, 6)
    y = ops.instance_norm_relu(x)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    yr = F.relu(F.instance_norm(xr, eps=1e-5))
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 2e-5


@pytest.mark.parametrize("shape", [(3, 4, 4, 4, 6), (3, 4, 12, 24, 20)])   # one workgroup per channel / chunked two-launch path
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_relu(training, shape):
    x = (_rand(shape, 7) + 0.3).requires_grad_()
    ga = (1 + 0.1 * _rand((4,), 8)).requires_grad_()
    be = (0.1 * _rand((4,), 9)).requires_grad_()
    rm, rv = 0.1 * _rand((4,), 10), 1 + 0.2 * _rand((4,), 11).abs()
    rm2, rv2 = rm.clone(), rv.clone()
    gy = _rand(shape, 12)
    y = ops.batch_norm_relu(x, ga, be, rm, rv, training)
    y.backward(gy)
    xr, gr, br = (t.detach().clone().requires_grad_() for t in (x, ga, be))
    yr = F.relu(F.batch_norm(xr, rm2, rv2, gr, br, training, 0.1, 1e-5))
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 2e-5
    if training:
        assert rel_err(ga.grad, gr.grad) < 2e-5 and rel_err(be.grad, br.grad) < 2e-5
        assert rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5


def test_maxpool_ties_and_grad():
    x = _rand((2, 3, 4, 6, 8), 13)
    x = torch.relu(x)  # many exact-zero ties, like the post-ReLU activations of the backbone
    x.requires_grad_()
    y = ops.max_pool3d_2(x)
    gy = _rand(tuple(y.shape), 14)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    yr = F.max_pool3d(xr, 2)
    yr.backward(gy)
    assert torch.equal(y.detach(), yr.detach())
    assert torch.equal(x.grad, xr.grad)


@pytest.mark.parametrize("ins,outs", [((3, 4, 5), (6, 8, 10)), ((2, 2, 2), (12, 12, 12)), ((6, 6, 6), (24, 24, 24)),
                                      ((4, 4, 8), (8, 12, 16)), ((5, 3, 12), (7, 6, 20)),
                                      ((3, 6, 6), (12, 24, 48)),      # >= 4x along x: LDS-staged x pass of the backward, ragged last row block
                                      # exact 2x: the specialised one-pass kernels (bwd needs W % 4 == 0), incl. size-1 / size-2 axes
                                      ((4, 6, 8), (8, 12, 16)), ((1, 2, 4), (2, 4, 8)), ((6, 6, 6), (12, 12, 12)), ((3, 5, 12), (6, 10, 24))])
def test_trilinear(ins, outs):
    x = _rand((2, 2) + ins, 15, True)
    y = ops.trilinear_resize(x, outs)
    gy = _rand(tuple(y.shape), 16)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    yr = F.interpolate(xr, size=list(outs), mode="trilinear", align_corners=False)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-6
    assert rel_err(x.grad, xr.grad) < 1e-5


def test_upsample_concat():
    skip = _rand((2, 3, 4, 6, 8), 17, True)
    deep = _rand((2, 5, 2, 3, 4), 18, True)
    y = ops.upsample2x_concat(skip, deep)
    gy = _rand(tuple(y.shape), 19)
    y.backward(gy)
    sr, dr = skip.detach().clone().requires_grad_(), deep.detach().clone().requires_grad_()
    yr = torch.cat([sr, F.interpolate(dr, scale_factor=(2, 2, 2), mode="trilinear")], 1)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-6
    assert rel_err(skip.grad, sr.grad) < 1e-6
    assert rel_err(deep.grad, dr.grad) < 1e-5


@pytest.mark.parametrize("nc", [2, 3, 16])
def test_fused_losses(nc):
    from icl_amd.utils.hashfill import synthetic_labels
    B, S = 2, (6, 5, 7)
    lab = synthetic_labels((B,) + S, 31, nc)
    a = (_rand((B, nc) + S, 32) * 2).requires_grad_()
    ar = a.detach().clone().requires_grad_()
    # CE + hard dice from logits
    ce, dc = ops.cross_entropy_dice_parts(a, lab, nc)
    (1.3 * ce + 0.7 * dc).backward()
    pr = torch.softmax(ar, 1)
    dr = 0
    for c in range(nc):
        t = (lab == c).float()
        dr = dr + (1 - (2 * (pr[:, c] * t).sum() + 1e-5) / ((pr[:, c] ** 2).sum() + t.sum() + 1e-5))
    dr = dr / nc
    cr = F.cross_entropy(ar, lab)
    (1.3 * cr + 0.7 * dr).backward()
    assert abs(float(ce) - float(cr)) < 1e-5 and abs(float(dc) - float(dr)) < 1e-5
    assert rel_err(a.grad, ar.grad) < 1e-4
    # hard dice on probabilities with class weights
    w = [0.5 + 0.1 * i for i in range(nc)]
    p = torch.softmax(_rand((B, nc) + S, 33), 1).requires_grad_()
    prr = p.detach().clone().requires_grad_()
    d = ops.dice_loss(p, lab, nc, softmax=False, weight=w)
    d.backward()
    drr = 0
    for c in range(nc):
        t = (lab == c).float()
        drr = drr + w[c] * (1 - (2 * (prr[:, c] * t).sum() + 1e-5) / ((prr[:, c] ** 2).sum() + t.sum() + 1e-5))
    drr = drr / nc
    drr.backward()
    assert abs(float(d) - float(drr)) < 1e-5 and rel_err(p.grad, prr.grad) < 1e-4
    # soft dice and softmax-MSE against a second logit tensor
    b = _rand((B, nc) + S, 34) * 2
    a2 = (_rand((B, nc) + S, 35) * 2).requires_grad_()
    a2r = a2.detach().clone().requires_grad_()
    sd, ms = ops.soft_dice_loss(a2, b), ops.softmax_mse(a2, b)
    (sd + 10 * ms).backward()
    sa, sb = torch.softmax(a2r, 1), torch.softmax(b, 1)
    sdr = 0
    for c in range(nc):
        sdr = sdr + (1 - (2 * (sa[:, c] * sb[:, c]).sum() + 1e-5) / (sa[:, c].sum() + sb[:, c].sum() + 1e-5))
    sdr = sdr / nc
    msr = ((sa - sb) ** 2).mean()
    (sdr + 10 * msr).backward()
    assert abs(float(sd) - float(sdr)) < 1e-5 and abs(float(ms) - float(msr)) < 1e-6
    assert rel_err(a2.grad, a2r.grad) < 1e-4


@pytest.mark.parametrize("shape", [(3, 4, 5, 6, 7), (2, 3, 12, 12, 12)])      # one chunk / several chunks of the weight gradient
def test_depthwise_conv(shape):
    x = _rand(shape, 44, True)
    c = shape[1]
    w = (_rand((c, 1, 3, 3, 3), 45) * 0.3).requires_grad_()
    y = ops.depthwise_conv3d(x, w)
    gy = _rand(tuple(y.shape), 46)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    yr = F.conv3d(xr, wr, None, padding=1, groups=c)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-5 and rel_err(w.grad, wr.grad) < 1e-5


def test_depthwise_and_dropout():
    x = _rand((3, 4, 5, 6, 7), 41, True)
    w = (_rand((4, 1, 3, 3, 3), 42) * 0.3).requires_grad_()
    y = ops.depthwise_conv3d(x, w)
    gy = _rand(tuple(y.shape), 43)
    y.backward(gy)
    xr, wr = x.detach().clone().requires_grad_(), w.detach().clone().requires_grad_()
    yr = F.conv3d(xr, wr, None, padding=1, groups=4)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-5 and rel_err(w.grad, wr.grad) < 1e-5
    xo = torch.ones(1 << 14, requires_grad=True)
    yo = ops.dropout(xo, 0.3, seed=7)
    assert abs((yo != 0).float().mean().item() - 0.7) < 0.03
    yo.sum().backward()
    assert torch.equal(xo.grad != 0, yo.detach() != 0)


def test_drop_path_is_per_sample_and_consistent_in_backward():
    x = torch.ones(256, 3, 5, requires_grad=True)
    y = ops.drop_path(x, 0.25, seed=77)
    flat = y.detach().reshape(256, -1)
    assert bool((flat.max(dim=1).values == flat.min(dim=1).values).all())          # one decision per sample
    kept = flat[:, 0] != 0
    assert 0.65 < float(kept.float().mean()) < 0.85 and abs(float(flat[kept][0, 0]) - 1 / 0.75) < 1e-6
    y.sum().backward()
    assert torch.equal(x.grad != 0, y.detach() != 0)                               # same mask in backward
    assert not torch.equal(ops.drop_path(x.detach(), 0.25, seed=78) != 0, y.detach() != 0)


def test_step_counter_masks_are_decorrelated_across_steps_ranks_and_seeds():
    """Dropout under graph replay (ops.StepRNG): the device-resident step counter is hashed into the seed, so the mask of step t is
    not the mask of step t-1 shifted by one element (the defect of a linear mix), per-sample DropPath decisions do not slide along
    the batch from step to step, every data-parallel rank draws its own masks and torch.manual_seed selects the stream."""
    x = torch.ones(1 << 14)

    def masks(steps, rank=0, seed=1234):
        torch.manual_seed(seed)
        ops.StepRNG.enable(x.device, rank=rank)
        out = []
        for _ in range(steps):
            ops.StepRNG.begin_step()
            out.append((ops.dropout(x, 0.3) != 0, ops.drop_path(torch.ones(512, 4), 0.3)[:, 0] != 0))
            ops.StepRNG.end_step()
        ops.StepRNG.tensor = None
        return out

    try:
        m = masks(4)
        for t in range(1, 4):
            a, b = m[t - 1][0], m[t][0]
            keep = float(a.float().mean())
            assert 0.66 < keep < 0.74
            for shift in (0, 1, 2, 3):          # agreement with shifted copies of the previous mask stays at the chance level
                agree = float((a[shift:] == b[:len(b) - shift]).float().mean())
                assert abs(agree - (keep ** 2 + (1 - keep) ** 2)) < 0.03, (t, shift, agree)
            pa, pb = m[t - 1][1], m[t][1]
            assert float((pa[1:] == pb[:-1]).float().mean()) < 0.75      # sample k at step t is not sample k+1 at step t-1
        again = masks(2)
        assert torch.equal(again[1][0], m[1][0])                          # same seed, same rank: same stream
        other_rank, other_seed = masks(2, rank=1), masks(2, seed=99)
        assert not torch.equal(other_rank[0][0], m[0][0]) and not torch.equal(other_seed[0][0], m[0][0])
    finally:
        ops.StepRNG.tensor = None


def test_packed_weight_cache_repacks_all_weights_in_one_launch():
    """ops.PackedWeights: first bracketed step packs per call and records the Parameters, later begin_step() calls repack all of
    them at once (after an in-place weight update, as FusedSGD does) — results equal the uncached path, gradients included."""
    ws = [torch.nn.Parameter(_rand((5, 3, 3, 3, 3), 101) * 0.2), torch.nn.Parameter(_rand((20, 5, 3, 3, 3), 102) * 0.2),
          torch.nn.Parameter(_rand((4, 20, 1, 1, 1), 103) * 0.2)]
    x = _rand((2, 3, 6, 5, 7), 104, True)

    def run():
        x.grad = None
        for w in ws:
            w.grad = None
        h = x
        for w in ws:
            h = ops.conv3d(h, w, None)
        h.backward(torch.ones_like(h))
        return h.detach().clone(), x.grad.clone(), [w.grad.clone() for w in ws]

    cache = ops.PackedWeights()
    for step in range(3):
        ref = run()                                   # uncached: packs per call
        cache.begin_step()
        got = run()
        cache.end_step()
        assert ops.PackedWeights.current is None and len(cache.entries) == 3
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        assert all(torch.equal(a, b) for a, b in zip(got[2], ref[2]))
        with torch.no_grad():
            for w in ws:
                w.mul_(0.9).add_(0.01)                # the weights change between steps


def test_deferred_bias_gradients_match_immediate_ones():
    """ops.DeferredBiasGrads: bias gradients of owner-Linear layers queued during backward and reduced in one launch equal the
    per-layer reductions, also for a layer used twice in the step and with an existing .grad to accumulate into."""
    from icl_amd.networks.aligner import Linear
    torch.manual_seed(5)
    la, lb = Linear(12, 20), Linear(20, 7)
    x1, x2 = _rand((3, 15, 12), 111), _rand((4, 12), 112)

    def loss():
        return lb(torch.tanh(la(x1))).pow(2).sum() + la(x2).sum() * 0.5        # la is used twice

    for p in list(la.parameters()) + list(lb.parameters()):
        p.grad = None
    loss().backward()
    ref = [p.grad.clone() for p in (la.weight, la.bias, lb.weight, lb.bias)]
    for p in (la.weight, la.bias, lb.weight, lb.bias):
        p.grad = None
    lb.bias.grad = torch.ones(7)                                               # pre-existing gradient: must be accumulated
    ops.DeferredBiasGrads.begin()
    loss().backward()
    assert la.bias.grad is None and len(ops.DeferredBiasGrads.pending) == 3    # nothing reduced yet
    ops.DeferredBiasGrads.flush()
    assert ops.DeferredBiasGrads.pending is None
    got = [la.weight.grad, la.bias.grad, lb.weight.grad, lb.bias.grad - 1.0]
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 1e-6


def test_drop_path_add_matches_separate_ops():
    """res + drop_path(x) fused (also with res is x) equals the two-kernel form, forward and both gradients."""
    x = _rand((6, 3, 5), 81, True)
    r = _rand((6, 3, 5), 82, True)
    gy = _rand((6, 3, 5), 83)
    y = ops.drop_path_add(r, x, 0.4, seed=91)
    y.backward(gy)
    x2, r2 = x.detach().clone().requires_grad_(), r.detach().clone().requires_grad_()
    y2 = r2 + ops.drop_path(x2, 0.4, seed=91)
    y2.backward(gy)
    assert torch.equal(y.detach(), y2.detach()) and torch.equal(x.grad, x2.grad) and torch.equal(r.grad, r2.grad)
    assert 0 < int((x.grad == 0).all(dim=(1, 2)).sum()) < 6           # some samples dropped, some kept
    q = _rand((6, 4), 84, True)
    z = ops.drop_path_add(q, q, 0.4, seed=92)
    z.backward(torch.ones_like(z))
    q2 = q.detach().clone().requires_grad_()
    z2 = q2 + ops.drop_path(q2, 0.4, seed=92)
    z2.backward(torch.ones_like(z2))
    assert torch.equal(z.detach(), z2.detach()) and torch.equal(q.grad, q2.grad)
    assert torch.equal(ops.drop_path_add(r.detach(), x.detach(), 0.4, training=False), r.detach() + x.detach())


def test_fused_sgd_matches_torch():
    from icl_amd.optim import FusedSGD
    torch.manual_seed(0)
    shapes = [(5, 7), (33,), (1 << 20,), (3, 3, 3, 2, 2), (1,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    skip = torch.nn.Parameter(torch.randn(4))  # never receives a gradient
    skip_ref = skip.detach().clone()
    a = FusedSGD(ps + [skip], lr=0.01, momentum=0.9, weight_decay=1e-4)
    b = torch.optim.SGD(qs, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for it in range(3):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        a.step()
        b.step()
        for grp in a.param_groups:
            grp["lr"] = 0.01 * (1 - it / 10) ** 0.9
        for grp in b.param_groups:
            grp["lr"] = 0.01 * (1 - it / 10) ** 0.9
    for p, q in zip(ps, qs):
        assert rel_err(p.detach(), q.detach()) < 1e-6
    assert torch.equal(skip.detach(), skip_ref)


@pytest.mark.parametrize("rows1", [19, 4, 70])
def test_factored_gradient_sgd_matches_dense(monkeypatch, rows1):
    """ops.FactoredGrads: the weight gradient of a huge Linear stays (g, x); FusedSGD applies g^T x + wd*p without forming it.
    Three steps (first step initialises the momentum), two uses of the same weight in one backward (row blocks concatenate),
    against F.linear + torch.optim.SGD with the dense gradient.  2*rows1 + 5 factor rows: 43 and 145 take the MFMA kernel
    (two and five 32-row chunks), 13 the small-M VALU kernel."""
    from icl_amd.networks.aligner import Linear
    from icl_amd.optim import FusedSGD
    monkeypatch.setattr(ops.FactoredGrads, "min_elems", 1000)
    torch.manual_seed(1)
    lin = Linear(300, 70)            # weight [70, 300]: partial row block (70 = 4*16 + 6) and a partial 256-column slice
    ref_w = torch.nn.Parameter(lin.weight.detach().clone())
    ref_b = torch.nn.Parameter(lin.bias.detach().clone())
    a = FusedSGD(lin.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2)
    b = torch.optim.SGD([ref_w, ref_b], lr=0.05, momentum=0.9, weight_decay=1e-2)
    for it in range(3):
        x1, x2 = _rand((2, rows1, 300), 90 + it), _rand((5, 300), 95 + it)
        a.zero_grad()
        b.zero_grad()
        with ops.FactoredGrads(True):
            y = lin(x1).pow(2).sum() + lin(x2).sum()
            y.backward()
        assert lin.weight.grad is None and len(lin.weight._icl_factors) == 2 and lin.bias.grad is not None
        yr = F.linear(x1, ref_w, ref_b).pow(2).sum() + F.linear(x2, ref_w, ref_b).sum()
        yr.backward()
        a.step()
        b.step()
        assert lin.weight._icl_factors is None
        assert rel_err(lin.weight.detach(), ref_w.detach()) < 1e-5, it
        assert rel_err(lin.bias.detach(), ref_b.detach()) < 1e-5
    # switch off: dense gradient as usual
    lin(x2).sum().backward()
    assert lin.weight.grad is not None and lin.weight._icl_factors is None


@pytest.mark.parametrize("rows,n,k,first", [(70, 70, 300, 0), (128, 130, 264, 1), (33, 64, 256, 0)])
def test_factored_sgd_on_split_products(rows, n, k, first):
    """icl_sgd_step_factored_split (gathered factors / nc = 16: many rows): d = g^T x from exact three-way bf16 splits on the bf16
    matrix pipe, then the SGD rule — against fp64, with the tolerance of the fp32-MFMA kernel; ragged row block, partial 64-row and
    256-column blocks, momentum initialisation."""
    L = _lib.lib()
    g, x = _rand((rows, n), 201) * 0.3, _rand((rows, k), 202)
    p0, m0 = _rand((n, k), 203) * 0.05, _rand((n, k), 204) * 0.01
    lr, mom, wd = 0.02, 0.9, 1e-3
    p, m = p0.clone(), (torch.full_like(m0, float("nan")) if first else m0.clone())
    ws = torch.empty(max(1, L.icl_sgd_factored_split_ws_bytes(rows, n, k) // 4))
    rc = L.icl_sgd_step_factored_split(p.data_ptr(), m.data_ptr(), g.data_ptr(), x.data_ptr(), ws.data_ptr(), rows, n, k, lr, mom, wd, first,
                                       None, None)
    assert rc == 0, _lib.last_error()
    d = g.double().t() @ x.double() + wd * p0.double()
    m_ref = d if first else mom * m0.double() + d
    p_ref = p0.double() - lr * m_ref
    assert rel_err(m, m_ref.float()) < 2e-6
    assert float((p - p_ref.float()).abs().max()) < 2e-6 * float(p_ref.abs().max()) + 1e-7
    # and the fp32-MFMA kernel on the same data is no closer
    p2, m2 = p0.clone(), (torch.full_like(m0, float("nan")) if first else m0.clone())
    assert L.icl_sgd_step_factored(p2.data_ptr(), m2.data_ptr(), g.data_ptr(), x.data_ptr(), rows, n, k, lr, mom, wd, first, None, None) == 0
    e_split, e_fp32 = float((m.double() - m_ref).abs().max()), float((m2.double() - m_ref).abs().max())
    assert e_split <= 4.0 * e_fp32 + 1e-7 * float(m_ref.abs().max()), (e_split, e_fp32)


@pytest.mark.parametrize("rows,n,k,first,wgs", [(8, 40, 1300, 0, 3), (16, 19, 2048, 1, 2), (4, 16, 260, 0, 7)])
def test_narrow_persistent_factored_sgd_is_bit_identical_to_the_wide_launch(rows, n, k, first, wgs):
    """Round 6 (FusedSGD.update_placement "deep"): icl_sgd_step_factored_narrow — a few fat workgroups walking 16 x 1,024 tiles, so that the
    update stream of a 13,824^2 matrix can run under the deep backward levels without taking their CUs — applies exactly the update of
    icl_sgd_step_factored: same sums in the same order, every bit of p and of the momentum buffer.  Ragged rows / columns, a partial
    last column group, momentum initialisation, more workgroups than tiles; more than 16 factor rows are refused (return code 1)."""
    L = _lib.lib()
    g, x = _rand((rows, n), 301) * 0.3, _rand((rows, k), 302)
    p0, m0 = _rand((n, k), 303) * 0.05, _rand((n, k), 304) * 0.01
    lr, mom, wd = 0.02, 0.9, 1e-3
    pa, ma = p0.clone(), (torch.full_like(m0, float("nan")) if first else m0.clone())
    pb, mb = p0.clone(), (torch.full_like(m0, float("nan")) if first else m0.clone())
    assert L.icl_sgd_step_factored(pa.data_ptr(), ma.data_ptr(), g.data_ptr(), x.data_ptr(), rows, n, k, lr, mom, wd, first, None, None) == 0
    assert L.icl_sgd_step_factored_narrow(pb.data_ptr(), mb.data_ptr(), g.data_ptr(), x.data_ptr(), rows, n, k, lr, mom, wd, first, None, wgs, None) == 0
    assert torch.equal(pa, pb) and torch.equal(ma, mb)
    d = g.double().t() @ x.double() + wd * p0.double()
    m_ref = d if first else mom * m0.double() + d
    assert rel_err(mb, m_ref.float()) < 2e-6
    g17, x17 = _rand((17, n), 305), _rand((17, k), 306)
    assert L.icl_sgd_step_factored_narrow(pb.data_ptr(), mb.data_ptr(), g17.data_ptr(), x17.data_ptr(), 17, n, k, lr, mom, wd, 0, None, wgs, None) == 1


@pytest.mark.parametrize("shape", [
    (3, 5, 37),        # one chunk, ragged columns of one lane column
    (2, 70, 64),       # 140 rows: five chunks of 32, the last one ragged; rows in flight past the chunk end
    (1, 33, 128),      # two columns per lane
    (1, 9, 200),       # four columns per lane, two rows in flight
    (1, 6, 300), (1, 5, 768),      # one row in flight (Swin widths)
    (2, 3, 1100),      # one workgroup per row, the row in registers (norm3's long rows): 16 columns per thread
    (1, 2, 5000), (1, 2, 13824),      # 32 / 64 columns per thread
    (1, 1, 16500),     # longer than the register form holds: the plain one-workgroup-per-row kernels
])
def test_layernorm_gelu(shape):
    c = shape[-1]
    x = (_rand(shape, 61) * 2 + 0.3).requires_grad_()
    w = (1 + 0.1 * _rand((c,), 62)).requires_grad_()
    b = (0.1 * _rand((c,), 63)).requires_grad_()
    gy = _rand(shape, 64)
    y = ops.gelu(ops.layer_norm(x, w, b))
    y.backward(gy)
    xr, wr, br = (t.detach().clone().requires_grad_() for t in (x, w, b))
    yr = F.gelu(F.layer_norm(xr, (c,), wr, br, 1e-5))
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-4 and rel_err(w.grad, wr.grad) < 1e-4 and rel_err(b.grad, br.grad) < 1e-4


@pytest.mark.parametrize("B,h,nc,d,N", [(2, 2, 3, 8, 70), (1, 4, 2, 16, 300), (1, 1, 16, 16, 64), (1, 2, 2, 16, 4200)])      # a long token axis
def test_prototype_attention(B, h, nc, d, N):
    C = h * d
    qh = _rand((B, h, nc, d), 71, True)
    kv = _rand((B, N, 2 * C), 72, True)
    go, gl = _rand((B, h, nc, d), 73), _rand((B, nc, h, N), 74)      # logits come back class-major (attn1.permute(0, 2, 1, 3))
    scale = d ** -0.5
    out, logits = ops.prototype_attention(qh, kv, h, scale)
    assert logits.shape == (B, nc, h, N) and logits.is_contiguous()
    ((out * go).sum() + (logits * gl).sum()).backward()
    qr, kr = qh.detach().clone().requires_grad_(), kv.detach().clone().requires_grad_()
    kvp = kr.reshape(B, N, 2, h, d).permute(2, 0, 3, 1, 4)
    lr4 = (qr @ kvp[0].transpose(-2, -1)) * scale
    orf = lr4.softmax(dim=-1) @ kvp[1]
    lr = lr4.permute(0, 2, 1, 3)
    ((orf * go).sum() + (lr * gl).sum()).backward()
    assert rel_err(out.detach(), orf.detach()) < 1e-5 and rel_err(logits.detach(), lr.detach()) < 1e-5
    assert rel_err(qh.grad, qr.grad) < 1e-4 and rel_err(kv.grad, kr.grad) < 1e-4


def test_2d_ops_through_3d_kernels():
    """2-D U-Net ICL building blocks (config 1) run as D = 1 volumes."""
    x = _rand((2, 4, 16, 16), 81, True)
    w = (_rand((6, 4, 3, 3), 82) * 0.2).requires_grad_()
    b = (_rand((6,), 83) * 0.1).requires_grad_()
    ga, be = (1 + 0.1 * _rand((6,), 84)).requires_grad_(), (0.1 * _rand((6,), 85)).requires_grad_()
    rm, rv = torch.zeros(6), torch.ones(6)
    y = ops.batch_norm_act(ops.conv2d(x, w, b), ga, be, rm, rv, True, 2)
    y = ops.bilinear_resize(ops.max_pool2d_2(y), (16, 16), align_corners=True)
    gy = _rand(tuple(y.shape), 86)
    y.backward(gy)
    xr, wr, br, gr, ber = (t.detach().clone().requires_grad_() for t in (x, w, b, ga, be))
    rm2, rv2 = torch.zeros(6), torch.ones(6)
    yr = F.leaky_relu(F.batch_norm(F.conv2d(xr, wr, br, padding=1), rm2, rv2, gr, ber, True, 0.1, 1e-5))
    yr = F.interpolate(F.max_pool2d(yr, 2), size=[16, 16], mode="bilinear", align_corners=True)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    for a, r in ((x, xr), (w, wr), (ga, gr), (be, ber)):
        assert rel_err(a.grad, r.grad) < 1e-4
    assert rel_err(rm, rm2) < 1e-5 and rel_err(rv, rv2) < 1e-5
    # bilinear align_corners=False (loss resize) and depthwise 3x3
    m = _rand((2, 3, 5, 7), 87, True)
    z = ops.bilinear_resize(m, (20, 21))
    gz = _rand(tuple(z.shape), 88)
    z.backward(gz)
    mr = m.detach().clone().requires_grad_()
    zr = F.interpolate(mr, size=[20, 21], mode="bilinear")
    zr.backward(gz)
    assert rel_err(z.detach(), zr.detach()) < 1e-6 and rel_err(m.grad, mr.grad) < 1e-5
    d = _rand((3, 4, 6, 8), 89, True)
    dw = (_rand((4, 1, 3, 3), 90) * 0.3).requires_grad_()
    o = ops.depthwise_conv2d(d, dw)
    go = _rand(tuple(o.shape), 91)
    o.backward(go)
    dr, dwr = d.detach().clone().requires_grad_(), dw.detach().clone().requires_grad_()
    orf = F.conv2d(dr, dwr, None, padding=1, groups=4)
    orf.backward(go)
    assert rel_err(o.detach(), orf.detach()) < 1e-5 and rel_err(d.grad, dr.grad) < 1e-5 and rel_err(dw.grad, dwr.grad) < 1e-5


@pytest.mark.parametrize("n,cin,cout,d,h,w", [(2, 128, 128, 3, 3, 3), (1, 130, 128, 2, 3, 4)])
def test_conv3d_tiny_volume_gemm_path(n, cin, cout, d, h, w):
    """<= 6^3 voxels with >= 128x128 channels: im2col3 / col2im3 + the dense products of csrc/kernels/gemm.h instead of the tile kernels."""
    assert cin * cout * 27 >= ops.SMALL_CONV_MIN_WEIGHTS
    _conv_check(n, cin, cout, d, h, w, 3)


@pytest.mark.parametrize("n,cin,cout,d,h,w", [(2, 20, 50, 10, 12, 20), (1, 96, 48, 4, 32, 36), (3, 1, 48, 6, 14, 18)])
def test_conv1x1_wgrad_big_volume_path(n, cin, cout, d, h, w):
    """1x1x1 convolution with >= 4096 voxels: channel-major row-split MFMA reduction (conv1x1_wgrad_kernel): ragged
    channel blocks, a ragged 16-voxel tail group (S % 16 != 0), several batch items."""
    assert n * d * h * w >= ops.CONV1X1_WGRAD_MIN_VOXELS and (d * h * w) % 4 == 0
    _conv_check(n, cin, cout, d, h, w, 1)


@pytest.mark.parametrize("n,cout,d,h,w", [(2, 16, 5, 7, 12), (1, 20, 4, 6, 8)])
def test_first_conv_weight_gradient_through_shifted_planes(monkeypatch, n, cout, d, h, w):
    """One input channel on a big volume: the weight gradient runs as 27 shifted planes + the 1x1x1 reduction (threshold lowered
    so that the emulator reaches that branch)."""
    monkeypatch.setattr(ops, "CONV1X1_WGRAD_MIN_VOXELS", 1)
    _conv_check(n, 1, cout, d, h, w, 3)


@pytest.mark.parametrize("n,cout,d,h,w,zero_bias", [(2, 16, 5, 9, 12, True), (1, 12, 3, 8, 24, True), (1, 16, 4, 17, 16, True),
                                                   (1, 16, 3, 6, 10, True), (2, 5, 2, 8, 8, False)])
def test_first_conv_dedicated_kernels(n, cout, d, h, w, zero_bias):
    """conv_cin1.h: the first convolution of the backbones (one input channel, the volume itself: no input gradient) forward and weight
    gradient on their own kernels — partial x groups (W = 24, 12), H not a multiple of the 8-row slab, fewer than 16 output channels, a
    width that is not a multiple of 4 and a bias whose gradient is wanted (both: weight gradient on the shifted-planes path)."""
    x = _rand((n, 1, d, h, w), 21)
    wt = (_rand((cout, 1, 3, 3, 3), 22) * 0.3).requires_grad_()
    b = (_rand((cout,), 23) * 0.1).requires_grad_()
    gy = _rand((n, cout, d, h, w), 24)
    y = ops.conv3d(x, wt, b, zero_bias_grad=zero_bias)
    assert _lib.lib().icl_last_kernel_name().decode() == "conv_cin1_fwd_kernel"
    y.backward(gy)
    wr, br = wt.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    yr = F.conv3d(x, wr, br, padding=1)
    yr.backward(gy)
    assert rel_err(y.detach(), yr.detach()) < 1e-5
    assert rel_err(wt.grad, wr.grad) < 1e-5
    if not zero_bias:
        assert rel_err(b.grad, br.grad) < 1e-5
    else:
        assert float(b.grad.abs().max()) == 0.0


def test_conv1x1_big_volume_runs_as_batched_product(monkeypatch):
    """>= 65536 voxels (threshold lowered here): forward / input gradient of a 1x1x1 convolution with many channels are one
    batched product of csrc/kernels/gemm.h on the channel-major volume (bias indexed by the output row), with <= 16 channels the
    voxel-streaming VALU kernel (conv1x1_stream_kernel: all outputs of a voxel quad per thread)."""
    monkeypatch.setattr(ops, "CONV1X1_GEMM_MIN_VOXELS", 4096)
    _conv_check(2, 20, 50, 10, 12, 20, 1)
    x = _rand((1, 6, 16, 16, 16), 5, True)
    w = (_rand((3, 6, 1, 1, 1), 6) * 0.3).requires_grad_()
    y = ops.conv3d(x, w, None)
    y.sum().backward()
    assert rel_err(y.detach(), F.conv3d(x.detach(), w.detach())) < 1e-5 and x.grad is not None and w.grad is not None


@pytest.mark.parametrize("n,cin,cout", [(1, 16, 2), (2, 16, 16), (1, 5, 3), (1, 2, 16)])
def test_conv1x1_stream_kernel_final_conv_class(n, cin, cout):
    """The `final` convolution class (unet_3D_icl.py:65): <= 16 channels on >= 65536 voxels, forward and input gradient."""
    _conv_check(n, cin, cout, 32, 32, 64 // n, 1)


@pytest.mark.timeout(1200)
def test_fused_sgd_step_scope_equals_the_plain_loop_with_torch_sgd():
    """The one-line optimiser swap of INTEGRATION.md §2 on the kernel emulation: the reference loop body
    (`outputs = model(..); losses; optimizer.zero_grad(); loss.backward(); optimizer.step()`) on a small `unet_3D_icl` with
    `FusedSGD(model.parameters(), ...)` — which finds the model through its tagged parameters and opens ICLTrainer's step scope (packed
    weights, factored token-axis gradients, deferred bias gradients, lane bookkeeping) from a forward pre-hook — leaves every parameter
    where `torch.optim.SGD` on dense gradients leaves it, over two iterations (momentum included); the scope is closed after `step()`,
    an evaluation forward does not open it, and a forward whose step never comes leaves a scope that `abandon_step()` discards."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.optim import FusedSGD, tag_model_parameters
    from icl_amd.utils import losses
    from icl_amd.utils.hashfill import synthetic_labels
    nc = 2
    old_min = ops.FactoredGrads.min_elems
    ops.FactoredGrads.min_elems = 64 * 64
    try:
        vol = synthetic_volume((2, 1, 16, 16, 16), 900)
        lab = synthetic_labels((2, 16, 16, 16), 950, nc)

        def run(fused):
            torch.manual_seed(7)
            model = tag_model_parameters(unet_3D_icl(feature_scale=16, n_classes=nc, in_channels=1, icl_in_resolutions=(1, 2, 4), icl_heads=(8, 4, 2)))
            for m in model.modules():           # parity mode: no dropout / drop-path randomness between the two runs
                if hasattr(m, "p") and m.__class__.__name__ == "Dropout3":
                    m.p = 0.0
                if hasattr(m, "drop_prob"):
                    m.drop_prob = 0.0
            opt = (FusedSGD if fused else torch.optim.SGD)(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2)
            ce = torch.nn.CrossEntropyLoss()
            dice, aux, pse = losses.DiceLoss(nc), losses.AuxLoss3D(nc, (16, 16, 16)), losses.PseudoSoftLoss3D(nc, (16, 16, 16))
            seen = []
            for it in range(2):
                outputs = model(vol[:1], vol[1:])
                if fused:
                    seen.append((ops.PackedWeights.current is not None, ops.FactoredGrads.enabled, ops.WgradLane.uses is not None))
                loss = (dice(torch.softmax(outputs[0], 1), lab[:1].unsqueeze(1)) + ce(outputs[0], lab[:1]) + aux(outputs[2], lab[:1])
                        + pse(outputs[3], outputs[1]) + 10 * losses.softmax_mse_loss(outputs[3], outputs[4]))
                opt.zero_grad()
                loss.backward()
                if fused and it == 0:
                    fac = [k for k, p in model.named_parameters() if getattr(p, "_icl_factors", None)]
                    assert sum("mlp2" in k for k in fac) == 4, fac          # the token-axis gradients stayed factored
                opt.step()
                if fused:
                    assert ops.PackedWeights.current is None and not ops.FactoredGrads.enabled and ops.FactoredGrads.uses is None
                    assert ops.WgradLane.uses is None and not ops.WgradLane.open and ops.DeferredBiasGrads.pending is None
            if fused:
                assert seen == [(True, True, True)] * 2, seen
                with torch.no_grad():
                    model.eval()
                    model(vol[:1], inference=True)
                    model.train()
                assert ops.PackedWeights.current is None
                model(vol[:1], vol[1:])                   # opens the scope; no backward, no step
                assert ops.PackedWeights.current is not None
                model(vol[:1], vol[1:])                   # ... the next forward keeps it open (round 6: accumulation; use counts continue)
                assert ops.PackedWeights.current is not None
                opt.abandon_step()
                assert ops.PackedWeights.current is None and not ops.FactoredGrads.enabled
            return {k: p.detach().clone() for k, p in model.named_parameters()}, float(loss)

        a, la = run(False)
        b, lb = run(True)
        assert abs(la - lb) < 1e-5 * max(1.0, abs(la)), (la, lb)
        # (the gradients of the first iteration are bit-equal; the two optimisers round their updates differently — an ulp per weight — and
        # the second iteration of this 16^3, 4-channel network (InstanceNorm over 4096 voxels) amplifies that to a few 1e-4 of a layer's
        # largest weight: the band is the path's 1e-3, relative to the layer)
        bad = [k for k in a if float((a[k] - b[k]).abs().max()) > 1e-3 * float(a[k].abs().max()) + 1e-6]
        assert not bad, bad[:8]
    finally:
        ops.FactoredGrads.min_elems = old_min


@pytest.mark.timeout(1200)
def test_fused_sgd_gradients_are_complete_after_backward_and_accumulate_over_micro_batches(monkeypatch):
    """ADVICE round 5 (optim.py).  (1) When `loss.backward()` returns, every `.grad` is whole: the lane of the deep levels' weight gradients
    is joined and the deferred bias / LayerNorm gradients are reduced by an end-of-backward callback (ops.BackwardEnd), not in `step()` —
    code between backward and step (`clip_grad_norm_`, logging) sees what it sees with torch.optim.SGD.  (2) forward, backward, forward,
    backward, step — gradient accumulation with `update_in_backward=False` — ends where torch.optim.SGD ends on the summed dense
    gradients (factors, bias gradients and use counts keep accumulating; nothing is abandoned).  (3) With `update_in_backward=True` a
    second forward after a backward that already updated the big matrices raises instead of training on half-updated weights.
    (4) `torch.save`-style pickling of a factory-tagged, optimiser-hooked model works."""
    import io
    import pickle
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.optim import FusedSGD, model_of, tag_model_parameters
    from icl_amd.utils import losses
    from icl_amd.utils.hashfill import synthetic_labels
    nc = 2
    old_min = ops.FactoredGrads.min_elems
    ops.FactoredGrads.min_elems = 64 * 64
    try:
        vols = [synthetic_volume((2, 1, 16, 16, 16), 900 + i) for i in range(2)]
        labs = [synthetic_labels((2, 16, 16, 16), 950 + i, nc) for i in range(2)]

        def build():
            torch.manual_seed(7)
            model = tag_model_parameters(unet_3D_icl(feature_scale=16, n_classes=nc, in_channels=1, icl_in_resolutions=(1, 2, 4), icl_heads=(8, 4, 2)))
            for m in model.modules():
                if hasattr(m, "p") and m.__class__.__name__ == "Dropout3":
                    m.p = 0.0
                if hasattr(m, "drop_prob"):
                    m.drop_prob = 0.0
            return model

        ce = torch.nn.CrossEntropyLoss()
        dice, aux, pse = losses.DiceLoss(nc), losses.AuxLoss3D(nc, (16, 16, 16)), losses.PseudoSoftLoss3D(nc, (16, 16, 16))

        def loss_of(model, vol, lab):
            outputs = model(vol[:1], vol[1:])
            return (dice(torch.softmax(outputs[0], 1), lab[:1].unsqueeze(1)) + ce(outputs[0], lab[:1]) + aux(outputs[2], lab[:1])
                    + pse(outputs[3], outputs[1]) + 10 * losses.softmax_mse_loss(outputs[3], outputs[4]))

        def run(fused):
            model = build()
            opt = (FusedSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2, update_in_backward=False) if fused
                   else torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2))
            opt.zero_grad()
            norms = None
            for mb in range(2):
                loss_of(model, vols[mb], labs[mb]).backward()
                if mb == 0:
                    # between backward and step: every gradient the dense path has exists already (factored ones as factors)
                    have = {k for k, p in model.named_parameters() if p.grad is not None or getattr(p, "_icl_factors", None)}
                    norms = (have, torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1e9))
                    if fused:
                        assert not ops.DeferredBiasGrads.pending and not ops.WgradLane._used
            opt.step()
            return {k: p.detach().clone() for k, p in model.named_parameters()}, norms, model, opt

        a, (have_a, _), _, _ = run(False)
        b, (have_b, _), model, opt = run(True)
        assert have_a == have_b, sorted(have_a ^ have_b)[:8]
        bad = [k for k in a if float((a[k] - b[k]).abs().max()) > 1e-3 * float(a[k].abs().max()) + 1e-6]
        assert not bad, bad[:8]

        # (3) update inside backward + a second forward before step(): refused loudly
        model2 = build()
        # (the size threshold of the one-pass input gradient + update is the 13,824^2 matrices'; lowered to this model's 64^2 ones)
        monkeypatch.setattr(FusedSGD, "can_update_in_backward",
                            lambda self, p, rows: rows <= 32 and p.numel() >= 64 * 64 and id(p) in self._group_of())
        opt2 = FusedSGD(model2.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-2)
        loss_of(model2, vols[0], labs[0]).backward()
        assert opt2._updated_in_backward, "the token-axis matrices were expected to take the update-in-backward path"
        with pytest.raises(RuntimeError, match="update_in_backward"):
            model2(vols[1][:1], vols[1][1:])
        opt2.abandon_step()

        # (4) the tag is a registry entry, not a weak reference on the tensor; the scope hook pickles as a no-op
        assert model_of(next(model.parameters())) is model
        buf = io.BytesIO()
        pickle.dump(model, buf)
        clone = pickle.loads(buf.getvalue())
        assert sorted(clone.state_dict()) == sorted(model.state_dict())
    finally:
        ops.FactoredGrads.min_elems = old_min
        ops.BackwardEnd.hook = None


def test_unet_blocks_take_the_materialised_path_when_they_or_their_parents_are_hooked():
    """ADVICE round 5 (layers.py): `UnetConv3.forward(lazy=True)` calls `conv2.forward_lazy` directly and may return an ops.LazyAct; with
    a forward hook on the block, on its `conv2`, on the enclosing `UnetUp3_CT`, or a global module hook, the ordinary `__call__` path runs —
    the hook fires and sees a tensor — and the results equal the unhooked ones."""
    from icl_amd.networks.layers import UnetConv3, UnetUp3_CT
    torch.manual_seed(5)
    blk = UnetConv3(4, 8)
    up = UnetUp3_CT(16, 8)
    x = _rand((1, 4, 4, 8, 16), 9)
    skip, deep = _rand((1, 8, 4, 8, 16), 10), _rand((1, 16, 2, 4, 8), 11)
    with torch.no_grad():
        ref = ops.materialized(blk(x, lazy=True))
        ref_up = ops.materialized(up(skip, deep, lazy=True))
        for target, call, want in ((blk, lambda: blk(x, lazy=True), ref), (blk.conv2, lambda: blk(x, lazy=True), ref),
                                   (up, lambda: up(skip, deep, lazy=True), ref_up), (up.conv, lambda: up(skip, deep, lazy=True), ref_up)):
            seen = []
            h = target.register_forward_hook(lambda m, i, o: seen.append(type(o)))
            out = call()
            h.remove()
            assert seen == [torch.Tensor], seen
            assert isinstance(out, torch.Tensor) and torch.allclose(out, want, rtol=1e-5, atol=1e-6)
        seen = []
        h = torch.nn.modules.module.register_module_forward_hook(lambda m, i, o: seen.append(type(o)) if m is blk.conv2 else None)
        try:
            out = blk(x, lazy=True)
        finally:
            h.remove()
        assert seen == [torch.Tensor] and isinstance(out, torch.Tensor)


def test_conv_block_falls_back_to_its_children_when_they_are_hooked_or_swapped():
    """ConvBlock.forward runs its three children as one fused operator only while they are the stock modules with no
    forward hooks; a hook on a child fires (and sees the child's output), a swapped child is honoured — results equal the fused path."""
    from icl_amd.networks.layers import ConvBlock
    torch.manual_seed(3)
    blk = ConvBlock(4, 8)
    x = _rand((1, 4, 4, 8, 16), 5)
    with torch.no_grad():
        fused = blk(x)
        seen = []
        h = blk[1].register_forward_hook(lambda m, i, o: seen.append(o.shape))
        hooked = blk(x)
        h.remove()
        assert seen == [fused.shape] and torch.allclose(hooked, fused, rtol=1e-5, atol=1e-6)
        assert blk._stock()
        blk[2] = torch.nn.Hardtanh(0.0, 0.5)          # swapped through the reference-compatible index
        assert not blk._stock()
        assert torch.allclose(blk(x), fused.clamp(0.0, 0.5), rtol=1e-5, atol=1e-6)


@pytest.mark.timeout(900)
def test_deferred_instance_norm_equals_the_materialised_path(monkeypatch):
    """Round 5: inside a step scope the `conv2` blocks of the U-Net's upper levels hand their consumers the RAW convolution output plus
    (scale, shift) (ops.LazyAct): max-pooling, the skip / up-sampling into the concat buffer and `final(dropout2(.))` apply
    relu(fma(y, scale, shift)) while they read it.  On a two-level U-Net built from the backbone's own blocks (threshold lowered so that
    both levels defer; every consumer kind is reached: pooling of a LazyAct, a LazyAct skip, a LazyAct deep map, the lazy 1x1x1
    convolution with and without dropout) logits, the input gradient and every parameter gradient equal the path that writes the
    normalised tensors (ICL_LAZY_NORM=0)."""
    from icl_amd.networks.layers import Conv3d, UnetConv3, UnetUp3_CT
    monkeypatch.setenv("ICL_CONV_SPLIT_MIN", "1")
    x = _rand((2, 1, 8, 16, 16), 21)

    class Mini(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.conv2 = UnetConv3(1, 16), UnetConv3(16, 32)
            self.up = UnetUp3_CT(32, 16)
            self.final = Conv3d(16, 2, 1, kaiming_normal=True)

        def forward(self, x, p):
            c1, p1 = ops.skip_and_pool(self.conv1(x, lazy=True))
            c2 = self.conv2(p1, lazy=True)
            kinds = [isinstance(c1, ops.LazyAct), isinstance(c2, ops.LazyAct)]
            up1 = self.up(c1, c2, lazy=True)
            kinds.append(isinstance(up1, ops.LazyAct))
            if isinstance(up1, ops.LazyAct):
                return ops.conv1x1_lazy(up1, self.final.weight, self.final.bias, p, seed=77), kinds
            return (ops.dropout_conv1x1(up1, self.final.weight, self.final.bias, p, seed=77) if p > 0 else self.final(up1)), kinds

    def run(lazy, p):
        monkeypatch.setenv("ICL_LAZY_NORM", "1" if lazy else "0")
        monkeypatch.setenv("ICL_LAZY_NORM_MIN", "32")
        torch.manual_seed(5)
        model = Mini()
        packed = ops.PackedWeights()
        xin = x.clone().requires_grad_()
        for it in range(2):          # the second iteration runs with packed + split weights: the epilogue statistics exist
            packed.begin_step()
            for q in model.parameters():
                q.grad = None
            xin.grad = None
            y, kinds = model(xin, p)
            (y * _rand(tuple(y.shape), 22)).sum().backward()
            packed.end_step()
        return y.detach(), xin.grad.clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None}, kinds

    for p in (0.3,):      # (without dropout: test_lazy_final_convolution_fused_kernels[0.0] and every GPU golden test)
        y0, gx0, g0, k0 = run(False, p)
        y1, gx1, g1, k1 = run(True, p)
        assert k0 == [False, False, False] and k1 == [True, True, True], (k0, k1)          # the deferred path really ran
        # (n * s < 65536 here: the lazy 1x1x1 convolution materialises and runs the ordinary operators; its fused kernels are GPU-tested)
        assert rel_err(y1, y0) < 2e-5 and rel_err(gx1, gx0) < 2e-4, (p, rel_err(y1, y0), rel_err(gx1, gx0))
        assert sorted(g0) == sorted(g1)
        # (conv biases in front of an InstanceNorm have a true gradient of 0: both paths hold rounding noise there)
        bad = [(k, rel_err(g1[k], g0[k])) for k in g0 if not k.endswith(".0.bias") and rel_err(g1[k], g0[k]) > 5e-4]
        assert not bad, (p, bad[:6])


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_lazy_final_convolution_fused_kernels(p):
    """ops.conv1x1_lazy on a volume big enough for the fused kernels (icl_conv1x1_dropout_norm / icl_conv1x1_wgrad_dropout_norm):
    y = conv1x1(dropout(relu(fma(t, scale, shift)))) in one pass over the raw tensor == the same operators on the materialised tensor,
    forward, input gradient (the gradient of the normalised activation) and weight / bias gradients; same dropout mask (same seed)."""
    n, cin, cout, r = 2, 16, 2, 32
    t = _rand((n, cin, r, r, r), 31)
    ss = torch.stack([0.5 + 0.1 * _rand((n * cin,), 32).abs(), 0.2 * _rand((n * cin,), 33)], 1).contiguous()
    w = (_rand((cout, cin, 1, 1, 1), 34) * 0.3)
    b = _rand((cout,), 35)
    gy = _rand((n, cout, r, r, r), 36)
    res = []
    for lazy in (True, False):
        tt = t.clone().requires_grad_()
        ww, bb = w.clone().requires_grad_(), b.clone().requires_grad_()
        la = ops.LazyAct(tt, ss)
        if lazy:
            y = ops.conv1x1_lazy(la, ww, bb, p, seed=91)
        else:
            a = la.materialize()
            y = ops.dropout_conv1x1(a, ww, bb, p, seed=91) if p > 0 else ops.conv3d(a, ww, bb)
        y.backward(gy)
        res.append((y.detach(), tt.grad.clone(), ww.grad.clone(), bb.grad.clone()))
    for a, b_ in zip(*res):
        assert rel_err(a, b_) < 2e-5
    ref = torch.clamp_min(t * ss[:, 0].view(n, cin, 1, 1, 1) + ss[:, 1].view(n, cin, 1, 1, 1), 0)
    if p == 0.0:
        assert rel_err(res[0][0], F.conv3d(ref, w, b)) < 2e-5


def test_conv3d_weight_gradient_with_exchanged_roles(monkeypatch):
    """One cout block, three cin blocks (up_concat1.conv1's class, 48 -> 16): ops._Conv3d.backward hands dY to the z-column weight-gradient
    kernel as the halo operand and x as the three plain blocks, and writes the [cin][cout] result back transposed with mirrored taps.
    Against torch, with the bias gradient off (the convolution feeds an InstanceNorm: zero_bias_grad) — and equal to the direct path."""
    monkeypatch.setattr(ops, "WGRAD_SWAP_MIN_VOXELS", 1)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("ICL_WGRAD_SWAP", flag)
        x = _rand((1, 48, 8, 8, 16), 41, True)
        wt = (_rand((16, 48, 3, 3, 3), 42) * 0.2).requires_grad_()
        b = (_rand((16,), 43) * 0.1).requires_grad_()
        gy = _rand((1, 16, 8, 8, 16), 44)
        y = ops.conv3d(x, wt, b, zero_bias_grad=True)
        y.backward(gy)
        got[flag] = wt.grad.clone()
        xr, wr = x.detach().clone().requires_grad_(), wt.detach().clone().requires_grad_()
        F.conv3d(xr, wr, b.detach(), padding=1).backward(gy)
        assert rel_err(wt.grad, wr.grad) < 1e-5 and rel_err(x.grad, xr.grad) < 1e-5, flag
        assert float(b.grad.abs().max()) == 0.0
    assert rel_err(got["1"], got["0"]) < 1e-6


@pytest.mark.timeout(900)
@pytest.mark.parametrize("nc,bs,ba,drop", [(2, 2, 1, 0.0), (3, 2, 1, 0.25), (5, 4, 2, 0.0)])
def test_fused_query_chain_equals_the_operator_by_operator_aligner(nc, bs, ba, drop):
    """Round 6 (csrc/kernels/qchain.h, ops.query_attend): the query half of Class_Decoder as fused stages — LayerNorm / GELU / bias / the
    drop-path residual forms / the batch broadcast of the guided query on the products they feed or follow, every parameter gradient of a
    level in one launch — against the operator-by-operator mirror it replaces (ICL_QCHAIN off), on a small InherentConsistent: the own-query
    pair call (`forward_labeled_pair`: both batch halves as one batch, queries handed down the levels), the guided call with and without
    its updated queries, and the plain labeled call; attention maps, updated queries, and the gradient of EVERY parameter and of the
    three feature maps.  nc = 5, bs = 4: 20 rows = three passes of eight rows through every stage; drop = 0.25: both per-sample drop-path
    sites active (the two paths draw the same seeds in the same order)."""
    from icl_amd.networks.aligner import InherentConsistent
    chans, res, heads = (64, 32, 16), (2, 2, 4), (4, 2, 2)

    def build():
        torch.manual_seed(11)
        m = InherentConsistent(in_chans=chans, depths=(2, 2, 2), input_resolution=res, num_classes=nc, num_heads=heads, drop_path_rate=drop)
        with torch.no_grad():
            getattr(m, "guided_Q").copy_(_rand((1, nc, chans[0]), 77))
            for k, p in m.named_parameters():
                if p.dim() == 1 and ("norm" in k) and k.endswith("weight"):
                    p.copy_(1.0 + 0.1 * _rand(p.shape, 3 + p.numel()))
        m.train()
        return m

    def run(fused, mode):
        ops.QCHAIN = fused
        ops.QC_FUSE_MAX_ROWS = 32      # (the product takes the fused chain for one pass of eight rows only; the stages hold 32)
        ops.StepRNG.tensor = None
        torch.manual_seed(5)      # the seeds of the drop-path sites come from torch's generator when no step RNG is active
        m = build()
        feats = [_rand((bs, c, r, r, r), 20 + i, grad=True) for i, (c, r) in enumerate(zip(chans, res))]
        if mode == "pair":
            (maps_a, qs_a), (maps_b, qs_b) = m.forward_labeled_pair(feats, ba)
            outs = maps_a + maps_b + qs_a + qs_b
        elif mode == "labeled":
            maps, qs = m(feats, None, "labeled")
            outs = maps + qs
        else:
            gq = [_rand((1, nc, c), 40 + i, grad=True) for i, c in enumerate(chans)]
            maps, qs = m(feats, gq, "unlabeled", need_queries=(mode == "guided+q"))
            outs = maps + ([q for q in qs if q is not None] if mode == "guided+q" else [])      # "guided": only the maps are read
            feats = feats + gq
        loss = sum((o * _rand(o.shape, 60 + i)).sum() for i, o in enumerate(outs))
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        return [o.detach().clone() for o in outs], grads, [f.grad.clone() for f in feats]

    try:
        for mode in ("pair", "labeled", "guided", "guided+q"):
            if mode == "guided" and drop > 0:
                continue      # (without its query half the fused guided call draws fewer drop-path seeds: other masks in the map chains)
            o0, g0, f0 = run(False, mode)
            o1, g1, f1 = run(True, mode)
            assert len(o0) == len(o1)
            for a, b in zip(o0, o1):
                assert rel_err(b, a) < 2e-5, (mode, rel_err(b, a))
            assert {k for k, v in g0.items() if v is None} == {k for k, v in g1.items() if v is None}, mode
            # (max-norm relative to the tensor, with an absolute floor: level 0 has ONE token, so its token-axis LayerNorm and everything
            # behind it carry exact zeros in one evaluation order and 1e-9 in another)
            close = lambda b, a: float((b - a).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-6      # noqa: E731
            for k in g0:
                if g0[k] is not None:
                    assert close(g1[k], g0[k]), (mode, k, float((g1[k] - g0[k]).abs().max()), float(g0[k].abs().max()))
            for a, b in zip(f0, f1):
                assert close(b, a), (mode, float((b - a).abs().max()), float(a.abs().max()))
    finally:
        ops.QCHAIN = True
        ops.QC_FUSE_MAX_ROWS = ops.QC_ROWS_PER_PASS


@pytest.mark.parametrize("nc", [2, 5])
def test_all_loss_terms_in_one_launch_are_bit_identical_to_the_single_term_calls(nc):
    """Round 6 (ops.fused_losses, icl_loss_fwd_multi / icl_loss_bwd_multi): the ten reductions of the trainer's objective — CE + Dice on the
    logits, three resized AuxLoss3D maps, three PseudoSoftLoss3D maps, three softmax-MSE pairs at their own resolutions — in one
    statistics launch, one finalize launch and one gradient launch: every term and every gradient equals the single-term `_FusedLoss`
    call bit for bit (job j of the multi kernels is launch j of the single ones), with some outputs unused (no gradient for them)."""
    B, S = 2, (6, 8, 10)
    lab = torch.randint(0, nc, (B,) + S, dtype=torch.int64)
    mk = lambda shape, seed: _rand((B, nc) + shape, seed)      # noqa: E731
    small = [(3, 4, 5), (6, 4, 5), S]

    def build():
        logits = mk(S, 1).requires_grad_()
        aux = [mk(S, 10 + i).requires_grad_() for i in range(3)]
        pse = [mk(S, 20 + i).requires_grad_() for i in range(3)]
        con_a = [mk(sh, 30 + i).requires_grad_() for i, sh in enumerate(small)]
        con_b = [mk(sh, 40 + i) for i, sh in enumerate(small)]
        tgt = mk(S, 50)
        terms = [(logits, lab, 1)] + [(a, lab, 1) for a in aux] + [(a, tgt, 2) for a in pse] + [(a, b, 3) for a, b in zip(con_a, con_b)]
        return terms, [logits] + aux + pse + con_a

    def total(pairs):
        # every first term, the Dice of the hard / soft-Dice terms; the MSE terms' second output and the soft-Dice terms' first stay unused
        w = [0.3 + 0.1 * j for j in range(len(pairs))]
        return sum(w[j] * (p[0] if terms_modes[j] in (1, 3) else p[1]) + (0.5 * p[1] if terms_modes[j] == 1 else 0.0) for j, p in enumerate(pairs))

    terms, leaves = build()
    terms_modes = [m for _, _, m in terms]
    single = [ops._FusedLoss.apply(a, (t.long() if m <= 1 else t.detach()), None, m, False) for a, t, m in terms]
    total(single).backward()
    g_single = [x.grad.clone() for x in leaves]
    terms2, leaves2 = build()
    multi = ops.fused_losses(terms2)
    assert len(multi) == len(single)
    for p, q in zip(single, multi):
        assert torch.equal(p[0].detach(), q[0].detach()) and torch.equal(p[1].detach(), q[1].detach())
    total(multi).backward()
    for a, b in zip(g_single, (x.grad for x in leaves2)):
        assert torch.equal(a, b)
