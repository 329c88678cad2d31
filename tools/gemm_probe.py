"""Time the dense-product kernels of csrc/kernels/gemm.h against the library GEMM (torch.matmul -> rocBLAS/hipBLASLt) on the
shapes of the ICL step: the skinny 13,824^2 / 1,728^2 token-axis MLP products (weight streaming) and the tiled general product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402

dev = torch.device("cuda")


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def err(a, b):
    return float((a - b).abs().max() / b.abs().max())


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "stream"):
    for N in (13824, 1728):
        w = torch.randn(N, N, device=dev) * 0.01
        b = torch.randn(N, device=dev)
        gb = w.numel() * 4 / 1e9
        for m in (8, 12, 16, 24, 32):
            x = torch.randn(m, N, device=dev)
            g = torch.randn(m, N, device=dev)
            y, yr = ops.linear_forward_raw(x, w, b), torch.nn.functional.linear(x, w, b)
            gx, gxr = ops.linear_dgrad_raw(g, w), g @ w
            f1, f0 = t(lambda: ops.linear_forward_raw(x, w, b)), t(lambda: torch.nn.functional.linear(x, w, b))
            d1, d0 = t(lambda: ops.linear_dgrad_raw(g, w)), t(lambda: g @ w)
            print(f"N={N:6d} M={m:3d}  fwd ours {f1:7.1f} us ({gb / f1 * 1e3:5.2f} TB/s) lib {f0:7.1f} us ({gb / f0 * 1e3:5.2f})  err {err(y, yr):.1e} | "
                  f"dgrad ours {d1:7.1f} us ({gb / d1 * 1e3:5.2f} TB/s) lib {d0:7.1f} us ({gb / d0 * 1e3:5.2f})  err {err(gx, gxr):.1e}", flush=True)

if which in ("all", "tiled"):
    shapes = [  # (rows, in, out, note)
        (27648, 64, 128, "fc_kv scale 2"), (27648, 64, 64, "token proj scale 2"), (3456, 128, 256, "fc_kv scale 1"),
        (432, 3456, 256, "center.conv1 im2col"), (432, 6912, 256, "center.conv2 im2col"),
        (128, 13824, 13824, "mlp2 nc=16"), (64, 13824, 13824, "mlp2 nc=16 uscl"), (256, 1728, 1728, "mlp2 scale 1 nc=16"),
        (32, 216, 216, "mlp2 scale 0"), (4, 256, 1024, "mlp.fc1"), (4, 1024, 256, "mlp.fc2"),
        (221184, 48, 144, "swin qkv s0"), (221184, 48, 192, "swin mlp fc1 s0"), (221184, 192, 48, "swin mlp fc2 s0"),
        (27648, 96, 288, "swin qkv s1"), (27648, 384, 96, "swin mlp fc2 s1"), (3456, 192, 576, "swin qkv s2"),
        (432, 384, 1152, "swin qkv s3"), (432, 1536, 384, "swin mlp fc2 s3"), (54, 3072, 768, "swin mlp fc2 s4"),
    ]
    for rows, i, o, note in shapes:
        x = torch.randn(rows, i, device=dev)
        w = torch.randn(o, i, device=dev) * 0.05
        b = torch.randn(o, device=dev)
        g = torch.randn(rows, o, device=dev)
        y, yr = ops.linear_forward_raw(x, w, b), torch.nn.functional.linear(x, w, b)
        gx, gxr = ops.linear_dgrad_raw(g, w), g @ w
        fl = 2.0 * rows * i * o / 1e6
        f1, f0 = t(lambda: ops.linear_forward_raw(x, w, b)), t(lambda: torch.nn.functional.linear(x, w, b))
        d1, d0 = t(lambda: ops.linear_dgrad_raw(g, w)), t(lambda: g @ w)
        line = (f"{note:22s} [{rows:6d},{i:5d}]->{o:5d}  fwd ours {f1:7.1f} us ({fl / f1:6.1f} TF) lib {f0:7.1f} ({fl / f0:6.1f})  err {err(y, yr):.1e} | "
                f"dgrad ours {d1:7.1f} ({fl / d1:6.1f} TF) lib {d0:7.1f} ({fl / d0:6.1f})  err {err(gx, gxr):.1e}")
        if rows < 2048:
            dw, dwr = ops._tall_atb(g, x, False)[0], g.t() @ x
            w1, w0 = t(lambda: ops._tall_atb(g, x, False)), t(lambda: g.t() @ x)
            line += f" | wgrad ours {w1:7.1f} ({fl / w1:6.1f}) lib {w0:7.1f} ({fl / w0:6.1f}) err {err(dw, dwr):.1e}"
        print(line, flush=True)

if which in ("all", "convt"):
    import torch.nn.functional as F
    for (b, cin, cout, r) in ((2, 768, 384, 3), (2, 384, 192, 6), (2, 192, 96, 12), (2, 96, 48, 24), (2, 48, 48, 48)):
        x = torch.randn(b, cin, r, r, r, device=dev, requires_grad=True)
        w = (torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.05).requires_grad_()
        y = ops.conv_transpose3d_k2s2(x, w)
        yr = F.conv_transpose3d(x, w, stride=2)
        gy = torch.randn_like(y)
        gx, gw = torch.autograd.grad(y, (x, w), gy)
        gxr, gwr = torch.autograd.grad(yr, (x, w), gy)
        f1 = t(lambda: ops.conv_transpose3d_k2s2(x, w))
        def fb():
            yy = ops.conv_transpose3d_k2s2(x, w)
            torch.autograd.grad(yy, (x, w), gy)
        fb1 = t(fb)
        fl = 2.0 * b * r ** 3 * cin * cout * 8 / 1e6
        print(f"convT {cin:4d}->{cout:4d} @{r:2d}^3  fwd {f1:7.1f} us ({fl / f1:6.1f} TF)  fwd+bwd {fb1:7.1f} us ({3 * fl / fb1:6.1f} TF)  "
              f"err y {err(y, yr):.1e} gx {err(gx, gxr):.1e} gw {err(gw, gwr):.1e}", flush=True)

if which == "one":
    rows, i, o = 128, 13824, 13824
    x = torch.randn(rows, i, device=dev)
    w = torch.randn(o, i, device=dev) * 0.05
    b = torch.randn(o, device=dev)
    f1 = t(lambda: ops.linear_forward_raw(x, w, b))
    print(f"[{rows},{i}]->{o} fwd {f1:7.1f} us ({2.0 * rows * i * o / 1e6 / f1:6.1f} TF)")

if which in ("fused",):
    # input gradient + SGD step of a token-axis matrix: one pass (linear_dgrad_sgd_kernel) vs linear_dgrad + sgd_factored_kernel
    import ctypes
    from icl_amd import _lib
    L = _lib.lib()
    for N in (13824, 1728):
        w = torch.randn(N, N, device=dev) * 0.01
        mo = torch.randn(N, N, device=dev) * 0.001
        for m in (4, 8, 16, 32):
            x = torch.randn(m, N, device=dev)
            g = torch.randn(m, N, device=dev) * 0.01
            gx = torch.empty(m, N, device=dev)
            ws = torch.empty(max(1, L.icl_linear_ws_bytes(m, N, N, 3) // 4), device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

            def fused():
                _lib.check(L.icl_linear_dgrad_sgd(g.data_ptr(), x.data_ptr(), w.data_ptr(), mo.data_ptr(), gx.data_ptr(), ws.data_ptr(), m, N, N,
                                                  1e-3, 0.9, 1e-4, 0, None, st), "fused")

            def separate():
                ops.linear_dgrad_raw(g, w)
                _lib.check(L.icl_sgd_step_factored(w.data_ptr(), mo.data_ptr(), g.data_ptr(), x.data_ptr(), m, N, N, 1e-3, 0.9, 1e-4, 0, None, st), "sep")

            w0, m0 = w.clone(), mo.clone()
            fused()
            wa, ma, gxa = w.clone(), mo.clone(), gx.clone()
            w.copy_(w0), mo.copy_(m0)
            gxb = ops.linear_dgrad_raw(g, w)
            _lib.check(L.icl_sgd_step_factored(w.data_ptr(), mo.data_ptr(), g.data_ptr(), x.data_ptr(), m, N, N, 1e-3, 0.9, 1e-4, 0, None, st), "sep")
            e = (err(gxa, gxb), err(wa, w), err(ma, mo))
            tf, ts = t(fused), t(separate)
            gb = 16.0 * N * N / 1e9
            print(f"N={N:6d} M={m:3d}  one pass {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s of 16 B/weight)   dgrad + update {ts:7.1f} us   err gx {e[0]:.1e} w {e[1]:.1e} m {e[2]:.1e}",
                  flush=True)
