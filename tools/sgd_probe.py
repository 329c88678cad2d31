"""Time the factored SGD kernel (13824^2 weight) for several factor row counts M (M = rows per rank x ranks after the gather)."""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib  # noqa: E402

dev = torch.device("cuda")
L = _lib.lib()
N = K = 13824
p = torch.randn(N, K, device=dev)
m = torch.zeros_like(p)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for M in [int(a) for a in sys.argv[1:]] or (8, 16, 24, 64, 128, 192, 512, 1024):
    g = torch.randn(M, N, device=dev) * 1e-3
    x = torch.randn(M, K, device=dev)
    ws = torch.empty(max(1, L.icl_sgd_factored_split_ws_bytes(M, N, K) // 4), device=dev)

    def run():
        _lib.check(L.icl_sgd_step_factored(p.data_ptr(), m.data_ptr(), g.data_ptr(), x.data_ptr(), M, N, K, 0.01, 0.9, 1e-4, 0, None, st), "sgd")

    def run_split():     # factor split + update on split products (bf16 matrix pipe, fp32 accuracy)
        _lib.check(L.icl_sgd_step_factored_split(p.data_ptr(), m.data_ptr(), g.data_ptr(), x.data_ptr(), ws.data_ptr(), M, N, K, 0.01, 0.9, 1e-4,
                                                 0, None, st), "sgd split")

    def t(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5 * 1e3
    # agreement of the two forms on one step from the same state
    p0, m0 = p.clone(), m.clone()
    run()
    pa, ma = p.clone(), m.clone()
    p.copy_(p0), m.copy_(m0)
    run_split()
    err = float((m - ma).abs().max() / ma.abs().max())
    p.copy_(p0), m.copy_(m0)
    del p0, m0, pa, ma
    us, us2 = t(run), t(run_split)
    print(f"M={M:5d}: fp32 MFMA {us:8.1f} us ({16 * N * K / us / 1e6:5.2f} TB/s, {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s)   split products "
          f"{us2:8.1f} us ({16 * N * K / us2 / 1e6:5.2f} TB/s)   momentum rel diff {err:.1e}", flush=True)
