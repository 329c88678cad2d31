"""Time the library GEMMs of the aligner's token-axis MLP (13824^2 weights) for several skinny row counts."""
import sys
import torch

dev = torch.device("cuda")
N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 13824
w = torch.randn(N, K, device=dev)
b = torch.randn(N, device=dev)


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


for m in (4, 8, 12, 16, 24, 32, 48, 64, 96):
    x = torch.randn(m, K, device=dev)
    g = torch.randn(m, N, device=dev)
    fwd = t(lambda: torch.nn.functional.linear(x, w, b))
    dx = t(lambda: g @ w)
    dw = t(lambda: g.t() @ x)
    gb = w.numel() * 4 / 1e9
    print(f"M={m:3d}  fwd {fwd:7.1f} us ({gb / fwd * 1e6 / 1e3:5.2f} TB/s)   dx {dx:7.1f} us ({gb / dx * 1e6 / 1e3:5.2f} TB/s)   dW {dw:7.1f} us ({gb / dw * 1e6 / 1e3:5.2f} TB/s)", flush=True)
