"""Which co-running kernel makes the PACKED-add LayerNorm backward (the library before the round-5 fix) lose a row's term?  Run with
ICL_HIP_LIB=<a library built from this tree with commit 41b7d15's icl_amd/csrc/kernels/token.h: hipcc --offload-arch=gfx950 -O3 -std=c++17
-shared -fPIC -I <csrc copy> -o libicl_hip_packed.so <csrc copy>/icl_hip.hip> and again with the shipped library.
The LayerNorm backward (1,728 x 128, the uscl level-1 shape) runs 80 times on one stream while a candidate runs on a second one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rows, c = 1728, 128
x = torch.randn(rows, c, device=dev)
w = torch.randn(c, device=dev).requires_grad_()
b = torch.randn(c, device=dev).requires_grad_()
gy = torch.randn(rows, c, device=dev) * 1e-3
vol16 = torch.randn(2, 16, 96, 96, 96, device=dev)
w16 = (torch.randn(16, 16, 3, 3, 3, device=dev) * 0.05).requires_grad_()
g16 = torch.randn(2, 16, 96, 96, 96, device=dev)
vol32 = torch.randn(2, 32, 48, 48, 48, device=dev)
w32 = (torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05).requires_grad_()
g32 = torch.randn(2, 32, 48, 48, 48, device=dev)
vol128 = torch.randn(2, 128, 12, 12, 12, device=dev)
w128 = (torch.randn(128, 128, 3, 3, 3, device=dev) * 0.05).requires_grad_()
g128 = torch.randn(2, 128, 12, 12, 12, device=dev)
big = torch.randn(64, 13824, device=dev)
wbig = torch.randn(13824, 13824, device=dev)
tok = torch.randn(1728, 128, device=dev)
wl = (torch.randn(512, 128, device=dev) * 0.05).requires_grad_()
x0 = torch.randn(13824, 64, device=dev)
w0 = torch.randn(64, device=dev).requires_grad_()
b0 = torch.randn(64, device=dev).requires_grad_()
g0 = torch.randn(13824, 64, device=dev)

def ln_once():
    xx = x.clone().requires_grad_()
    y = ops.layer_norm(xx, w, b)
    w.grad = b.grad = None
    y.backward(gy)
    return b.grad.clone(), w.grad.clone(), xx.grad.clone()

def wgrad(v, wt, g):
    y = ops.conv3d(v, wt, None)
    torch.autograd.grad(y, wt, g)

def co_conv_fwd():
    with torch.no_grad():
        ops.conv3d(vol16, w16, None)

def co_ln0():
    xx = x0.clone().requires_grad_()
    y = ops.layer_norm(xx, w0, b0)
    torch.autograd.grad(y, (xx, w0, b0), g0)

def co_linear():
    y = ops.linear(tok, wl, None)
    torch.autograd.grad(y, wl, torch.ones_like(y))

CANDS = {
    "alone": None,
    "conv forward 16->16 @96^3": co_conv_fwd,
    "conv forward + weight gradient 16->16 @96^3": lambda: wgrad(vol16, w16, g16),
    "conv forward + weight gradient 32->32 @48^3": lambda: wgrad(vol32, w32, g32),
    "conv forward + weight gradient 128->128 @12^3": lambda: wgrad(vol128, w128, g128),
    "weight stream 64 x 13824^2": lambda: ops.linear(big, wbig, None),
    "LayerNorm fwd+bwd 13824 x 64": co_ln0,
    "linear fwd + weight gradient 1728 x 128 -> 512": co_linear,
    "elementwise torch add 2 x 16 x 96^3": lambda: vol16.add(1.0),
}
ref = ln_once()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print("library:", os.environ.get("ICL_HIP_LIB", "shipped"))
for name, co in CANDS.items():
    bad = [0, 0, 0]
    outs = []
    cols = set()
    for it in range(80):
        if co is not None:
            with torch.cuda.stream(sb):
                co()
        with torch.cuda.stream(sa):
            outs.append(ln_once())
    torch.cuda.synchronize()
    for o in outs:
        for j in range(3):
            bad[j] += int(not torch.equal(o[j], ref[j]))
        cols.update((o[0] != ref[0]).nonzero().flatten().tolist())
    print(f"TRIG {name}: dbeta differs in {bad[0]} of 80 runs (columns {min(cols) if cols else '-'}..{max(cols) if cols else '-'}), dgamma {bad[1]}, dx {bad[2]}")
