"""Is the LayerNorm backward (1,728 x 128, the uscl level-1 shape) reproducible while other kernels run on other streams?"""
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
vol = torch.randn(2, 16, 96, 96, 96, device=dev)
wt = torch.randn(16, 16, 3, 3, 3, device=dev) * 0.05
big = torch.randn(64, 13824, device=dev)
wbig = torch.randn(13824, 13824, device=dev)

def ln_once():
    xx = x.clone().requires_grad_()
    y = ops.layer_norm(xx, w, b)
    w.grad = b.grad = None
    y.backward(gy)
    return b.grad.clone(), w.grad.clone(), xx.grad.clone()

ref = ln_once()
torch.cuda.synchronize()
sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("alone", "beside convolutions", "beside weight streams", "beside both"):
    bad = [0, 0, 0]
    outs = []
    for it in range(60):
        if mode in ("beside convolutions", "beside both"):
            with torch.cuda.stream(sb), torch.no_grad():
                ops.conv3d(vol, wt, None)
        if mode in ("beside weight streams", "beside both"):
            with torch.cuda.stream(sc), torch.no_grad():
                ops.linear(big, wbig, None)
        with torch.cuda.stream(sa):
            outs.append(ln_once())
    torch.cuda.synchronize()
    for o in outs:
        for j in range(3):
            bad[j] += int(not torch.equal(o[j], ref[j]))
    print(f"{mode}: dbeta differs in {bad[0]} of 60 runs, dgamma {bad[1]}, dx {bad[2]}")
