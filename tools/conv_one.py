#!/usr/bin/env python3
"""Run one conv layer shape repeatedly (for rocprofv3 --pmc): python tools/conv_one.py cin cout side [fwd|dgrad|wgrad] [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
what = sys.argv[4] if len(sys.argv) > 4 else "fwd"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
nb = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda", 0)
L = _lib.lib()
x = torch.randn(nb, cin, s, s, s, device=dev)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
gy = torch.randn(nb, cout, s, s, s, device=dev)
y, gx, gw, gb = torch.empty_like(gy), torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
wp, wpt = ops.pack_weights(w, 0), ops.pack_weights(w, 1)
S = s ** 3
ws = torch.empty(L.icl_conv3d_wgrad_ws_bytes(nb, cin, cout, 3) // 4, device=dev)
for _ in range(iters):
    if what == "fwd":
        ops.conv3d_forward_raw(x, wp, b, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S)
    elif what == "dgrad":
        ops.conv3d_forward_raw(gy, wpt, None, nb, cout, cin, s, s, s, 3, cout * S, gx, cin * S)
    else:
        _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), nb, cin, cout, s, s, s, 3,
                                      cin * S, cout * S, ops._stream(x)))
torch.cuda.synchronize()

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    if what == "fwd":
        ops.conv3d_forward_raw(x, wp, b, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S)
    elif what == "dgrad":
        ops.conv3d_forward_raw(gy, wpt, None, nb, cout, cin, s, s, s, 3, cout * S, gx, cin * S)
    else:
        _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), nb, cin, cout, s, s, s, 3,
                                      cin * S, cout * S, ops._stream(x)))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"{cin}->{cout} @{s}^3 n={nb} {what}: {us:.1f} us  {2*27*cin*cout*S*nb/us/1e6:.1f} TFLOP/s  ksplit={os.environ.get('ICL_CONV_KSPLIT','auto')}  kernel={L.icl_last_kernel_name().decode()}")
