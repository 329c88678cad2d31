#!/usr/bin/env python3
"""Phase timeline of the transposing-read weight-gradient kernel from in-kernel stamps (a build with -DWGTR_DEBUG loaded through
ICL_HIP_LIB):   ICL_HIP_LIB=$PWD/gpurun_in/libicl_wgtr.so python3 tools/wgtr_stamps.py 16 16 96
Waves 0 (S -> M: split + store the next tile, issue loads, multiply) and 4 (M -> S) of workgroup 0 share a SIMD; cycles between the
stamps of phases 4..7."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nb = 2
dev = torch.device("cuda", 0)
L = _lib.lib()
x = torch.randn(nb, cin, s, s, s, device=dev)
gy = torch.randn(nb, cout, s, s, s, device=dev)
gw = torch.empty(cout, cin, 3, 3, 3, device=dev)
ws = torch.empty(max(L.icl_conv3d_wgrad_ws_bytes(nb, cin, cout, 3), 4) // 4, device=dev)
S = s ** 3
for _ in range(5):
    _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), nb, cin, cout, s, s, s, 3, cin * S, cout * S,
                                  ops._stream(x)))
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * 4 * 8))()
raw = ctypes.CDLL(_lib.lib_path())
assert raw.icl_debug_wgtr_stamps(buf) == 0
names = [["start", "split+store", "loads issued", "multiply", "barrier"], ["start", "multiply", "split+store", "loads issued", "barrier"]]
for wv in range(2):
    for ph in range(4):
        st = [buf[(wv * 4 + ph) * 8 + k] for k in range(5)]
        print(f"wave {wv * 4} phase {ph + 4}: total {st[4] - st[0]}  " + " ".join(f"{names[wv][k]}:{st[k] - st[k - 1]}" for k in range(1, 5)))
