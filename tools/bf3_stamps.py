#!/usr/bin/env python3
"""Phase timeline of the split-product forward kernel from in-kernel stamps (a build with -DBF3_DEBUG=16, loaded through ICL_HIP_LIB):
    ICL_HIP_LIB=$PWD/gpurun_in/libicl_dbg16.so ICL_CONV_SPLIT_V=8 python3 tools/bf3_stamps.py 16 16 96
Prints, for waves 0 and 4 of workgroup 0 (one SIMD) and work items 2..4, the cycles between consecutive stamps:
0 item start, 1 after barrier A, 2 after split + LDS stores, 3 after barrier B, 4 after the next item's loads were issued,
5..18 after tap pair 0..13, 20 before the epilogue, 21 item end."""
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
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
y = torch.empty(nb, cout, s, s, s, device=dev)
wp = ops.pack_weights(w, 0)
S = s ** 3
for _ in range(5):
    ops.conv3d_forward_raw(x, wp, None, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (2 * 3 * 32))()
raw = ctypes.CDLL(_lib.lib_path())
rc = raw.icl_debug_bf3_stamps(buf)
assert rc == 0, rc
print("kernel:", L.icl_last_kernel_name().decode())
names = {0: "start", 1: "barrier A", 2: "split+store", 3: "barrier B", 4: "loads issued", 20: "pairs done", 21: "epilogue"}
for wv in range(2):
    for it in range(3):
        st = [buf[(wv * 3 + it) * 32 + k] for k in range(32)]
        t0 = st[0]
        seq = [k for k in list(range(0, 5)) + list(range(5, 19)) + [20, 21] if st[k]]
        line, prev = [], t0
        for k in seq[1:]:
            line.append(f"{names.get(k, 'p' + str(k - 5))}:{st[k] - prev}")
            prev = st[k]
        print(f"wave {wv * 4} item {it + 2}: total {st[21] - t0}  " + " ".join(line))
