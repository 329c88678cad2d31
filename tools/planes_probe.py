#!/usr/bin/env python3
"""Decides between the two readings of the load term: the split-product forward kernel fed with operand planes
that a producer has written (16-byte loads, no split VALU) against the shipped kernel that splits fp32 while staging.
Needs a probe build:  hipcc ... -DBF3_PLANES_PROBE -o gpurun_in/libicl_planes.so   (the kernel of that build ONLY takes planes)
    python3 tools/planes_probe.py 16 16 96"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nb, S = 2, int(sys.argv[3]) ** 3
dev = torch.device("cuda", 0)
x = torch.randn(nb, cin, s, s, s, device=dev)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05


def timed(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def run(libpath, planes):
    from icl_amd import _lib, ops
    _lib._use_library_for_tests(libpath, host_pointers=False)      # (a probe tool: two builds of the ABI in one process)
    L = _lib.lib()
    wp = ops.pack_weights(w, 0)
    y = torch.empty(nb, cout, s, s, s, device=dev)
    src = x
    if planes:
        raw = ctypes.CDLL(libpath)
        pl = torch.empty(3 * nb * (cin // 8) * S * 4, dtype=torch.int32, device=dev)
        t_split = timed(lambda: raw.icl_debug_split_planes(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(pl.data_ptr()), nb, cin,
                                                           ctypes.c_int64(S), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        src = pl.view(torch.float32)
    else:
        t_split = 0.0
    t = timed(lambda: ops.conv3d_forward_raw(src, wp, None, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S))
    return t, t_split, y.clone(), L.icl_last_kernel_name().decode()


base = os.path.join(ROOT, "icl_amd", "libicl_hip.so")
probe = os.path.join(ROOT, "gpurun_in", "libicl_planes.so")
t0, _, y0, k0 = run(base, False)
t1, ts, y1, k1 = run(probe, True)
err = float((y0 - y1).abs().max() / y0.abs().max())
print(f"{cin}->{cout} @{s}^3 batch 2: fp32 input, split while staging {t0:7.1f} us ({k0});  planes input {t1:7.1f} us ({k1}; the standalone split "
      f"kernel that wrote them: {ts:6.1f} us);  rel diff of the outputs {err:.1e}")
