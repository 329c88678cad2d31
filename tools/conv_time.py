#!/usr/bin/env python3
"""Per-launch time of the convolution kernels on a list of layer shapes, for A/B runs of two BUILDS of the library (the library path is
fixed per process: run this once per build, alternating, e.g. ICL_HIP_LIB=icl_amd/libicl_hip_noslp.so python tools/conv_time.py).

    python tools/conv_time.py [--rounds 7] [--reps 10] [--batch 2] [--shapes "16,16,96,fwd;..."]
Prints one line per shape: median and min of the per-round HIP-event times (us per launch) and the kernel name."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

DEFAULT = ("16,16,96,fwd;48,16,96,fwd;32,32,48,fwd;96,32,48,fwd;64,64,24,fwd;192,64,24,fwd;48,48,96,fwd;"
           "16,16,96,wgrad;48,16,96,wgrad;32,32,48,wgrad;96,32,48,wgrad;64,64,24,wgrad;48,48,96,wgrad")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--shapes", default=DEFAULT)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    nb = a.batch
    print("# library", _lib.lib_path())
    for spec in a.shapes.split(";"):
        cin, cout, s, what = spec.split(",")
        cin, cout, s = int(cin), int(cout), int(s)
        S = s ** 3
        x = torch.randn(nb, cin, s, s, s, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        gy = torch.randn(nb, cout, s, s, s, device=dev)
        y, gw = torch.empty_like(gy), torch.empty_like(w)
        wp = ops.pack_weights(w, 0)
        wsplit = None
        if cin % 16 == 0 and L.icl_conv3d_split_ws_bytes(cin, cout):      # the step's path: bf16 planes split once, outside the timed calls
            import ctypes
            wsplit = torch.empty(L.icl_conv3d_split_ws_bytes(cin, cout) // 4, dtype=torch.float32, device=dev)
            arr, iarr = ctypes.c_void_p * 1, ctypes.c_int32 * 1
            _lib.check(L.icl_conv3d_split_weights_multi(arr(wp.data_ptr()), arr(wsplit.data_ptr()), iarr(cin), iarr(cout), 1, ops._stream(x)))
        ws = torch.empty(max(L.icl_conv3d_wgrad_ws_bytes(nb, cin, cout, 3) // 4, 1), device=dev)

        def run():
            if what == "fwd":
                ops.conv3d_forward_raw(x, wp, None, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S, wsplit=wsplit)
            else:
                _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), nb, cin, cout, s, s, s, 3,
                                              cin * S, cout * S, ops._stream(x)))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        fl = 2.0 * 27 * cin * cout * S * nb
        med = statistics.median(ts)
        print(f"{cin:4d}->{cout:<4d}@{s:<3d} {what:5s} median {med:8.1f} us  min {min(ts):8.1f} us  {fl / med / 1e6:6.1f} TFLOP/s  {L.icl_last_kernel_name().decode()[:60]}")


if __name__ == "__main__":
    main()
