#!/usr/bin/env python3
"""Per-layer timing of the 3D U-Net convolutions (SURVEY.md Appendix B shapes) on the GPU:
forward, input-gradient and weight-gradient launches, HIP events, TFLOP/s against the fp32 MFMA peak."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

LAYERS = [(1, 16, 96), (16, 16, 96), (16, 32, 48), (32, 32, 48), (32, 64, 24), (64, 64, 24), (64, 128, 12), (128, 128, 12),
          (128, 256, 6), (256, 256, 6), (384, 128, 12), (192, 64, 24), (96, 32, 48), (48, 16, 96)]


if os.environ.get("ICL_CONV_BENCH_LAYERS"):      # "cin,cout,side;cin,cout,side;..."
    LAYERS = [tuple(int(v) for v in item.split(",")) for item in os.environ["ICL_CONV_BENCH_LAYERS"].split(";")]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # batch (bench.py runs 2 volumes per step)
    rows = []
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for cin, cout, s in LAYERS:
        x = torch.randn(nb, cin, s, s, s, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        gy = torch.randn(nb, cout, s, s, s, device=dev)
        y = torch.empty_like(gy)
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        gb = torch.empty_like(b)
        wp, wpt = ops.pack_weights(w, 0), ops.pack_weights(w, 1)
        S = s ** 3
        ws = torch.empty(L.icl_conv3d_wgrad_ws_bytes(nb, cin, cout, 3) // 4, device=dev)
        fl = 2.0 * 27 * cin * cout * S * nb
        t_f = timeit(lambda: ops.conv3d_forward_raw(x, wp, b, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S))
        t_d = timeit(lambda: ops.conv3d_forward_raw(gy, wpt, None, nb, cout, cin, s, s, s, 3, cout * S, gx, cin * S)) if cin > 1 else 0.0
        t_w = timeit(lambda: _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), ops._ptr(gb), ops._ptr(ws), nb, cin, cout,
                                                           s, s, s, 3, cin * S, cout * S, ops._stream(x))))
        mult = 2 if (cin, cout, s) in [(16, 16, 96), (32, 32, 48), (64, 64, 24), (128, 128, 12)] else 1
        for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[k] += t * mult
        rows.append(dict(cin=cin, cout=cout, side=s, gflop=round(fl / 1e9, 3), fwd_ms=round(t_f, 4), dgrad_ms=round(t_d, 4), wgrad_ms=round(t_w, 4),
                         fwd_tf=round(fl / t_f / 1e9, 2), dgrad_tf=round(fl / t_d / 1e9, 2) if t_d else None, wgrad_tf=round(fl / t_w / 1e9, 2)))
        print(rows[-1], flush=True)
    print("per-volume totals (ms, x2 for the repeated shapes):", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
    json.dump(rows, open(os.path.join("gpurun_out", os.environ.get("ICL_CONV_BENCH_OUT", "conv_bench.json")), "w"), indent=1)


if __name__ == "__main__":
    main()
