#!/usr/bin/env python3
"""A/B timing of convolution kernel variants inside ONE process (interleaved rounds, median and min of per-round HIP-event times;
cdna_hip_programming.md §5.4 rule 24).  Variants are sets of environment switches the launchers read per call.

    python tools/conv_ab.py [--rounds 7] [--reps 10] [--shapes "16,16,96,fwd;48,16,96,fwd;..."] --var "ICL_CONV_SPLIT_V=0" --var "ICL_CONV_SPLIT_V=1"
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

DEFAULT_SHAPES = ("16,16,96,fwd;48,16,96,fwd;16,48,96,fwd;32,32,48,fwd;96,32,48,fwd;16,32,48,fwd;64,64,24,fwd;192,64,24,fwd;"
                  "16,16,96,wgrad;48,16,96,wgrad;32,32,48,wgrad;96,32,48,wgrad;16,32,48,wgrad;64,64,24,wgrad;192,64,24,wgrad")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--shapes", default=DEFAULT_SHAPES)
    ap.add_argument("--var", action="append", default=[], help="comma-separated KEY=VALUE switches of one variant")
    a = ap.parse_args()
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in (a.var or [""])]
    keys = sorted({k for v in variants for k in v})
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    nb = a.batch
    for spec in a.shapes.split(";"):
        cin, cout, s, what = spec.split(",")
        cin, cout, s = int(cin), int(cout), int(s)
        S = s ** 3
        x = torch.randn(nb, cin, s, s, s, device=dev)
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        gy = torch.randn(nb, cout, s, s, s, device=dev)
        y, gw = torch.empty_like(gy), torch.empty_like(w)
        wp = ops.pack_weights(w, 0)
        ws = torch.empty(max(L.icl_conv3d_wgrad_ws_bytes(nb, cin, cout, 3), 4) // 4, device=dev)

        def run():
            if what == "fwd":
                ops.conv3d_forward_raw(x, wp, None, nb, cin, cout, s, s, s, 3, cin * S, y, cout * S)
            else:
                _lib.check(L.icl_conv3d_wgrad(ops._ptr(x), ops._ptr(gy), ops._ptr(gw), None, ops._ptr(ws), nb, cin, cout, s, s, s, 3,
                                              cin * S, cout * S, ops._stream(x)))
        times = [[] for _ in variants]
        names = [None] * len(variants)
        outs = [None] * len(variants)
        for rnd in range(a.rounds + 1):
            for vi, v in enumerate(variants):
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(v)
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if rnd:      # round 0 = warm-up
                    times[vi].append(e0.elapsed_time(e1) * 1e3 / a.reps)
                names[vi] = L.icl_last_kernel_name().decode()
                outs[vi] = (y if what == "fwd" else gw).clone()
        fl = 2.0 * 27 * cin * cout * S * nb
        for vi, v in enumerate(variants):
            med, mn = statistics.median(times[vi]), min(times[vi])
            d = float((outs[vi] - outs[0]).abs().max() / outs[0].abs().max())
            print(f"{cin:>3}->{cout:<3} @{s}^3 n={nb} {what:5s} {str(v):40s} median {med:8.1f} us  min {mn:8.1f} us  {fl / med / 1e6:6.1f} TF  "
                  f"maxdiff_vs_first {d:.1e}  {names[vi]}", flush=True)
        for k in keys:
            os.environ.pop(k, None)


if __name__ == "__main__":
    main()
