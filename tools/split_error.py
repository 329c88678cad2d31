#!/usr/bin/env python3
"""Per-product error of the six-term bf16x3 product (csrc/kernels/conv_bf16x3.h) against the exact product, for round-to-nearest
and truncation splits, in units of 2^-24 |x w| (numpy emulation; fp64 holds every bf16 x bf16 product and their sum exactly enough)."""
import numpy as np


def bf16_rn(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    return r.view(np.float32)


def bf16_trunc(v):
    return (v.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split(v, f):
    s1 = f(v)
    r = (v - s1).astype(np.float32)
    s2 = f(r)
    s3 = f((r - s2).astype(np.float32))
    assert np.array_equal((s1.astype(np.float64) + s2 + s3).astype(np.float32), v), "split is not exact"
    return [s.astype(np.float64) for s in (s1, s2, s3)]


def six_terms(x, w, f):
    a, b = split(x, f), split(w, f)
    return a[0] * b[0] + (a[0] * b[1] + a[1] * b[0]) + (a[0] * b[2] + a[1] * b[1] + a[2] * b[0])


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n = 1 << 22
    for name, x, w in (("N(0,1) x N(0,1)", rng.standard_normal(n), rng.standard_normal(n)),
                       ("|N| x |N| (post-ReLU x positive weights)", np.abs(rng.standard_normal(n)), np.abs(rng.standard_normal(n))),
                       ("log-uniform 1e-15 .. 1e15", 10.0 ** rng.uniform(-15, 15, n), 10.0 ** rng.uniform(-15, 15, n) * rng.choice([-1, 1], n))):
        x, w = x.astype(np.float32), w.astype(np.float32)
        exact = x.astype(np.float64) * w.astype(np.float64)
        ulp = np.abs(exact) * 2.0 ** -24
        print(name)
        for label, f in (("round-to-nearest", bf16_rn), ("truncation", bf16_trunc)):
            e = (six_terms(x, w, f) - exact) / ulp
            print(f"  {label:17s} mean |e| {np.abs(e).mean():.3f}  max |e| {np.abs(e).max():.2f}  mean signed e*sign(xw) {np.mean(e * np.sign(exact)):+.4f}")
        e = ((x * w).astype(np.float64) - exact) / ulp
        print(f"  {'fp32 multiply':17s} mean |e| {np.abs(e).mean():.3f}  max |e| {np.abs(e).max():.2f}  mean signed {np.mean(e * np.sign(exact)):+.4f}")
