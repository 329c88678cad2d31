"""Counter-based, platform-independent tensor filler.

Every element is a pure function of (seed, flat index), computed with 32-bit
integer arithmetic only, so numpy on any host and torch on any device produce
bit-identical fp32 values.  It is what lets the parity tests compare a 785 M
parameter model against golden vectors captured from the reference without
storing a single weight (SURVEY.md §8c "golden-vector plan").

    u   = mix32(index * 0x9E3779B1 + seed * 0x85EBCA77 + 0x6A09E667)
    val = ((u >> 8) * 2^-24 - 0.5) * 2 * bound          # uniform in [-bound, bound)

`fill_like_reference_init` maps a parameter name/shape to `bound` so that the
synthetic weights have the variance of the reference's initialisers
(kaiming fan-in for >=2-D weights, `networks_other.py:64-75`).
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

_M32 = 0xFFFFFFFF


def _mix32_np(x: np.ndarray) -> np.ndarray:
    # lowbias32-style finaliser on uint32 lanes (wrap-around arithmetic)
    x = x.astype(np.uint32, copy=False)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def hash_uniform_np(n: int, seed: int, bound: float = 1.0, offset: int = 0) -> np.ndarray:
    """n fp32 values in [-bound, bound), element i uses counter offset+i."""
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 24
    with np.errstate(over="ignore"):
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            idx = np.arange(offset + s, offset + e, dtype=np.uint64).astype(np.uint32)
            x = idx * np.uint32(0x9E3779B1) + np.uint32((seed * 0x85EBCA77 + 0x6A09E667) & _M32)
            u = _mix32_np(x)
            f = (u >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) - np.float32(0.5)
            out[s:e] = f * np.float32(2.0 * bound)
    return out


def hash_uniform_torch(n: int, seed: int, bound: float = 1.0, device="cpu", offset: int = 0) -> torch.Tensor:
    """Same values as `hash_uniform_np`, produced with torch int64 ops on `device`."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = 1 << 24
    c0 = (seed * 0x85EBCA77 + 0x6A09E667) & _M32
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.arange(offset + s, offset + e, dtype=torch.int64, device=device) & _M32
        x = (x * 0x9E3779B1 + c0) & _M32
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & _M32
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & _M32
        x = x ^ (x >> 16)
        f = (x >> 8).to(torch.float32) * (2.0 ** -24) - 0.5
        out[s:e] = f * (2.0 * bound)
    return out


def name_seed(name: str, base_seed: int = 1337) -> int:
    return (zlib.crc32(name.encode()) ^ (base_seed * 0x9E3779B1)) & _M32


def reference_init_bound(name: str, shape) -> tuple[float, float]:
    """(centre, bound) of the synthetic value range for one parameter tensor."""
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return 0.0, math.sqrt(6.0 / max(fan_in, 1))
    if name.endswith("weight"):  # norm scales
        return 1.0, 0.1
    return 0.0, 0.1


@torch.no_grad()
def fill_like_reference_init(named_tensors, base_seed: int = 1337) -> None:
    """Overwrite every (name, tensor) in place with hash-filled values."""
    for name, t in named_tensors:
        if not t.is_floating_point():
            continue
        centre, bound = reference_init_bound(name, t.shape)
        n = t.numel()
        if t.device.type == "cpu":
            v = torch.from_numpy(hash_uniform_np(n, name_seed(name, base_seed), bound))
        else:
            v = hash_uniform_torch(n, name_seed(name, base_seed), bound, device=t.device)
        if centre:
            v = v + centre
        t.copy_(v.view(t.shape))


def synthetic_volume(shape, seed: int, device="cpu") -> torch.Tensor:
    """Roughly N(0,1)-scaled synthetic image: sum of 3 uniforms (variance 1)."""
    n = 1
    for s in shape:
        n *= s
    gen = hash_uniform_torch if str(device) != "cpu" else None
    acc = None
    for k in range(3):
        if gen is None:
            v = torch.from_numpy(hash_uniform_np(n, seed * 3 + k, 1.0))
        else:
            v = gen(n, seed * 3 + k, 1.0, device=device)
        acc = v if acc is None else acc + v
    return acc.view(*shape)


def synthetic_labels(shape, seed: int, num_classes: int, device="cpu") -> torch.Tensor:
    """int64 labels; nc=2 -> Bernoulli(0.3) foreground, else uniform."""
    n = 1
    for s in shape:
        n *= s
    if str(device) == "cpu":
        u = torch.from_numpy(hash_uniform_np(n, seed, 0.5)) + 0.5
    else:
        u = hash_uniform_torch(n, seed, 0.5, device=device) + 0.5
    if num_classes == 2:
        lab = (u < 0.3).to(torch.int64)
    else:
        lab = torch.clamp((u * num_classes).to(torch.int64), 0, num_classes - 1)
    return lab.view(*shape)
