#!/usr/bin/env python3
"""Time the fused window-attention kernels on the SwinUNETR stage shapes (batch 2, 96^3 input):
python tools/winattn_probe.py [stage]   stage 1: 686 windows x 343 tokens x 3 heads, 2: 128 x 343 x 6, 3: 16 x 343 x 12, 4: 2 x 216 x 24."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops  # noqa: E402
from icl_amd.networks import swinunetr as SW  # noqa: E402

dev = torch.device("cuda", 0)
STAGES = {1: ((48, 48, 48), 3), 2: ((24, 24, 24), 6), 3: ((12, 12, 12), 12), 4: ((6, 6, 6), 24)}
for st in ([int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]):
    dims, heads = STAGES[st]
    c = 16 * heads
    ws, ss = SW.get_window_size(dims, (7, 7, 7), (3, 3, 3))
    padded = [-(-d // w) * w for d, w in zip(dims, ws)]
    n = ws[0] * ws[1] * ws[2]
    nw = (padded[0] // ws[0]) * (padded[1] // ws[1]) * (padded[2] // ws[2])
    regions = SW.window_regions(padded, ws, ss, dev) if any(s > 0 for s in ss) else None
    attn = SW.WindowAttention(c, heads, (7, 7, 7), device=dev)
    qkv = torch.randn(2 * nw, n, 3 * c, device=dev, requires_grad=True)
    index = attn.relative_position_index[:n, :n].contiguous()
    table = attn.relative_position_bias_table

    def fwd():
        return ops.window_attention(qkv, table, index, regions, heads, 16 ** -0.5)

    y = fwd()
    g = torch.randn_like(y)
    y.backward(g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(5):
        y = fwd()
    e[1].record()
    for _ in range(5):
        y = fwd()
        y.backward(g)
    e[2].record()
    torch.cuda.synchronize()
    tf = e[0].elapsed_time(e[1]) / 5 * 1e3
    tb = e[1].elapsed_time(e[2]) / 5 * 1e3 - tf
    fl = 4.0 * n * n * 16 * 2 * nw * heads
    print(f"stage {st}: {2 * nw} windows x {n} tokens x {heads} heads: fwd {tf:7.1f} us ({fl / tf / 1e6:5.1f} TFLOP/s)  bwd {tb:7.1f} us "
          f"({3.5 * fl / tb / 1e6:5.1f} TFLOP/s)", flush=True)
