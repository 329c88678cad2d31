"""Data-parallel gradient reduction for the ICL step: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference is single-GPU; this is a capability the build adds.  Semantics
(SURVEY.md §8e): W ranks == W independent reference steps whose gradients are averaged.
 - volumes shard batch-wise: each rank gets its own labeled + unlabeled volumes; BatchNorm statistics,
   ``updated_Qs`` batch means and batch-wide Dice sums stay per rank by design;
 - parameters whose ``.grad`` is None after backward (33 tensors in an ICL step) are skipped,
   never zero-filled — torch SGD skips them too, so weight decay must not touch them;
 - gradients travel as fp32 in a few large flat buckets (the payload is 3.14 GB, 99 % of it the four
   13,824^2 ``mlp2`` matrices): large buckets keep every xGMI link busy and amortise launch latency.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, model: torch.nn.Module, world_size: int, bucket_bytes: int = 512 << 20):
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.world = world_size
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._flat = {}

    def broadcast_parameters(self, src: int = 0):
        for p in self.params:
            dist.broadcast(p.data, src)

    def _buckets(self):
        cur, n = [], 0
        for p in self.params:
            if p.grad is None:
                continue
            if cur and n + p.grad.numel() > self.bucket_elems:
                yield cur
                cur, n = [], 0
            cur.append(p)
            n += p.grad.numel()
        if cur:
            yield cur

    def reduce_gradients(self):
        """Average .grad over ranks in place.  Every rank must hold the same set of non-None grads
        (true for ICL: the set is a property of the graph, not of the data)."""
        if self.world == 1:
            return
        inv = 1.0 / self.world
        for bucket in self._buckets():
            if len(bucket) == 1:
                g = bucket[0].grad
                dist.all_reduce(g)
                g.mul_(inv)
                continue
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            dist.all_reduce(flat)
            flat.mul_(inv)
            off = 0
            for p in bucket:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
