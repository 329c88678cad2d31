"""Data-parallel gradient reduction for the ICL step: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference is single-GPU; this is a capability the build adds.  Semantics
(SURVEY.md §8e): W ranks == W independent reference steps whose gradients are averaged.
 - volumes shard batch-wise: each rank gets its own labeled + unlabeled volumes; BatchNorm statistics,
   ``updated_Qs`` batch means and batch-wide Dice sums stay per rank by design;
 - parameters whose ``.grad`` is None after backward (33 tensors in an ICL step) are skipped,
   never zero-filled — torch SGD skips them too, so weight decay must not touch them;
 - gradients travel as fp32.  The payload is 3.14 GB and 99 % of it is twelve token-axis ``mlp2`` matrices
   (four of them 764 MB each).  Those are produced FIRST in backward (the aligners are the last thing in forward),
   so every large gradient is handed to RCCL the moment autograd has finished accumulating it
   (``register_post_accumulate_grad_hook``) and its all-reduce runs on RCCL's stream underneath the whole
   backbone backward; only the small tensors (~6 M elements, one flat bucket) are reduced after backward.
   xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, very large messages keep all links busy.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, model: torch.nn.Module, world_size: int, bucket_bytes: int = 256 << 20, overlap_min_elems: int = 1 << 22):
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.world = world_size
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap_min = overlap_min_elems
        self._handles = []
        self._early = set()
        self._avg = dist.is_initialized() and dist.get_backend() == "nccl"   # RCCL averages in the collective
        if world_size > 1:
            for p in self.params:
                if p.numel() >= overlap_min_elems:
                    p.register_post_accumulate_grad_hook(self._on_grad_ready)

    def broadcast_parameters(self, src: int = 0):
        for p in self.params:
            dist.broadcast(p.data, src)

    # -- large tensors: start the all-reduce as soon as the gradient is complete, overlap with the rest of backward
    def _on_grad_ready(self, p: torch.nn.Parameter):
        if p.grad is None:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._handles.append((dist.all_reduce(p.grad, op=op, async_op=True), p))
        self._early.add(id(p))

    def _buckets(self):
        cur, n = [], 0
        for p in self.params:
            if p.grad is None or id(p) in self._early:
                continue
            if cur and n + p.grad.numel() > self.bucket_elems:
                yield cur
                cur, n = [], 0
            cur.append(p)
            n += p.grad.numel()
        if cur:
            yield cur

    def _gather_factored(self):
        """Parameters whose gradient is still factored (ops.FactoredGrads: dW = g^T x with a few dozen rows): the mean over
        ranks of g_r^T x_r is [g_1/W; ...; g_W/W]^T [x_1; ...; x_W], so the ranks all-gather the row blocks (~1 MB per
        13,824^2 matrix) instead of all-reducing 764 MB.  Every rank must hold the same number of rows."""
        inv = 1.0 / self.world
        for p in self.params:
            fac = getattr(p, "_icl_factors", None)
            if not fac:
                continue
            g = fac[0][0] if len(fac) == 1 else torch.cat([f[0] for f in fac], 0)
            x = fac[0][1] if len(fac) == 1 else torch.cat([f[1] for f in fac], 0)
            g = (g * inv).contiguous()
            x = x.contiguous()
            gs = [torch.empty_like(g) for _ in range(self.world)]
            xs = [torch.empty_like(x) for _ in range(self.world)]
            dist.all_gather(gs, g)
            dist.all_gather(xs, x)
            p._icl_factors = [(torch.cat(gs, 0), torch.cat(xs, 0))]

    def reduce_gradients(self):
        """Call after ``loss.backward()``: reduces the remaining (small) gradients and waits for the overlapped ones.
        Every rank must hold the same set of non-None grads (true for ICL: a property of the graph, not of the data)."""
        if self.world == 1:
            return
        self._gather_factored()
        inv = 1.0 / self.world
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        for bucket in self._buckets():
            if len(bucket) == 1:
                g = bucket[0].grad
                dist.all_reduce(g, op=op)
                if not self._avg:
                    g.mul_(inv)
                continue
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            dist.all_reduce(flat, op=op)
            if not self._avg:
                flat.mul_(inv)
            off = 0
            for p in bucket:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        for h, p in self._handles:
            h.wait()
            if not self._avg:
                p.grad.mul_(inv)
        self._handles.clear()
        self._early.clear()
