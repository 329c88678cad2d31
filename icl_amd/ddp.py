"""Data-parallel gradient reduction for the ICL step: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The reference is single-GPU; this is a capability the build adds.  Semantics
(SURVEY.md §8e): W ranks == W independent reference steps whose gradients are averaged.
 - volumes shard batch-wise: each rank gets its own labeled + unlabeled volumes; BatchNorm statistics,
   ``updated_Qs`` batch means and batch-wide Dice sums stay per rank by design;
 - parameters whose ``.grad`` is None after backward (33 tensors in an ICL step) are skipped,
   never zero-filled — torch SGD skips them too, so weight decay must not touch them;
 - gradients travel as fp32.  The payload is 3.14 GB and 99 % of it is twelve token-axis ``mlp2`` matrices
   (four of them 764 MB each).  With factored gradients (the trainer's default) they travel as ~6 MB of factor rows per rank;
   with dense gradients every large tensor is all-reduced on its own, in place (no bucket copy), the small tensors
   (~6 M elements) in flat buckets.  All collectives are issued from ``communicate()``, after backward: a gradient that
   autograd accumulates more than once per step (a weight used twice, micro-batch accumulation) is complete by then — an
   all-reduce started from an accumulation hook would reduce a partial sum and race the second accumulation.
   xGMI is point-to-point (7 links x ~153 GB/s per GPU): few, very large messages keep all links busy.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


# Factored SGD update of ONE 13,824^2 matrix on one MI355X as a function of the gathered factor rows (tools/sgd_probe.py, rounds 2-3:
# fp32 MFMA below 192 rows, split products from there on), ms.  Other matrix sizes scale with their element count (the update is a
# stream over p and m plus rows x n x k multiply-adds).
_UPDATE_ROWS_MS = ((16, 0.52), (32, 0.55), (64, 0.70), (128, 0.80), (256, 1.03), (512, 1.77), (1024, 3.2), (1536, 4.8))
_REF_ELEMS = 13824 * 13824
ASSUMED_ALLGATHER_GBPS = 300.0      # per rank, before the node has been measured (calibrate())


def update_ms(rows: float, elems: int = _REF_ELEMS) -> float:
    """Projected time of the factored update of an `elems`-element matrix from `rows` gathered factor rows."""
    t = _UPDATE_ROWS_MS
    if rows <= t[0][0]:
        ms = t[0][1]
    elif rows >= t[-1][0]:
        ms = t[-1][1] * rows / t[-1][0]
    else:
        for (r0, m0), (r1, m1) in zip(t, t[1:]):
            if r0 <= rows <= r1:
                ms = m0 + (m1 - m0) * (rows - r0) / (r1 - r0)
                break
    return ms * elems / _REF_ELEMS


def exchange_plan(rows_gathered: int, elems: int, world: int, allgather_gbps: float, allreduce_gbps: float = None, divisible: bool = True):
    """The three ways a rank can apply the averaged gradient of a token-axis matrix, priced with the node's measured rates:
    whole   every rank applies the rank-(rows W) update of the whole matrix from the gathered factors;
    shard   rank r updates rows [r N/W, (r+1) N/W) and the ranks all-gather the updated rows (4 B per weight over xGMI);
    dense   the layer forms its dense gradient, the ranks all-reduce it (8 (W-1)/W B per weight) and apply the 20 B/weight update.
    Returns (mode, {mode: projected ms})."""
    nbytes = 4.0 * elems
    cost = {"whole": update_ms(rows_gathered, elems)}
    if divisible and world > 1:
        cost["shard"] = update_ms(rows_gathered, elems) / world + nbytes / (allgather_gbps * 1e6)
    if allreduce_gbps and world > 1:
        local_rows = rows_gathered / world
        form = 2.0 * local_rows * elems / 100e12 * 1e3                       # dW = g^T x on the fp32 MFMA tile kernel (~100 TFLOP/s)
        cost["dense"] = form + 2.0 * (world - 1) / world * nbytes / (allreduce_gbps * 1e6) + 20.0 * elems / 5.5e12 * 1e3
    mode = min(cost, key=cost.get)
    return mode, {k: round(v, 3) for k, v in cost.items()}


def project_world(matrices, world: int, base_ms: float, allgather_gbps: float = ASSUMED_ALLGATHER_GBPS, allreduce_gbps: float = 250.0,
                  small_grad_bytes: float = 34e6, factor_bytes_per_rank: float = 0.0, shard_min_rows: int = None, _nested: bool = False):
    """Projected step time on `world` ranks from a ONE-rank measurement (`base_ms`: the data-parallel step on a one-rank group, whose
    token-axis updates are the whole factored updates at the one-rank row counts) — the arithmetic behind ``config.ddp_plan.projection``
    of bench.py, stated so that it can be checked once a multi-GPU node is available:
      * per token-axis matrix (``matrices``: dicts with ``rows`` = factor rows of ONE rank and ``elems``): the mode `exchange_plan`
        picks at rows x world gathered rows (or `shard_min_rows` if given) and what its update then costs against the one-rank update;
      * the all-gathers of row-sharded matrices (4 B per weight each) are started after those matrices' updates and run under the
        updates of the remaining matrices (ICLTrainer._step_body): only what exceeds them is exposed;
      * exposed after backward: the all-reduce of the small dense gradients (2 (W-1)/W x bytes) and the all-gather of the factor rows.
    Returns a dict with every term; all rates in GB/s per rank."""
    extra = ag_total = rest = 0.0
    rows_out = []
    for m in matrices:
        rows, elems = int(m["rows"]), int(m["elems"])
        gathered = rows * world
        divisible = m.get("out_rows", world) % world == 0
        mode, cost = exchange_plan(gathered, elems, world, allgather_gbps, allreduce_gbps, divisible)
        if shard_min_rows is not None:
            mode = "shard" if (shard_min_rows > 0 and gathered >= shard_min_rows and divisible and world > 1) else "whole"
        one = update_ms(rows, elems)
        if mode == "shard":
            upd = update_ms(gathered, elems) / world
            ag_total += 4.0 * elems / (allgather_gbps * 1e6)
        elif mode == "dense":
            upd = cost["dense"]
        else:
            upd = update_ms(gathered, elems)
            rest += upd
        extra += upd - one
        rows_out.append({"rows_gathered": gathered, "mode": mode, "update_ms": round(upd, 3), "one_rank_update_ms": round(one, 3)})
    exposed_ag = max(0.0, ag_total - rest)
    small = (2.0 * (world - 1) / world * small_grad_bytes / (allreduce_gbps * 1e6) if world > 1 else 0.0)
    factors = world * factor_bytes_per_rank / (allgather_gbps * 1e6) if world > 1 else 0.0
    total = base_ms + extra + exposed_ag + small + factors
    # the link rate at which this configuration still reaches the target of 6x on 8 ranks (efficiency 0.75): the projection re-run at scaled
    # rates (all-reduce kept in proportion) — how much margin the ASSUMED rates leave
    break_even = None
    if world > 1 and not _nested:
        lo, hi = 5.0, max(allgather_gbps, 5.0)
        eff = lambda r: project_world(matrices, world, base_ms, r, allreduce_gbps * r / allgather_gbps, small_grad_bytes, factor_bytes_per_rank,     # noqa: E731
                                      shard_min_rows, _nested=True)["projected_scaling_efficiency"]
        want = 6.0 / 8.0 if world == 8 else 0.75
        if eff(hi) >= want:
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                if eff(mid) >= want:
                    hi = mid
                else:
                    lo = mid
            break_even = {"target_efficiency": want, "allgather_gbps": round(hi, 1), "allreduce_gbps": round(allreduce_gbps * hi / allgather_gbps, 1),
                          "margin_vs_assumed": round(allgather_gbps / hi, 2)}
        else:
            break_even = {"target_efficiency": want, "allgather_gbps": None, "note": "the target is not reached at the assumed rates"}
    return {"world": world, "base_ms_one_rank_group": round(base_ms, 3), "update_extra_ms": round(extra, 3),
            **({"break_even_link_rate": break_even} if break_even else {}),
            "row_allgather_ms": round(ag_total, 3), "row_allgather_exposed_ms": round(exposed_ag, 3),
            "small_gradient_allreduce_ms": round(small, 3), "factor_row_allgather_ms": round(factors, 3),
            "projected_ms_per_step": round(total, 3), "projected_scaling_efficiency": round(base_ms / total, 3),
            "rates_gbps": {"allgather": allgather_gbps, "allreduce": allreduce_gbps}, "matrices": rows_out}


class GradientReducer:
    """``reduce_gradients()`` = ``pack()`` (device-side preparation) + ``communicate()`` (the collectives, nothing else) +
    ``rebind()`` (Python-side: ``p.grad`` / ``p._icl_factors`` now name the reduced buffers).  ICLTrainer.capture() records
    pack() at the end of the forward/backward hipGraph and the optimiser in a second graph, so that a data-parallel step is
    two graph replays with only the collectives issued eagerly in between."""

    def __init__(self, model: torch.nn.Module, world_size: int, bucket_bytes: int = 256 << 20, overlap_min_elems: int = 1 << 22,
                 force: bool = False, shard_min_rows: int = None):
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.world = world_size
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap_min = overlap_min_elems
        self.force = force       # run the collectives even with a single rank (exercises the captured path on one GPU)
        self.static = False      # set by ICLTrainer.capture(): the packed buffers persist (they are baked into the graphs)
        self._captured = False   # set once the graphs exist: pack() may not run eagerly any more
        self._early = set()      # ids of the large parameters of this step: all-reduced alone, in place
        self._flat = []          # [(flat buffer, [params])]
        self._fac = []           # [(param, G_all, X_all)]
        self._avg = dist.is_initialized() and dist.get_backend() == "nccl"   # RCCL averages in the collective
        self._send = self._recv = None   # all factor rows of a step, packed / gathered
        # Crossover of the factored exchange: every rank applies the rank-(rows x W) update of a matrix, which is MFMA-bound beyond a
        # few hundred gathered rows (13,824^2: 0.8 ms at 128, 2.1 at 512, 4.0 at 1024; tools/sgd_probe.py).  From `shard_min_rows`
        # gathered rows on, rank r updates only rows [r N/W, (r+1) N/W) of the matrix (1/W of the product and of the p / m stream)
        # and `post_update()` all-gathers the updated rows in place (764 MB per 13,824^2 matrix over xGMI).  Which side wins depends
        # on the all-gather rate RCCL reaches on the node (2.2 ms per matrix at 300 GB/s per rank -> crossover ~600 rows; 1 TB/s ->
        # ~100; one 153 GB/s link -> ~1300): default 768, ICL_DDP_SHARD_ROWS to tune on hardware, 0 = never.
        import os
        # an explicit choice (constructor argument or ICL_DDP_SHARD_ROWS / ICL_DDP_DENSE_ROWS) wins over what calibrate() measures
        self._explicit_shard = shard_min_rows is not None or "ICL_DDP_SHARD_ROWS" in os.environ
        if shard_min_rows is None:
            shard_min_rows = int(os.environ.get("ICL_DDP_SHARD_ROWS", "768"))
        self.shard_min_rows = shard_min_rows
        # gathered factor rows above which a layer forms its dense gradient instead (ops.FactoredGrads.max_rows_gathered is the switch
        # the layers read; it is set from HERE for the duration of a step by ICLTrainer, not globally: ADVICE round 4)
        from . import ops as _ops
        self._explicit_dense = "ICL_DDP_DENSE_ROWS" in os.environ
        self.max_rows_gathered = int(os.environ.get("ICL_DDP_DENSE_ROWS", str(_ops.FactoredGrads.max_rows_gathered)))
        self._sharded = []       # parameters whose update of this step is row-sharded
        self._names = {id(p): n for n, p in model.named_parameters()}
        self.rates = {"allgather_gbps": ASSUMED_ALLGATHER_GBPS, "allreduce_gbps": None, "measured": False}
        self.last_plan = []      # per factored matrix of the last step: mode and projected ms (bench.py prints it)

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def calibrate(self, nbytes: int = 4 * _REF_ELEMS, iters: int = 2):
        """Collective (every rank, outside graph capture): measures what the node's RCCL actually delivers on the two exchanges the
        crossover depends on — the in-place all-gather of a parameter-sized buffer (764 MB) and the all-reduce of the same buffer —
        takes the MAX time over ranks, and derives `shard_min_rows` (smallest gathered row count from which the row-sharded update +
        all-gather beats the whole update) and ops.FactoredGrads.max_rows_gathered (where forming and all-reducing the dense gradient
        would win) from the measured rates instead of the assumed 300 GB/s.  With one rank there is nothing to measure."""
        if self.world <= 1 or not dist.is_initialized():
            return self.rates
        import time
        p0 = self.params[0]
        n = nbytes // 4 // self.world * self.world
        buf = torch.empty(n, dtype=torch.float32, device=p0.device)
        mine = buf[dist.get_rank() * (n // self.world):(dist.get_rank() + 1) * (n // self.world)]

        def sync():
            if buf.is_cuda:
                torch.cuda.synchronize(buf.device)

        def timed(fn):
            fn()                                                        # warm-up: connection set-up is not the steady rate
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            sync()
            t = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=p0.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        if self._avg:
            t_ag = timed(lambda: dist.all_gather_into_tensor(buf, mine))
        else:
            t_ag = timed(lambda: dist.all_gather(list(buf.chunk(self.world)), mine.clone()))
        t_ar = timed(lambda: dist.all_reduce(buf))
        # (a loaded host under gloo can be slower than 0.05 GB/s: keep the digits, a rate of 0.0 would divide by zero in exchange_plan)
        self.rates = {"allgather_gbps": max(round(4.0 * n / t_ag / 1e9, 4), 1e-4), "allreduce_gbps": max(round(4.0 * n / t_ar / 1e9, 4), 1e-4), "measured": True,
                      "allgather_ms_764MB": round(t_ag * 1e3, 3), "allreduce_ms_764MB": round(t_ar * 1e3, 3)}
        del buf
        # thresholds for the 13,824^2 matrices (they are 97 % of the model) from the priced alternatives
        shard_from = dense_from = None
        for rows in range(16 * self.world, 4097, 16 * self.world):
            mode, _ = exchange_plan(rows, _REF_ELEMS, self.world, self.rates["allgather_gbps"], self.rates["allreduce_gbps"])
            if shard_from is None and mode == "shard":
                shard_from = rows
            if dense_from is None and mode == "dense":
                dense_from = rows
        # the thresholds live on THIS reducer (ICLTrainer hands max_rows_gathered to the layers per step); explicit settings win
        if not self._explicit_shard:
            self.shard_min_rows = shard_from if shard_from is not None else 0
        if dense_from is not None and not self._explicit_dense:
            self.max_rows_gathered = dense_from - 1
        self.rates["shard_min_rows"] = self.shard_min_rows
        self.rates["max_rows_gathered"] = self.max_rows_gathered
        self.rates["explicit"] = {"shard_min_rows": self._explicit_shard, "max_rows_gathered": self._explicit_dense}
        return self.rates

    def state_dict(self):
        """The exchange plan of this run: which matrices take the factored path, which updates are row-sharded.  Save it with the
        checkpoint and ``load_state_dict`` it on resume — a re-measured crossover would otherwise put a resumed run on a different
        arithmetic path (same mathematics, different summation order) than the run that wrote the checkpoint."""
        return {"world": self.world, "shard_min_rows": self.shard_min_rows, "max_rows_gathered": self.max_rows_gathered,
                "rates": dict(self.rates)}

    def load_state_dict(self, sd):
        if sd.get("world") != self.world:
            raise RuntimeError(f"GradientReducer: the saved exchange plan is for {sd.get('world')} ranks, this group has {self.world}")
        self.shard_min_rows, self.max_rows_gathered = int(sd["shard_min_rows"]), int(sd["max_rows_gathered"])
        self._explicit_shard = self._explicit_dense = True       # a loaded plan is a choice: calibrate() must not replace it
        self.rates = dict(sd.get("rates", self.rates))

    def broadcast_parameters(self, src: int = 0):
        for p in self.params:
            dist.broadcast(p.data, src)

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

    def pack(self):
        """Device-side preparation, no communication: the small gradients of a bucket are concatenated into one flat buffer;
        factored gradients (ops.FactoredGrads: dW = g^T x with a few dozen rows) are scaled for the mean — the mean over ranks
        of g_r^T x_r is [g_1/W; ...; g_W/W]^T [x_1; ...; x_W], so the ranks exchange the row blocks (~1 MB per 13,824^2
        matrix) instead of all-reducing 764 MB — and the gather targets are allocated."""
        if self.static and self._captured:
            raise RuntimeError("GradientReducer: the packed buffers belong to the captured hipGraphs; an eager step would rebind "
                               "them and later replays would communicate on buffers the graphs never touch")
        self._flat, self._fac = [], []
        if not self.active:
            return
        # large dense gradients: reduced alone and in place (no concatenation copy of a 764 MB tensor)
        self._early = {id(p) for p in self.params if p.grad is not None and p.grad.numel() >= self.overlap_min and p.grad.is_contiguous()}
        for p in self.params:
            if id(p) in self._early:
                self._flat.append((p.grad.view(-1), None))
        for bucket in self._buckets():
            if len(bucket) == 1 and bucket[0].grad.is_contiguous():
                self._flat.append((bucket[0].grad.view(-1), None))          # reduced in place
            else:
                self._flat.append((torch.cat([p.grad.reshape(-1) for p in bucket]), bucket))
        inv = 1.0 / self.world
        pieces = []
        for p in self.params:
            fac = getattr(p, "_icl_factors", None)
            if not fac:
                continue
            g = fac[0][0] if len(fac) == 1 else torch.cat([f[0] for f in fac], 0)
            x = fac[0][1] if len(fac) == 1 else torch.cat([f[1] for f in fac], 0)
            G = torch.empty((self.world * g.shape[0], g.shape[1]), dtype=g.dtype, device=g.device)
            X = torch.empty((self.world * x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
            self._fac.append((p, G, X))
            pieces += [(g * inv).reshape(-1), x.reshape(-1)]
        self._send = self._recv = None
        if pieces:   # ONE exchange for all factor rows of the step: [g_1/W | x_1 | g_2/W | x_2 | ...]
            self._send = torch.cat(pieces)
            self._recv = torch.empty(self.world * self._send.numel(), dtype=self._send.dtype, device=self._send.device)

    def communicate(self):
        """The collectives on the packed buffers (every rank must hold the same set of non-None grads and the same number of
        factor rows — true for ICL: a property of the graph, not of the data) and the wait for the overlapped ones."""
        if not self.active:
            return
        inv = 1.0 / self.world
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        for flat, _ in self._flat:
            dist.all_reduce(flat, op=op)
            if not self._avg:
                flat.mul_(inv)
        if self._send is not None:
            if self._avg:     # RCCL: one contiguous gather
                dist.all_gather_into_tensor(self._recv, self._send)
            else:
                dist.all_gather(list(self._recv.chunk(self.world, 0)), self._send)

    def unpack(self):
        """Device-side, after ``communicate()``: the gathered block [rank][g_1 | x_1 | g_2 | ...] is split into the per-matrix row
        blocks [g_i of rank 0; g_i of rank 1; ...] that ``sgd_factored_kernel`` reads (one multi-tensor copy)."""
        if self._send is None:
            return
        t = self._send.numel()
        recv = self._recv.view(self.world, t)
        dsts, srcs, off = [], [], 0
        for _, G, X in self._fac:
            for buf in (G, X):
                n = buf.numel() // self.world
                dsts.append(buf.view(self.world, n))
                srcs.append(recv[:, off:off + n])
                off += n
        assert off == t
        torch._foreach_copy_(dsts, srcs)

    def rebind(self):
        """``p.grad`` becomes a view into its bucket's reduced buffer (no copy back) and ``p._icl_factors`` the gathered rows."""
        for flat, bucket in self._flat:
            if bucket is None:
                continue
            off = 0
            for p in bucket:
                n = p.grad.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        self._sharded = []
        self.last_plan = []
        for p, G, X in self._fac:
            p._icl_factors = [(G, X)]
            shard = (self.world > 1 and self.shard_min_rows > 0 and G.shape[0] >= self.shard_min_rows and p.shape[0] % self.world == 0
                     and p.is_contiguous())
            if p.numel() >= (1 << 22):      # the plan line of bench.py: what this matrix does and what the alternatives would cost
                _, cost = exchange_plan(G.shape[0], p.numel(), self.world, self.rates["allgather_gbps"], self.rates["allreduce_gbps"],
                                        p.shape[0] % max(self.world, 1) == 0)
                self.last_plan.append({"param": self._names.get(id(p), "?"), "shape": list(p.shape), "rows_gathered": int(G.shape[0]),
                                       "mode": "shard" if shard else "whole", "projected_ms": cost})
            # consumed (and cleared) by FusedSGD._step_factored together with these factors: the decision never outlives its step
            p._icl_shard = (dist.get_rank(), self.world) if shard else None
            if shard:
                self._sharded.append(p)

    def sharded_params(self):
        """The matrices whose update of this step is row-sharded (decided in rebind(): a function of shapes and world size only)."""
        return list(self._sharded)

    def post_update(self, async_op: bool = False):
        """After the update of the row-sharded matrices: the ranks exchange the rows they updated (in place, no staging copy).
        ``async_op``: the all-gathers are only STARTED — ordered after what is queued on the current stream, i.e. after those updates —
        and travel over xGMI while the caller queues the rest of the optimiser step; ``finish_post_update()`` orders the current stream
        after them.  (Round 5: 2 x 764 MB at nc = 16 on 8 ranks are 5.1 ms at 300 GB/s per rank, the updates they now run under 4.3 ms.)"""
        self._pending = []
        for p in self._sharded:
            rows = p.shape[0] // self.world
            r = dist.get_rank()
            if self._avg:      # RCCL: in-place all-gather, this rank's rows are already where they belong
                w = dist.all_gather_into_tensor(p.data.view(-1), p.data[r * rows:(r + 1) * rows].reshape(-1), async_op=async_op)
            else:
                w = dist.all_gather([p.data[i * rows:(i + 1) * rows] for i in range(self.world)], p.data[r * rows:(r + 1) * rows].clone(),
                                    async_op=async_op)
            if async_op:
                self._pending.append(w)

    def finish_post_update(self):
        for w in getattr(self, "_pending", []):
            w.wait()
        self._pending = []

    def reduce_gradients(self):
        """Call after ``loss.backward()``: averages every gradient over the ranks (parameters whose grad is None are skipped)."""
        if not self.active:
            return
        self.pack()
        self.communicate()
        self.unpack()
        self.rebind()
        if not self.static:
            self._flat, self._fac = [], []
            self._send = self._recv = None
