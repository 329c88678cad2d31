"""FusedSGD — torch.optim.SGD(momentum, weight_decay) semantics on one HIP pass per parameter.

Counterpart of ``optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=1e-4)`` in the reference trainers
(/root/reference/code/train_inherent_consistent_unet_3D_BraTS.py:85-86).  It is a ``torch.optim.Optimizer`` so the
trainers' ``for g in optimizer.param_groups: g['lr'] = lr_`` schedule (:117-119) keeps working.  Parameters whose
``.grad`` is None are skipped entirely — no weight decay, no momentum — exactly like torch.
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import _lib

_BIG = 1 << 20
SPLIT_MIN_ROWS = 192    # factor rows from which d = g^T x runs on split products (tools/sgd_probe.py, 13,824^2: 128 rows 769 -> 720 us
                        # but +2 small launches: a wash inside the step; 256 rows 1244 -> 1029 us, 512 rows 2158 -> 1770 us)


_MODEL_OF = {}     # id(parameter) -> (weak reference to the parameter, weak reference to its model)


def tag_model_parameters(model: torch.nn.Module) -> torch.nn.Module:
    """Called by the model factories (networks/net_factory*.py): the model every parameter belongs to is remembered, so that an optimiser
    built from ``model.parameters()`` alone — the one line of the reference trainers — can find the module to hook (FusedSGD step scope).
    The map lives HERE, not on the tensors: a weak reference in a Parameter's ``__dict__`` made ``torch.save(model)`` / pickle of every
    factory-built model raise.  Entries die with their parameter."""
    ref = weakref.ref(model)
    for p in model.parameters():
        key = id(p)
        _MODEL_OF[key] = (weakref.ref(p, lambda _, k=key: _MODEL_OF.pop(k, None)), ref)
    return model


def model_of(p):
    e = _MODEL_OF.get(id(p))
    if e is None or e[0]() is not p:
        return None
    return e[1]()


def _no_hook(module, args):
    return None


class _ScopePreHook:
    """The forward pre-hook that opens FusedSGD's step scope.  A module-level class holding a weak reference (the model does not keep
    its optimiser alive); pickling a hooked model stores a no-op in its place."""

    def __init__(self, opt):
        self.opt = weakref.ref(opt)

    def __call__(self, module, args):
        opt = self.opt()
        if opt is not None:
            opt._open_scope(module)

    def __reduce__(self):
        return (_load_no_hook, ())


def _load_no_hook():
    return _no_hook


class FusedSGD(torch.optim.Optimizer):
    """``step_scope`` (default on): when the parameters belong to ONE icl_amd model built by ``net_factory_3d`` / ``net_factory``, the
    optimiser opens — from a forward pre-hook of that model, in training mode with gradients enabled — the step-scoped machinery that
    ``ICLTrainer`` opens around its iteration, and closes it in ``step()``: packed convolution weights (``ops.PackedWeights``), factored
    gradients of the token-axis matrices and their SGD step inside backward (``ops.FactoredGrads``), deferred bias gradients and
    BatchNorm counters, weight gradients of the deep levels on a lane (``ops.WgradLane``).  The unchanged reference loop
    (``outputs = model(..); ..; optimizer.zero_grad(); loss.backward(); optimizer.step()``,
    /root/reference/code/train_inherent_consistent_unet_3D_BraTS.py:103-115) then runs the fast path with the one-line optimiser swap of
    INTEGRATION.md §2.  Requirement it inherits from ``update_in_backward``: every ``loss.backward()`` of a training-mode forward is
    followed by ``optimizer.step()`` (the big matrices are updated inside their backward pass); pass ``update_in_backward=False`` for
    loops that accumulate gradients over several backward passes, ``step_scope=False`` to get a plain fused optimiser."""

    def __init__(self, params, lr=0.01, momentum=0.9, weight_decay=0.0, step_scope: bool = True, update_in_backward: bool = True,
                 graph: bool = False, graph_warmup: int = 3):
        if momentum <= 0:
            raise ValueError("FusedSGD implements the momentum form used by the ICL trainers")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._scope_model = None
        self._scope_open = False
        self._scope_update_in_backward = bool(update_in_backward)
        self._scope_packed = None
        self._scope_hook = None
        # graph=True (round 5, needs the step scope): after `graph_warmup` eager training iterations the model's forward and backward
        # are captured into two hipGraphs (the step scope's bookkeeping inside them) and every later `model(x_lab, x_unlab)` /
        # `loss.backward()` of the UNCHANGED loop replays them; the losses, `optimizer.zero_grad()`, `optimizer.step()` and the
        # `.item()` reads of the loop stay eager.  See _GraphedStep.
        self._graph_on = bool(graph) and bool(step_scope)
        self._graph_warmup = max(int(graph_warmup), 1)
        self._graph_state = None
        self._graph_failed = None
        self._train_calls = 0
        self._cap_stream = None      # the stream of every training forward before AND during the capture (see _install_graphed_forward)
        if step_scope:
            models = {id(m): m for m in (model_of(p) for g in self.param_groups for p in g["params"]) if m is not None}
            if len(models) == 1:
                (model,) = models.values()
                self._scope_model = weakref.ref(model)
                self._scope_hook = model.register_forward_pre_hook(_ScopePreHook(self))
                if self._graph_on:
                    _install_graphed_forward(self, model)
        self.lr_dev = None   # optional device scalar read by the kernels instead of group['lr'] (hipGraph replay)
        self._groups = None  # id(parameter) -> its group (update_in_backward)
        self._updated_in_backward = set()   # ids of the parameters whose update of this step already ran inside backward
        # Where the update of a big once-used matrix runs on a single rank (update_in_backward):
        #   "fused"  inside its backward, in the pass that computes the input gradient (one read of the matrix, 20 B/weight and step)
        #            — what "tail" (the default) does for every matrix the model has not marked
        #   "gated"  the backward only computes the input gradient (4 B/weight); the update (16 B/weight) is queued on an update stream
        #            that starts when the model opens the gate (flush_deferred() from a gradient hook where the chip-filling part of
        #            the backward ends) and is joined in step(): 24 B/weight and step, but the HBM stream runs next to the small
        #            deep-level kernels instead of next to the 96^3 / 48^3 convolutions (tools/critical_path.py)
        #   "free"   like "gated" without the gate: the update stream only waits for the input-gradient kernel of its matrix
        #   "tail"   "gated" for the matrices the model marks (`_icl_tail_update`: the own-query aligner's 24^3 level), with the gate where
        #            that level's map chain is done and only the serial query chain is left (~1 ms of small dependent launches that leave
        #            the chip idle): their update streams run beside that chain; every other matrix stays fused
        #   "deep"   (round 6) the marked matrices' updates start at the gate where the DEEP part of the backward pass begins (the up3 hook
        #            of the backbone, unet_3D._open_update_gate) as NARROW persistent launches (icl_sgd_step_factored_narrow: `update_wgs`
        #            workgroups of 1,024 threads, one CU each): the deep levels' launches fill 72-144 of the 256 CUs and leave the HBM
        #            idle, so the 16 B / weight streams run under them, and the forked phase ends with the aligners' (now fused) query
        #            chain alone instead of beside two 3 GB streams
        self.update_placement = os.environ.get("ICL_UPDATE_PLACEMENT", "deep")
        self.update_wgs = int(os.environ.get("ICL_UPDATE_WGS", "128"))
        # "deep", early gate: the model may open the gate BEFORE the deferring backward nodes run (unet_3D: when the decoder's own gradient
        # of up3 is complete — autograd processes the upper decoder before the aligners); `open_gate()` then records an event and every
        # "deep" entry is queued the moment it is deferred, behind that event: the narrow streams start while the step's stream still waits
        # for the aligners' query chain instead of after it
        # Measured and NOT the default (profiles/r6_qchain_ab.txt, item 6): 10.82-11.51 against 10.25-10.37 ms — beside the aligners' chain of
        # dependent 5 us launches even a half-chip stream costs more (HBM latency under load) than the 0.58 ms head start returns
        self.update_early = os.environ.get("ICL_UPDATE_EARLY", "0") != "0"
        self._gate_event = None
        self._deferred = []          # (parameter, g, x, event after the input-gradient kernel)
        self._update_stream = None
        self._update_stream_used = False

    def zero_grad(self, set_to_none: bool = True):
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_icl_factors", None) is not None:
                    p._icl_factors = None
        # a step that was skipped or failed after its backward pass must not make the next one raise "updated twice"
        self._updated_in_backward.clear()
        if self._deferred or self._update_stream_used:
            raise RuntimeError("FusedSGD.zero_grad(): updates queued by the previous backward pass were never applied; call step() "
                               "after every backward pass when update_placement is 'tail' (the default), 'gated' or 'free' — or "
                               "abandon_step() to discard what a failed iteration left behind")
        super().zero_grad(set_to_none=set_to_none)

    def abandon_step(self):
        """Discard what an iteration that will not reach ``step()`` left behind (an exception between backward and step, a skipped
        step): queued updates of the update stream are dropped — the stream is joined first, so nothing of them is still running —,
        the factors of factored gradients and the step scope are cleared.  Matrices whose update already ran inside their backward pass
        (placement 'fused') keep it: that pass cannot be undone.  ICLTrainer calls this when its step raises."""
        if self._update_stream is not None and self._update_stream_used:
            torch.cuda.current_stream(self._update_stream.device).wait_stream(self._update_stream)
        self._deferred = []
        self._gate_event = None
        self._update_stream_used = False
        self._updated_in_backward.clear()
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_icl_factors", None) is not None:
                    p._icl_factors = None
        if self._scope_open:
            self._close_scope(flush=False)

    # ---- step scope (see the class docstring)
    def _open_scope(self, module, capturing: bool = False):
        from . import ops
        from .networks.layers import BatchNormAct
        if not module.training or not torch.is_grad_enabled():
            return
        if self._graph_state is not None and not capturing:
            return                       # the captured graphs carry the scope's work; nothing runs eagerly
        if self._scope_open:
            # A second training forward before step(): gradient accumulation over micro-batches, or several forward
            # passes whose losses are summed before ONE backward.  The scope stays open — use counts, queued bias gradients and
            # factored gradients keep accumulating, exactly as `.grad` does — and the packed weights are refreshed (one launch; the
            # weights may have been touched between the calls).  The one thing that cannot continue: a backward pass that has already
            # applied this step's update of the big matrices (update_in_backward) followed by another forward without step().
            if self._updated_in_backward or self._deferred or self._update_stream_used:
                raise RuntimeError(
                    "FusedSGD: a training forward pass started after a backward pass that already applied this step's update of the "
                    "token-axis matrices (update_in_backward=True) and before optimizer.step(): the second micro-batch would see "
                    "half-updated weights.  Call optimizer.step() after every loss.backward(), or build the optimiser with "
                    "update_in_backward=False for loops that accumulate gradients over several backward passes (or abandon_step() "
                    "to discard a failed iteration)")
            self._scope_packed.begin_step()
            return
        if ops.PackedWeights.current is not None or ops.FactoredGrads.uses is not None:
            return                       # somebody else (ICLTrainer) has a step open
        if self._scope_packed is None:
            self._scope_packed = ops.PackedWeights()
        self._scope_open = True
        self._scope_packed.begin_step()
        BatchNormAct.defer_counters()
        multi = False
        try:
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        except Exception:                # noqa: BLE001
            multi = False
        ops.WgradLane.begin_step()
        self._scope_prev_factored = ops.FactoredGrads.enabled
        self._scope_multi = multi
        if not multi:
            # single process: weight gradients of the deep levels on a lane, bias / LayerNorm gradients reduced by one launch — both
            # completed when the backward pass ENDS (ops.BackwardEnd), so `.grad` is whole when loss.backward() returns
            ops.DeferredBiasGrads.begin()
            ops.DeferredWgradReduce.begin()
            ops.WgradLane.open = True    # (only backward functions consult it)
            me = weakref.ref(self)
            ops.BackwardEnd.hook = lambda: (me() is not None and me()._end_of_backward())
            ops.FactoredGrads.enabled = True
            ops.FactoredGrads.world = 1
            ops.FactoredGrads.fused_optimizer = self if self._scope_update_in_backward else None
            ops.FactoredGrads.uses = {} if self._scope_update_in_backward else None
        # multi-process (a data-parallel wrapper around the unchanged loop): dense `.grad` tensors that pass through autograd's own
        # accumulation — a wrapper's accumulator hooks fire for every parameter and bucket nothing that another stream is still
        # writing — so neither the lane nor the bias deferral nor the factored gradients are opened; packed weights and the
        # BatchNorm counters (forward-only) still are

    def _end_of_backward(self):
        """Runs when an autograd pass that used the scope's lane / deferrals finishes (on the stream of the caller of backward())."""
        from . import ops
        if not self._scope_open:
            return
        ops.WgradLane.join()
        if ops.DeferredWgradReduce.pending is not None:
            ops.DeferredWgradReduce.flush(keep_open=True)
        if ops.DeferredBiasGrads.pending is not None:
            ops.DeferredBiasGrads.flush(keep_open=True)

    def _close_scope(self, flush: bool = True):
        from . import ops
        from .networks.layers import BatchNormAct
        ops.BackwardEnd.hook = None
        ops.WgradLane.join()
        if ops.DeferredWgradReduce.pending is not None:
            if flush:
                ops.DeferredWgradReduce.flush()
            ops.DeferredWgradReduce.pending = None
        ops.WgradLane.open = False
        ops.WgradLane.uses = None
        ops.FactoredGrads.enabled = self._scope_prev_factored
        ops.FactoredGrads.fused_optimizer = None
        ops.FactoredGrads.uses = None
        if flush:
            BatchNormAct.flush_counters()
            if ops.DeferredBiasGrads.pending is not None:
                ops.DeferredBiasGrads.flush()
        else:
            BatchNormAct.deferred = None
            ops.DeferredBiasGrads.pending = None
            ops.DeferredBiasGrads.producers = None
        self._scope_packed.end_step()
        self._scope_open = False

    def _step_factored(self, L, p, factors, lr, mom, wd, narrow: int = 0):
        """dW = sum over entries of g^T x, applied without forming it (ops.FactoredGrads, csrc/kernels/optim.h)."""
        g = factors[0][0] if len(factors) == 1 else torch.cat([f[0] for f in factors], 0)
        x = factors[0][1] if len(factors) == 1 else torch.cat([f[1] for f in factors], 0)
        n, k = p.shape
        if g.shape[1] != n or x.shape[1] != k or g.shape[0] != x.shape[0] or not p.is_contiguous():
            raise RuntimeError("FusedSGD: factored gradient does not match its parameter")
        if not p.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("FusedSGD needs device tensors (no CPU fallback)")
        st = self.state[p]
        shard = getattr(p, "_icl_shard", None)
        p._icl_shard = None      # the decision belongs to THIS step's factors (ddp.GradientReducer.rebind sets it per step)
        first = 0
        if "momentum_buffer" not in st:
            # row-sharded: the other ranks' rows are never written here, so they must be zeros, not uninitialised memory
            st["momentum_buffer"] = torch.zeros_like(p) if shard is not None else torch.empty_like(p)
            first = 1
        elif st.get("momentum_shard") != shard and st.get("momentum_shard") is not None:
            self._gather_momentum(p)      # the shard decision flipped: make every row of the buffer valid first
        m = st["momentum_buffer"]
        if shard is not None:
            st["momentum_shard"] = tuple(shard)      # only rows [r N/W, (r+1) N/W) of the buffer are this parameter's momentum
        stream = ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream) if p.is_cuda else None
        lrp = self.lr_dev.data_ptr() if self.lr_dev is not None else None
        pv, mv = p, m
        if shard is not None:
            # data-parallel crossover (ddp.GradientReducer): this rank updates only its block of rows from the gathered factors —
            # 1/W of the MFMA-bound rank-(M W) product — and the ranks all-gather the updated rows afterwards (post_update)
            rank, world = shard
            r0, r1 = rank * (n // world), (rank + 1) * (n // world)
            pv, mv, g = p[r0:r1], m[r0:r1], g[:, r0:r1]
            n = r1 - r0
        g, x = g.contiguous(), x.contiguous()
        from . import ops   # KernelTimer bracket: the update streams p and m in and out (16 B per weight), HBM-bound
        rows = g.shape[0]
        with ops._timed("sgd_factored_kernel", 2.0 * rows * n * k, 16.0 * n * k, p):
            if narrow and shard is None and pv.is_contiguous():
                rc = L.icl_sgd_step_factored_narrow(pv.data_ptr(), mv.data_ptr(), g.data_ptr(), x.data_ptr(), rows, n, k, lr, mom, wd, first,
                                                    lrp, int(narrow), stream)
                if rc == 0:
                    return
                if rc != 1:      # 1: more factor rows than the narrow form takes
                    _lib.check(rc, "sgd_step_factored_narrow")
            if rows >= SPLIT_MIN_ROWS and (n * k) >= (1 << 22) and os.environ.get("ICL_SGD_SPLIT", "1") != "0":
                # many factor rows (gathered factors of a data-parallel step; nc = 16): d = g^T x from exact bf16 splits on the bf16
                # matrix pipe (csrc/kernels/optim.h sgd_factored_split_kernel) so that the update stays an HBM stream
                ws = ops._ws(L.icl_sgd_factored_split_ws_bytes(rows, n, k), p)
                _lib.check(L.icl_sgd_step_factored_split(pv.data_ptr(), mv.data_ptr(), g.data_ptr(), x.data_ptr(), ws.data_ptr(), rows, n, k,
                                                         lr, mom, wd, first, lrp, stream), "sgd_step_factored_split")
            else:
                _lib.check(L.icl_sgd_step_factored(pv.data_ptr(), mv.data_ptr(), g.data_ptr(), x.data_ptr(), rows, n, k, lr, mom, wd,
                                                   first, lrp, stream), "sgd_step_factored")

    def _gather_momentum(self, p):
        """Row-sharded data-parallel updates (ddp.GradientReducer, `_icl_shard`) keep only this rank's rows of the momentum buffer
        current.  Before the buffer is used whole again — the shard decision flips, or the state is saved — the ranks exchange
        their rows in place (a collective: every rank of the group gets here in the same step, the decision is a function of
        shapes and world size only)."""
        import torch.distributed as dist
        st = self.state[p]
        rank, world = st.pop("momentum_shard")
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != world:
            raise RuntimeError("FusedSGD: a row-sharded momentum buffer can only be completed by the process group that sharded it "
                               f"(world size {world}); call consolidate_momentum() before leaving the group")
        m = st["momentum_buffer"]
        rows = m.shape[0] // world
        mine = m[rank * rows:(rank + 1) * rows]
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(m.view(-1), mine.reshape(-1))
        else:
            dist.all_gather([m[i * rows:(i + 1) * rows] for i in range(world)], mine.clone())

    def consolidate_momentum(self):
        """Collective: completes every row-sharded momentum buffer (all ranks call it together)."""
        for group in self.param_groups:
            for p in group["params"]:
                if self.state.get(p, {}).get("momentum_shard") is not None:
                    self._gather_momentum(p)

    def state_dict(self):
        """LOCAL, like torch.optim.Optimizer.state_dict (the usual `if rank == 0: torch.save(opt.state_dict())` must not deadlock).
        If a data-parallel step has row-sharded a momentum buffer, only this rank's rows of it are current: saving that silently
        would write a wrong checkpoint, so this raises and names the fix — every rank calls ``consolidate_momentum()`` (a
        collective, outside any graph capture) first; ``ICLTrainer.optimizer_state_dict()`` does both."""
        pending = [i for g in self.param_groups for i, p in enumerate(g["params"]) if self.state.get(p, {}).get("momentum_shard") is not None]
        if pending:
            raise RuntimeError(f"FusedSGD.state_dict(): {len(pending)} momentum buffer(s) are row-sharded over the data-parallel group; "
                               "call optimizer.consolidate_momentum() on EVERY rank first (collective), then state_dict() on any rank")
        return super().state_dict()

    def can_update_in_backward(self, p, rows: int) -> bool:
        n, k = p.shape
        # 13,824^2: 0.55 ms for both against 0.13 + 0.54 apart; 1,728^2 matrices are launch-shaped (74 us against 41): left to step()
        return (rows <= 32 and n * k >= (1 << 26) and n % 4 == 0 and k % 4 == 0 and p.is_contiguous() and p.dtype == torch.float32
                and id(p) in self._group_of() and (p.is_cuda or _lib.host_pointers_ok()))

    def _group_of(self):
        if self._groups is None:
            self._groups = {id(p): g for g in self.param_groups for p in g["params"]}
        return self._groups

    @torch.no_grad()
    def update_in_backward(self, p, g, x):
        """Called from the backward of a big skinny Linear whose weight is used once per step (ops.FactoredGrads.fused_optimizer):
        returns gx = g W_old and applies this step's update of W from the factors (g, x) in the same pass over the matrix
        (icl_linear_dgrad_sgd).  ``step()`` then skips the parameter: it has neither a dense nor a factored gradient."""
        from . import ops
        L = _lib.lib()
        g, x = g.contiguous(), x.contiguous()
        if self.update_placement != "fused" and p.is_cuda and (self.update_placement not in ("tail", "deep") or getattr(p, "_icl_tail_update", False)):
            gx = ops.linear_dgrad_raw(g, p)
            ev = torch.cuda.Event()
            ev.record()
            # "deep" needs the narrow launch, which holds <= 16 factor rows: a matrix with more (nc = 16) keeps the tail gate
            self._deferred.append((p, g, x, ev, "deep" if (self.update_placement == "deep" and g.shape[0] <= 16) else "tail"))
            if self._gate_event is not None and self._deferred[-1][4] == "deep":
                self.flush_deferred(gate=False, only="deep", after=self._gate_event)
            torch.autograd.graph.increment_version(p)
            self._updated_in_backward.add(id(p))
            if self.update_placement == "free":
                self.flush_deferred(gate=False)
            return gx
        group = self._group_of()[id(p)]
        lr, mom, wd = float(group["lr"]), float(group["momentum"]), float(group["weight_decay"])
        n, k = p.shape
        rows = g.shape[0]
        st = self.state[p]
        first = 0
        if "momentum_buffer" not in st:
            st["momentum_buffer"] = torch.empty_like(p)
            first = 1
        m = st["momentum_buffer"]
        gx = torch.empty((rows, k), dtype=torch.float32, device=g.device)
        ws = ops._ws(L.icl_linear_ws_bytes(rows, k, n, 3), g)
        stream = ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream) if p.is_cuda else None
        lrp = self.lr_dev.data_ptr() if self.lr_dev is not None else None
        with ops._timed("linear_dgrad_sgd_kernel", 4.0 * rows * n * k, 16.0 * n * k, p):
            _lib.check(L.icl_linear_dgrad_sgd(g.data_ptr(), x.data_ptr(), p.data_ptr(), m.data_ptr(), gx.data_ptr(), ws.data_ptr(), rows, k, n,
                                              lr, mom, wd, first, lrp, stream), "linear_dgrad_sgd")
        # the kernel wrote the weight through its raw pointer: tell autograd, so that any other node of this graph that saved the
        # weight (a second use that the use count missed: F.linear, a tied weight, a hook) raises instead of differentiating
        # through the already-updated matrix
        torch.autograd.graph.increment_version(p)
        self._updated_in_backward.add(id(p))
        return gx

    @torch.no_grad()
    def open_gate(self):
        """Called by the model from a gradient hook at the point of the backward pass from which "deep" updates may run (the current
        stream's work up to here is what they wait for).  Entries deferred later are queued at once; entries already deferred now."""
        if self.update_placement != "deep" or not self.update_early:
            return
        ev = torch.cuda.Event()
        ev.record()
        self._gate_event = ev
        self.flush_deferred(gate=False, only="deep", after=ev)

    def flush_deferred(self, gate: bool = True, only: str = None, after=None):
        """Queue the updates that ``update_in_backward`` deferred on the update stream.  ``gate``: the stream also waits for everything
        queued so far on the CURRENT stream (the caller is a gradient hook at the point of the backward pass from which on the HBM is
        idle); every update waits for the input-gradient kernel that read its matrix.  ``only``: flush the entries of that gate ("tail":
        the aligner's hook where the serial query chain begins) and keep the others for a later call.  step() joins the stream."""
        todo = [e for e in self._deferred if only is None or e[4] == only]
        if not todo:
            return
        self._deferred = [e for e in self._deferred if not (only is None or e[4] == only)]
        L = _lib.lib()
        dev = todo[0][0].device
        if self._update_stream is None:
            # (stream priorities measured without effect on this step: profiles/r4_schedule_experiments.txt)
            self._update_stream = torch.cuda.Stream(device=dev)
        s = self._update_stream
        if after is not None:
            s.wait_event(after)
        if gate:
            s.wait_stream(torch.cuda.current_stream(dev))
        for p, g, x, ev, where in todo:
            s.wait_event(ev)
        with torch.cuda.stream(s):
            for p, g, x, ev, where in todo:
                group = self._group_of()[id(p)]
                self._step_factored(L, p, [(g, x)], float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]),
                                    narrow=self.update_wgs if where == "deep" else 0)
                g.record_stream(s)
                x.record_stream(s)
        self._update_stream_used = True

    @torch.no_grad()
    def step_subset(self, params):
        """The factored update of the given parameters only, now; ``step()`` afterwards finds nothing left to do for them.  Used by the
        data-parallel step for the ROW-SHARDED matrices: their updates run first so that the all-gather of the updated rows
        (ddp.GradientReducer.post_update) travels over xGMI while ``step()`` updates everything else."""
        L = _lib.lib()
        groups = self._group_of()
        for p in params:
            fac = getattr(p, "_icl_factors", None)
            if not fac:
                continue
            if p.grad is not None or id(p) in self._updated_in_backward:
                continue              # a mixed / already applied gradient: left to step(), which knows how to merge or refuse it
            group = groups[id(p)]
            self._step_factored(L, p, fac, float(group["lr"]), float(group["momentum"]), float(group["weight_decay"]))
            p._icl_factors = None

    @torch.no_grad()
    def step(self, closure=None):
        L = _lib.lib()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._scope_open:
            self._close_scope()                 # lane joined, deferred bias gradients assigned, packed weights released
        if self._graph_state is not None and self.lr_dev is not None and not (
                self.lr_dev.is_cuda and torch.cuda.is_current_stream_capturing()):
            # graph mode: a batch whose shape the graphs do not match ran eagerly — its update must not read the learning rate the
            # last REPLAYED backward left in the device scalar
            self.lr_dev.fill_(float(self.param_groups[0]["lr"]))
        self.flush_deferred(gate=True)      # a model without a gate hook: the updates start here
        self._gate_event = None
        if self._update_stream_used:
            torch.cuda.current_stream(self._update_stream.device).wait_stream(self._update_stream)
            self._update_stream_used = False
        for group in self.param_groups:
            lr, mom, wd = float(group["lr"]), float(group["momentum"]), float(group["weight_decay"])
            small = {0: [], 1: []}
            for p in group["params"]:
                g = p.grad
                fac = getattr(p, "_icl_factors", None)
                if id(p) in self._updated_in_backward and (g is not None or fac):
                    raise RuntimeError("FusedSGD: a parameter that was updated inside its backward pass (update_in_backward) also "
                                       "received a gradient from another use in the same step; this step's update would be applied "
                                       "twice.  Turn ICLConfig.update_in_backward off for models that reuse such a weight")
                if fac and g is not None:
                    # a weight with a factored use AND a dense one in the same step (ops.linear with few rows and with many): the
                    # factors are multiplied out into the dense gradient (csrc/kernels/gemm.h) and the ordinary update runs
                    from . import ops
                    for gf, xf in fac:
                        g = g + ops._tall_atb(gf, xf, False)[0]
                    p.grad = g
                    p._icl_factors = fac = None
                if fac:
                    self._step_factored(L, p, fac, lr, mom, wd)
                    p._icl_factors = None
                    continue
                if g is None:
                    continue
                if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32):
                    raise RuntimeError("FusedSGD needs contiguous fp32 parameters and gradients")
                if not p.is_cuda and not _lib.host_pointers_ok():
                    raise RuntimeError("FusedSGD needs device tensors (no CPU fallback)")
                st = self.state[p]
                first = 0
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.empty_like(p)
                    first = 1
                m = st["momentum_buffer"]
                if p.numel() >= _BIG:
                    stream = ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream) if p.is_cuda else None
                    lrp = self.lr_dev.data_ptr() if self.lr_dev is not None else None
                    _lib.check(L.icl_sgd_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), p.numel(), lr, mom, wd, first, lrp, stream),
                               "sgd_step")
                else:
                    small[first].append((p, g, m))
            for first, items in small.items():
                if not items:
                    continue
                n = len(items)
                arr = ctypes.c_void_p * n
                P_ = arr(*[it[0].data_ptr() for it in items])
                G_ = arr(*[it[1].data_ptr() for it in items])
                M_ = arr(*[it[2].data_ptr() for it in items])
                N_ = (ctypes.c_int64 * n)(*[it[0].numel() for it in items])
                p0 = items[0][0]
                stream = ctypes.c_void_p(torch.cuda.current_stream(p0.device).cuda_stream) if p0.is_cuda else None
                lrp = self.lr_dev.data_ptr() if self.lr_dev is not None else None
                _lib.check(L.icl_sgd_step_multi(P_, G_, M_, N_, n, lr, mom, wd, first, lrp, stream), "sgd_step_multi")
        self._updated_in_backward.clear()
        return loss


# ------------------------------------------------------------------------------------------------ graphed forward / backward of the unchanged loop
def _flatten_outputs(out):
    """(tensors, rebuild): the tensors of a nested tuple / list output in order, and a function that rebuilds the structure."""
    flat = []

    def walk(o):
        if isinstance(o, torch.Tensor):
            flat.append(o)
            return ("t", len(flat) - 1)
        if isinstance(o, (list, tuple)):
            return ("l" if isinstance(o, list) else "u", [walk(v) for v in o])
        return ("c", o)

    spec = walk(out)

    def rebuild(ts, node=spec):
        kind, v = node
        if kind == "t":
            return ts[v]
        if kind == "c":
            return v
        seq = [rebuild(ts, n) for n in v]
        return seq if kind == "l" else tuple(seq)

    return flat, rebuild


class _GraphedFn(torch.autograd.Function):
    """Connects the replayed forward graph's static outputs to autograd: backward copies the incoming gradients into the static
    gradient buffers, replays the backward graph and re-binds the parameters' (static) gradients and factored gradients — the loop's
    `optimizer.zero_grad()` between forward and backward has dropped the references, the buffers themselves belong to the graphs."""

    @staticmethod
    def forward(ctx, state, anchor, *outs):
        ctx.state = state
        ctx.set_materialize_grads(False)
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        st = ctx.state
        with torch.no_grad():
            for buf, g in zip(st.gouts, gs):
                if g is None:
                    buf.zero_()
                else:
                    buf.copy_(g)
            opt = st.opt()
            opt.lr_dev.fill_(float(opt.param_groups[0]["lr"]))
            st.gb.replay()
            for p, g in st.grads:
                p.grad = g
            for p, f in st.factors:
                p._icl_factors = list(f)
        return (None, None) + (None,) * len(gs)


class _GraphedStep:
    """Forward and backward of ONE icl_amd model as two hipGraphs, driven by the unchanged reference loop (FusedSGD(graph=True)).

    Capture (at the first training forward after the warm-up iterations, on the tensors of that call): graph F = the step scope's opening
    (all convolution weights packed and split in one launch each, dropout counter) + `model.forward`; graph B = `torch.autograd.backward`
    of the outputs against static gradient buffers + the scope's closing (lane join, BatchNorm counters, deferred bias gradients) + the
    join of FusedSGD's update stream.  The big token-axis matrices are updated INSIDE graph B (update_in_backward), everything else by the
    loop's own eager `optimizer.step()` from gradients that live in the graphs' memory pool.  Two graphs, not one, because forked streams
    that wait on each other in both directions crash hipStreamEndCapture on ROCm 7.2 (trainer.ICLTrainer.capture)."""

    def __init__(self, opt, module, orig_forward, args):
        from . import ops
        self.opt = weakref.ref(opt)
        dev = args[0].device
        if opt._scope_open:
            opt._close_scope(flush=False)       # the pre-hook opened an eager scope for this call: the captured one replaces it
        params = [p for g in opt.param_groups for p in g["params"]]
        for p in params:
            p.grad = None
            if getattr(p, "_icl_factors", None) is not None:
                p._icl_factors = None
        torch.cuda.synchronize(dev)
        if ops.StepRNG.tensor is None or ops.StepRNG.tensor.device != dev:
            ops.StepRNG.enable(dev)
        if opt.lr_dev is None:
            opt.lr_dev = torch.full((1,), float(opt.param_groups[0]["lr"]), dtype=torch.float32, device=dev)
        self.static_in = [a.detach().clone() for a in args]
        self.shapes = [tuple(a.shape) for a in args]
        pool = torch.cuda.graph_pool_handle()
        self.gf, self.gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gf, pool=pool, stream=opt._cap_stream):
            ops.StepRNG.begin_step()
            opt._open_scope(module, capturing=True)
            out = orig_forward(*self.static_in)
        flat, self.rebuild = _flatten_outputs(out)
        if not flat or not all(t.requires_grad for t in flat):
            raise RuntimeError("graphed step: every output of the training forward is expected to require grad")
        self.outs = flat
        self.souts = [t.detach() for t in flat]      # the static output buffers, cut loose from the autograd graph of the capture
        self.gouts = [torch.zeros_like(t) for t in flat]
        with torch.cuda.graph(self.gb, pool=pool, stream=opt._cap_stream):
            torch.autograd.backward(flat, self.gouts)
            if opt._scope_open:
                opt._close_scope()
            opt.flush_deferred(gate=True)
            if opt._update_stream_used:
                torch.cuda.current_stream(dev).wait_stream(opt._update_stream)
            ops.StepRNG.end_step()
        # nothing of the capture has run: forget the Python-side traces of a backward pass that never executed
        opt._deferred = []
        opt._gate_event = None
        opt._update_stream_used = False
        opt._updated_in_backward.clear()
        self.grads = [(p, p.grad) for p in params if p.grad is not None]
        self.factors = [(p, list(p._icl_factors)) for p in params if getattr(p, "_icl_factors", None)]
        self.anchor = next(p for p in params if p.requires_grad)
        self.outs = None      # (drops the autograd graph of the capture: its kernels live in the hipGraphs, its buffers in their pool)
        for p in params:
            p.grad = None
            if getattr(p, "_icl_factors", None) is not None:
                p._icl_factors = None

    def matches(self, args) -> bool:
        return len(args) == len(self.shapes) and all(isinstance(a, torch.Tensor) and tuple(a.shape) == s for a, s in zip(args, self.shapes))

    def run(self, args):
        with torch.no_grad():
            for buf, a in zip(self.static_in, args):
                if a.data_ptr() != buf.data_ptr():
                    buf.copy_(a)
        self.gf.replay()
        outs = _GraphedFn.apply(self, self.anchor, *self.souts)
        return self.rebuild(list(outs))


def _install_graphed_forward(opt, model):
    """Instance-level `forward` of the model: eager for inference / evaluation / the warm-up iterations, graph replay afterwards."""
    orig = model.forward
    me = weakref.ref(opt)

    def forward(*args, **kwargs):
        o = me()
        train = (o is not None and o._graph_on and o._graph_failed is None and model.training and torch.is_grad_enabled() and not kwargs
                 and len(args) == 2 and all(isinstance(a, torch.Tensor) and a.is_cuda for a in args))
        if not train:
            return orig(*args, **kwargs)
        st = o._graph_state
        if st is None:
            o._train_calls += 1
            dev = args[0].device
            if o._cap_stream is None:
                o._cap_stream = torch.cuda.Stream(device=dev)
            if o._train_calls <= o._graph_warmup:
                # Eager iterations (allocator warm-up, momentum buffers, one-time kernel attributes) — on the stream the capture will
                # use.  A parameter's AccumulateGrad node keeps the stream of the forward pass that created it, and the loop's own
                # variables (`outputs`, `loss` of the previous iteration) keep that node alive into the next forward: created on the
                # default stream it would run there during the capture of the backward graph, which hipStreamEndCapture does not survive.
                cur = torch.cuda.current_stream(dev)
                o._cap_stream.wait_stream(cur)
                with torch.cuda.stream(o._cap_stream):
                    out = orig(*args)
                cur.wait_stream(o._cap_stream)
                for t in _flatten_outputs(out)[0]:
                    t.record_stream(cur)
                return out
            try:
                st = _GraphedStep(o, model, orig, args)
            except Exception as e:          # noqa: BLE001 — a model that cannot be captured keeps training eagerly
                o._graph_failed = repr(e)
                o._graph_state = None
                import warnings
                warnings.warn(f"FusedSGD(graph=True): capture failed, staying eager: {e!r}")
                # nothing of the aborted capture has run: drop every Python-side trace of it (queued updates, "updated in backward"
                # marks, gradients and factors that point into the failed capture's pool, the device learning rate the captured
                # kernels would have read) before the eager scope reopens — the loop's next zero_grad() / step() must find a clean
                # optimiser
                o._deferred = []
                o._update_stream_used = False
                o._updated_in_backward.clear()
                if o._scope_open:
                    o._close_scope(flush=False)
                for g_ in o.param_groups:
                    for p_ in g_["params"]:
                        p_.grad = None
                        if getattr(p_, "_icl_factors", None) is not None:
                            p_._icl_factors = None
                o.lr_dev = None
                o._open_scope(model)
                return orig(*args)
            o._graph_state = st
        if not st.matches(args):
            return orig(*args)              # another batch shape: eager (the pre-hook's scope stayed closed: plain autograd path)
        return st.run(args)

    model.forward = forward
