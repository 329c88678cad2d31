"""torch.autograd wrappers around the C-ABI HIP kernels (include/icl_hip.h).

PyTorch is plumbing here: tensors own the device memory, autograd owns the graph, the current
torch stream is handed to every launch.  All arithmetic happens in libicl_hip.so.
Each function cites the reference operator it replaces (paths relative to /root/reference/code).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional, Sequence

import torch

from . import _lib

_vp = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else _vp(t.data_ptr())


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return _vp(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _require(*ts: Optional[torch.Tensor]):
    for t in ts:
        if t is None:
            continue
        if t.dtype not in (torch.float32, torch.uint8, torch.int64, torch.int32):
            raise TypeError(f"icl_amd: unsupported dtype {t.dtype}")
        if not t.is_cuda and not _lib.host_pointers_ok():
            raise RuntimeError("icl_amd: HIP kernels need device tensors (no CPU fallback)")


class SideStream:
    """Fork/join of the aligner heads onto a second HIP stream.

    The aligners (three calls, ~600 small launches forward+backward and the HBM-bound streaming of the 764 MB ``mlp2`` weights) need
    only the three deepest decoder maps; the rest of the decoder (the 48^3 / 96^3 convolutions, compute-bound on the matrix cores) does
    not depend on them.  ``with SideStream(inputs) as side:`` runs its body on a second stream ordered after everything queued on the
    current one; ``side.join(outputs)`` orders the current stream after it.  Autograd runs every backward node on the stream of its
    forward and synchronises across streams itself, so the aligner backward overlaps the backward of the upper decoder too; inside a
    captured step (ICLTrainer.capture) the two streams become parallel branches of the hipGraph.  Tensors crossing streams are
    recorded with the caching allocator (``record_stream``).  ``ICL_ALIGNER_STREAM=0`` (or a CPU tensor) runs the body in line."""
    enabled = os.environ.get("ICL_ALIGNER_STREAM", "1") != "0"
    # which branches take a lane: bit 0 the attention-map chains of sspa (own queries), bit 1 the guided levels of uscl (A/B runs:
    # with the three lanes 11.94 ms, only bit 0 13.95, only bit 1 13.62 on one box)
    lane_mask = 3
    lanes = int(os.environ.get("ICL_ALIGNER_LANES", "3"))   # further streams for the per-level branches inside the aligners (0: none)
    # Round 6: one more stream for the TOKEN side of the own-query aligner (1x1x1 projection, two LayerNorms, fc_kv of every level: needs
    # only the feature maps).  Forward: it runs beside the query chain, which waits for a level's k / v when it gets there.  Backward: the
    # token side of level i (130 us of 13,824-row kernels at 24^3) used to sit BETWEEN the chain's calls for levels i and i - 1 on the
    # aligner stream; on its own stream it runs beside them.  It forks from the step's stream (never from the aligner stream) and is
    # joined there: per captured graph every edge between forked streams keeps one direction (see join()).
    token_lane_on = os.environ.get("ICL_TOKEN_LANE", "1") != "0"
    _streams = {}
    _outer = None

    def __init__(self, inputs, lane: int = 0):
        """``lane`` 0 is the aligner stream itself; lanes 1..``SideStream.lanes`` carry branches forked from it (the attention-map
        chain of one resolution level: drop-path, LayerNorm, the token-axis ``mlp2``, the separable convolutions — independent of
        the query chain that links the levels, unet_3D_icl.py:209-222), so that the many small launches of the 6^3 / 12^3 levels
        and the HBM-bound 13,824^2 weight streams of the 24^3 level overlap instead of queueing behind each other.  A lane beyond
        ``SideStream.lanes`` runs its body in line on the current stream."""
        self.stream = None
        self.children, self.pending = [], []
        t0 = inputs[0]
        if SideStream.enabled and t0.is_cuda and torch.is_grad_enabled() and lane <= SideStream.lanes:
            dev = t0.device
            s = SideStream._streams.get((dev.index, lane))
            if s is None:
                s = SideStream._streams[(dev.index, lane)] = torch.cuda.Stream(device=dev)
            self.main = torch.cuda.current_stream(dev)
            s.wait_stream(self.main)
            self.tok = None
            if lane == 0:
                # the lanes fork from and join to the stream the aligner stream itself forks from (see join())
                for k in range(1, SideStream.lanes + 1):
                    sk = SideStream._streams.get((dev.index, k))
                    if sk is None:
                        sk = SideStream._streams[(dev.index, k)] = torch.cuda.Stream(device=dev)
                    sk.wait_stream(self.main)
                    self.children.append(sk)
            for t in inputs:
                t.record_stream(s)
            self.stream = s
            self._guard = torch.cuda.stream(s)

    @staticmethod
    def token_stream():
        """The token-side stream of the aligner block that is open on this thread, or None.  Forked from the step's stream on FIRST
        request (and joined there with the lanes): a block that never asks for it — nc = 16, SwinUNETR, whose aligners take the
        operator-by-operator path — captures no empty fork / join of it (a forked-but-unused token stream in the THIRD capture of a
        process made the replay of that graph crash inside hipGraphLaunch on ROCm 7.2: bench.py's default run, round 6)."""
        outer = SideStream._outer
        if outer is None or outer.stream is None or not SideStream.token_lane_on or SideStream.lanes <= 0:
            return None
        if outer.tok is None:
            dev = outer.stream.device
            sk = SideStream._streams.get((dev.index, "tok"))
            if sk is None:
                sk = SideStream._streams[(dev.index, "tok")] = torch.cuda.Stream(device=dev)
            sk.wait_stream(outer.main)
            outer.children.append(sk)
            outer.tok = sk
        return outer.tok

    def __enter__(self):
        if self.stream is not None:
            self._guard.__enter__()
            if self.children:
                SideStream._outer = self
        return self

    def __exit__(self, *a):
        if self.stream is not None:
            if SideStream._outer is self:
                SideStream._outer = None
            self._guard.__exit__(*a)

    def join(self, outputs):
        """Order the forking stream after this one.  A lane joined inside the aligner stream's block is NOT waited for there: the
        aligner stream's own join (to the stream everything forked from) waits for every lane — the lanes' results are consumed by the
        losses only, and a lane that the aligner stream waits on after it waited on the aligner stream is the two-way dependency
        between forked streams that a stream capture on ROCm 7.2 does not survive (trainer.capture)."""
        if self.stream is None:
            return
        outer = SideStream._outer
        if outer is not None and outer is not self:
            outer.pending.extend(outputs)
            return
        self.main.wait_stream(self.stream)
        for sk in self.children:
            self.main.wait_stream(sk)
        for t in list(outputs) + self.pending:
            t.record_stream(self.main)
        self.pending = []



class BackwardEnd:
    """End-of-backward work of a step scope that is NOT driven by ICLTrainer (FusedSGD's scope inside the unchanged reference loop).
    The lane of the deep levels' weight gradients is joined and the deferred bias / LayerNorm gradients are reduced when the autograd
    pass that produced them FINISHES (engine.queue_callback, armed by the first backward node that defers something), not in
    ``optimizer.step()``: code between ``loss.backward()`` and ``step()`` — ``clip_grad_norm_``, gradient logging, a GradScaler — then
    sees complete ``.grad`` tensors on the caller's stream.  ``hook`` is None unless such a scope is open."""
    hook = None
    _armed = False

    @classmethod
    def arm(cls):
        if cls.hook is None or cls._armed:
            return
        try:
            torch.autograd.Variable._execution_engine.queue_callback(cls._fire)
            cls._armed = True
        except RuntimeError:      # not inside a backward pass (a Function.backward called by hand): the scope's close does the work
            pass

    @classmethod
    def _fire(cls):
        cls._armed = False
        h = cls.hook
        if h is not None:
            h()


class WgradLane:
    """Weight gradients of the deep levels on a second stream.  From `up3` down (24^3, 12^3, 6^3) a layer's input-gradient and weight-gradient
    launches occupy 54-144 workgroups each on a chip of 256 CUs, and nothing else runs in that part of the backward pass (the aligner
    branches are done: tools/critical_path.py).  The weight gradient is a leaf of the graph — only the optimiser reads it — so inside
    ICLTrainer's step (``open``) `_Conv3d.backward` launches it on a lane ordered after the producer of dY and goes on with the input
    gradient on its own stream; ``join()`` (ICLTrainer, after backward) orders the step's stream after the lane.  Captured steps: the lane
    is a parallel branch of the hipGraph.  ``ICL_WGRAD_LANE=0`` keeps everything in line."""
    enabled = os.environ.get("ICL_WGRAD_LANE", "1") != "0"
    max_voxels = int(os.environ.get("ICL_WGRAD_LANE_MAX_VOXELS", str(24 ** 3)))
    open = False
    _stream = None
    _used = False
    # forward uses of every weight whose gradient may take the lane, counted while the step is open (ICLTrainer resets it per step).
    # The lane's only join is after loss.backward(), so a gradient computed there must reach the optimiser untouched: autograd has to
    # ADOPT the returned tensor (weight.grad is None, one use per step, no hooks on the parameter).  Anything else — a weight applied
    # twice, a .grad left from an earlier backward, parameter hooks — reads or adds to the gradient on the step's own stream during
    # backward; ``adoptable`` is False then and the caller orders its stream after the lane before it returns the gradient.
    uses = None

    @classmethod
    def begin_step(cls):
        cls.uses = {}
        cls.open = False

    @classmethod
    def note_use(cls, weight):
        if cls.uses is not None and weight is not None:
            cls.uses[id(weight)] = cls.uses.get(id(weight), 0) + 1

    @classmethod
    def adoptable(cls, weight) -> bool:
        if cls.uses is None or cls.uses.get(id(weight), 0) != 1 or weight.grad is not None:
            return False
        if getattr(weight, "_backward_hooks", None) or getattr(weight, "_post_accumulate_grad_hooks", None):
            return False
        return True

    @classmethod
    def sync_to_current(cls, like: torch.Tensor):
        """Order the current stream after everything queued on the lane (a gradient that is not simply adopted)."""
        torch.cuda.current_stream(like.device).wait_stream(cls._stream)

    @classmethod
    def wants(cls, x: torch.Tensor, voxels: int) -> bool:
        return cls.enabled and cls.open and x.is_cuda and voxels <= cls.max_voxels

    @classmethod
    def fork_point(cls, like: torch.Tensor):
        """Order the lane after everything queued so far on the current stream (call where dY is complete, BEFORE the input gradient
        is queued, so that the two gradients of the layer run side by side)."""
        dev = like.device
        if cls._stream is None or cls._stream.device != dev:
            # an aligner lane if there is one (idle in this part of the backward pass, and known to run beside the step's stream)
            cls._stream = SideStream._streams.get((dev.index, 1)) or torch.cuda.Stream(device=dev)
        cls._stream.wait_stream(torch.cuda.current_stream(dev))
        cls._used = True
        BackwardEnd.arm()

    @classmethod
    def on_lane(cls, *tensors):
        for t in tensors:
            if t is not None:
                t.record_stream(cls._stream)
        return torch.cuda.stream(cls._stream)

    @classmethod
    def join(cls):
        if cls._used:
            torch.cuda.current_stream(cls._stream.device).wait_stream(cls._stream)
            cls._used = False

class DeferredWgradReduce:
    """While open (a step scope: ICLTrainer, or FusedSGD's from the model's forward hook), the slab sums of the convolution weight
    gradients are not launched behind their kernels: `_Conv3d.backward` runs the weight-gradient kernel only (icl_conv3d_wgrad_slabs),
    hands autograd the still-unwritten gradient tensor and queues (slabs, gradient); ``flush()`` sums all of them in ONE launch
    (icl_conv3d_wgrad_reduce_multi) when the backward pass has ended and the weight-gradient lane is joined.  Only for a gradient that
    autograd ADOPTS untouched (one use of the weight in the step, no .grad yet, no hooks: WgradLane.adoptable) — anything that reads or
    accumulates the gradient during backward gets it complete, as before.  ``ICL_WGRAD_DEFER_REDUCE=0`` keeps every sum in line."""
    enabled = os.environ.get("ICL_WGRAD_DEFER_REDUCE", "1") != "0"
    pending = None

    @classmethod
    def begin(cls):
        cls.pending = [] if cls.enabled else None

    @classmethod
    def wants(cls, weight) -> bool:
        return cls.pending is not None and isinstance(weight, torch.nn.Parameter) and WgradLane.adoptable(weight)

    @classmethod
    def defer(cls, ws, gw, cout, cin, ks, nslabs, weight=None):
        # an ALIAS of the gradient's memory, not the tensor autograd is handed: AccumulateGrad adopts a gradient only while nobody else
        # holds it (use count) and clones it otherwise — it would clone the unwritten bytes and the sum would land in the original
        # (its STORAGE and address: a view would keep `gw` itself alive through `_base`)
        cls.pending.append((ws, (gw.untyped_storage(), gw.data_ptr(), gw.device, weight, tuple(gw.shape)), cout, cin, ks, nslabs))
        BackwardEnd.arm()

    @classmethod
    def flush(cls, keep_open: bool = False):
        items, cls.pending = cls.pending, ([] if (keep_open and cls.enabled) else None)
        if not items:
            return
        dev = items[0][1][2]
        here = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        n = len(items)
        arr, iarr = _vp * n, ctypes.c_int32 * n
        if here is not None:
            for ws, _gw, *_ in items:
                ws.record_stream(here)      # (the gradients themselves live until the optimiser has read them on this stream)
        def target(rec):
            # belt and braces: should autograd have CLONED the unwritten tensor after all (a condition `adoptable` does not foresee),
            # the parameter's .grad is that clone — the only gradient it has (it had none before): the sum goes there
            _st, ptr, _dev, weight, shape = rec
            g = getattr(weight, "grad", None) if weight is not None else None
            if g is not None and g.data_ptr() != ptr and tuple(g.shape) == shape and g.is_contiguous() and g.dtype == torch.float32:
                return g.data_ptr()
            return ptr
        _lib.check(_lib.lib().icl_conv3d_wgrad_reduce_multi(arr(*[it[0].data_ptr() for it in items]), arr(*[target(it[1]) for it in items]),
                                                           iarr(*[it[2] for it in items]), iarr(*[it[3] for it in items]),
                                                           iarr(*[it[4] for it in items]), iarr(*[it[5] for it in items]), n,
                                                           _vp(here.cuda_stream) if here is not None else None), "wgrad_reduce_multi")


class KernelTimer:
    """HIP-event timing of individual kernel launches on the stream they are launched on (bench.py roofline).
    Usage: ``with KernelTimer() as kt: step()``; ``kt.summary()`` -> {name: (launches, ms_total, flops, bytes)}."""

    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *a):
        KernelTimer.active = None

    def region(self, name, flops, nbytes, like):
        return _TimedRegion(self, name, flops, nbytes, like)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, fl, by in self.records:
            n, ms, f, b = out.get(name, (0, 0.0, 0.0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), f + fl, b + by)
        return out


def event_bracket_overhead_us(device, reps: int = 40) -> float:
    """Median HIP-event duration of a bracket around a one-element fill kernel: what an event pair adds to the kernel it brackets on an
    otherwise idle stream (dispatch latency of the kernel packet behind the event packet).  bench.py reports it next to the raw numbers."""
    t = torch.zeros(1, device=device)
    s = torch.cuda.current_stream(device)
    vals = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        t.fill_(1.0)
        e1.record(s)
        torch.cuda.synchronize(device)
        vals.append(e0.elapsed_time(e1) * 1e3)
    vals.sort()
    return vals[len(vals) // 2]


class _TimedRegion:
    def __init__(self, kt, name, flops, nbytes, like):
        self.kt, self.name, self.flops, self.nbytes, self.like = kt, name, flops, nbytes, like

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record(torch.cuda.current_stream(self.like.device))

    def __exit__(self, *a):
        self.e1.record(torch.cuda.current_stream(self.like.device))
        name = self.name
        if name.startswith("conv3d"):  # the launcher reports which template instantiation it picked
            kn = _lib.lib().icl_last_kernel_name()
            if kn:
                name = kn.decode()
        self.kt.records.append((name, self.e0, self.e1, self.flops, self.nbytes))


class _NullRegion:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullRegion()


def _timed(name, flops, nbytes, like):
    kt = KernelTimer.active
    if kt is None or not like.is_cuda:
        return _NULL
    return kt.region(name, flops, nbytes, like)


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 4) + 3) // 4, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------------------
# Conv3d (k=3 pad 1 or k=1), stride 1 — nn.Conv3d in networks/utils.py:104,107, unet_3D_icl.py:65
# --------------------------------------------------------------------------------------

def pack_weights(w: torch.Tensor, mode: int) -> torch.Tensor:
    L = _lib.lib()
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    n = L.icl_conv3d_packed_elems(cout, cin, ks, mode)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    _lib.check(L.icl_conv3d_pack_weights(_ptr(w), _ptr(wp), cout, cin, ks, mode, _stream(w)), "pack_weights")
    return wp


class PackedWeights:
    """Step-level cache of the packed convolution weights (forward + dgrad layout).  The weights only change in the optimiser
    step, so ICLTrainer brackets its iteration with ``begin_step()`` / ``end_step()``: the first iteration packs per call (as a
    bare module does) and records which Parameters asked; every later ``begin_step()`` packs all of them in ONE launch into
    persistent buffers (fixed addresses: the launch is part of the captured hipGraph).  Outside the bracket nothing is cached."""
    current: Optional["PackedWeights"] = None

    def __init__(self):
        self.entries = {}     # id(parameter) -> [parameter, wp, wpt, valid, wsplit_fwd, wsplit_dgrad, planes valid, wants planes]

    def begin_step(self):
        PackedWeights.current = self
        live = list(self.entries.values())
        if not live:
            return
        L = _lib.lib()
        n = len(live)
        arr = ctypes.c_void_p * n
        iarr = ctypes.c_int32 * n
        _lib.check(L.icl_conv3d_pack_weights_multi(arr(*[e[0].data_ptr() for e in live]), arr(*[e[1].data_ptr() for e in live]),
                                                   arr(*[e[2].data_ptr() for e in live]), iarr(*[e[0].shape[0] for e in live]),
                                                   iarr(*[e[0].shape[1] for e in live]), iarr(*[e[0].shape[2] for e in live]), n,
                                                   _stream(live[0][0])), "pack_weights_multi")
        for e in live:
            e[3] = True
        # the three bf16 planes of every pack that can run on the split-product kernels (csrc/kernels/conv_bf16x3.h), again in one
        # launch: a convolution call then starts its kernel directly instead of a small split launch in front of it
        jobs = []
        for e in live:
            cout, cin = e[0].shape[0], e[0].shape[1]
            if e[0].shape[2] != 3 or not e[7]:      # only weights that have met a volume the split-product kernels take
                continue
            if e[4] is None and e[5] is None:
                for slot, (ci, co) in ((4, (cin, cout)), (5, (cout, cin))):
                    nb = L.icl_conv3d_split_ws_bytes(ci, co)
                    if nb:
                        e[slot] = torch.empty(nb // 4, dtype=torch.float32, device=e[0].device)
            if e[4] is not None:
                jobs.append((e[1], e[4], cin, cout))
            if e[5] is not None:
                jobs.append((e[2], e[5], cout, cin))
        if jobs:
            m = len(jobs)
            arr, iarr = ctypes.c_void_p * m, ctypes.c_int32 * m
            _lib.check(L.icl_conv3d_split_weights_multi(arr(*[j[0].data_ptr() for j in jobs]), arr(*[j[1].data_ptr() for j in jobs]),
                                                        iarr(*[j[2] for j in jobs]), iarr(*[j[3] for j in jobs]), m, _stream(live[0][0])),
                       "split_weights_multi")
        for e in live:
            e[6] = True

    @staticmethod
    def split_of(weight):
        """(fwd planes, dgrad planes) of a weight whose packs were split by this step's ``begin_step()``; (None, None) otherwise."""
        cache = PackedWeights.current if isinstance(weight, torch.nn.Parameter) else None
        e = cache.entries.get(id(weight)) if cache is not None else None
        if e is None or e[0] is not weight or not e[3] or not e[6]:
            return None, None
        return e[4], e[5]

    def end_step(self):
        for e in self.entries.values():
            e[3] = e[6] = False
        if PackedWeights.current is self:
            PackedWeights.current = None

    @staticmethod
    def get(weight):
        """(wp, wpt) of ``weight`` (contiguous [Cout, Cin, k, k, k]), packed now unless this step's begin_step() already did."""
        L = _lib.lib()
        cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
        cache = PackedWeights.current if isinstance(weight, torch.nn.Parameter) else None
        e = cache.entries.get(id(weight)) if cache is not None else None
        if e is not None and e[0] is not weight:
            e = None
        if e is not None and e[3]:
            return e[1], e[2]
        if e is not None:
            wp, wpt = e[1], e[2]
        else:
            wp = torch.empty(L.icl_conv3d_packed_elems(cout, cin, ks, 0), dtype=torch.float32, device=weight.device)
            wpt = torch.empty(L.icl_conv3d_packed_elems(cout, cin, ks, 1), dtype=torch.float32, device=weight.device)
        _lib.check(L.icl_conv3d_pack_weights_both(_ptr(weight), _ptr(wp), _ptr(wpt), cout, cin, ks, _stream(weight)), "pack_weights_both")
        if cache is not None:
            if e is not None:
                e[1], e[2], e[3], e[6] = wp, wpt, True, False       # packed by this call: the split planes are stale
            else:
                cache.entries[id(weight)] = [weight, wp, wpt, True, None, None, False, False]
        return wp, wpt

    @staticmethod
    def note_volume(weight, voxels: int):
        """Called by the convolution with the size of the volume it runs on: weights that meet a volume of the split-product
        kernels (>= 6^3 voxels — the launcher decides per shape — unless ICL_CONV_SPLIT_MIN says otherwise) get their bf16 planes from
        the next begin_step() on."""
        cache = PackedWeights.current if isinstance(weight, torch.nn.Parameter) else None
        e = cache.entries.get(id(weight)) if cache is not None else None
        if e is not None and e[0] is weight and voxels >= int(os.environ.get("ICL_CONV_SPLIT_MIN", 6 ** 3)):
            e[7] = True


def conv3d_forward_raw(x, wp, bias, n, cin, cout, d, h, w, ks, x_bstride, y, y_bstride, wsplit=None, want_stats=False):
    """``wsplit``: the pack's bf16 planes when the step's PackedWeights.begin_step() has split them already.
    ``want_stats``: returns the InstanceNorm statistics of y — [n * cout, slots, 3] (count, mean, M2) summaries written by the
    convolution's epilogue (icl_conv3d_fwd_presplit_stats) — or None when this shape / path does not produce them."""
    L = _lib.lib()
    s = d * h * w
    # algorithmic work of this launch: 2*taps*Cin*Cout FLOP per voxel; 4*(I+O+W) bytes
    flops = 2.0 * ks ** 3 * cin * cout * s * n
    nbytes = 4.0 * (n * s * (cin + cout) + ks ** 3 * cin * cout)
    need = L.icl_conv3d_fwd_ws_bytes(n, cin, cout, d, h, w, ks)
    ws = _ws(need, x) if need else None
    with _timed("conv3d_mfma_fwd_kernel", flops, nbytes, x):
        if want_stats and wsplit is not None and ks == 3 and os.environ.get("ICL_CONV_STATS", "1") != "0":
            slots = L.icl_conv3d_fwd_stats_slots(n, cin, cout, d, h, w)
            if slots > 0:
                stats = torch.empty((n * cout, slots, 3), dtype=torch.float32, device=x.device)
                rc = L.icl_conv3d_fwd_presplit_stats(_ptr(x), _ptr(wsplit), _ptr(bias), _ptr(y), _ptr(stats), n, cin, cout, d, h, w,
                                                     x_bstride, y_bstride, _stream(x))
                if rc != 1:
                    _lib.check(rc, "conv3d_fwd_presplit_stats")
                    return stats
        if wsplit is not None and ks == 3:
            # deep levels (rows of 12 / 24 voxels): the launcher may split the channel chunks over workgroups and needs a slab workspace
            kneed = L.icl_conv3d_fwd_presplit_ws_bytes(n, cin, cout, d, h, w)
            if kneed:
                kws = _ws(kneed, x)
                rc = L.icl_conv3d_fwd_presplit_ws(_ptr(x), _ptr(wsplit), _ptr(bias), _ptr(y), _ptr(kws), n, cin, cout, d, h, w,
                                                  x_bstride, y_bstride, _stream(x))
            else:
                rc = L.icl_conv3d_fwd_presplit(_ptr(x), _ptr(wsplit), _ptr(bias), _ptr(y), n, cin, cout, d, h, w, x_bstride, y_bstride,
                                               _stream(x))
            if rc != 1:         # 1: this shape does not run on the split-product kernel -> the fp32 pack below
                _lib.check(rc, "conv3d_fwd_presplit")
                return
        _lib.check(L.icl_conv3d_fwd(_ptr(x), _ptr(wp), _ptr(bias), _ptr(y), _ptr(ws), n, cin, cout, d, h, w, ks,
                                    x_bstride, y_bstride, _stream(x)), "conv3d_fwd")


_ZERO_CACHE = {}


def _cached_zeros(n: int, device) -> torch.Tensor:
    key = (n, str(device))
    z = _ZERO_CACHE.get(key)
    if z is None:
        z = _ZERO_CACHE[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return z


CONV1X1_WGRAD_MIN_VOXELS = 4096
FIRST_CONV_PLANES_MAX_K = 32      # Cin * 27 <= 32, i.e. one input channel
CONV1X1_GEMM_MIN_VOXELS = 65536
CONV1X1_SMALL_MAX_CHANNELS = 16       # 1x1x1 convolutions with <= 16 -> <= 16 channels below the GEMM threshold: icl_conv1x1_small


class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, zero_bias_grad=False, want_stats=False):
        """want_stats: returns (y, stats) — stats = the InstanceNorm summaries of y from the convolution's epilogue, or an empty
        tensor when this call did not produce them (the normalisation then runs its own statistics pass)."""
        y = _Conv3d._forward(ctx, x, weight, bias, zero_bias_grad, want_stats)
        if not want_stats:
            return y
        ctx.set_materialize_grads(False)      # (no zero tensor — a fill launch per call — for the non-differentiable statistics output)
        stats = ctx.stats if getattr(ctx, "stats", None) is not None else torch.empty(0, dtype=torch.float32, device=x.device)
        ctx.stats = None
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def _forward(ctx, x, weight, bias, zero_bias_grad, want_stats):
        _require(x, weight, bias)
        x = x.contiguous()
        weight = weight.contiguous()
        ctx.zero_bias_grad = zero_bias_grad
        n, cin, d, h, w = x.shape
        cout, ks = weight.shape[0], weight.shape[2]
        assert weight.shape[1] == cin and weight.shape[2] == weight.shape[3] == weight.shape[4]
        s = d * h * w
        ctx.save_for_backward(x, weight)
        if ks == 3:
            WgradLane.note_use(weight)
        ctx.has_bias = bias is not None
        small_ch = cin <= CONV1X1_SMALL_MAX_CHANNELS and cout <= CONV1X1_SMALL_MAX_CHANNELS and s % 4 == 0
        ctx.pointwise_gemm = ks == 1 and n * s >= CONV1X1_GEMM_MIN_VOXELS and not small_ch
        if ctx.pointwise_gemm:
            # 1x1x1 convolution with many channels on a big volume = one batched product  W [Cout,Cin] x x[b] [Cin,S]  on the
            # tiled fp32-MFMA kernel (csrc/kernels/gemm.h), reading the channel-major volume in place (k-strided B operand)
            y = torch.empty((n, cout, d, h, w), dtype=torch.float32, device=x.device)
            gemm(weight, x, cout, s, cin, cin, s, True, False, out=y, ldc=s, bias=bias, act=16, batch=n, b_bstride=cin * s,
                 c_bstride=cout * s)       # act 16: the bias is indexed by the output row (= channel)
            return y
        y = torch.empty((n, cout, d, h, w), dtype=torch.float32, device=x.device)
        ctx.wpt = None
        ctx.pointwise_small = ks == 1 and small_ch
        if ctx.pointwise_small:
            # <= 16 channels (the aligner's h -> h / h -> 1 maps, the `final` 16 -> num_classes convolution on 96^3 voxels): VALU
            # kernels on the natural weight layout, HBM-bound on big volumes
            _lib.check(_lib.lib().icl_conv1x1_small(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), n, cin, cout, s, cin, 1, _stream(x)),
                       "conv1x1_small")
            return y
        ctx.cin1 = (ks == 3 and cin == 1 and cout <= 16 and not ctx.needs_input_grad[0] and 120 * (w + 2) <= 65536
                    and os.environ.get("ICL_CONV_CIN1", "1") != "0")
        if ctx.cin1:
            # the first convolution of the backbones (one input channel): K = 27, an HBM stream of the output (csrc/kernels/conv_cin1.h)
            with _timed("conv_cin1_fwd_kernel", 2.0 * 27 * cout * s * n, 4.0 * n * s * (1 + cout), x):
                _lib.check(_lib.lib().icl_conv3d_cin1_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), n, cout, d, h, w, cin * s, cout * s,
                                                          _stream(x)), "conv3d_cin1_fwd")
            return y
        wsf = ctx.wsd = None
        if ctx.needs_input_grad[0]:
            # the input gradient will need the flipped/transposed packing too: one launch for both, kept for backward
            wp, ctx.wpt = PackedWeights.get(weight)
            wsf, ctx.wsd = PackedWeights.split_of(weight)
            if ks == 3:
                PackedWeights.note_volume(weight, s)
        else:
            wp = pack_weights(weight, 0)
        ctx.stats = conv3d_forward_raw(x, wp, bias, n, cin, cout, d, h, w, ks, cin * s, y, cout * s, wsplit=wsf, want_stats=want_stats)
        return y

    @staticmethod
    def backward(ctx, gy, gstats=None):
        if gy is None:
            return None, None, None, None, None
        x, weight = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        n, cin, d, h, w = x.shape
        cout, ks = weight.shape[0], weight.shape[2]
        s = d * h * w
        gx = gw = gb = None
        lane = (ks == 3 and WgradLane.wants(x, s) and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not getattr(ctx, "cin1", False)
                and not (cin * 27 <= FIRST_CONV_PLANES_MAX_K))
        if lane:
            WgradLane.fork_point(gy)
        if ctx.needs_input_grad[0]:
            if ctx.pointwise_gemm:
                # gx[b] [Cin,S] = W^T gy[b]: A = W read transposed (k-strided), B = gy[b] [Cout,S] (k-strided)
                gx = torch.empty_like(x)
                gemm(weight, gy, cin, s, cout, cin, s, False, False, out=gx, ldc=s, batch=n, b_bstride=cout * s, c_bstride=cin * s)
            elif ctx.pointwise_small:
                gx = torch.empty_like(x)
                _lib.check(L.icl_conv1x1_small(_ptr(gy), _ptr(weight), None, _ptr(gx), n, cout, cin, s, 1, cin, _stream(x)),
                           "conv1x1_small dgrad")
            else:
                gx = torch.empty_like(x)
                wpt = ctx.wpt if ctx.wpt is not None else pack_weights(weight, 1)
                conv3d_forward_raw(gy, wpt, None, n, cout, cin, d, h, w, ks, cout * s, gx, cin * s, wsplit=getattr(ctx, "wsd", None))
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            gb_arg = gb
            if ctx.has_bias and ctx.zero_bias_grad:
                # the conv feeds an InstanceNorm: its output is invariant to the bias, so dL/dbias == 0 exactly
                # (the reference holds ~1e-8 rounding noise there); skip the reduction pass over dY.  The zero vector is a
                # cached constant (nothing ever writes a non-zero into it), not a fill kernel per layer and step; returned as a
                # fresh view so that AccumulateGrad adopts it instead of cloning a tensor it sees other references to.
                # Only inside ICLTrainer's step bracket, where FusedSGD is the sole consumer of the gradient; anywhere else (stock
                # optimisers, user code writing into .grad) every bias gets its own zero tensor.
                gb = _cached_zeros(cout, x.device).view(cout) if PackedWeights.current is not None else x.new_zeros(cout)
                gb_arg = None
            flops = 2.0 * ks ** 3 * cin * cout * s * n
            nbytes = 4.0 * (n * s * (cin + cout) + 2 * ks ** 3 * cin * cout)
            if ks == 1 and s % 4 == 0 and n * s >= CONV1X1_WGRAD_MIN_VOXELS:
                # big-volume 1x1x1 convolution: HBM-bound channel-major reduction, rows split over many waves
                ws = _ws(L.icl_conv1x1_wgrad_ws_bytes(n, s, cin, cout), x)
                with _timed("conv1x1_wgrad_kernel", flops, nbytes, x):
                    _lib.check(L.icl_conv1x1_wgrad(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb_arg), _ptr(ws), n, cin, cout, s, cin * s,
                                                   cout * s, _stream(x)), "conv1x1_wgrad")
            elif getattr(ctx, "cin1", False) and w % 4 == 0 and gb_arg is None and gy.data_ptr() % 16 == 0:
                ws = _ws(L.icl_conv3d_cin1_wgrad_ws_bytes(n, d, h), x)
                with _timed("conv_cin1_wgrad_kernel", flops, nbytes, x):
                    _lib.check(L.icl_conv3d_cin1_wgrad(_ptr(x), _ptr(gy), _ptr(gw), _ptr(ws), n, cout, d, h, w, cin * s, cout * s,
                                                       _stream(x)), "conv3d_cin1_wgrad")
            elif ks == 3 and cin * 27 <= FIRST_CONV_PLANES_MAX_K and w % 4 == 0 and n * s >= CONV1X1_WGRAD_MIN_VOXELS * 64:
                # first convolution (one input channel, big volume): 27 shifted planes + the same HBM-bound reduction; the implicit
                # GEMM pads Cin to 16 and takes 300 us for 0.8 GFLOP (batch 2, 96^3), this takes ~130
                planes = torch.empty((n, cin * 27, s), dtype=torch.float32, device=x.device)
                _lib.check(L.icl_im2col3_planes(_ptr(x), _ptr(planes), n, cin, d, h, w, _stream(x)), "im2col3_planes")
                ws = _ws(L.icl_conv1x1_wgrad_ws_bytes(n, s, cin * 27, cout), x)
                with _timed("conv1x1_wgrad_kernel", flops, nbytes, x):
                    _lib.check(L.icl_conv1x1_wgrad(_ptr(planes), _ptr(gy), _ptr(gw), _ptr(gb_arg), _ptr(ws), n, cin * 27, cout, s,
                                                   cin * 27 * s, cout * s, _stream(x)), "conv1x1_wgrad")
            else:
                # One cout block but three (or six, nine ...) cin blocks — up_concat1.conv1, 48 -> 16 @96^3, the U-Net's largest weight
                # gradient: with the roles of x and dY exchanged, dW[co][ci][t] = sum_q dY[co][q - t] x[ci][q], the z-column kernel stages
                # ONE halo'd dY block for three x blocks instead of one halo'd x block per dY block (matrix pipe 52 -> 68 %,
                # profiles/r5_pmc_conv.md); the result comes out as [cin][cout] with mirrored taps and is written back by two tiny copies.
                tiny6 = ks == 3 and (d, h, w) == (6, 6, 6) and _f6_split_ok(x, weight)
                swap = (ks == 3 and gb_arg is None and cout == 16 and cin % 48 == 0 and h % 8 == 0 and d >= 8 and w % 4 == 0
                        and s >= WGRAD_SWAP_MIN_VOXELS and os.environ.get("ICL_WGRAD_SWAP", "1") != "0")

                def wgrad():
                    if tiny6:
                        # dW[cout][cin * 27] = dY^T [cout, n S] x im2col(x) [n S, cin * 27]: the tall / skinny product of ops.linear's
                        # weight gradient (the form _conv3d_tiny_volume gives it), here beside the split-product input gradient
                        cols = torch.empty((n * s, cin * 27), dtype=torch.float32, device=x.device)
                        _lib.check(L.icl_im2col3(_ptr(x), _ptr(cols), n, cin, d, h, w, _stream(x)), "im2col3")
                        g2 = gy.reshape(n, cout, s).transpose(1, 2).reshape(n * s, cout).contiguous()
                        gw2, gb2 = _tall_atb(g2, cols, gb_arg is not None)
                        gw.copy_(gw2.view_as(gw))
                        if gb_arg is not None:
                            gb_arg.copy_(gb2)
                        return
                    if swap:
                        ws = _ws(L.icl_conv3d_wgrad_ws_bytes(n, cout, cin, ks), x)
                        gws = torch.empty((cin, cout, ks, ks, ks), dtype=torch.float32, device=x.device)
                        with _timed("conv3d_mfma_wgrad_kernel", flops, nbytes, x):
                            _lib.check(L.icl_conv3d_wgrad(_ptr(gy), _ptr(x), _ptr(gws), None, _ptr(ws), n, cout, cin, d, h, w, ks,
                                                          cout * s, cin * s, _stream(x)), "conv3d_wgrad (roles exchanged)")
                        gw.copy_(gws.flip(2, 3, 4).transpose(0, 1))
                        return
                    ws = _ws(L.icl_conv3d_wgrad_ws_bytes(n, cin, cout, ks), x)
                    if gb_arg is None and DeferredWgradReduce.wants(weight):
                        # the slab sum is a leaf of the pass: queued, one launch for all of them when backward has ended
                        ns = ctypes.c_int32(0)
                        with _timed("conv3d_mfma_wgrad_kernel", flops, nbytes, x):
                            _lib.check(L.icl_conv3d_wgrad_slabs(_ptr(x), _ptr(gy), _ptr(ws), n, cin, cout, d, h, w, ks, cin * s, cout * s,
                                                                ctypes.byref(ns), _stream(x)), "conv3d_wgrad_slabs")
                        DeferredWgradReduce.defer(ws, gw, cout, cin, ks, ns.value, weight)
                        return
                    with _timed("conv3d_mfma_wgrad_kernel", flops, nbytes, x):
                        _lib.check(L.icl_conv3d_wgrad(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb_arg), _ptr(ws), n, cin, cout, d, h, w, ks,
                                                      cin * s, cout * s, _stream(x)), "conv3d_wgrad")
                if lane:
                    # deep level: beside the input gradient instead of behind it (the lane was forked where dY was complete, before
                    # the input gradient above was queued on this stream)
                    with WgradLane.on_lane(x, gy, gw, gb_arg):
                        wgrad()
                    if not WgradLane.adoptable(weight):
                        WgradLane.sync_to_current(gw)      # accumulated / hooked gradient: no race with the lane
                else:
                    wgrad()
        return gx, gw, gb, None, None


class _Im2Col3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        n, c, d, h, w = x.shape
        cols = torch.empty((n * d * h * w, c * 27), dtype=torch.float32, device=x.device)
        _lib.check(L.icl_im2col3(_ptr(x), _ptr(cols), n, c, d, h, w, _stream(x)), "im2col3")
        ctx.shape = (n, c, d, h, w)
        return cols

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        n, c, d, h, w = ctx.shape
        g = g.contiguous()
        dx = torch.empty((n, c, d, h, w), dtype=torch.float32, device=g.device)
        _lib.check(L.icl_col2im3(_ptr(g), _ptr(dx), n, c, d, h, w, _stream(g)), "col2im3")
        return dx


SMALL_CONV_MAX_VOXELS = 216      # 6^3
WGRAD_SWAP_MIN_VOXELS = 24 ** 3    # exchanged-roles weight gradient (see _Conv3d.backward): only where the split-product kernels run
SMALL_CONV_MIN_WEIGHTS = 128 * 128 * 27


def _f6_split_ok(x, weight) -> bool:
    """A 3^3 convolution on a whole 6^3 volume that the split-product kernel takes as ONE flat tile, split over the channel chunks
    (csrc/kernels/conv_bf16x3.h Bf3F6, round 6): the centre block of the 3-D U-Net.  Forward and input gradient then are one launch + a
    slab sum each instead of im2col / col2im around a skinny product; the weight gradient keeps the product form."""
    cout, cin = weight.shape[0], weight.shape[1]
    return (tuple(x.shape[2:]) == (6, 6, 6) and cin % 16 == 0 and cin >= 32 and ((cout + 15) // 16 * 16) % 32 == 0
            and os.environ.get("ICL_CONV_SPLIT", "1") != "0" and os.environ.get("ICL_CONV_SPLIT_FLAT6", "1") != "0"
            and os.environ.get("ICL_CONV_SPLIT_KSPLIT", "-1") != "0" and int(os.environ.get("ICL_CONV_SPLIT_MIN", 216)) <= 216)


def _conv3d_tiny_volume(x, weight, bias):
    """3^3 convolution on <= 6^3 voxels with >= 128x128 channels: a skinny GEMM that streams the weights once
    (csrc/kernels/misc.h im2col3 / col2im3 + the weight-streaming / tiled products of csrc/kernels/gemm.h through ``linear``;
    dW through its tall/skinny paths)."""
    n, cin, d, h, w = x.shape
    cout = weight.shape[0]
    y = linear(_Im2Col3.apply(x), weight.flatten(1), bias, lane_wgrad=True)            # [n*S, cout]; dW beside the input gradient (WgradLane)
    return y.view(n, d * h * w, cout).permute(0, 2, 1).reshape(n, cout, d, h, w)


def conv3d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
           zero_bias_grad: bool = False) -> torch.Tensor:
    """Conv3d, kernel 3 (pad 1) or 1 (pad 0), stride 1.  ``zero_bias_grad``: the caller guarantees the output goes
    straight into a mean-removing normalisation, so the bias gradient is identically zero."""
    if (weight.shape[2] == 3 and x.shape[2] * x.shape[3] * x.shape[4] <= SMALL_CONV_MAX_VOXELS
            and weight.numel() >= SMALL_CONV_MIN_WEIGHTS and not _f6_split_ok(x, weight)):
        return _conv3d_tiny_volume(x, weight, bias)
    return _Conv3d.apply(x, weight, bias, zero_bias_grad)


def conv3d_instance_norm_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 1, eps: float = 1e-5):
    """Conv3d(k=3, pad=1) -> InstanceNorm3d(affine=False) -> act (1 = ReLU): the block of UnetConv3
    (/root/reference/code/networks/utils.py:104-106).  Where the convolution runs on the split-product kernels inside a trainer step,
    its epilogue hands the normalisation the per-(sample, channel) statistics and the stand-alone statistics pass is skipped; everywhere
    else this is conv3d followed by instance_norm_act."""
    if (weight.shape[2] == 3 and x.shape[2] * x.shape[3] * x.shape[4] <= SMALL_CONV_MAX_VOXELS
            and weight.numel() >= SMALL_CONV_MIN_WEIGHTS and not _f6_split_ok(x, weight)):
        return _NormAct.apply(_conv3d_tiny_volume(x, weight, bias), None, None, None, None, 0, True, int(act), eps, 0.0, None)
    y, stats = _Conv3d.apply(x, weight, bias, True, True)
    return _NormAct.apply(y, None, None, None, None, 0, True, int(act), eps, 0.0, stats if stats.numel() else None)


# --------------------------------------------------------------------------------------
# InstanceNorm3d / BatchNorm3d (+ReLU) — networks/utils.py:105-109, unet_3D_icl.py:325-340
# --------------------------------------------------------------------------------------

class _NormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, mode, use_batch_stats, act, eps, momentum, stats=None):
        """stats: [rows, slots, 3] (count, mean, M2) summaries of x written by its producer (conv3d_forward_raw) — the statistics
        pass over x is skipped."""
        _require(x, gamma, beta, running_mean, running_var)
        L = _lib.lib()
        x = x.contiguous()
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        groups = c if mode == 1 else n * c
        y = torch.empty_like(x)
        ws = _ws(L.icl_norm_ws_bytes(n, c, s), x)
        if use_batch_stats:
            mean = torch.empty(groups, dtype=torch.float32, device=x.device)
            rstd = torch.empty(groups, dtype=torch.float32, device=x.device)
            rm, rv = running_mean, running_var
        else:  # eval-mode BatchNorm: fixed statistics
            assert mode == 1 and running_mean is not None
            mean = running_mean.detach().clone()
            rstd = torch.empty(groups, dtype=torch.float32, device=x.device)
            _lib.check(L.icl_rstd_from_var(_ptr(running_var), _ptr(rstd), c, eps, _stream(x)), "rstd_from_var")
            rm = rv = None
        if stats is not None and use_batch_stats:
            assert mode == 0 and stats.shape[0] == n * c and stats.is_contiguous()
            _lib.check(L.icl_norm_fwd_given_stats(_ptr(x), None, _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv),
                                                  n, c, s, mode, act, eps, momentum, _ptr(stats), stats.shape[1], _stream(x)),
                       "norm_fwd_given_stats")
        else:
            _lib.check(L.icl_norm_fwd(_ptr(x), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv),
                                      n, c, s, mode, int(use_batch_stats), act, eps, momentum, _ptr(ws), _stream(x)), "norm_fwd")
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.cfg = (mode, int(use_batch_stats), act)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        mode, ubs, act = ctx.cfg
        L = _lib.lib()
        gy = gy.contiguous()
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        gx = torch.empty_like(x)
        dg = db = None
        if gamma is not None:
            dg = torch.empty_like(gamma)
            db = torch.empty_like(beta)
        ws = _ws(L.icl_norm_ws_bytes(n, c, s), x)
        _lib.check(L.icl_norm_bwd(_ptr(gy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(gx), _ptr(dg),
                                  _ptr(db), n, c, s, mode, ubs, act, _ptr(ws), _stream(x)), "norm_bwd")
        return gx, dg, db, None, None, None, None, None, None, None, None


def instance_norm_relu(x: torch.Tensor, relu: bool = True, eps: float = 1e-5) -> torch.Tensor:
    """InstanceNorm3d(affine=False, no running stats) [+ ReLU] fused."""
    return _NormAct.apply(x, None, None, None, None, 0, True, int(relu), eps, 0.0)


def instance_norm_act(x: torch.Tensor, act: int, eps: float = 1e-5) -> torch.Tensor:
    """InstanceNorm3d(affine=False) fused with act 0 none / 1 ReLU / 2 LeakyReLU(0.01) (MONAI UnetResBlock norm1+lrelu, norm3)."""
    return _NormAct.apply(x, None, None, None, None, 0, True, int(act), eps, 0.0)


# --------------------------------------------------------------------------------------
# Deferred InstanceNorm3d + ReLU (round 5) — Conv3d -> InstanceNorm3d -> ReLU of UnetConv3 (networks/utils.py:107-109) whose output
# feeds MaxPool3d, the skip concatenation, nn.Upsample or the `final` convolution (unet_3D_icl.py:41-53,116-117; utils.py:264,276)
# --------------------------------------------------------------------------------------

def _lazy_norm_min_voxels() -> int:
    """Smallest volume whose normalisation is deferred; ICL_LAZY_NORM=0 turns the deferral off (both read per call: the tests switch them)."""
    if os.environ.get("ICL_LAZY_NORM", "1") == "0":
        return 1 << 62
    return int(os.environ.get("ICL_LAZY_NORM_MIN", str(48 ** 3)))


class LazyAct:
    """A convolution output whose InstanceNorm + ReLU is deferred to its consumers.  ``t`` holds the RAW convolution output — in autograd
    terms it already IS the normalised activation (its gradient is the gradient of relu(norm(y)); `_NormActLazy.backward` turns it into
    the gradient of y) — and ``ss`` the per-(sample, channel) (scale, shift) pairs with which ``skip_and_pool``, ``upsample2x_concat`` and
    ``dropout_conv1x1`` / ``conv1x1_lazy`` apply relu(fma(t, scale, shift)) while they load it.  Never hand ``t`` to anything else:
    ``materialize()`` writes the normalised tensor out (one pass, what the non-deferred path always does).  Measured upper bound of not
    writing the 96^3 / 48^3 activations of a step: profiles/r5_skip_norm_fwd_upper_bound.txt."""
    __slots__ = ("t", "ss")

    def __init__(self, t: torch.Tensor, ss: torch.Tensor):
        self.t, self.ss = t, ss

    @property
    def shape(self):
        return self.t.shape

    def materialize(self) -> torch.Tensor:
        return _LazyMaterialize.apply(self.t, self.ss)


def materialized(x):
    return x.materialize() if isinstance(x, LazyAct) else x


class _NormActLazy(torch.autograd.Function):
    """(y, summaries) -> (y as the virtual normalised activation, ss).  Backward: InstanceNorm + ReLU backward exactly as _NormAct's."""

    @staticmethod
    def forward(ctx, y, stats, eps):
        _require(y, stats)
        L = _lib.lib()
        n, c = y.shape[0], y.shape[1]
        mean = torch.empty(n * c, dtype=torch.float32, device=y.device)
        rstd = torch.empty(n * c, dtype=torch.float32, device=y.device)
        ss = torch.empty((n * c, 2), dtype=torch.float32, device=y.device)
        _lib.check(L.icl_norm_finalize_stats(_ptr(stats), n, c, stats.shape[1], eps, _ptr(mean), _ptr(rstd), _ptr(ss), _stream(y)),
                   "norm_finalize_stats")
        ctx.save_for_backward(y, mean, rstd)
        ctx.mark_non_differentiable(ss)
        ctx.set_materialize_grads(False)
        return y.view_as(y), ss

    @staticmethod
    def backward(ctx, ga, gss=None):
        if ga is None:
            return None, None, None
        y, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        ga = ga.contiguous()
        n, c = y.shape[0], y.shape[1]
        s = y.numel() // (n * c)
        gy = torch.empty_like(y)
        ws = _ws(L.icl_norm_ws_bytes(n, c, s), y)
        _lib.check(L.icl_norm_bwd(_ptr(ga), _ptr(y), _ptr(mean), _ptr(rstd), None, None, _ptr(gy), None, None, n, c, s, 0, 1, 1, _ptr(ws),
                                  _stream(y)), "norm_bwd")
        return gy, None, None


class _LazyMaterialize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, ss):
        L = _lib.lib()
        n, c = t.shape[0], t.shape[1]
        s = t.numel() // (n * c)
        if s % 4:
            sc = ss[:, 0].reshape(n, c, *([1] * (t.dim() - 2)))
            sh = ss[:, 1].reshape(n, c, *([1] * (t.dim() - 2)))
            return torch.clamp_min(torch.addcmul(sh, t, sc), 0.0)
        out = torch.empty_like(t)
        _lib.check(L.icl_norm_apply(_ptr(t), _ptr(ss), _ptr(out), n, c, s, _stream(t)), "norm_apply")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None      # `t` already stands for the normalised activation in the autograd graph


def conv3d_instance_norm_act_lazy(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float = 1e-5):
    """``conv3d_instance_norm_act(..., act=1)`` returning a ``LazyAct`` when the convolution's epilogue produced the statistics and the
    volume is big enough for the normalisation pass to matter (>= 48^3 voxels), the normalised tensor otherwise."""
    s = x.shape[2] * x.shape[3] * x.shape[4]
    if weight.shape[2] != 3 or s < _lazy_norm_min_voxels() or s % 4 or not torch.is_grad_enabled():
        return conv3d_instance_norm_act(x, weight, bias, act=1, eps=eps)
    y, stats = _Conv3d.apply(x, weight, bias, True, True)
    if not stats.numel():
        return _NormAct.apply(y, None, None, None, None, 0, True, 1, eps, 0.0, None)
    av, ss = _NormActLazy.apply(y, stats, eps)
    return LazyAct(av, ss)


class _InstanceNormAddAct(torch.autograd.Function):
    """act(InstanceNorm3d(x) + res): the tail of MONAI UnetResBlock.forward in one pass (and one backward pass that
    produces both gx and gres)."""

    @staticmethod
    def forward(ctx, x, res, act, eps, stats=None):
        _require(x, res)
        L = _lib.lib()
        x, res = x.contiguous(), res.contiguous()
        assert x.shape == res.shape
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        y = torch.empty_like(x)
        mean = torch.empty(n * c, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n * c, dtype=torch.float32, device=x.device)
        if stats is not None:       # the producing convolution's epilogue wrote the (count, mean, M2) summaries (conv3d_forward_raw)
            assert stats.shape[0] == n * c and stats.is_contiguous()
            _lib.check(L.icl_norm_fwd_given_stats(_ptr(x), _ptr(res), _ptr(y), _ptr(mean), _ptr(rstd), None, None, None, None, n, c, s, 0, act,
                                                  eps, 0.0, _ptr(stats), stats.shape[1], _stream(x)), "norm_fwd_given_stats")
        else:
            ws = _ws(L.icl_norm_ws_bytes(n, c, s), x)
            _lib.check(L.icl_norm_res_fwd(_ptr(x), _ptr(res), _ptr(y), _ptr(mean), _ptr(rstd), None, None, None, None, n, c, s, 0, 1, act,
                                          eps, 0.0, _ptr(ws), _stream(x)), "norm_res_fwd")
        ctx.save_for_backward(x, res, mean, rstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, res, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        n, c = x.shape[0], x.shape[1]
        s = x.numel() // (n * c)
        gx = torch.empty_like(x)
        gres = torch.empty_like(x)
        ws = _ws(L.icl_norm_ws_bytes(n, c, s), x)
        _lib.check(L.icl_norm_res_bwd(_ptr(gy), _ptr(x), _ptr(res), _ptr(mean), _ptr(rstd), None, None, _ptr(gx), _ptr(gres), None,
                                      None, n, c, s, 0, 1, ctx.act, _ptr(ws), _stream(x)), "norm_res_bwd")
        return gx, gres, None, None, None


def instance_norm_add_act(x: torch.Tensor, res: torch.Tensor, act: int = 2, eps: float = 1e-5) -> torch.Tensor:
    return _InstanceNormAddAct.apply(x, res, int(act), eps)


def conv3d_instance_norm_add_act(x: torch.Tensor, weight: torch.Tensor, res: torch.Tensor, act: int = 2, eps: float = 1e-5) -> torch.Tensor:
    """act(InstanceNorm3d(Conv3d(x)) + res): the second half of MONAI's UnetResBlock (swinunetr_icl.py:128-229), the convolution's
    epilogue handing the normalisation its statistics where it can (see conv3d_instance_norm_act)."""
    if (weight.shape[2] == 3 and x.shape[2] * x.shape[3] * x.shape[4] <= SMALL_CONV_MAX_VOXELS
            and weight.numel() >= SMALL_CONV_MIN_WEIGHTS and not _f6_split_ok(x, weight)):
        return _InstanceNormAddAct.apply(_conv3d_tiny_volume(x, weight, None), res, int(act), eps)
    y, stats = _Conv3d.apply(x, weight, None, True, True)
    return _InstanceNormAddAct.apply(y, res, int(act), eps, stats if stats.numel() else None)


def batch_norm_relu(x, gamma, beta, running_mean, running_var, training: bool, relu: bool = True,
                    eps: float = 1e-5, momentum: float = 0.1) -> torch.Tensor:
    """BatchNorm3d (batch statistics + running-stat update in training, running stats in eval) [+ ReLU]."""
    return _NormAct.apply(x, gamma, beta, running_mean, running_var, 1, bool(training), int(relu), eps, momentum)


def batch_norm_act(x, gamma, beta, running_mean, running_var, training: bool, act: int,
                   eps: float = 1e-5, momentum: float = 0.1) -> torch.Tensor:
    """BatchNorm{2,3}d fused with act: 0 none, 1 ReLU, 2 LeakyReLU(0.01) (2-D ConvBlock, networks/unet_icl.py:46-54).
    A 4-D input [N,C,H,W] is normalised over (N,H,W) exactly like a 5-D one over (N,D,H,W)."""
    return _NormAct.apply(x, gamma, beta, running_mean, running_var, 1, bool(training), int(act), eps, momentum)


def _embed_2d_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Co,Ci,3,3] -> [Co,Ci,3,3,3] with the 2-D stencil in the dz = 1 plane (autograd slices the gradient back).
    A 3x3 convolution of a D = 1 volume with this kernel and padding 1 is exactly the 2-D convolution."""
    co, ci, kh, kw = weight.shape
    if kh == 1:
        return weight.unsqueeze(2)
    w3 = weight.new_zeros((co, ci, 3, kh, kw))
    w3[:, :, 1] = weight
    return w3


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, zero_bias_grad: bool = False) -> torch.Tensor:
    """nn.Conv2d k=3 pad 1 / k=1 on [N,C,H,W] (networks/unet_icl.py:46-52,82,173) through the 3-D MFMA kernels."""
    return conv3d(x.unsqueeze(2), _embed_2d_weight(weight), bias, zero_bias_grad).squeeze(2)


def depthwise_conv2d(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """Depthwise 3x3 (SeparableConv2d.depthwise, networks/unet_icl.py:100-103) on the 3-D stencil kernel."""
    return depthwise_conv3d(x.unsqueeze(2), _embed_2d_weight(weight)).squeeze(2)


# --------------------------------------------------------------------------------------
# MaxPool3d(2) — networks/unet_3D_icl.py:41-53
# --------------------------------------------------------------------------------------

class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pd):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        n, c, d, h, w = x.shape
        assert d % pd == 0 and h % 2 == 0 and w % 2 == 0, "max-pool kernels need even extents"
        y = torch.empty((n, c, d // pd, h // 2, w // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        _lib.check(L.icl_maxpool2_fwd(_ptr(x), _ptr(y), _ptr(idx), n * c, d // pd, h // 2, w // 2, pd, _stream(x)), "maxpool2_fwd")
        ctx.save_for_backward(idx)
        ctx.pd = pd
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        n, c, d, h, w = gy.shape
        gx = torch.empty((n, c, d * ctx.pd, h * 2, w * 2), dtype=torch.float32, device=gy.device)
        _lib.check(L.icl_maxpool2_bwd(_ptr(gy), _ptr(idx), _ptr(gx), n * c, d, h, w, ctx.pd, _stream(gy)), "maxpool2_bwd")
        return gx, None


class _SkipAndPool(torch.autograd.Function):
    """(x, maxpool2(x)) for a tensor that feeds a skip connection AND the pooling of the next level: the backward adds the two
    gradients inside the pooling backward's pass (icl_maxpool2_bwd_add) instead of a pooling backward plus an autograd `add` over the
    full-resolution tensor.  The skip gradient may be a channel slice of a concat gradient (batch-strided view, _UpCat.backward)."""

    @staticmethod
    def forward(ctx, x):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        n, c, d, h, w = x.shape
        assert d % 2 == 0 and h % 2 == 0 and w % 2 == 0, "max-pool kernels need even extents"
        y = torch.empty((n, c, d // 2, h // 2, w // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        _lib.check(L.icl_maxpool2_fwd(_ptr(x), _ptr(y), _ptr(idx), n * c, d // 2, h // 2, w // 2, 2, _stream(x)), "maxpool2_fwd")
        ctx.save_for_backward(idx)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gskip, gy):
        (idx,) = ctx.saved_tensors
        L = _lib.lib()
        n, c, d, h, w = idx.shape
        if gy is None:
            return gskip
        gy = gy.contiguous()
        gx = torch.empty((n, c, d * 2, h * 2, w * 2), dtype=torch.float32, device=gy.device)
        plane = 8 * d * h * w
        ok = (gskip is not None and gskip.dtype == torch.float32 and tuple(gskip.shape) == tuple(gx.shape)
              and gskip.stride()[1:] == (plane, 4 * h * w, 2 * w, 1) and gskip.stride(0) % 2 == 0 and gskip.data_ptr() % 8 == 0)
        if ok:
            _lib.check(L.icl_maxpool2_bwd_add(_ptr(gy), _ptr(idx), _vp(gskip.data_ptr()), _ptr(gx), n, c, d, h, w, 2, gskip.stride(0),
                                              _stream(gy)), "maxpool2_bwd_add")
            return gx
        _lib.check(L.icl_maxpool2_bwd(_ptr(gy), _ptr(idx), _ptr(gx), n * c, d, h, w, 2, _stream(gy)), "maxpool2_bwd")
        return gx if gskip is None else gx + gskip


class _SkipAndPoolLazy(torch.autograd.Function):
    """_SkipAndPool on a deferred normalisation: the pooling reads the raw tensor and normalises on load; the skip output stays deferred."""

    @staticmethod
    def forward(ctx, t, ss):
        L = _lib.lib()
        n, c, d, h, w = t.shape
        y = torch.empty((n, c, d // 2, h // 2, w // 2), dtype=torch.float32, device=t.device)
        idx = torch.empty(y.shape, dtype=torch.uint8, device=t.device)
        _lib.check(L.icl_maxpool2_fwd_norm(_ptr(t), _ptr(ss), _ptr(y), _ptr(idx), n * c, d // 2, h // 2, w // 2, 2, _stream(t)), "maxpool2_fwd_norm")
        ctx.save_for_backward(idx)
        return t.view_as(t), y

    @staticmethod
    def backward(ctx, gskip, gy):
        return _SkipAndPool.backward(ctx, gskip, gy), None


def skip_and_pool(x):
    """Returns (x, MaxPool3d(2)(x)); use the FIRST output for the skip connection so that the backward can fuse the two gradients.
    ``x`` may be a ``LazyAct``: the skip output then is one too."""
    if isinstance(x, LazyAct):
        d, h, w = x.t.shape[2:]
        if d % 2 == 0 and h % 2 == 0 and w % 2 == 0 and x.t.is_contiguous():
            skip, pooled = _SkipAndPoolLazy.apply(x.t, x.ss)
            return LazyAct(skip, x.ss), pooled
        x = x.materialize()
    return _SkipAndPool.apply(x)


def max_pool3d_2(x: torch.Tensor) -> torch.Tensor:
    return _MaxPool2.apply(x, 2)


def max_pool2d_2(x: torch.Tensor) -> torch.Tensor:
    """nn.MaxPool2d(2) on [N,C,H,W] (networks/unet_icl.py:64), run as a D = 1 volume."""
    return _MaxPool2.apply(x.unsqueeze(2), 1).squeeze(2)


# --------------------------------------------------------------------------------------
# trilinear resize (align_corners=False) — networks/utils.py:264, utils/losses.py:263,292
# --------------------------------------------------------------------------------------

class _Trilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        n, c, di, hi, wi = x.shape
        do, ho, wo = size
        y = torch.empty((n, c, do, ho, wo), dtype=torch.float32, device=x.device)
        _lib.check(L.icl_trilinear_fwd(_ptr(x), _ptr(y), n, c, di, hi, wi, do, ho, wo, c * do * ho * wo, align, _stream(x)),
                   "trilinear_fwd")
        ctx.in_shape = (n, c, di, hi, wi)
        ctx.align = align
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        gy = gy.contiguous()
        n, c, di, hi, wi = ctx.in_shape
        do, ho, wo = gy.shape[2:]
        gx = torch.empty(ctx.in_shape, dtype=torch.float32, device=gy.device)
        ws = _ws(L.icl_trilinear_bwd_ws_bytes(n, c, di, hi, wi, do, ho, wo), gy)
        _lib.check(L.icl_trilinear_bwd(_ptr(gy), _ptr(gx), _ptr(ws), n, c, di, hi, wi, do, ho, wo, c * do * ho * wo, ctx.align,
                                       _stream(gy)), "trilinear_bwd")
        return gx, None, None


def trilinear_resize(x: torch.Tensor, size: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    return _Trilinear.apply(x, tuple(int(s) for s in size), int(bool(align_corners)))


def bilinear_resize(x: torch.Tensor, size: Sequence[int], align_corners: bool = False) -> torch.Tensor:
    """F.interpolate(mode='bilinear') / nn.Upsample(bilinear) on [N,C,H,W], run as a D = 1 volume."""
    return _Trilinear.apply(x.unsqueeze(2), (1, int(size[0]), int(size[1])), int(bool(align_corners))).squeeze(2)


class _UpCat(torch.autograd.Function):
    """cat([skip, upsample2x(deep)], 1) without materialising the upsampled tensor:
    the resize kernel writes straight into the channel slice of the concat buffer
    (UnetUp3_CT.forward, networks/utils.py:271-276)."""

    @staticmethod
    def forward(ctx, skip, deep):
        _require(skip, deep)
        L = _lib.lib()
        skip = skip.contiguous()
        deep = deep.contiguous()
        n, cs, d, h, w = skip.shape
        cd = deep.shape[1]
        assert deep.shape[0] == n and tuple(deep.shape[2:]) == (d // 2, h // 2, w // 2)
        s = d * h * w
        out = torch.empty((n, cs + cd, d, h, w), dtype=torch.float32, device=skip.device)
        _lib.check(L.icl_copy_rows(_ptr(skip), _ptr(out), n, cs * s, cs * s, (cs + cd) * s, _stream(skip)), "copy_rows")
        up_view = out[:, cs:]
        _lib.check(L.icl_trilinear_fwd(_ptr(deep), _vp(up_view.data_ptr()), n, cd, d // 2, h // 2, w // 2, d, h, w,
                                       (cs + cd) * s, 0, _stream(skip)), "trilinear_fwd")
        ctx.dims = (n, cs, cd, d, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        g = g.contiguous()
        n, cs, cd, d, h, w = ctx.dims
        s = d * h * w
        gskip = gdeep = None
        if ctx.needs_input_grad[0]:
            # a VIEW of the concat gradient, not a copy: the skip tensor also feeds the max-pool, so autograd adds this to the pooling
            # gradient right away (one strided read instead of a copy pass plus a read); a sole consumer makes it contiguous itself
            gskip = g[:, :cs]
        if ctx.needs_input_grad[1]:
            gdeep = torch.empty((n, cd, d // 2, h // 2, w // 2), dtype=torch.float32, device=g.device)
            gv = g[:, cs:]
            ws = _ws(L.icl_trilinear_bwd_ws_bytes(n, cd, d // 2, h // 2, w // 2, d, h, w), g)
            _lib.check(L.icl_trilinear_bwd(_vp(gv.data_ptr()), _ptr(gdeep), _ptr(ws), n, cd, d // 2, h // 2, w // 2, d, h, w,
                                           (cs + cd) * s, 0, _stream(g)), "trilinear_bwd")
        return gskip, gdeep


class _UpCatLazy(torch.autograd.Function):
    """_UpCat with either source given as a deferred normalisation (raw tensor + (scale, shift), applied while it is copied / up-sampled
    into the concat buffer)."""

    @staticmethod
    def forward(ctx, skip, skip_ss, deep, deep_ss):
        L = _lib.lib()
        n, cs, d, h, w = skip.shape
        cd = deep.shape[1]
        out = torch.empty((n, cs + cd, d, h, w), dtype=torch.float32, device=skip.device)
        _lib.check(L.icl_upsample2x_concat_norm(_ptr(skip), _ptr(skip_ss), _ptr(deep), _ptr(deep_ss), _ptr(out), n, cs, cd, d, h, w,
                                                _stream(skip)), "upsample2x_concat_norm")
        ctx.dims = (n, cs, cd, d, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        ctx2 = ctx
        needs = ctx.needs_input_grad

        class _View:      # _UpCat.backward reads needs_input_grad[0] / [1] of (skip, deep)
            dims = ctx2.dims
            needs_input_grad = (needs[0], needs[2])
        gskip, gdeep = _UpCat.backward(_View, g)
        return gskip, None, gdeep, None


def upsample2x_concat(skip, deep) -> torch.Tensor:
    """cat([skip, upsample2x(deep)], 1); either argument may be a ``LazyAct`` (its normalisation is applied while it is read)."""
    if isinstance(skip, LazyAct) or isinstance(deep, LazyAct):
        st, dt = (skip.t if isinstance(skip, LazyAct) else skip), (deep.t if isinstance(deep, LazyAct) else deep)
        n, cs, d, h, w = st.shape
        ok = (d % 2 == 0 and h % 2 == 0 and w % 4 == 0 and tuple(dt.shape[2:]) == (d // 2, h // 2, w // 2) and dt.shape[0] == n
              and st.is_contiguous() and dt.is_contiguous() and st.dtype == torch.float32 and (st.is_cuda or _lib.host_pointers_ok()))
        if ok:
            return _UpCatLazy.apply(st, skip.ss if isinstance(skip, LazyAct) else None, dt, deep.ss if isinstance(deep, LazyAct) else None)
        skip, deep = materialized(skip), materialized(deep)
    return _UpCat.apply(skip, deep)


# --------------------------------------------------------------------------------------
# depthwise Conv3d 3^3 (groups = C, no bias) — SeparableConv3d.depthwise, unet_3D_icl.py:320-323
# --------------------------------------------------------------------------------------

class _DWConv3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        _require(x, weight)
        L = _lib.lib()
        x = x.contiguous()
        weight = weight.contiguous()
        n, c, d, h, w = x.shape
        assert tuple(weight.shape) == (c, 1, 3, 3, 3)
        y = torch.empty_like(x)
        _lib.check(L.icl_dwconv3_fwd(_ptr(x), _ptr(weight), _ptr(y), n, c, d, h, w, 0, _stream(x)), "dwconv3_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        n, c, d, h, w = x.shape
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.icl_dwconv3_fwd(_ptr(gy), _ptr(weight), _ptr(gx), n, c, d, h, w, 1, _stream(x)), "dwconv3_dgrad")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            ws = _ws(L.icl_dwconv3_wgrad_ws_bytes(n, c, d, h, w), x)
            _lib.check(L.icl_dwconv3_wgrad(_ptr(x), _ptr(gy), _ptr(gw), _ptr(ws), n, c, d, h, w, _stream(x)), "dwconv3_wgrad")
        return gx, gw


def depthwise_conv3d(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return _DWConv3.apply(x, weight)


# --------------------------------------------------------------------------------------
# Dropout — nn.Dropout(p=0.3), unet_3D_icl.py:67-68
# --------------------------------------------------------------------------------------

def _mix32(x: int) -> int:
    """The 32-bit finaliser of csrc/kernels/misc.h (mix32), on the host."""
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


class StepRNG:
    """Device-resident step counter for dropout under hipGraph replay: a captured launch bakes its scalar arguments in,
    so the per-step variation of the mask comes from this counter (incremented on the device at the end of every step and
    hashed into the seed by the kernel) and the per-call variation from a host-side call index that is identical in every
    replay.  The stream is keyed by ``torch.initial_seed()`` (so ``torch.manual_seed`` selects it) and by the data-parallel rank
    (every rank draws its own masks, as W independent reference processes would)."""
    tensor: Optional[torch.Tensor] = None
    calls = 0
    base = 0

    @classmethod
    def enable(cls, device, base_seed: Optional[int] = None, rank: Optional[int] = None):
        cls.tensor = torch.zeros(1, dtype=torch.int32, device=device)
        cls.calls = 0
        if base_seed is None:
            base_seed = torch.initial_seed()
        if rank is None:
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        cls.base = _mix32(_mix32(base_seed & 0xFFFFFFFF) ^ _mix32((base_seed >> 32) + 0x9E3779B1) ^ _mix32(rank + 0x85EBCA77))

    @classmethod
    def begin_step(cls):
        cls.calls = 0

    @classmethod
    def end_step(cls):
        if cls.tensor is not None:
            cls.tensor += 1

    @classmethod
    def next_seed(cls) -> int:
        """Seed of the next stochastic call of the step: a hash of (stream key, call index)."""
        seed = _mix32(cls.base ^ _mix32(cls.calls + 0x2545F491))
        cls.calls += 1
        return seed


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_dev, group=1):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(L.icl_drop_path(_ptr(x), _ptr(y), x.numel(), group, seed, p, _ptr(seed_dev), _stream(x)), "dropout")
        ctx.cfg = (p, seed, group)
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        p, seed, group = ctx.cfg
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        _lib.check(L.icl_drop_path(_ptr(gy), _ptr(gx), gy.numel(), group, seed, p, _ptr(ctx.seed_dev), _stream(gy)), "dropout_bwd")
        return gx, None, None, None, None


class _DropPathAdd(torch.autograd.Function):
    """y = res + drop_path(x) in one kernel; ``res is x`` gives x + drop_path(x).  d/dres is the identity (the incoming gradient is
    handed on as is), d/dx the same per-sample mask applied to the gradient."""

    @staticmethod
    def forward(ctx, x, res, p, seed, seed_dev):
        _require(x, res)
        L = _lib.lib()
        x = x.contiguous()
        same = res is x or (res.data_ptr() == x.data_ptr() and res.shape == x.shape and res.is_contiguous())
        res = x if same else res.contiguous()
        assert res.shape == x.shape
        group = x.numel() // x.shape[0]
        y = torch.empty_like(x)
        _lib.check(L.icl_drop_path_add(_ptr(x), _ptr(res), _ptr(y), x.numel(), group, seed, p, _ptr(seed_dev), _stream(x)), "drop_path_add")
        ctx.cfg = (p, seed, group)
        ctx.seed_dev = seed_dev
        ctx.same = same
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        p, seed, group = ctx.cfg
        gx = None
        if ctx.same and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # x + drop_path(x) (the doubling of Class_Decoder, unet_3D_icl.py:264,266): both gradients belong to the same tensor —
            # gy + mask * gy in the forward kernel's one pass, handed back once, instead of a mask kernel plus an autograd add
            gy = gy.contiguous()
            gx = torch.empty_like(gy)
            _lib.check(L.icl_drop_path_add(_ptr(gy), _ptr(gy), _ptr(gx), gy.numel(), group, seed, p, _ptr(ctx.seed_dev), _stream(gy)),
                       "drop_path_add_bwd")
            return gx, None, None, None, None
        if ctx.needs_input_grad[0]:
            gy = gy.contiguous()
            gx = torch.empty_like(gy)
            _lib.check(L.icl_drop_path(_ptr(gy), _ptr(gx), gy.numel(), group, seed, p, _ptr(ctx.seed_dev), _stream(gy)), "drop_path_bwd")
        return gx, (gy if ctx.needs_input_grad[1] else None), None, None, None


def _step_seed(x, seed):
    seed_dev = None
    if seed is None:
        if StepRNG.tensor is not None and StepRNG.tensor.device == x.device:
            seed = StepRNG.next_seed()
            seed_dev = StepRNG.tensor
        else:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return int(seed) & 0xFFFFFFFF, seed_dev


def drop_path_add(res: torch.Tensor, x: torch.Tensor, p: float, training: bool = True, seed: Optional[int] = None) -> torch.Tensor:
    """res + DropPath(p)(x) — the residual update of a transformer block — as one kernel in training (a plain add otherwise).
    Same per-sample mask and seeding as ``drop_path``."""
    if p == 0.0 or not training:
        return res + x
    seed, seed_dev = _step_seed(x, seed)
    return _DropPathAdd.apply(x, res, float(p), seed, seed_dev)


def dropout(x: torch.Tensor, p: float, seed: Optional[int] = None) -> torch.Tensor:
    """Training-mode dropout.  Eager: the seed advances with torch's CPU generator (reproducible under manual_seed).
    With StepRNG enabled (graph capture): seed = call index, varied per step by the device-resident counter."""
    seed_dev = None
    if seed is None:
        if StepRNG.tensor is not None and StepRNG.tensor.device == x.device:
            seed = StepRNG.next_seed()
            seed_dev = StepRNG.tensor
        else:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _Dropout.apply(x, float(p), int(seed) & 0xFFFFFFFF, seed_dev)


class _DropoutConv1x1(torch.autograd.Function):
    """conv1x1(dropout(x)) with the mask applied on the fly: forward (mask on the convolution's input loads), input gradient (mask on its
    stores) and weight gradient (mask on the x operand) each run as ONE pass — the dropped tensor never exists."""

    @staticmethod
    def forward(ctx, x, weight, bias, p, seed, seed_dev):
        _require(x, weight, bias)
        L = _lib.lib()
        x = x.contiguous()
        n, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        s = x.numel() // (n * cin)
        y = torch.empty((n, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        _lib.check(L.icl_conv1x1_dropout(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), n, cin, cout, s, cin, 1, 1, seed, p, _ptr(seed_dev),
                                         _stream(x)), "conv1x1_dropout")
        ctx.save_for_backward(x, weight)
        ctx.cfg = (p, seed, bias is not None)
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        p, seed, has_bias = ctx.cfg
        L = _lib.lib()
        gy = gy.contiguous()
        n, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        s = x.numel() // (n * cin)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.icl_conv1x1_dropout(_ptr(gy), _ptr(weight), None, _ptr(gx), n, cout, cin, s, 1, cin, 2, seed, p, _ptr(ctx.seed_dev),
                                             _stream(x)), "conv1x1_dropout dgrad")
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(weight)
            gb = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
            ws = _ws(L.icl_conv1x1_wgrad_ws_bytes(n, s, cin, cout), x)
            with _timed("conv1x1_wgrad_kernel", 2.0 * cin * cout * s * n, 4.0 * n * s * (cin + cout), x):
                _lib.check(L.icl_conv1x1_wgrad_dropout(_ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), n, cin, cout, s, cout * s, seed, p,
                                                       _ptr(ctx.seed_dev), _stream(x)), "conv1x1_wgrad_dropout")
        return gx, gw, gb, None, None, None


class _DropoutConv1x1Lazy(torch.autograd.Function):
    """_DropoutConv1x1 on a deferred normalisation: y = conv1x1(dropout(relu(fma(t, scale, shift)), p)) — InstanceNorm + ReLU, the dropout
    mask and the convolution in one pass over the raw tensor (p = 0: no mask); the weight gradient applies both on its x operand."""

    @staticmethod
    def forward(ctx, t, ss, weight, bias, p, seed, seed_dev):
        L = _lib.lib()
        n, cin = t.shape[0], t.shape[1]
        cout = weight.shape[0]
        s = t.numel() // (n * cin)
        y = torch.empty((n, cout) + tuple(t.shape[2:]), dtype=torch.float32, device=t.device)
        _lib.check(L.icl_conv1x1_dropout_norm(_ptr(t), _ptr(ss), _ptr(weight), _ptr(bias), _ptr(y), n, cin, cout, s, cin, 1, seed, p,
                                              _ptr(seed_dev), _stream(t)), "conv1x1_dropout_norm")
        ctx.save_for_backward(t, ss, weight)
        ctx.cfg = (p, seed, bias is not None)
        ctx.seed_dev = seed_dev
        return y

    @staticmethod
    def backward(ctx, gy):
        t, ss, weight = ctx.saved_tensors
        p, seed, has_bias = ctx.cfg
        L = _lib.lib()
        gy = gy.contiguous()
        n, cin = t.shape[0], t.shape[1]
        cout = weight.shape[0]
        s = t.numel() // (n * cin)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(t)      # the gradient of the NORMALISED activation (what `t` stands for in the autograd graph)
            if p > 0.0:
                _lib.check(L.icl_conv1x1_dropout(_ptr(gy), _ptr(weight), None, _ptr(gx), n, cout, cin, s, 1, cin, 2, seed, p, _ptr(ctx.seed_dev),
                                                 _stream(t)), "conv1x1_dropout dgrad")
            else:
                _lib.check(L.icl_conv1x1_small(_ptr(gy), _ptr(weight), None, _ptr(gx), n, cout, cin, s, 1, cin, _stream(t)), "conv1x1_small dgrad")
        if ctx.needs_input_grad[2] or (has_bias and ctx.needs_input_grad[3]):
            gw = torch.empty_like(weight)
            gb = torch.empty(cout, dtype=torch.float32, device=t.device) if has_bias else None
            ws = _ws(L.icl_conv1x1_wgrad_ws_bytes(n, s, cin, cout), t)
            with _timed("conv1x1_wgrad_kernel", 2.0 * cin * cout * s * n, 4.0 * n * s * (cin + cout), t):
                _lib.check(L.icl_conv1x1_wgrad_dropout_norm(_ptr(t), _ptr(ss), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws), n, cin, cout, s, cout * s,
                                                            seed, p, _ptr(ctx.seed_dev), _stream(t)), "conv1x1_wgrad_dropout_norm")
        return gx, None, gw, gb, None, None, None


def conv1x1_lazy(x, weight: torch.Tensor, bias: Optional[torch.Tensor], p: float = 0.0, seed: Optional[int] = None) -> torch.Tensor:
    """``conv3d(dropout(relu(norm(y)), p), weight, bias)`` for a 1x1x1 convolution whose input is a ``LazyAct`` — `final(dropout2(up1))`
    (unet_3D_icl.py:116-117) in one pass over the RAW output of up1's last convolution; p = 0: `final(up1)`.  Shapes the fused kernels do
    not take are materialised and run the ordinary operators."""
    if not isinstance(x, LazyAct):
        return dropout_conv1x1(x, weight, bias, p, seed) if p > 0.0 else conv3d(x, weight, bias)
    t = x.t
    n, cin = t.shape[0], t.shape[1]
    s = t.numel() // max(n * cin, 1)
    fusable = ((t.is_cuda or _lib.host_pointers_ok()) and weight.shape[2:].numel() == 1 and cin <= 16 and weight.shape[0] <= 16
               and s % 4 == 0 and n * s >= 65536 and t.is_contiguous())
    if not fusable:
        a = x.materialize()
        return dropout_conv1x1(a, weight, bias, p, seed) if p > 0.0 else conv3d(a, weight, bias)
    seed_dev = None
    if p > 0.0 and seed is None:
        if StepRNG.tensor is not None and StepRNG.tensor.device == t.device:
            seed = StepRNG.next_seed()
            seed_dev = StepRNG.tensor
        else:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _DropoutConv1x1Lazy.apply(t, x.ss, weight.contiguous(), bias, float(p), int(seed or 0) & 0xFFFFFFFF, seed_dev)


def dropout_conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], p: float, seed: Optional[int] = None) -> torch.Tensor:
    """``conv3d(dropout(x, p), weight, bias)`` for a 1x1x1 (or 1x1) convolution of <= 16 channels on a big volume — `final(dropout2(up1))` of
    the backbones (unet_3D_icl.py:67-68,117) — without materialising the dropped tensor.  Same seed protocol and the same mask as
    ``dropout`` (so the results are those of the two-op form, bit for bit); other shapes take the two-op form."""
    n, cin = x.shape[0], x.shape[1]
    s = x.numel() // max(n * cin, 1)
    fusable = (x.is_cuda or _lib.host_pointers_ok()) and weight.shape[2:].numel() == 1 and cin <= 16 and weight.shape[0] <= 16 \
        and s % 4 == 0 and n * s >= 65536 and torch.is_grad_enabled()
    if not fusable:
        return conv3d(dropout(x, p, seed), weight, bias)
    seed_dev = None
    if seed is None:
        if StepRNG.tensor is not None and StepRNG.tensor.device == x.device:
            seed = StepRNG.next_seed()
            seed_dev = StepRNG.tensor
        else:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _DropoutConv1x1.apply(x, weight, bias, float(p), int(seed) & 0xFFFFFFFF, seed_dev)


def drop_path(x: torch.Tensor, p: float, seed: Optional[int] = None) -> torch.Tensor:
    """Per-sample stochastic depth (timm / MONAI DropPath): each sample of the batch keeps its branch with probability 1-p (scaled
    by 1/(1-p)) or drops it entirely — one kernel instead of empty + bernoulli_ + div + mul; same seeding scheme as ``dropout``."""
    seed_dev = None
    if seed is None:
        if StepRNG.tensor is not None and StepRNG.tensor.device == x.device:
            seed = StepRNG.next_seed()
            seed_dev = StepRNG.tensor
        else:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _Dropout.apply(x, float(p), int(seed) & 0xFFFFFFFF, seed_dev, x.numel() // x.shape[0])


# --------------------------------------------------------------------------------------
# Aligner token ops (unet_3D_icl.py:244-315): LayerNorm, GELU and the prototype attention are HIP kernels
# (csrc/kernels/token.h); the Linear layers (incl. the 13,824^2 mlp2) run on csrc/kernels/gemm.h — the skinny mlp2 products
# stream the weight matrix once per launch (5.4-5.8 TB/s, profiles/), the others use the LDS-tiled fp32 MFMA product.
# --------------------------------------------------------------------------------------

LINEAR_WGRAD_MIN_ROWS = 2048
LINEAR_BWD_ONE_LAUNCH = True


def linear_forward_raw(x2: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0) -> torch.Tensor:
    """act(x2 W^T + b) for row-major x2 [rows, in], W [out, in] (csrc/kernels/gemm.h: weight streaming for <= 32 rows and
    >= 2^20 weights, LDS-tiled fp32 MFMA product otherwise)."""
    _require(x2, weight, bias)
    L = _lib.lib()
    rows, i = x2.shape
    o = weight.shape[0]
    y = torch.empty((rows, o), dtype=torch.float32, device=x2.device)
    need = L.icl_linear_ws_bytes(rows, i, o, 0)
    ws = _ws(need, x2) if need else None
    big = rows <= 32 and i * o >= (1 << 26)      # the 13,824^2 matrices: reported separately (bench.py roofline.mlp2_weight_stream)
    with _timed("linear_stream_fwd" if big else "linear_fwd", 2.0 * rows * i * o, 4.0 * (i * o if big else rows * (i + o) + i * o), x2):
        _lib.check(L.icl_linear_fwd(_ptr(x2), _ptr(weight), _ptr(bias), _ptr(y), _ptr(ws), rows, i, o, act, _stream(x2)), "linear_fwd")
    return y


def linear_dgrad_raw(g2: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """g2 W for row-major g2 [rows, out], W [out, in] -> [rows, in]."""
    _require(g2, weight)
    L = _lib.lib()
    rows, o = g2.shape
    i = weight.shape[1]
    gx = torch.empty((rows, i), dtype=torch.float32, device=g2.device)
    need = L.icl_linear_ws_bytes(rows, i, o, 1)
    ws = _ws(need, g2) if need else None
    big = rows <= 32 and i * o >= (1 << 26)
    with _timed("linear_stream_dgrad" if big else "linear_dgrad", 2.0 * rows * i * o, 4.0 * (i * o if big else rows * (i + o) + i * o), g2):
        _lib.check(L.icl_linear_dgrad(_ptr(g2), _ptr(weight), _ptr(gx), _ptr(ws), rows, i, o, _stream(g2)), "linear_dgrad")
    return gx


def gemm(a: torch.Tensor, b: torch.Tensor, m: int, n: int, k: int, lda: int, ldb: int, a_kcontig: bool, b_kcontig: bool,
         out: Optional[torch.Tensor] = None, ldc: Optional[int] = None, bias: Optional[torch.Tensor] = None, act: int = 0, batch: int = 1,
         a_bstride: int = 0, b_bstride: int = 0, c_bstride: int = 0) -> torch.Tensor:
    """C[b] = act(A[b] B[b] + bias) through icl_gemm (include/icl_hip.h): operands are described by their pitches, so transposed
    views are consumed in place (no transpose copies)."""
    _require(a, b, bias)
    L = _lib.lib()
    if out is None:
        out = torch.empty((batch, m, n) if batch > 1 else (m, n), dtype=torch.float32, device=a.device)
        ldc, c_bstride = n, m * n
    need = L.icl_gemm_ws_bytes(m, n, k, batch)
    ws = _ws(need, a) if need else None
    with _timed("gemm", 2.0 * batch * m * n * k, 4.0 * batch * (m * k + k * n + m * n), a):
        _lib.check(L.icl_gemm(_ptr(a), _ptr(b), _ptr(out), _ptr(bias), _ptr(ws), m, n, k, lda, ldb, ldc, int(a_kcontig), int(b_kcontig),
                              act, batch, a_bstride, b_bstride, c_bstride, _stream(a)), "gemm")
    return out


def _tall_atb(a: torch.Tensor, b: torch.Tensor, want_colsum: bool):
    """a^T b for row-major a [rows, A], b [rows, B] (-> [A, B]) and, optionally, the column sums of a (-> [A]).
    rows >= LINEAR_WGRAD_MIN_ROWS: csrc/kernels/linear_wgrad.h (rows split over many waves); otherwise the tiled product of
    csrc/kernels/gemm.h with both operands k-strided."""
    rows, A = a.shape
    B = b.shape[1]
    _require(a, b)
    L = _lib.lib()
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((A, B), dtype=torch.float32, device=a.device)
    if rows < LINEAR_WGRAD_MIN_ROWS:
        need = L.icl_linear_ws_bytes(rows, B, A, 2)
        ws = _ws(need, a) if need else None
        with _timed("linear_wgrad_small", 2.0 * rows * A * B, 4.0 * (rows * (A + B) + A * B), a):
            _lib.check(L.icl_linear_wgrad_small(_ptr(a), _ptr(b), _ptr(out), _ptr(ws), rows, B, A, _stream(a)), "linear_wgrad_small")
        return out, (a.sum(0) if want_colsum else None)
    colsum = torch.empty(A, dtype=torch.float32, device=a.device) if want_colsum else None
    ws = _ws(L.icl_linear_wgrad_ws_bytes(rows, A, B), a)
    with _timed("linear_wgrad_kernel", 2.0 * rows * A * B, 4.0 * rows * (A + B), a):
        _lib.check(L.icl_linear_wgrad(_ptr(a), _ptr(b), _ptr(out), _ptr(colsum), _ptr(ws), rows, A, B, _stream(a)), "linear_wgrad")
    return out, colsum


class DeferredBiasGrads:
    """While open (ICLTrainer brackets forward/backward with ``begin()`` / ``flush()``), the Linear layers that know their module
    (``owner``) and see few rows do not reduce their bias gradient inside backward: they queue (bias, dY) and ``flush()`` computes
    all column sums in ONE launch and assigns (or accumulates into) ``bias.grad``.  Closed, every bias gradient is reduced on
    the spot as usual."""
    pending = None
    producers = None      # the streams the queued tensors were produced on (the aligner lanes): flush() orders its stream after them

    @classmethod
    def begin(cls):
        cls.pending = []
        cls.producers = set()

    @classmethod
    def note_producer(cls, t):
        """The queued tensor is written by work on the CURRENT stream; flush() runs on the step's stream.  Autograd joins the streams of
        the AccumulateGrad nodes that ran — a lane whose parameters are all deferred or adopted is none of them — so the flush orders
        itself after every producing stream explicitly (round 5)."""
        if cls.producers is not None and t.is_cuda:
            cls.producers.add(torch.cuda.current_stream(t.device))

    @classmethod
    def defer(cls, bias, g2) -> bool:
        if cls.pending is None or not isinstance(bias, torch.nn.Parameter) or g2.shape[0] >= LINEAR_WGRAD_MIN_ROWS:
            return False
        g2 = g2.contiguous()
        cls.pending.append((bias, g2))
        cls.note_producer(g2)
        BackwardEnd.arm()
        return True

    @classmethod
    def flush(cls, keep_open: bool = False):
        """``keep_open``: reduce what is queued and stay open for the next backward pass of the same step scope (BackwardEnd)."""
        items, cls.pending = cls.pending, ([] if keep_open else None)
        producers, cls.producers = cls.producers, (set() if keep_open else None)
        if not items:
            return
        if producers:
            here = torch.cuda.current_stream(items[0][1].device)
            for s in producers:
                if s != here:
                    here.wait_stream(s)
            for _, g in items:
                g.record_stream(here)
        L = _lib.lib()
        n = len(items)
        sizes = [g.shape[1] for _, g in items]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=items[0][1].device)
        outs, off = [], 0
        for c in sizes:
            outs.append(flat[off:off + c])
            off += c
        arr = ctypes.c_void_p * n
        iarr = ctypes.c_int32 * n
        _lib.check(L.icl_colsum_multi(arr(*[g.data_ptr() for _, g in items]), arr(*[o.data_ptr() for o in outs]),
                                      iarr(*[g.shape[0] for _, g in items]), iarr(*sizes), n, _stream(flat)), "colsum_multi")
        for (bias, _), o in zip(items, outs):
            if bias.grad is None:
                bias.grad = o
            else:
                bias.grad = bias.grad + o      # a bias used by several calls of the step


class _Linear(torch.autograd.Function):
    """y = x W^T + b.  Forward and dx: csrc/kernels/gemm.h (fp32 MFMA; the weight is streamed once for skinny inputs); dW/db of TALL
    inputs (>= 2048 rows: the Swin token grids) use csrc/kernels/linear_wgrad.h, of short ones the k-strided tiled product."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner=None, lane_wgrad=False):
        weight = weight.contiguous()
        x2 = x.reshape(-1, weight.shape[1]).contiguous()
        ctx.save_for_backward(x2, weight)
        ctx.lane_wgrad = bool(lane_wgrad)
        # the gradient is returned for `weight` as passed (a view of the convolution weight in _conv3d_tiny_volume): adoption is
        # decided on the Parameter behind it
        ctx.lane_param = owner.weight if owner is not None and getattr(owner, "weight", None) is not None else (
            weight._base if weight._base is not None else weight)
        if lane_wgrad:
            WgradLane.note_use(ctx.lane_param)
        ctx.has_bias = bias is not None
        # the Parameter behind the bias (for DeferredBiasGrads: all small bias gradients of a step reduced in one launch): the
        # owner's, or the argument itself when the caller passed a Parameter without naming its module
        ctx.bias_param = (owner.bias if (owner is not None and bias is not None and getattr(owner, "bias", None) is bias)
                          else (bias if isinstance(bias, torch.nn.Parameter) else None))
        ctx.x_shape = x.shape
        return linear_forward_raw(x2, weight, bias).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        o, i = weight.shape
        gx = gw = gb = None
        g2 = gy.reshape(-1, o).contiguous()
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if need_b and ctx.bias_param is not None and DeferredBiasGrads.defer(ctx.bias_param, g2):
            need_b = False      # reduced with all the other bias gradients of the step (DeferredBiasGrads.flush)
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and g2.shape[0] <= 32 and LINEAR_BWD_ONE_LAUNCH:
            # few rows (the aligner's query-side layers): input gradient and weight gradient in ONE launch — these layers sit on the
            # backward chains the step waits for, where a dependent launch costs 10-15 us (profiles/r4_timeline.md)
            _require(g2, weight, x2)
            L = _lib.lib()
            rows = g2.shape[0]
            gx2 = torch.empty((rows, i), dtype=torch.float32, device=g2.device)
            gw2 = torch.empty((o, i), dtype=torch.float32, device=g2.device)
            need = L.icl_linear_ws_bytes(rows, i, o, 1)
            ws = _ws(need, g2) if need else None
            rc = L.icl_linear_bwd_small(_ptr(g2), _ptr(weight), _ptr(x2), _ptr(gx2), _ptr(gw2), _ptr(ws), rows, i, o, _stream(g2))
            if rc == 0:
                return gx2.view(ctx.x_shape), gw2, (g2.sum(0) if need_b else None), None, None
            if rc != 1:      # 1: shapes that do not take that path
                _lib.check(rc, "linear_bwd_small")
        # the product form of a 6^3 convolution (conv3d on tiny volumes): its weight gradient goes to the lane of the deep levels' weight gradients
        lane = ctx.lane_wgrad and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not need_b and WgradLane.wants(g2, 0)
        if lane:
            WgradLane.fork_point(g2)
        if ctx.needs_input_grad[0]:
            gx = linear_dgrad_raw(g2, weight).view(ctx.x_shape)
        if ctx.needs_input_grad[1] or need_b:
            if ctx.needs_input_grad[1]:
                if lane:
                    with WgradLane.on_lane(g2, x2):
                        gw, gb = _tall_atb(g2, x2, False)
                    if not WgradLane.adoptable(ctx.lane_param):
                        WgradLane.sync_to_current(gw)
                else:
                    gw, gb = _tall_atb(g2, x2, need_b)
            elif need_b:
                gb = g2.sum(0)
        return gx, gw, gb, None, None


class _ConvTransposeGemm(torch.autograd.Function):
    """y[b, s, :] = sum_ci x[b, ci, s] W[ci, :] — the k2s2 transposed convolution as ONE batched product whose A operand is the
    channel-major activation itself (k-strided: no transpose copy); gx comes back channel-major, gW is summed over the batch."""

    @staticmethod
    def forward(ctx, x, w2):
        x = x.contiguous()
        w2 = w2.contiguous()
        b, cin = x.shape[0], x.shape[1]
        s = x.numel() // (b * cin)
        j = w2.shape[1]
        ctx.save_for_backward(x, w2)
        return gemm(x, w2, s, j, cin, s, j, False, False, batch=b, a_bstride=cin * s)        # [b, s, j]

    @staticmethod
    def backward(ctx, gy):
        x, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        b, cin = x.shape[0], x.shape[1]
        s = x.numel() // (b * cin)
        j = w2.shape[1]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # gx[b][ci][s] = sum_j W[ci][j] gy[b][s][j]
            gx = torch.empty_like(x)
            gemm(w2, gy, cin, s, j, j, j, True, True, out=gx, ldc=s, batch=b, b_bstride=s * j, c_bstride=cin * s)
        if ctx.needs_input_grad[1]:
            # gW[ci][j] = sum_{b,s} x[b][ci][s] gy[b][s][j]: one product per sample, summed
            gw = torch.empty_like(w2)
            gemm(x, gy, cin, j, s, s, j, True, False, out=gw, ldc=j, batch=b, a_bstride=cin * s, b_bstride=s * j, c_bstride=0)
        return gx, gw


class FactoredGrads:
    """Switch for keeping the weight gradient of HUGE Linear layers factored (ICLTrainer turns it on around its step).

    The four 13,824^2 ``Class_Decoder.mlp2`` weights are 97 % of the ICL model and their gradient is a sum of
    M = batch*classes*heads outer products, dW = g^T x.  With the switch on, backward does not form dW (764 MB each): it
    appends the pair (g [M, out], x [M, in]) to ``weight._icl_factors`` and leaves ``weight.grad`` as None;
    ``icl_amd.optim.FusedSGD`` applies the update straight from the factors (csrc/kernels/optim.h) and
    ``icl_amd.ddp.GradientReducer`` all-gathers the ~1 MB factors instead of all-reducing the matrix.
    Off (the default), ``weight.grad`` is a dense tensor as usual, e.g. for ``torch.optim.SGD`` in the reference trainers."""

    enabled = False
    min_elems = 1 << 21      # the four 13,824^2 and the four 1,728^2 token-axis matrices of the 3-D ICL models
    max_rows = 512           # rows of ONE backward call
    # Data-parallel crossover: the ranks all-gather the factor rows, so the update every rank applies has rows * world rows and is
    # MFMA-bound beyond a few hundred (tools/sgd_probe.py, 13,824^2: 0.54 ms at 16 rows, 0.80 at 128, 2.1 at 512, 4.0 at 1024,
    # 5.9 at 1536 — ~100 TFLOP/s).  The dense alternative costs the 764 MB gradient (0.5 ms to form), its all-reduce (>= 5 ms at the
    # ~150 GB/s of one xGMI link per ring hop) and the 20 B/parameter update (0.64 ms): above ~1536 gathered rows the layer's
    # gradient is formed densely instead.  ICLTrainer sets `world`.
    world = 1
    max_rows_gathered = 1536
    # Single rank: the factors of a layer are complete when its backward runs, so the optimiser step of a weight that is used ONCE
    # per step can ride on the pass that computes the input gradient (csrc/kernels/gemm.h linear_dgrad_sgd_kernel: the 764 MB
    # matrix is read once for gx AND the update instead of twice).  ICLTrainer sets `fused_optimizer` (its FusedSGD) and resets the
    # use counts at the start of each step; None = the update stays in FusedSGD.step().
    fused_optimizer = None
    uses = None

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = FactoredGrads.enabled
        FactoredGrads.enabled = self.on
        return self

    def __exit__(self, *a):
        FactoredGrads.enabled = self.prev


class _LinearFactored(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, owner):
        weight = weight.contiguous()
        x2 = x.reshape(-1, weight.shape[1]).contiguous()
        ctx.save_for_backward(x2, weight)
        ctx.owner = owner
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return linear_forward_raw(x2, weight, bias).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        o, i = weight.shape
        g2 = gy.reshape(-1, o).contiguous()
        param = ctx.owner.weight
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2] and not DeferredBiasGrads.defer(getattr(ctx.owner, "bias", None), g2):
            gb = g2.sum(0)
        opt = FactoredGrads.fused_optimizer
        if (opt is not None and ctx.needs_input_grad[0] and FactoredGrads.uses is not None and FactoredGrads.uses.get(id(param)) == 1
                and weight.data_ptr() == param.data_ptr() and opt.can_update_in_backward(param, g2.shape[0])):
            # input gradient from the old weight and the SGD step of the weight, one pass over the matrix
            return opt.update_in_backward(param, g2, x2).view(ctx.x_shape), None, gb, None
        gx = linear_dgrad_raw(g2, weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        if getattr(param, "_icl_factors", None) is None:
            param._icl_factors = []
        param._icl_factors.append((g2, x2))
        return gx, None, gb, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], owner=None, lane_wgrad: bool = False) -> torch.Tensor:
    """F.linear.  ``owner``: the module whose ``.weight`` this is — lets the gradient stay factored (see FactoredGrads)."""
    if FactoredGrads.uses is not None:
        # every use of a weight in this step, whichever autograd path it takes below: the update-inside-backward of
        # _LinearFactored is only legal for a weight that is read exactly once per step
        key = id(owner.weight) if owner is not None and getattr(owner, "weight", None) is not None else id(weight)
        FactoredGrads.uses[key] = FactoredGrads.uses.get(key, 0) + 1
    if (owner is not None and FactoredGrads.enabled and weight.requires_grad and weight.numel() >= FactoredGrads.min_elems
            and x.numel() // x.shape[-1] <= FactoredGrads.max_rows
            and (x.numel() // x.shape[-1]) * FactoredGrads.world <= FactoredGrads.max_rows_gathered
            and weight.shape[1] % 4 == 0 and torch.is_grad_enabled()):
        return _LinearFactored.apply(x, weight, bias, owner)
    return _Linear.apply(x, weight, bias, owner, lane_wgrad)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _require(x, weight, bias)
        L = _lib.lib()
        x = x.contiguous()
        c = x.shape[-1]
        rows = x.numel() // c
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(L.icl_layernorm_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), _ptr(mean), _ptr(rstd), rows, c, eps, _stream(x)),
                   "layernorm_fwd")
        ctx.save_for_backward(x, weight, mean, rstd)
        both = isinstance(weight, torch.nn.Parameter) and isinstance(bias, torch.nn.Parameter)
        ctx.params = (weight, bias) if both and weight.requires_grad and bias.requires_grad else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        c = x.shape[-1]
        rows = x.numel() // c
        gx = torch.empty_like(x)
        dg = db = ws = None       # no-affine: F.layer_norm(x, [C]) (swinunetr_icl.py:1214)
        defer = False
        if weight is not None:
            ws = _ws(L.icl_layernorm_bwd_ws_bytes(rows, c), x)
            # inside ICLTrainer's step the chunk partials of all LayerNorm layers are summed by ONE launch at the end of backward
            # (DeferredBiasGrads.flush); elsewhere on the spot.  Either way in a fixed order: no float atomics.
            defer = DeferredBiasGrads.pending is not None and ctx.params is not None
            if not defer:
                dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
                dg, db = dgb[0], dgb[1]
        _lib.check(L.icl_layernorm_bwd(_ptr(gy), _ptr(x), _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(dg), _ptr(db), _ptr(ws), rows, c,
                                       _stream(x)), "layernorm_bwd")
        if defer:
            part = ws[:(ws.numel() // (2 * c)) * 2 * c].view(2, -1, c)
            DeferredBiasGrads.pending.append((ctx.params[0], part[0]))
            DeferredBiasGrads.pending.append((ctx.params[1], part[1]))
            DeferredBiasGrads.note_producer(part)
            BackwardEnd.arm()
        return gx, dg, db, None


def layer_norm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    return _LayerNorm.apply(x, weight, bias, eps)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require(x)
        L = _lib.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(L.icl_gelu_fwd(_ptr(x), _ptr(y), x.numel(), _stream(x)), "gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        L = _lib.lib()
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        _lib.check(L.icl_gelu_bwd(_ptr(gy), _ptr(x), _ptr(gx), x.numel(), _stream(x)), "gelu_bwd")
        return gx


def gelu(x: torch.Tensor) -> torch.Tensor:
    return _Gelu.apply(x)


# --------------------------------------------------------------------------------------
# SwinUNETR pieces (networks/swinunetr_icl.py)
# --------------------------------------------------------------------------------------

class _DepthToSpaceCat(torch.autograd.Function):
    """[depth_to_space(yt) | skip] along channels in one buffer: the GEMM result of the k2s2 transposed convolution is moved
    straight into the channel slice of the concat buffer MONAI's UnetrUpBlock builds (``torch.cat((out, skip), 1)``)."""

    @staticmethod
    def forward(ctx, yt, skip, dims):
        _require(yt, skip)
        L = _lib.lib()
        n, d, h, w, cout = dims
        yt = yt.contiguous()
        so = 8 * d * h * w
        cs = 0 if skip is None else skip.shape[1]
        out = torch.empty((n, cout + cs, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=yt.device)
        _lib.check(L.icl_depth_to_space2(_ptr(yt), _ptr(out), n, d, h, w, cout, (cout + cs) * so, _stream(yt)), "depth_to_space2")
        if cs:
            skip = skip.contiguous()
            _lib.check(L.icl_copy_rows(_ptr(skip), _vp(out[:, cout:].data_ptr()), n, cs * so, cs * so, (cout + cs) * so, _stream(yt)),
                       "copy_rows")
        ctx.dims = (n, d, h, w, cout, cs)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        n, d, h, w, cout, cs = ctx.dims
        g = g.contiguous()
        so = 8 * d * h * w
        gt = gskip = None
        if ctx.needs_input_grad[0]:
            gt = torch.empty((n, d * h * w, cout * 8), dtype=torch.float32, device=g.device)
            _lib.check(L.icl_space_to_depth2(_ptr(g), _ptr(gt), n, d, h, w, cout, (cout + cs) * so, _stream(g)), "space_to_depth2")
        if cs and ctx.needs_input_grad[1]:
            gskip = torch.empty((n, cs, 2 * d, 2 * h, 2 * w), dtype=torch.float32, device=g.device)
            _lib.check(L.icl_copy_rows(_vp(g[:, cout:].data_ptr()), _ptr(gskip), n, cs * so, (cout + cs) * so, cs * so, _stream(g)),
                       "copy_rows")
        return gt, gskip, None


def conv_transpose3d_k2s2(x: torch.Tensor, weight: torch.Tensor, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.ConvTranspose3d(cin, cout, kernel 2, stride 2, no bias) — MONAI UnetrUpBlock.transp_conv — optionally followed by
    ``torch.cat((up, skip), 1)``.  Kernel == stride, so the output voxels do not overlap:
    out[b, co, 2z+i, 2y+j, 2x+k] = sum_ci x[b, ci, z, y, x] * W[ci, co, i, j, k], i.e. ONE product
    [B*S, Cin] x [Cin, Cout*8] on the fp32 matrix cores (csrc/kernels/gemm.h, reading the channel-major activation in place)
    followed by a depth-to-space move (csrc/kernels/pool_resize.h) that writes straight into the concat buffer."""
    b, cin, d, h, w = x.shape
    cout = weight.shape[1]
    y = _ConvTransposeGemm.apply(x, weight.flatten(1))          # [B, S, Cout*8]
    return _DepthToSpaceCat.apply(y, skip, (b, d, h, w, cout))


class _GatherRows(torch.autograd.Function):
    """out[b, m] = src[b, idx[m]] (zero row for idx[m] < 0); ``idx_back`` is the index that undoes it (gradient = gather again)."""

    @staticmethod
    def forward(ctx, src, idx, idx_back):
        _require(src, idx, idx_back)
        L = _lib.lib()
        src = src.contiguous()
        b, s, c = src.shape
        m = idx.shape[0]
        assert idx_back.shape[0] == s
        out = torch.empty((b, m, c), dtype=torch.float32, device=src.device)
        _lib.check(L.icl_gather_rows(_ptr(src), _ptr(idx), _ptr(out), b, s, m, c, _stream(src)), "gather_rows")
        ctx.save_for_backward(idx, idx_back)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, idx_back = ctx.saved_tensors
        L = _lib.lib()
        g = g.contiguous()
        b, m, c = g.shape
        s = idx_back.shape[0]
        gs = torch.empty((b, s, c), dtype=torch.float32, device=g.device)
        _lib.check(L.icl_gather_rows(_ptr(g), _ptr(idx_back), _ptr(gs), b, m, s, c, _stream(g)), "gather_rows")
        return gs, None, None


def gather_rows(src: torch.Tensor, idx: torch.Tensor, idx_back: torch.Tensor) -> torch.Tensor:
    """src [B, S, C] -> [B, M, C] with out[:, m] = src[:, idx[m]] (zeros where idx[m] < 0).  ``idx`` must reference every source
    row exactly once (a permutation with optional padding slots) and ``idx_back`` [S] must be its inverse."""
    return _GatherRows.apply(src, idx, idx_back)


class _SplitBatch(torch.autograd.Function):
    """(x[:k], x[k:]) whose gradient is ONE concatenation; two Python slices cost two zero fills, two copies and an add."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k, ctx.n = int(k), x.shape[0]
        ctx.tail = tuple(x.shape[1:])
        ctx.set_materialize_grads(False)
        return x[:k], x[k:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        na, nb = ctx.k * math.prod(ctx.tail), (ctx.n - ctx.k) * math.prod(ctx.tail)
        ok = ref.dtype == torch.float32 and na % 4 == 0 and nb % 4 == 0 and (ref.is_cuda or _lib.host_pointers_ok())
        if ok:
            # one launch, whichever halves exist (a missing half is written as zeros)
            ga = ga.contiguous() if ga is not None else None
            gb = gb.contiguous() if gb is not None else None
            out = torch.empty((ctx.n,) + ctx.tail, dtype=torch.float32, device=ref.device)
            if all(t is None or t.data_ptr() % 16 == 0 for t in (ga, gb)):
                _lib.check(_lib.lib().icl_concat2(_ptr(ga), na, _ptr(gb), nb, _ptr(out), _stream(ref)), "concat2")
                return out, None
        if ga is None:
            ga = ref.new_zeros((ctx.k,) + ctx.tail)
        if gb is None:
            gb = ref.new_zeros((ctx.n - ctx.k,) + ctx.tail)
        return torch.cat([ga, gb], 0), None


def cat_batch(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], 0) — without the copies when the two already ARE one batch: contiguous slices `v[:k]`, `v[k:]` of one tensor
    that need no gradient (the trainers split their batch that way, train_inherent_consistent_unet_3D_BraTS.py:103-104, and the ICL models
    run both streams as one batch): a view over both instead of two device copies at the head of every step."""
    if (a.dim() == b.dim() and a.shape[1:] == b.shape[1:] and a.dtype == b.dtype and a.device == b.device and a.is_contiguous()
            and b.is_contiguous() and not a.requires_grad and not b.requires_grad and a.numel() > 0 and b.numel() > 0
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()):
        return a.as_strided((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
    return torch.cat([a, b], 0)


def split_batch(x: torch.Tensor, k: int):
    """x[:k], x[k:] (views) with a single-kernel gradient."""
    return _SplitBatch.apply(x, k)


class _GatherRowsDup(torch.autograd.Function):
    """Row gather whose index may repeat or omit source rows (PatchMerging); ``idx_back2`` [S, 2] lists for every source row the
    (at most two) output rows that read it, -1 where absent, so the gradient is again a deterministic gather."""

    @staticmethod
    def forward(ctx, src, idx, idx_back2):
        _require(src, idx, idx_back2)
        L = _lib.lib()
        src = src.contiguous()
        b, s, c = src.shape
        m = idx.shape[0]
        assert tuple(idx_back2.shape) == (s, 2)
        out = torch.empty((b, m, c), dtype=torch.float32, device=src.device)
        _lib.check(L.icl_gather_rows(_ptr(src), _ptr(idx), _ptr(out), b, s, m, c, _stream(src)), "gather_rows")
        ctx.save_for_backward(idx_back2)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx_back2,) = ctx.saved_tensors
        L = _lib.lib()
        g = g.contiguous()
        b, m, c = g.shape
        s = idx_back2.shape[0]
        gs = torch.empty((b, s, c), dtype=torch.float32, device=g.device)
        _lib.check(L.icl_gather_rows_sum2(_ptr(g), _ptr(idx_back2), _ptr(gs), b, m, s, c, _stream(g)), "gather_rows_sum2")
        return gs, None, None


def gather_rows_dup(src: torch.Tensor, idx: torch.Tensor, idx_back2: torch.Tensor) -> torch.Tensor:
    """src [B, S, C] -> [B, M, C], out[:, m] = src[:, idx[m]]; every source row is read by at most two output rows, listed in
    ``idx_back2`` [S, 2] (int32, -1 = none)."""
    return _GatherRowsDup.apply(src, idx, idx_back2)


_RELPOS_INV = {}


def _relpos_inverse(index: torch.Tensor, n: int, table_rows: int):
    """(inv, offs) of ``relative_position_index[:n, :n]``: the padded positions i * npad + j sorted by table row (stable) and the
    row offsets — what icl_relpos_bias_bwd_sum walks to add the bias gradient in a fixed order.  Built once per (index buffer, n) on the
    host (a buffer that never changes); the first call must not happen under graph capture (every trainer warms up eagerly)."""
    key = (index.data_ptr(), int(index._version), n, table_rows, str(index.device))
    hit = _RELPOS_INV.get(key)
    if hit is None:
        if index.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("icl_amd window attention: the inverse of relative_position_index must be built before graph capture "
                               "(run one eager backward first)")
        npad = (n + 15) // 16 * 16
        idx = index[:n, :n].reshape(-1).to("cpu", torch.int64)
        order = torch.argsort(idx, stable=True)
        inv = ((order // n) * npad + order % n).to(torch.int32)
        offs = torch.zeros(table_rows + 1, dtype=torch.int64)
        offs[1:] = torch.cumsum(torch.bincount(idx, minlength=table_rows), 0)
        hit = _RELPOS_INV[key] = (inv.to(index.device), offs.to(torch.int32).to(index.device))
    return hit


class _WindowAttention(torch.autograd.Function):
    """One fused MFMA kernel per direction (csrc/kernels/winattn.h): scores, bias, shift mask, softmax and P@V never leave
    the CU; backward recomputes the scores from the saved log-sum-exp.  The relative-position bias is gathered from the
    table into the kernel's padded [heads, n, npad] layout by a small kernel (and its gradient scattered back)."""

    @staticmethod
    def forward(ctx, qkv, table, index, regions, heads, scale):
        _require(qkv, table, index, regions)
        L = _lib.lib()
        qkv = qkv.contiguous()
        b_, n, c3 = qkv.shape
        c = c3 // 3
        dh = c // heads
        if c != heads * dh or dh not in (16, 32):
            raise ValueError("icl_amd window attention: head dim must be 16 (SwinUNETR) or 32 (2-D Swin-UNet)")
        if table.shape[1] != heads or index.dim() != 2 or index.shape[0] != index.shape[1] or n > index.shape[0]:
            raise ValueError("icl_amd window attention: table [T, heads] / index [N, N] with n <= N expected")
        table, index = table.contiguous(), index.contiguous()
        npad = (n + 15) // 16 * 16
        bias_pad = torch.empty((heads, n, npad), dtype=torch.float32, device=qkv.device)
        _lib.check(L.icl_relpos_bias_fwd(_ptr(table), _ptr(index), _ptr(bias_pad), n, heads, index.shape[0], _stream(qkv)), "relpos_bias_fwd")
        nw = regions.shape[0] if regions is not None else 1
        out = torch.empty((b_, n, c), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((b_, heads, n), dtype=torch.float32, device=qkv.device)
        flops = 4.0 * b_ * heads * n * n * dh
        with _timed("window_attn_fwd_kernel", flops, 4.0 * (qkv.numel() + out.numel()), qkv):
            _lib.check(L.icl_window_attn_fwd(_ptr(qkv), _ptr(bias_pad), _ptr(regions), _ptr(out), _ptr(lse), b_, n, heads, nw, dh,
                                             scale, _stream(qkv)), "window_attn_fwd")
        ctx.save_for_backward(qkv, bias_pad, index, regions, out, lse)
        ctx.cfg = (heads, scale, nw, table.shape[0], dh)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, bias_pad, index, regions, out, lse = ctx.saved_tensors
        heads, scale, nw, trows, dh = ctx.cfg
        L = _lib.lib()
        gout = gout.contiguous()
        b_, n, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        need_table = ctx.needs_input_grad[1]
        chunks = L.icl_window_attn_bwd_chunks(b_, n, heads, nw, dh) if need_table else 0
        # one d(bias) slab per slice of windows (plain stores), summed in a fixed order below: no float atomics
        dbias = torch.empty((chunks,) + tuple(bias_pad.shape), dtype=torch.float32, device=qkv.device) if need_table else None
        flops = 14.0 * b_ * heads * n * n * dh
        with _timed("window_attn_bwd_kernel", flops, 4.0 * (2 * qkv.numel() + 2 * out.numel()), qkv):
            _lib.check(L.icl_window_attn_bwd(_ptr(qkv), _ptr(bias_pad), _ptr(regions), _ptr(out), _ptr(lse), _ptr(gout), _ptr(dqkv),
                                             _ptr(dbias), b_, n, heads, nw, dh, scale, _stream(qkv)), "window_attn_bwd")
        dtable = None
        if need_table:
            dtable = torch.empty((trows, heads), dtype=torch.float32, device=qkv.device)
            inv, offs = _relpos_inverse(index, n, trows)
            _lib.check(L.icl_relpos_bias_bwd_sum(_ptr(dbias), chunks, _ptr(inv), _ptr(offs), _ptr(dtable), trows, n, heads, _stream(qkv)),
                       "relpos_bias_bwd_sum")
        return dqkv, dtable, None, None, None, None


def window_attention(qkv: torch.Tensor, table: torch.Tensor, index: torch.Tensor, regions: Optional[torch.Tensor], heads: int,
                     scale: float) -> torch.Tensor:
    """WindowAttention.forward core (swinunetr_icl.py:728-747).  qkv [B_, n, 3C] (q|k|v, each head-major); table
    ``relative_position_bias_table`` [T, heads] and index ``relative_position_index`` [343, 343] (the bias of a window with
    n tokens is table[index[:n, :n]] — also for clipped windows, as the reference does); regions int32 [nW, n] or None
    (tokens of one window attend to each other iff their region ids match: the 0/-100 mask).  Returns [B_, n, C]."""
    return _WindowAttention.apply(qkv, table, index, regions, heads, float(scale))


class _ProtoAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qh, kv, heads, scale):
        _require(qh, kv)
        L = _lib.lib()
        qh = qh.contiguous()
        kv = kv.contiguous()
        B, h, nc, d = qh.shape
        N = kv.shape[1]
        assert h == heads and kv.shape[2] == 2 * h * d
        logits = torch.empty((B, nc, h, N), dtype=torch.float32, device=qh.device)       # class-major: the layout the reference returns
        out = torch.empty((B, h, nc, d), dtype=torch.float32, device=qh.device)
        stats = torch.empty((B, h, nc, 2), dtype=torch.float32, device=qh.device)
        _lib.check(L.icl_attn_fwd(_ptr(qh), _ptr(kv), _ptr(logits), _ptr(out), _ptr(stats), B, h, nc, N, d, scale, _stream(qh)), "attn_fwd")
        ctx.save_for_backward(qh, kv, logits, stats, out)
        ctx.scale = scale
        return out, logits

    @staticmethod
    def backward(ctx, gout, glog):
        qh, kv, logits, stats, out = ctx.saved_tensors
        L = _lib.lib()
        B, h, nc, d = qh.shape
        N = kv.shape[1]
        gout = gout.contiguous() if gout is not None else None
        glog = glog.contiguous() if glog is not None else None
        gq = torch.empty_like(qh)
        gkv = torch.empty_like(kv)
        # dQ from the shares the dK / dV pass writes per 256-token chunk (long token axes: the row-per-workgroup dQ kernel is a latency
        # chain on the backward's query chain); short axes keep the two-kernel form (one launch fewer than shares + sum... the same count)
        ws = _ws(L.icl_attn_bwd_ws_bytes(B, h, nc, N, d), qh) if N >= 1024 else None
        _lib.check(L.icl_attn_bwd_ws(_ptr(qh), _ptr(kv), _ptr(logits), _ptr(stats), _ptr(out), _ptr(gout), _ptr(glog), _ptr(gq), _ptr(gkv),
                                     _ptr(ws), B, h, nc, N, d, ctx.scale, _stream(qh)), "attn_bwd")
        return gq, gkv, None, None


def prototype_attention(qh: torch.Tensor, kv: torch.Tensor, heads: int, scale: float):
    """qh [B,h,nc,d]; kv [B,N,2*h*d] laid out (k|v, head, d).  Returns (softmax(QK^T*scale) V  [B,h,nc,d],
    scaled pre-softmax logits [B,nc,h,N] — already in the layout of the reference's ``attn1.permute(0, 2, 1, 3)``)."""
    return _ProtoAttention.apply(qh, kv, heads, float(scale))


# --------------------------------------------------------------------------------------
# The aligner's query chain as fused stages (csrc/kernels/qchain.h; networks/unet_3D_icl.py:258-264, 283-297, 299-315, 220-222)
# --------------------------------------------------------------------------------------

class _QcStage(ctypes.Structure):      # include/icl_hip.h IclQcStage
    _fields_ = [("x", _vp), ("x_rows", ctypes.c_int), ("w", _vp), ("bias", _vp), ("y", _vp),
                ("R", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int), ("trans", ctypes.c_int), ("nc", ctypes.c_int),
                ("npw", ctypes.c_int), ("pro", ctypes.c_int), ("epi", ctypes.c_int),
                ("pa", _vp), ("pb", _vp), ("pc", _vp), ("pd", _vp), ("pro_dp", ctypes.c_int),
                ("so0", _vp), ("so1", _vp), ("so2", _vp), ("ea", _vp), ("eb", _vp),
                ("ea_r0", ctypes.c_int), ("ea_r1", ctypes.c_int), ("eb_r0", ctypes.c_int), ("eb_r1", ctypes.c_int),
                ("dp_seed", ctypes.c_uint32 * 2), ("dp_thresh", ctypes.c_uint32 * 2), ("dp_scale", ctypes.c_float * 2),
                ("dp_seed_dev", _vp), ("eps", ctypes.c_float)]


QC_PRO_NONE, QC_PRO_LN, QC_PRO_GELU, QC_PRO_LNBWD, QC_PRO_SCALE = range(5)
QC_EPI_NONE, QC_EPI_DP1, QC_EPI_RES_DP, QC_EPI_GELUBWD, QC_EPI_ADDROWS, QC_EPI_SUMB = range(6)
QCHAIN = os.environ.get("ICL_QCHAIN", "1") != "0"
QC_MAX_ROWS, QC_MAX_K, QC_ROWS_PER_PASS = 32, 1024, 8
# Rows up to which the aligner takes the fused chain.  The stages walk the rows in passes of eight, re-reading the workgroup's weight slice per
# pass: with nc = 16 (32 / 16 rows) the fused chain measured SLOWER than the operator-by-operator path whose products run on the MFMA stream
# kernel (15.58 against 15.30 ms per step, profiles/r6_qchain_ab.txt), so it is taken for one pass only (nc = 2: 4 / 2 rows); the kernels and
# their tests cover 32 rows.
QC_FUSE_MAX_ROWS = int(os.environ.get("ICL_QCHAIN_MAX_ROWS", str(QC_ROWS_PER_PASS)))


def _qc_stage(like, dp, *, y, R, K, N=0, trans=0, nc=1, x=None, x_rows=0, w=None, bias=None, pro=QC_PRO_NONE, epi=QC_EPI_NONE, pa=None,
              pb=None, pc=None, pd=None, pro_dp=0, so0=None, so1=None, so2=None, ea=None, eb=None, ea_rows=(0, 0), eb_rows=(0, 0), eps=1e-5):
    """One launch of icl_qchain_stage.  ``dp``: ((seed, thresh, scale) of site 0, of site 1, seed_dev tensor or None)."""
    st = _QcStage()
    st.x, st.x_rows, st.w, st.bias, st.y = _ptr(x), x_rows, _ptr(w), _ptr(bias), _ptr(y)
    st.R, st.K, st.N, st.trans, st.nc, st.npw = R, K, N, trans, nc, 0
    st.pro, st.epi, st.pro_dp = pro, epi, pro_dp
    st.pa, st.pb, st.pc, st.pd = _ptr(pa), _ptr(pb), _ptr(pc), _ptr(pd)
    st.so0, st.so1, st.so2 = _ptr(so0), _ptr(so1), _ptr(so2)
    st.ea, st.eb = _ptr(ea), _ptr(eb)
    st.ea_r0, st.ea_r1 = ea_rows
    st.eb_r0, st.eb_r1 = eb_rows
    for i in (0, 1):
        st.dp_seed[i], st.dp_thresh[i], st.dp_scale[i] = dp[i]
    st.dp_seed_dev = _ptr(dp[2])
    st.eps = eps
    _lib.check(_lib.lib().icl_qchain_stage(ctypes.byref(st), _stream(like)), "qchain_stage")


def _qc_wgrad(like, jobs):
    """jobs: (g, x, dw, db or None, rows, n, k, kind)."""
    n = len(jobs)
    arr, iarr = _vp * n, ctypes.c_int32 * n
    _lib.check(_lib.lib().icl_qchain_wgrad(arr(*[j[0].data_ptr() for j in jobs]), arr(*[j[1].data_ptr() for j in jobs]),
                                          arr(*[j[2].data_ptr() for j in jobs]), arr(*[(j[3].data_ptr() if j[3] is not None else None) for j in jobs]),
                                          iarr(*[j[4] for j in jobs]), iarr(*[j[5] for j in jobs]), iarr(*[j[6] for j in jobs]),
                                          iarr(*[j[7] for j in jobs]), n, _stream(like)), "qchain_wgrad")


def query_chain_ok(rows: int, c: int, hidden: int, nc: int, like: torch.Tensor) -> bool:
    return (QCHAIN and rows <= min(QC_MAX_ROWS, QC_FUSE_MAX_ROWS) and c % 8 == 0 and hidden % 4 == 0 and max(c, hidden) <= QC_MAX_K and nc <= 16
            and (like.is_cuda or _lib.host_pointers_ok()))


class _QueryAttend(torch.autograd.Function):
    """One resolution level of the aligner's query half as fused stages: LayerNorm(norm1_query) -> fc_q -> [read-out of softmax(q k^T) v]
    -> proj -> q + drop_path(q) -> LayerNorm(norm2) -> fc1 -> GELU -> fc2 -> q + drop_path(.) -> query_convs.
    Inputs: q_in [B or 1, nc, C] (a [1, nc, C] query is broadcast over the batch inside the first stage), kv [B, N, 2C] (fc_kv of the
    normalised tokens), cfg, the fourteen parameters.  Outputs: logits [B, nc, h, N], and with cfg['full'] the query after the block as
    its two batch halves q2[:ba], q2[ba:] (views of one buffer) and the query handed down a level [B, nc, C/2].  ``full`` False (the
    guided call of the 3-D models, whose updated queries nobody reads): fc_q and the logits only."""

    @staticmethod
    def forward(ctx, q_in, kv, cfg, n1w, n1b, fqw, fqb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, qcw, qcb):
        L = _lib.lib()
        h, scale, ba, eps, dp, full = cfg["heads"], cfg["scale"], cfg["ba"], cfg["eps"], cfg["dp"], cfg["full"]
        q_in, kv = q_in.contiguous(), kv.contiguous()
        _require(q_in, kv, fqw)
        nc, C = q_in.shape[1], q_in.shape[2]
        B, N = kv.shape[0], kv.shape[1]
        R, d = B * nc, C // h
        dev = kv.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)      # noqa: E731
        xhat1, rstd1, n1, qf = new(R, C), new(R), new(R, C), new(R, C)
        _qc_stage(kv, dp, y=qf, R=R, K=C, N=C, nc=nc, x=q_in, x_rows=q_in.shape[0] * nc, w=fqw, bias=fqb, pro=QC_PRO_LN, pa=n1w, pb=n1b,
                  so0=xhat1, so1=rstd1, so2=n1, eps=eps)
        logits, a, stats = new(B, nc, h, N), new(R, C), new(B, h, nc, 2)
        # qf read as [B, h, nc, d] and the read-out written back as [B, nc, C]: the reference's reshape quirk (:287, :293) is a
        # reinterpretation of the same contiguous bytes
        _lib.check(L.icl_attn_fwd(_ptr(qf), _ptr(kv), _ptr(logits), _ptr(a), _ptr(stats), B, h, nc, N, d, scale, _stream(kv)), "attn_fwd")
        ctx.cfg, ctx.shape = cfg, (B, nc, C, N, tuple(q_in.shape))
        ctx.set_materialize_grads(False)      # an output nobody differentiates (the last level's hand-down, a discarded batch half) arrives as None
        if not full:
            ctx.save_for_backward(kv, logits, stats, a, qf, xhat1, rstd1, n1, n1w, fqw)
            return (logits,)
        H = f1w.shape[0]
        q1, xhat2, rstd2, n2, u, hh, q2, nxt = new(R, C), new(R, C), new(R), new(R, C), new(R, H), new(R, H), new(B, nc, C), new(B, nc, qcw.shape[0])
        _qc_stage(kv, dp, y=q1, R=R, K=C, N=C, nc=nc, x=a, w=pw, bias=pb, epi=QC_EPI_DP1)
        _qc_stage(kv, dp, y=u, R=R, K=C, N=H, nc=nc, x=q1, w=f1w, bias=f1b, pro=QC_PRO_LN, pa=n2w, pb=n2b, so0=xhat2, so1=rstd2, so2=n2, eps=eps)
        _qc_stage(kv, dp, y=q2, R=R, K=H, N=C, nc=nc, x=u, w=f2w, bias=f2b, pro=QC_PRO_GELU, so2=hh, epi=QC_EPI_RES_DP, ea=q1)
        _qc_stage(kv, dp, y=nxt, R=R, K=C, N=qcw.shape[0], nc=nc, x=q2, w=qcw, bias=qcb)
        ctx.save_for_backward(kv, logits, stats, a, qf, xhat1, rstd1, n1, n1w, fqw, q1, xhat2, rstd2, n2, u, hh, q2, n2w, pw, f1w, f2w, qcw)
        return logits, q2[:ba], q2[ba:], nxt

    @staticmethod
    def backward(ctx, glog, gq2a=None, gq2b=None, gnxt=None):
        L = _lib.lib()
        cfg = ctx.cfg
        h, scale, ba, dp, full = cfg["heads"], cfg["scale"], cfg["ba"], cfg["dp"], cfg["full"]
        B, nc, C, N, q_shape = ctx.shape
        R, d = B * nc, C // h
        sv = ctx.saved_tensors
        kv, logits, stats, a, qf, xhat1, rstd1, n1, n1w, fqw = sv[:10]
        dev = kv.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)      # noqa: E731
        jobs = []
        ga = None
        grads = [None] * 14
        if full and (gq2a is not None or gq2b is not None or gnxt is not None):
            q1, xhat2, rstd2, n2, u, hh, q2, n2w, pw, f1w, f2w, qcw = sv[10:]
            H = f1w.shape[0]
            gq2a = gq2a.contiguous() if gq2a is not None else None
            gq2b = gq2b.contiguous() if gq2b is not None else None
            dq2, g2, du, dn2, do, ga = new(R, C), new(R, C), new(R, H), new(R, C), new(R, C), new(R, C)
            add = dict(epi=QC_EPI_ADDROWS, ea=gq2a, ea_rows=(0, ba * nc), eb=gq2b, eb_rows=(ba * nc, R))
            if gnxt is not None:
                gnxt = gnxt.contiguous()
                _qc_stage(kv, dp, y=dq2, R=R, K=qcw.shape[0], N=C, trans=1, nc=nc, x=gnxt, w=qcw, **add)
            else:
                _qc_stage(kv, dp, y=dq2, R=R, K=C, trans=2, nc=nc, **add)
            # g2 = f1 * dq2 is fc2's output gradient; du = (g2 W2) * gelu'(u)
            _qc_stage(kv, dp, y=du, R=R, K=C, N=H, trans=1, nc=nc, x=dq2, w=f2w, pro=QC_PRO_SCALE, so2=g2, epi=QC_EPI_GELUBWD, ea=u)
            _qc_stage(kv, dp, y=dn2, R=R, K=H, N=C, trans=1, nc=nc, x=du, w=f1w)
            # do = (dq2 + LayerNorm'(dn2)) * (1 + f0) is proj's output gradient; ga = do Wp the read-out's
            _qc_stage(kv, dp, y=ga, R=R, K=C, N=C, trans=1, nc=nc, x=dn2, w=pw, pro=QC_PRO_LNBWD, pa=xhat2, pb=rstd2, pc=n2w, pd=dq2, pro_dp=1, so2=do)
            gpw, gpb, gn2w, gn2b, gf1w, gf1b, gf2w, gf2b = new(C, C), new(C), new(C), new(C), new(H, C), new(H), new(C, H), new(C)
            jobs += [(do, a, gpw, gpb, R, C, C, 0), (dn2, xhat2, gn2w, gn2b, R, C, C, 1), (du, n2, gf1w, gf1b, R, H, C, 0),
                     (g2, hh, gf2w, gf2b, R, C, H, 0)]
            grads[4:12] = [gpw, gpb, gn2w, gn2b, gf1w, gf1b, gf2w, gf2b]
            if gnxt is not None:
                gqcw, gqcb = new(qcw.shape[0], C), new(qcw.shape[0])
                jobs.append((gnxt, q2.view(R, C), gqcw, gqcb, R, qcw.shape[0], C, 0))
                grads[12:14] = [gqcw, gqcb]
        glog = glog.contiguous() if glog is not None else None
        gq, gkv = new(R, C), torch.empty_like(kv)
        ws = _ws(L.icl_attn_bwd_ws_bytes(B, h, nc, N, d), kv) if N >= 1024 else None
        _lib.check(L.icl_attn_bwd_ws(_ptr(qf), _ptr(kv), _ptr(logits), _ptr(stats), _ptr(a), _ptr(ga), _ptr(glog), _ptr(gq), _ptr(gkv),
                                     _ptr(ws), B, h, nc, N, d, scale, _stream(kv)), "attn_bwd")
        dn1 = new(R, C)
        _qc_stage(kv, dp, y=dn1, R=R, K=C, N=C, trans=1, nc=nc, x=gq, w=fqw)
        bcast = q_shape[0] == 1 and B > 1
        if bcast and R <= QC_ROWS_PER_PASS:
            gqin = new(1, nc, C)
            _qc_stage(kv, dp, y=gqin, R=R, K=C, trans=2, nc=nc, x=dn1, pro=QC_PRO_LNBWD, pa=xhat1, pb=rstd1, pc=n1w, epi=QC_EPI_SUMB)
        else:
            gqin = new(B, nc, C)
            _qc_stage(kv, dp, y=gqin, R=R, K=C, trans=2, nc=nc, x=dn1, pro=QC_PRO_LNBWD, pa=xhat1, pb=rstd1, pc=n1w)
            if bcast:
                gqin = gqin.sum(0, keepdim=True)
        gn1w, gn1b, gfqw, gfqb = new(C), new(C), new(C, C), new(C)
        jobs += [(dn1, xhat1, gn1w, gn1b, R, C, C, 1), (gq, n1, gfqw, gfqb, R, C, C, 0)]
        grads[0:4] = [gn1w, gn1b, gfqw, gfqb]
        _qc_wgrad(kv, jobs)
        return (gqin, gkv, None, *grads)


def query_attend(q_in, kv, heads, scale, ba, eps, dp, full, params):
    """See _QueryAttend.  ``params``: the fourteen parameter tensors in the Function's order."""
    cfg = dict(heads=int(heads), scale=float(scale), ba=int(ba), eps=float(eps), dp=dp, full=bool(full))
    return _QueryAttend.apply(q_in, kv, cfg, *params)


def drop_path_site(like, p: float, training: bool):
    """(seed, thresh, scale, seed_dev) of one drop-path site for the fused stages: the per-sample decision of ``drop_path`` (same hash, same
    seeding); an inactive site keeps everything at scale 1."""
    if p == 0.0 or not training:
        return (0, 0, 1.0), None
    seed, seed_dev = _step_seed(like, None)
    return (seed, int(p * 4294967296.0) & 0xFFFFFFFF, 1.0 / (1.0 - p)), seed_dev


# --------------------------------------------------------------------------------------
# Loss reductions (utils/losses.py L1-L5): one fused HIP pass per loss term, one more for its gradient
# --------------------------------------------------------------------------------------

LOSS_MAX_BLOCKS = 1024   # ICL_LOSS_MAX_BLOCKS (include/icl_hip.h)


class _FusedLoss(torch.autograd.Function):
    """(term0, term1) per icl_loss_fwd's mode table (include/icl_hip.h)."""

    @staticmethod
    def forward(ctx, a, target, weight, mode, a_is_prob):
        _require(a, target, weight)
        L = _lib.lib()
        a = a.contiguous()
        target = target.contiguous()
        B, nc = a.shape[0], a.shape[1]
        S = a.numel() // (B * nc)
        hard = mode <= 1
        if hard:
            assert target.dtype == torch.int64 and target.numel() == B * S, "labels must be int64 [B, ...]"
        else:
            assert target.shape == a.shape and target.dtype == torch.float32
        stats = torch.empty((3 * nc + 1) * (1 + LOSS_MAX_BLOCKS), dtype=torch.float32, device=a.device)   # ICL_LOSS_STATS_FLOATS
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        _lib.check(L.icl_loss_fwd(_ptr(a), None if hard else _ptr(target), _ptr(target) if hard else None, _ptr(weight),
                                  _ptr(stats), _ptr(out), B, nc, S, mode, int(a_is_prob), _stream(a)), "loss_fwd")
        ctx.save_for_backward(a, target, stats, weight)
        ctx.cfg = (mode, int(a_is_prob))
        ctx.set_materialize_grads(False)
        return out[0], out[1]      # two scalars: an unused one costs nothing in backward (no select_backward / zero fill)

    @staticmethod
    def backward(ctx, g0, g1):
        a, target, stats, weight = ctx.saved_tensors
        mode, aip = ctx.cfg
        L = _lib.lib()
        B, nc = a.shape[0], a.shape[1]
        S = a.numel() // (B * nc)
        hard = mode <= 1
        g0 = g0.contiguous() if g0 is not None else None
        g1 = g1.contiguous() if g1 is not None else None
        ga = torch.empty_like(a)
        _lib.check(L.icl_loss_bwd(_ptr(a), None if hard else _ptr(target), _ptr(target) if hard else None, _ptr(weight),
                                  _ptr(stats), _ptr(g0), _ptr(g1), _ptr(ga), B, nc, S, mode, aip, _stream(a)), "loss_bwd")
        return ga, None, None, None, None


class _LossJob(ctypes.Structure):      # include/icl_hip.h IclLossJob
    _fields_ = [("a", _vp), ("b", _vp), ("labels", _vp), ("weight", _vp), ("stats", _vp), ("out", _vp), ("gout0", _vp), ("gout1", _vp),
                ("ga", _vp), ("s", ctypes.c_int64), ("batch", ctypes.c_int), ("nc", ctypes.c_int), ("mode", ctypes.c_int),
                ("a_is_prob", ctypes.c_int)]


LOSS_MULTI_MAX = 12
LOSS_MULTI = os.environ.get("ICL_LOSS_MULTI", "1") != "0"


class _FusedLossMulti(torch.autograd.Function):
    """Several `_FusedLoss` terms — (a_j, target_j, mode_j) — in ONE statistics launch, one finalize launch and one gradient launch
    (icl_loss_fwd_multi / icl_loss_bwd_multi).  ``modes``: tuple of icl_loss_fwd modes; tensors: a_0, t_0, a_1, t_1, ...; returns
    (term0_0, term1_0, term0_1, term1_1, ...), each the 0-dim tensor `_FusedLoss` would have returned (bit-identical)."""

    @staticmethod
    def forward(ctx, modes, *tensors):
        L = _lib.lib()
        n = len(modes)
        assert 1 <= n <= LOSS_MULTI_MAX and len(tensors) == 2 * n
        a_s = [tensors[2 * j].contiguous() for j in range(n)]
        t_s = [tensors[2 * j + 1].contiguous() for j in range(n)]
        _require(*a_s, *t_s)
        dev = a_s[0].device
        nc = a_s[0].shape[1]
        per = (3 * nc + 1) * (1 + LOSS_MAX_BLOCKS)
        stats = torch.empty((n, per), dtype=torch.float32, device=dev)
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        jobs = (_LossJob * n)()
        for j in range(n):
            a, t, mode = a_s[j], t_s[j], modes[j]
            B = a.shape[0]
            S = a.numel() // (B * nc)
            hard = mode <= 1
            assert a.shape[1] == nc
            if hard:
                assert t.dtype == torch.int64 and t.numel() == B * S, "labels must be int64 [B, ...]"
            else:
                assert t.shape == a.shape and t.dtype == torch.float32
            q = jobs[j]
            q.a, q.b, q.labels, q.weight = a.data_ptr(), (None if hard else t.data_ptr()), (t.data_ptr() if hard else None), None
            q.stats, q.out = stats[j].data_ptr(), out[j].data_ptr()
            q.s, q.batch, q.nc, q.mode, q.a_is_prob = S, B, nc, mode, 0
        _lib.check(L.icl_loss_fwd_multi(jobs, n, _stream(a_s[0])), "loss_fwd_multi")
        ctx.save_for_backward(stats, *a_s, *t_s)
        ctx.modes = tuple(modes)
        ctx.set_materialize_grads(False)
        return tuple(out[j, k] for j in range(n) for k in (0, 1))

    @staticmethod
    def backward(ctx, *gs):
        L = _lib.lib()
        modes = ctx.modes
        n = len(modes)
        stats = ctx.saved_tensors[0]
        a_s, t_s = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:1 + 2 * n]
        nc = a_s[0].shape[1]
        gas = [None] * n
        keep = []
        live = [j for j in range(n) if ctx.needs_input_grad[1 + 2 * j] and (gs[2 * j] is not None or gs[2 * j + 1] is not None)]
        jobs = (_LossJob * max(len(live), 1))()
        for i, j in enumerate(live):
            a, t, mode = a_s[j], t_s[j], modes[j]
            B = a.shape[0]
            S = a.numel() // (B * nc)
            hard = mode <= 1
            g0 = gs[2 * j].contiguous() if gs[2 * j] is not None else None
            g1 = gs[2 * j + 1].contiguous() if gs[2 * j + 1] is not None else None
            keep += [g0, g1]
            gas[j] = torch.empty_like(a)
            q = jobs[i]
            q.a, q.b, q.labels, q.weight = a.data_ptr(), (None if hard else t.data_ptr()), (t.data_ptr() if hard else None), None
            q.stats, q.out = stats[j].data_ptr(), None
            q.gout0, q.gout1, q.ga = _ptr(g0), _ptr(g1), gas[j].data_ptr()
            q.s, q.batch, q.nc, q.mode, q.a_is_prob = S, B, nc, mode, 0
        if live:
            _lib.check(L.icl_loss_bwd_multi(jobs, len(live), _stream(a_s[0])), "loss_bwd_multi")
        res = [None]
        for j in range(n):
            res += [gas[j], None]
        return tuple(res)


def fused_losses(terms):
    """``terms``: list of (a, target, mode) with mode 1 = (cross-entropy, Dice(softmax)) on int64 labels, 2 = soft Dice against logits,
    3 = softmax MSE against logits (the targets of modes 2 / 3 are detached).  Returns the list of (term0, term1) pairs — one statistics
    launch, one finalize launch, one gradient launch for all of them when there are at most 12 with one class count."""
    if not terms:
        return []
    ncs = {t[0].shape[1] for t in terms}
    if not LOSS_MULTI or len(terms) > LOSS_MULTI_MAX or len(ncs) != 1:
        return [_FusedLoss.apply(a, (t.long() if m <= 1 else t.detach()), None, m, False) for a, t, m in terms]
    flat = []
    for a, t, m in terms:
        flat += [a, (t.long() if m <= 1 else t.detach())]
    out = _FusedLossMulti.apply(tuple(m for _, _, m in terms), *flat)
    return [(out[2 * j], out[2 * j + 1]) for j in range(len(terms))]


class _CombineScalars(torch.autograd.Function):
    """out[j] = sum_i W[j][i] * terms[i] for 0-dim fp32 tensors ``terms``: one launch (icl_scalar_combine); backward one launch with
    W^T.  The scalar arithmetic between loss terms (trainer :105-112, losses.py "(a + b + c) / 3") as torch ops is a chain of
    one-element kernels, forward and backward."""

    @staticmethod
    def forward(ctx, W, *terms):
        n, m = len(terms), len(W)
        ts = [t if (t.dtype == torch.float32 and t.dim() == 0) else t.float().reshape(()) for t in terms]
        _require(*ts)
        out = torch.empty(m, dtype=torch.float32, device=ts[0].device)
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        flat = (ctypes.c_float * (m * n))(*[float(W[j][i]) for j in range(m) for i in range(n)])
        _lib.check(_lib.lib().icl_scalar_combine(arr, flat, n, m, _ptr(out), _stream(out)), "scalar_combine")
        ctx.W = W
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, gout):
        W, n = ctx.W, ctx.n
        m = len(W)
        gout = gout.contiguous()
        gin = torch.empty(n, dtype=torch.float32, device=gout.device)
        arr = (ctypes.c_void_p * m)(*[gout.data_ptr() + 4 * j for j in range(m)])
        flat = (ctypes.c_float * (n * m))(*[float(W[j][i]) for i in range(n) for j in range(m)])       # W^T, [n][m]
        _lib.check(_lib.lib().icl_scalar_combine(arr, flat, m, n, _ptr(gin), _stream(gin)), "scalar_combine_bwd")
        return (None,) + tuple(gin[i] for i in range(n))


def combine_scalars(terms, W) -> torch.Tensor:
    """[sum_i W[j][i] * terms[i] for j] as one fp32 vector; ``terms``: 0-dim tensors on one device, ``W``: nested Python floats
    (at most 16 x 16)."""
    return _CombineScalars.apply(tuple(tuple(float(v) for v in row) for row in W), *terms)


def _weight_tensor(weight, like):
    if weight is None:
        return None
    return torch.as_tensor(weight, dtype=torch.float32, device=like.device)


def dice_loss(inputs: torch.Tensor, labels: torch.Tensor, n_classes: int, softmax: bool = False, weight=None):
    """DiceLoss.forward (losses.py:218-231); labels [B, ...] integer class ids."""
    assert inputs.shape[1] == n_classes
    return _FusedLoss.apply(inputs, labels.long(), _weight_tensor(weight, inputs), 0, not softmax)[1]


def cross_entropy_dice_parts(logits: torch.Tensor, labels: torch.Tensor, n_classes: int):
    """(CrossEntropyLoss()(logits, labels), DiceLoss(softmax=True)(logits, labels)) from ONE pass over the logits."""
    assert logits.shape[1] == n_classes
    return _FusedLoss.apply(logits, labels.long(), None, 1, False)


def cross_entropy_dice(logits: torch.Tensor, labels: torch.Tensor, n_classes: int):
    """CE + Dice(softmax=True) — one AuxLoss3D term (losses.py:268-269)."""
    ce, dc = cross_entropy_dice_parts(logits, labels, n_classes)
    return ce + dc


def soft_dice_loss(a: torch.Tensor, b: torch.Tensor):
    """softmax_dice_loss (losses.py:42-59): per class 1-(2*sum(sa*sb)+eps)/(sum(sa)+sum(sb)+eps), mean over classes.
    Gradient flows to ``a`` only (the reference detaches the target, losses.py:294)."""
    return _FusedLoss.apply(a, b.detach(), None, 2, False)[1]


def softmax_mse(a: torch.Tensor, b: torch.Tensor):
    """mean((softmax(a,1) - softmax(b,1))^2) — one scale of softmax_mse_loss (losses.py:82-87); target detached."""
    return _FusedLoss.apply(a, b.detach(), None, 3, False)[0]
