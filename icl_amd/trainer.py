"""One ICL training iteration — the build's counterpart of the hot loop in
/root/reference/code/train_inherent_consistent_unet_3D_BraTS.py:99-121.

    outputs = model(volume[:labeled_bs], volume[labeled_bs:])
    loss    = dice + ce + aux + w_pse*pse + w_con*consistency          (:105-112; AMOS: w_pse = 0.1)
    zero_grad -> backward -> [gradient all-reduce when world_size > 1] -> SGD(momentum 0.9, wd 1e-4) step
    lr      = base_lr * (1 - iter/max_iter)**0.9, computed from the pre-increment iter, used from the next step

The reference trainers can also be used unchanged with ``icl_amd.networks.net_factory_3d`` and
``icl_amd.utils.losses``; this module exists so bench.py / tests / DDP have one
function to call, without the per-iteration ``.item()`` syncs of the reference logging (:131-133).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
from torch.nn.modules.loss import CrossEntropyLoss

from . import ops
from .networks.layers import BatchNormAct
from .optim import FusedSGD
from .utils import losses as L


@dataclass
class ICLConfig:
    num_classes: int = 2
    labeled_bs: int = 1
    base_lr: float = 0.01
    max_iterations: int = 30000
    momentum: float = 0.9
    weight_decay: float = 1e-4
    w_pse: float = 1.0     # 0.1 in the AMOS trainer (train_..._AMOS22.py:230)
    w_con: float = 10.0    # 50 in the 2-D trainer (train_inherent_consistent_unet_2D.py:127)
    patch_size: tuple = (96, 96, 96)   # (256, 256) selects the 2-D losses (AuxLoss / PseudoSoftLoss)
    factored_mlp2_grads: bool = True   # keep the 13,824^2 mlp2 weight gradients factored (ops.FactoredGrads)
    update_in_backward: bool = True    # single rank: SGD step of those matrices inside their backward pass (one read of the matrix)


class ICLTrainer:
    def __init__(self, model: torch.nn.Module, cfg: ICLConfig, ddp=None):
        self.model, self.cfg, self.ddp = model, cfg, ddp
        self.optimizer = FusedSGD(model.parameters(), lr=cfg.base_lr, momentum=cfg.momentum,
                                  weight_decay=cfg.weight_decay, step_scope=False)      # this class opens the step scope itself
        self.ce_loss = CrossEntropyLoss()
        self.dice_loss = L.DiceLoss(cfg.num_classes)
        if len(cfg.patch_size) == 3:
            self.aux_loss = L.AuxLoss3D(cfg.num_classes, cfg.patch_size)
            self.pse_loss = L.PseudoSoftLoss3D(cfg.num_classes, cfg.patch_size)
        else:
            self.aux_loss = L.AuxLoss(cfg.num_classes, cfg.patch_size)
            self.pse_loss = L.PseudoSoftLoss(cfg.num_classes, cfg.patch_size)
        self.iter_num = 0
        self.packed = ops.PackedWeights()
        BatchNormAct.share_counters(model)   # one vector for all BatchNorm step counters: one add per step
        self.graph = None
        self.graph_forward = None
        self.graph_update = None
        self.graph_update_rest = None
        self.use_graph = True     # False: launch eagerly although a captured graph exists (bench.py --launch auto compares the two)
        # data-parallel replay only: set `time_phases` and every step records four events on its stream — start, forward + backward +
        # pack() done, collectives done, update done; `phase_times()` turns the last step's into milliseconds (bench.py's config.ddp_phases)
        self.time_phases = False
        self._phase_events = None
        self.lr_dev = None

    def compute_loss(self, outputs, label_batch):
        cfg = self.cfg
        lab = label_batch[:cfg.labeled_bs]
        # CrossEntropyLoss()(out0, y) and DiceLoss(softmax(out0), y) (…BraTS.py:105-108) from one fused pass
        n_terms = 1 + len(outputs[2]) + len(outputs[3]) + len(outputs[4])
        if hasattr(self.aux_loss, "terms") and hasattr(self.pse_loss, "terms") and n_terms <= ops.LOSS_MULTI_MAX:
            # round 6: the ten reductions of the objective — CE + Dice on the logits, the AuxLoss3D / PseudoSoftLoss3D maps, the
            # softmax-MSE pairs — in ONE statistics launch + one finalize (and one gradient launch in backward): ops.fused_losses
            terms = ([(outputs[0], lab, 1)] + self.aux_loss.terms(outputs[2], lab) + self.pse_loss.terms(outputs[3], outputs[1])
                     + [(a, b, 3) for a, b in zip(outputs[3], outputs[4])])
            pairs = ops.fused_losses(terms)
            na_, np_ = len(outputs[2]), len(outputs[3])
            loss_ce, loss_dice = pairs[0]
            aux = [v for p in pairs[1:1 + na_] for v in p]                     # [ce_0, dice_0, ce_1, ...]
            pse = [p[1] for p in pairs[1 + na_:1 + na_ + np_]]
            con = [p[0] for p in pairs[1 + na_ + np_:]]
        else:
            loss_ce, loss_dice = ops.cross_entropy_dice_parts(outputs[0], lab, cfg.num_classes)
            aux = self.aux_loss.leaves(outputs[2], lab)
            pse = self.pse_loss.leaves(outputs[3], outputs[1])
            con = L.softmax_mse_leaves(outputs[3], outputs[4])
        # every term of the five-part objective as a leaf scalar, the weighted sums (loss = dice + ce + aux + w_pse pse + w_con con,
        # aux / pse / con = means over the scales) in ONE launch: out = [loss, aux, pse, con]
        leaves = [loss_dice, loss_ce] + aux + pse + con
        if len(leaves) > 16:      # more scales than the combine kernel takes: the class-level means, then the sum
            loss_aux = self.aux_loss(outputs[2], lab)
            loss_pse = self.pse_loss(outputs[3], outputs[1])
            loss_con = L.softmax_mse_loss(outputs[3], outputs[4])
            loss = loss_dice + loss_ce + loss_aux + cfg.w_pse * loss_pse + cfg.w_con * loss_con
            return loss, dict(dice=loss_dice, ce=loss_ce, aux=loss_aux, pse=loss_pse, con=loss_con)
        na, npse, ncon = len(aux), len(pse), len(con)
        nmaps = max(len(outputs[2]), 1)
        wa, wp, wc = [1.0 / nmaps] * na, [1.0 / max(npse, 1)] * npse, [1.0 / max(ncon, 1)] * ncon
        z = lambda k: [0.0] * k      # noqa: E731
        W = [[1.0, 1.0] + wa + [cfg.w_pse * v for v in wp] + [cfg.w_con * v for v in wc],
             z(2) + wa + z(npse) + z(ncon), z(2) + z(na) + wp + z(ncon), z(2) + z(na) + z(npse) + wc]
        out = ops.combine_scalars(leaves, W)
        return out[0], dict(dice=loss_dice, ce=loss_ce, aux=out[1], pse=out[2], con=out[3])

    def _forward_backward(self, volume_batch, label_batch, boundary=None):
        """``boundary()`` (capture only) is called between the loss and ``loss.backward()``."""
        cfg = self.cfg
        ops.StepRNG.begin_step()
        self.packed.begin_step()        # every convolution weight packed once, in one launch (ops.PackedWeights)
        self.optimizer.zero_grad(set_to_none=True)
        BatchNormAct.defer_counters()
        ops.DeferredBiasGrads.begin()
        ops.WgradLane.begin_step()      # per-weight use counts of the step (a lane gradient must be adopted, not accumulated)
        ops.DeferredWgradReduce.begin()  # (needs those counts: only adopted gradients are deferred)
        try:
            ops.FactoredGrads.world = self.ddp.world if (self.ddp is not None and self.ddp.active) else 1
            rows_default = ops.FactoredGrads.max_rows_gathered
            if self.ddp is not None and self.ddp.active:
                ops.FactoredGrads.max_rows_gathered = self.ddp.max_rows_gathered      # this reducer's crossover, for this step only
            # without gradient exchange the factors of a layer are final when its backward runs: update there (FusedSGD.update_in_backward)
            fuse = cfg.factored_mlp2_grads and cfg.update_in_backward and not (self.ddp is not None and self.ddp.active)
            ops.FactoredGrads.fused_optimizer = self.optimizer if fuse else None
            ops.FactoredGrads.uses = {} if fuse else None
            with ops.FactoredGrads(cfg.factored_mlp2_grads):
                outputs = self.model(volume_batch[:cfg.labeled_bs], volume_batch[cfg.labeled_bs:])
                loss, parts = self.compute_loss(outputs, label_batch)
                if boundary is not None:
                    boundary()
                ops.WgradLane.open = True
                loss.backward()
                ops.WgradLane.join()
                ops.DeferredWgradReduce.flush()      # the slab sums of every convolution weight gradient of the pass: one launch
        finally:
            ops.DeferredWgradReduce.pending = None
            if "rows_default" in locals():
                ops.FactoredGrads.max_rows_gathered = rows_default
            ops.WgradLane.open = False
            ops.WgradLane.uses = None
            ops.FactoredGrads.fused_optimizer = None
            ops.FactoredGrads.uses = None
            BatchNormAct.flush_counters()     # all num_batches_tracked increments of the step in one launch
            ops.DeferredBiasGrads.flush()     # all small-Linear bias gradients of the step in one launch
        parts = {k: v.detach() for k, v in parts.items()}
        parts["loss"] = loss.detach()
        return parts

    def _apply_update(self, part: str = "all"):
        """``part``: "all" (one rank, or nothing row-sharded), or the two halves of a data-parallel update — "sharded" = the updates
        of the row-sharded matrices (GradientReducer.sharded_params()), "rest" = everything else; the all-gathers of the updated rows
        are started between the two and run under "rest"."""
        if part in ("all", "sharded") and self.ddp is not None and self.ddp.active:
            self.optimizer.step_subset(self.ddp.sharded_params())
        if part == "sharded":
            return
        self.optimizer.step()
        self.packed.end_step()
        ops.StepRNG.end_step()

    def _step_body(self, volume_batch, label_batch, boundary=None):
        try:
            parts = self._forward_backward(volume_batch, label_batch, boundary)
        except BaseException:
            # an iteration that will not reach optimizer.step(): drop its queued updates, or every later zero_grad() raises
            self.optimizer.abandon_step()
            self.packed.end_step()
            raise
        if self.ddp is not None and self.ddp.active:
            self.ddp.reduce_gradients()
            self._apply_update("sharded")
            self.ddp.post_update(async_op=True)      # row-sharded matrix updates: the all-gather of the updated rows starts here ...
            self._apply_update("rest")               # ... and runs under the rest of the optimiser step
            self.ddp.finish_post_update()
        else:
            self._apply_update()
        return parts

    def _advance_lr(self):
        cfg = self.cfg
        lr = cfg.base_lr * (1.0 - self.iter_num / cfg.max_iterations) ** 0.9   # pre-increment iter, used from the next step
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        self.iter_num += 1

    def step(self, volume_batch: torch.Tensor, label_batch: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One iteration; returns the (device-resident, un-synced) loss terms.  After ``capture()`` the whole iteration
        (forward, losses, backward, optimiser) is one hipGraph replay: ~1400 kernel launches per step otherwise cost
        ~21 ms of host time, on par with the GPU time."""
        if self.lr_dev is not None:       # after capture() the kernels read the learning rate from device memory, replayed or not
            self.lr_dev.fill_(self.optimizer.param_groups[0]["lr"])
        if self.graph_update is not None and not self.use_graph:
            raise RuntimeError("a data-parallel trainer that has been captured cannot step eagerly (the reducer's buffers belong to "
                               "the graphs); build a second trainer for eager steps")
        if self.graph is not None and self.use_graph:
            if volume_batch.data_ptr() != self.static_vol.data_ptr():
                self.static_vol.copy_(volume_batch)
            if label_batch.data_ptr() != self.static_lab.data_ptr():
                self.static_lab.copy_(label_batch)
            ev = None
            if self.time_phases and self.graph_update is not None:
                ev = self._phase_events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
            if self.graph_forward is not None:
                self.graph_forward.replay()
            self.graph.replay()
            if self.graph_update is not None:    # data-parallel: collectives between the graphs
                if ev:
                    ev[1].record()
                self.ddp.communicate()
                if ev:
                    ev[2].record()
                self.graph_update.replay()           # unpack + the updates of the row-sharded matrices
                self.ddp.post_update(async_op=True)  # their all-gathers, under ...
                self.graph_update_rest.replay()      # ... the rest of the optimiser step
                self.ddp.finish_post_update()
                if ev:
                    ev[3].record()
            parts = self.static_out
        else:
            parts = self._step_body(volume_batch, label_batch)
        self._advance_lr()
        return parts

    def phase_times(self):
        """Milliseconds of the last data-parallel replayed step's phases (needs `time_phases`; synchronises): forward + backward + pack(),
        the collectives as the step's stream sees them (what is NOT hidden under compute: they are issued after backward), the update
        (both optimiser graphs and whatever of the row all-gathers is not hidden under the second one)."""
        ev = self._phase_events
        if not ev:
            return None
        ev[3].synchronize()
        return {"forward_backward_ms": ev[0].elapsed_time(ev[1]), "exposed_collective_ms": ev[1].elapsed_time(ev[2]),
                "update_ms": ev[2].elapsed_time(ev[3])}

    def capture(self, volume_batch: torch.Tensor, label_batch: torch.Tensor, warmup: int = 3):
        """Capture one full iteration into a hipGraph (torch.cuda.CUDAGraph).  Every per-step scalar the kernels need —
        the learning rate and the dropout seed — lives in device memory (FusedSGD.lr_dev, ops.StepRNG) so that a replay
        is a real training step: new masks, scheduled lr, updated weights.

        The ``warmup`` steps before the capture are REAL training steps on this batch (weights, momentum, BatchNorm running
        statistics, ``iter_num`` and the lr schedule advance): capture at the start of training, on the first batch.  The loss
        terms returned by ``step()`` after a capture are the graph's static output tensors — every replay overwrites them, so
        ``.clone()`` (or ``.item()``) what must outlive the next step.

        Data-parallel (``ddp`` given): two graphs.  The first holds forward, losses, backward and GradientReducer.pack()
        (gradients concatenated into flat buffers, factor rows scaled); the RCCL collectives on those persistent buffers are
        issued eagerly; the second graph holds the optimiser, reading the reduced buffers.  Nothing is captured while a
        collective is in flight and no collective is captured."""
        if warmup < 1:
            raise ValueError("capture() needs at least one warm-up step: the momentum buffers must exist before the capture, "
                             "otherwise the captured SGD kernels would re-initialise them at every replay")
        dev = volume_batch.device
        ddp = self.ddp if (self.ddp is not None and self.ddp.active) else None
        self.static_vol = volume_batch.clone()
        self.static_lab = label_batch.clone()
        self.lr_dev = torch.full((1,), float(self.optimizer.param_groups[0]["lr"]), dtype=torch.float32, device=dev)
        self.optimizer.lr_dev = self.lr_dev
        ops.StepRNG.enable(dev)
        if ddp is not None:
            ddp.static = True
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):   # allocator warm-up, momentum buffers, one-time kernel attributes, RCCL communicators
                self.lr_dev.fill_(self.optimizer.param_groups[0]["lr"])
                self._step_body(self.static_vol, self.static_lab)
                self._advance_lr()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # RCCL's watchdog thread polls its events from another thread: keep its calls out of the capture's error scope
        mode = dict(capture_error_mode="thread_local") if ddp is not None else {}
        # Forward (+ losses) and backward (+ optimiser) are captured as TWO graphs replayed back to back (one memory pool).  Reason:
        # the branch streams inside the aligners (ops.SideStream lanes) wait on the aligner stream in the forward pass and the aligner
        # stream waits on them in the backward pass; hipStreamEndCapture (ROCm 7.2) crashes on a capture in which two forked streams
        # wait on each other in both directions (tools/capture_probe.py: "pingpong*", "nested*", "lanes_autograd").  Per graph every
        # cross-stream edge between forked streams has one direction.
        pool = torch.cuda.graph_pool_handle()
        graph_f, graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        cap_f, cap_b = torch.cuda.graph(graph_f, pool=pool, **mode), torch.cuda.graph(graph, pool=pool, **mode)
        active = [cap_f]

        def boundary():
            active.pop().__exit__(None, None, None)
            cap_b.__enter__()
            active.append(cap_b)

        cap_f.__enter__()
        try:
            if ddp is None:
                self.static_out = self._step_body(self.static_vol, self.static_lab, boundary)
            else:
                self.static_out = self._forward_backward(self.static_vol, self.static_lab, boundary)
                ddp.pack()
        finally:
            active.pop().__exit__(None, None, None)
        self.graph_forward = graph_f
        self.graph_update = None
        if ddp is not None:
            ddp.rebind()
            ddp._captured = True
            # the optimiser as TWO graphs (round 5): the updates of the row-sharded matrices first, so that the eager all-gathers of
            # their rows overlap the replay of everything else
            update, rest = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(update, pool=pool, **mode):
                ddp.unpack()
                self._apply_update("sharded")
            with torch.cuda.graph(rest, pool=pool, **mode):
                self._apply_update("rest")
            self.graph_update, self.graph_update_rest = update, rest
        self.graph = graph
        return self


def optimizer_state_dict(optimizer):
    """Checkpoint helper for data-parallel runs: EVERY rank calls it (``FusedSGD.consolidate_momentum()`` is a collective when a
    step has row-sharded a momentum buffer); the returned dict is complete on every rank and ``state_dict()`` itself stays local."""
    if hasattr(optimizer, "consolidate_momentum"):
        optimizer.consolidate_momentum()
    return optimizer.state_dict()


def backbone_state_dict(model: torch.nn.Module):
    """The checkpoint the reference trainers save: every key without 'sspa'/'uscl' (…BraTS.py:158-162)."""
    from collections import OrderedDict
    return OrderedDict((k, v) for k, v in model.state_dict().items() if "sspa" not in k and "uscl" not in k)
