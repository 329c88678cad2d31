#!/usr/bin/env python3
"""When does which stream of the captured ICL step reach which point?  One-thread stamp kernels (tools/probe/stamp.hip: the GPU's 100 MHz
clock) are launched at marks of the forward (current stream) and, through tensor hooks, of the backward (the stream of the node that just
produced the gradient); they are captured into the step's hipGraph with everything else, so the printed timeline is the REPLAYED step,
which rocprofv3's serialised queues cannot show.   python tools/critical_path.py  (needs tools/probe/libstamp.so, see stamp.hip)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icl_amd import ops  # noqa: E402
from icl_amd.networks import unet_3D as U  # noqa: E402
from icl_amd.networks.unet_3D_icl import unet_3D_icl  # noqa: E402
from icl_amd.trainer import ICLConfig, ICLTrainer  # noqa: E402
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume  # noqa: E402

S = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libstamp.so"))
S.probe_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
slots = torch.zeros(512, dtype=torch.int64, device=dev)
names = []


def mark(name):
    if name not in names:
        names.append(name)
    i = names.index(name)
    S.probe_stamp(slots.data_ptr() + 8 * i, torch.cuda.current_stream().cuda_stream)


def grad_mark(t, name):
    """Identity view of ``t`` whose gradient arrival is stamped."""
    v = t.view_as(t)
    v.register_hook(lambda g: mark(name))
    return v


def run_backbone(self, x, heads=None):
    mark("forward start")
    c1, p1 = ops.skip_and_pool(self.conv1(x))
    c2, p2 = ops.skip_and_pool(self.conv2(grad_mark(p1, "B gradient of pool1 ready (conv2..4 backward done)")))
    c3, p3 = ops.skip_and_pool(self.conv3(p2))
    c3 = grad_mark(c3, "B skip gradient of c3 ready (up_concat3's convolutions and concat backward done)")
    c4, p4 = ops.skip_and_pool(self.conv4(p3))
    c4 = grad_mark(c4, "B skip gradient of c4 ready (up_concat4 backward done)")
    p4 = grad_mark(p4, "B gradient of pool4 ready (center backward done)")
    mark("F encoder done")
    center = self.dropout1(self.center(p4))
    up4 = self.up_concat4(c4, center)
    up3 = self.up_concat3(c3, up4)
    mark("F up3 done, aligners fork")
    up3 = grad_mark(up3, "B up3 gradient complete (decoder + aligners), deep backward starts")
    up3.register_hook(U._open_update_gate)
    a_in = [grad_mark(center, "B aligner gradient of center ready"), grad_mark(up4, "B aligner gradient of up4 ready"),
            grad_mark(up3, "B aligner gradient of up3 ready")]
    if os.environ.get("CP_DETACH_HEADS", "0") != "0":      # what would the step take if the deep backward did not wait for the aligners?
        a_in = [t.detach() for t in a_in]
    extra = heads(a_in)
    up2 = self.up_concat2(c2, grad_mark(up3, "B decoder gradient of up3 ready (up2/up1/final backward done)"))
    up1 = self.dropout2(self.up_concat1(c1, up2))
    out = self.final(up1)
    mark("F final conv done (main)")
    return out, [center, up4, up3], extra


U.UNet3DBackbone.run_backbone = run_backbone
_join = ops.SideStream.join


def join(self, outputs):
    if self.stream is not None and ops.SideStream._outer is None:
        with torch.cuda.stream(self.stream):
            mark("F aligner stream done")
        for k, sk in enumerate(self.children):
            with torch.cuda.stream(sk):
                mark(f"F lane {k + 1} done")
    _join(self, outputs)
    if ops.SideStream._outer is None:
        mark("F joined")


ops.SideStream.join = join

torch.manual_seed(1337)
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, base_lr=0.01, w_pse=1.0), None)
if os.environ.get("CP_ALIGNER_DETAIL", "0") != "0":
    # when does the backward of each aligner level reach the input of its token-axis MLP / of its attention block?
    for head in ("sspa", "uscl"):
        for i, cd in enumerate(getattr(model, head).class_decoders):
            cd.mlp2.register_full_backward_hook(lambda m, gi, go, t=f"B {head} level {i}: mlp2 backward done": mark(t))
            cd.attn.register_full_backward_hook(lambda m, gi, go, t=f"B {head} level {i}: attention backward done": mark(t))
    if os.environ.get("CP_ALIGNER_DETAIL", "0") == "2":
        # the links of sspa's query chain (the serial tail of the backward's forked phase)
        for i, cd in enumerate(model.sspa.class_decoders):
            for name, mod in (("query_convs", model.sspa.query_convs[i]), ("mlp", cd.mlp), ("norm2", cd.norm2), ("attn.proj", cd.attn.proj),
                              ("attn.fc_q", cd.attn.fc_q), ("attn.fc_kv", cd.attn.fc_kv), ("norm1_query", cd.norm1_query), ("norm1", cd.norm1),
                              ("norm_layers", model.sspa.norm_layers[i])):
                mod.register_full_backward_hook(lambda m, gi, go, t=f"B   sspa level {i}: {name} backward done": mark(t))
if os.environ.get("CP_ALIGNER_DETAIL", "0") != "0":
    # round 6: the fused query chain (ops._QueryAttend) — when does each level's backward start and end?
    _qb, _qn = ops._QueryAttend.backward, [0]

    def _stamped_backward(ctx, *gs):
        _qn[0] += 1
        k = (_qn[0] - 1) % 6
        tag = f"B query chain call {k} (R={ctx.shape[0] * ctx.shape[1]}, C={ctx.shape[2]}, full={ctx.cfg['full']})"
        mark(tag + " start")
        out = _qb(ctx, *gs)
        mark(tag + " end")
        return out
    ops._QueryAttend.backward = staticmethod(_stamped_backward)
_loss = tr.compute_loss


def compute_loss(outputs, label):
    r = _loss(outputs, label)
    mark("losses done, backward starts")
    return r


tr.compute_loss = compute_loss
_upd = tr._apply_update


def apply_update():
    mark("B backward done (main), optimiser starts")
    us = tr.optimizer._update_stream
    if us is not None and tr.optimizer._update_stream_used:
        tr.optimizer.flush_deferred(gate=True)
        with torch.cuda.stream(us):
            mark("update stream done")
    _upd()
    mark("step end")


tr._apply_update = apply_update
vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2, device=dev)
tr.capture(vol, lab, warmup=3)
for _ in range(10):
    tr.step(vol, lab)
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.time()
for _ in range(20):
    tr.step(vol, lab)
torch.cuda.synchronize()
print(f"replayed step with the stamps in it: {(time.time() - t0) / 20 * 1e3:.2f} ms")
v = slots.cpu().tolist()
base = v[names.index("forward start")]
for t, n in sorted((v[i], n) for i, n in enumerate(names)):
    print(f"{(t - base) / 100.0:9.1f} us  {n}")

