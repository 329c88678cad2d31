"""Parameter-holding building blocks whose forward runs on the HIP kernels (icl_amd.ops).

Module/attribute names are chosen so that ``state_dict()`` keys equal the reference's
(/root/reference/code/networks/utils.py:99-123,260-276 and unet_3D_icl.py:155-345): checkpoints saved by
the reference trainers load here and vice versa.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


class Conv3d(nn.Module):
    """nn.Conv3d(cin, cout, k, stride 1, padding k//2) parameters; forward = MFMA implicit-GEMM kernel."""

    def __init__(self, cin, cout, ks=3, bias=True, groups=1, device=None, kaiming_normal=False, feeds_instance_norm=False):
        super().__init__()
        self.cin, self.cout, self.ks, self.groups = cin, cout, ks, groups
        self.feeds_instance_norm = feeds_instance_norm
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, ks, ks, ks, device=device))
        self.bias = nn.Parameter(torch.empty(cout, device=device)) if bias else None
        fan_in = (cin // groups) * ks ** 3
        with torch.no_grad():
            if kaiming_normal:  # networks_other.py:64-69 via init_weights('kaiming')
                self.weight.normal_(0.0, math.sqrt(2.0 / fan_in))
            else:  # torch default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                self.weight.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            if self.bias is not None:
                self.bias.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def forward(self, x):
        if self.groups != 1:
            return ops.depthwise_conv3d(x, self.weight)
        return ops.conv3d(x, self.weight, self.bias, self.feeds_instance_norm)


class InstanceNormReLU(nn.Module):
    """nn.InstanceNorm3d(c) (eps 1e-5, affine=False) followed by ReLU — one fused kernel pair."""

    def forward(self, x):
        return ops.instance_norm_relu(x, relu=True)


class _Identity(nn.Module):
    def forward(self, x):
        return x


def _hooked(*modules) -> bool:
    """True when a forward / forward-pre hook is attached to one of the modules, or globally (nn.modules.module.register_module_*):
    such hooks expect `Module.__call__` and a tensor output, so the callers below then take the materialised path."""
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks or getattr(_m, "_global_forward_hooks_always_called", None):
        return True
    return any(m._forward_hooks or m._forward_pre_hooks for m in modules)


class ConvBlock(nn.Sequential):
    """Sequential(Conv3d, InstanceNorm3d, ReLU): keys '0.weight', '0.bias' (utils.py:104-106).
    Index 1 fuses norm+ReLU; index 2 is kept as a no-op so the child layout matches the reference."""

    def __init__(self, cin, cout, device=None):
        super().__init__(Conv3d(cin, cout, 3, device=device, kaiming_normal=True, feeds_instance_norm=True),
                         InstanceNormReLU(), _Identity())

    def _stock(self) -> bool:
        """The fused operator stands for exactly these three children with nothing attached: a child swapped through the
        reference-compatible Sequential indices, or a forward hook on one of them, must keep working."""
        c, n, a = self[0], self[1], self[2]
        return (type(c) is Conv3d and type(n) is InstanceNormReLU and type(a) is _Identity and c.groups == 1 and c.ks == 3
                and c.feeds_instance_norm and not (c._forward_hooks or c._forward_pre_hooks or n._forward_hooks or n._forward_pre_hooks
                                                   or a._forward_hooks or a._forward_pre_hooks))

    def forward(self, x):
        if not self._stock():
            return super().forward(x)
        # the three children as ONE operator: the convolution's epilogue hands the normalisation its statistics (ops.py)
        return ops.conv3d_instance_norm_act(x, self[0].weight, self[0].bias, act=1)

    def forward_lazy(self, x):
        """The same block with its InstanceNorm + ReLU DEFERRED to the consumers of the output (``ops.LazyAct``) where that pays — big
        volumes inside a step scope — and the ordinary tensor otherwise.  Only for outputs whose consumers are ``ops.skip_and_pool``,
        ``ops.upsample2x_concat`` and ``ops.conv1x1_lazy`` (the `conv2` blocks of the 3-D U-Net's upper levels)."""
        if not self._stock():
            return super().forward(x)
        return ops.conv3d_instance_norm_act_lazy(x, self[0].weight, self[0].bias)


class UnetConv3(nn.Module):
    """Two ConvBlocks (networks/utils.py:99-123)."""

    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.conv1 = ConvBlock(cin, cout, device)
        self.conv2 = ConvBlock(cout, cout, device)

    def forward(self, x, lazy: bool = False):
        """``lazy``: the output may be an ``ops.LazyAct`` (see ConvBlock.forward_lazy); the input may be one never."""
        h = self.conv1(x)
        # the deferred form bypasses conv2's `__call__` and hands a LazyAct to whoever looks at this block's output: only when nothing
        # is hooked on this block, on conv2 or globally (hooks on conv2's three children: ConvBlock._stock)
        if lazy and not _hooked(self, self.conv2):
            return self.conv2.forward_lazy(h)
        return self.conv2(h)


class UnetUp3_CT(nn.Module):
    """Trilinear x2 of the deeper map written straight into the concat buffer, then UnetConv3
    (networks/utils.py:260-276); channel order [skip, upsampled]."""

    def __init__(self, in_size, out_size, device=None):
        super().__init__()
        self.conv = UnetConv3(in_size + out_size, out_size, device)
        self.up = _Identity()  # nn.Upsample has no parameters; kept for module-tree parity

    def forward(self, skip, deep, lazy: bool = False):
        """``skip`` / ``deep`` may be ``ops.LazyAct`` (their normalisation is applied while the concat buffer is written)."""
        return self.conv(ops.upsample2x_concat(skip, deep), lazy and not _hooked(self))


class Dropout3(nn.Module):
    """nn.Dropout(p) on the HIP counter-based mask kernel (unet_3D_icl.py:67-68)."""

    def __init__(self, p=0.3):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return ops.dropout(x, self.p)


class Conv2d(nn.Module):
    """nn.Conv2d(cin, cout, k in {1,3}, padding k//2) parameters (networks/unet_icl.py:46-52,82,173); the forward embeds
    the 2-D stencil in the dz = 1 plane of a 3-D kernel and runs the MFMA convolution on a D = 1 volume."""

    def __init__(self, cin, cout, ks=3, bias=True, groups=1, device=None, feeds_batch_norm=False):
        super().__init__()
        self.cin, self.cout, self.ks, self.groups = cin, cout, ks, groups
        self.feeds_batch_norm = feeds_batch_norm
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, ks, ks, device=device))
        self.bias = nn.Parameter(torch.empty(cout, device=device)) if bias else None
        fan_in = (cin // groups) * ks * ks
        with torch.no_grad():  # torch default init (the reference never re-initialises its 2-D nets)
            self.weight.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            if self.bias is not None:
                self.bias.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    def forward(self, x):
        if self.groups != 1:
            return ops.depthwise_conv2d(x, self.weight)
        # a batch-statistics BatchNorm removes the per-channel mean: the bias gradient is exactly zero in training
        return ops.conv2d(x, self.weight, self.bias, self.feeds_batch_norm and self.training)


class BatchNormAct(nn.Module):
    """nn.BatchNorm{2,3}d parameters/buffers fused with the activation that follows (act: 0 none, 1 ReLU, 2 LeakyReLU)."""

    def __init__(self, c, act, device=None):
        super().__init__()
        self.act = act
        self.weight = nn.Parameter(torch.ones(c, device=device))
        self.bias = nn.Parameter(torch.zeros(c, device=device))
        self.register_buffer("running_mean", torch.zeros(c, device=device))
        self.register_buffer("running_var", torch.ones(c, device=device))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long, device=device))

    # ICLTrainer collects the ``num_batches_tracked += 1`` of all BatchNorm layers of a step (18 one-element launches in the
    # 3-D ICL model) into one multi-tensor add: ``deferred`` maps id(counter) -> [counter, increments] while a step is open.
    deferred = None

    @classmethod
    def defer_counters(cls):
        cls.deferred = {}

    @classmethod
    def flush_counters(cls):
        pending, cls.deferred = cls.deferred, None
        if not pending:
            return
        pool = cls._shared
        if pool is not None and all(k in pool["index"] for k in pending):
            # all counters live in one int64 vector (share_counters): one add of a cached increment vector
            counts = [0] * pool["vector"].numel()
            for k, (_, n) in pending.items():
                counts[pool["index"][k]] = n
            key = tuple(counts)
            inc = pool["incs"].get(key)
            if inc is None:
                inc = pool["incs"][key] = torch.tensor(counts, dtype=torch.long, device=pool["vector"].device)
            pool["vector"].add_(inc)
        else:
            torch._foreach_add_([t for t, _ in pending.values()], [n for _, n in pending.values()])

    _shared = None

    @classmethod
    def share_counters(cls, model):
        """Re-home the ``num_batches_tracked`` buffers of all BatchNorm layers of ``model`` as 0-dim views of ONE int64 vector
        (same values, same ``state_dict`` keys), so that a step's increments are a single launch instead of one per layer."""
        mods = [m for m in model.modules() if isinstance(m, BatchNormAct)]
        if not mods:
            cls._shared = None
            return
        vec = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in mods]).to(torch.long)
        index = {}
        for i, m in enumerate(mods):
            m._buffers["num_batches_tracked"] = vec[i]
            index[id(m._buffers["num_batches_tracked"])] = i
        cls._shared = {"vector": vec, "index": index, "incs": {}}

    def forward(self, x):
        if self.training:
            d = BatchNormAct.deferred
            if d is None:
                self.num_batches_tracked += 1
            else:
                d.setdefault(id(self.num_batches_tracked), [self.num_batches_tracked, 0])[1] += 1
        return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.act)
