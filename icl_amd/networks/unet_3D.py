"""Plain 3D U-Net backbone on the HIP kernels — drop-in for the reference's ``unet_3D``
(/root/reference/code/networks/unet_3D.py:20-94): same constructor, same 38-key ``state_dict``,
``forward(inputs) -> logits``.  It is the inference/checkpoint target of the ICL trainers
(SURVEY.md §0.8)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .layers import Conv3d, Dropout3, UnetConv3, UnetUp3_CT, _Identity


def _open_update_gate(grad):
    from .. import ops
    opt = ops.FactoredGrads.fused_optimizer
    if opt is not None and hasattr(opt, "flush_deferred"):
        opt.flush_deferred(gate=True)
    return None


def _open_decoder_gate(grad):
    """The upper decoder's own gradient of up3 is complete: from here the step's stream only waits for the aligners (FusedSGD.open_gate)."""
    from .. import ops
    opt = ops.FactoredGrads.fused_optimizer
    if opt is not None and hasattr(opt, "open_gate"):
        opt.open_gate()
    return None


class UNet3DBackbone(nn.Module):
    """Everything ``unet_3D`` and ``unet_3D_icl`` share (unet_3D_icl.py:28-68): 5 encoder stages,
    4 decoder stages, 1x1x1 classifier, two Dropout(0.3)."""

    def __init__(self, feature_scale=4, n_classes=21, is_deconv=True, in_channels=3, is_batchnorm=True, device=None):
        super().__init__()
        if not is_batchnorm:
            raise NotImplementedError("the reference only ever builds the InstanceNorm variant (is_batchnorm=True)")
        self.is_deconv, self.in_channels = is_deconv, in_channels
        self.is_batchnorm, self.feature_scale = is_batchnorm, feature_scale
        f = [int(x / feature_scale) for x in (64, 128, 256, 512, 1024)]
        self.filters = f
        self.conv1 = UnetConv3(in_channels, f[0], device)
        self.maxpool1 = _Identity()  # pooling is parameter-free; see _pool below
        self.conv2 = UnetConv3(f[0], f[1], device)
        self.maxpool2 = _Identity()
        self.conv3 = UnetConv3(f[1], f[2], device)
        self.maxpool3 = _Identity()
        self.conv4 = UnetConv3(f[2], f[3], device)
        self.maxpool4 = _Identity()
        self.center = UnetConv3(f[3], f[4], device)
        self.up_concat4 = UnetUp3_CT(f[4], f[3], device)
        self.up_concat3 = UnetUp3_CT(f[3], f[2], device)
        self.up_concat2 = UnetUp3_CT(f[2], f[1], device)
        self.up_concat1 = UnetUp3_CT(f[1], f[0], device)
        self.final = Conv3d(f[0], n_classes, 1, device=device, kaiming_normal=True)
        self.dropout1 = Dropout3(0.3)
        self.dropout2 = Dropout3(0.3)

    def run_backbone(self, x, heads=None):
        """One stream: returns (logits, [center, up4, up3]) — unet_3D_icl.py:100-117.  ``heads`` (the ICL model's aligner calls) is
        invoked on the three deep maps as soon as they exist and its result returned as a third value."""
        from .. import ops
        # every encoder output feeds its level's skip connection AND the next level's pooling: ops.skip_and_pool returns both so that
        # the backward adds the two gradients inside the pooling backward's pass
        # Round 5: the encoder outputs and the two upper decoder outputs may come back as ops.LazyAct — the raw output of the block's
        # second convolution with its InstanceNorm + ReLU applied by the consumers (pooling, skip / up-sampling into the concat buffer,
        # the `final` convolution) while they read it; ops decides per tensor (>= 48^3 voxels, inside a step scope).  center / up4 / up3
        # feed the aligner heads and stay ordinary tensors.
        c1, p1 = ops.skip_and_pool(self.conv1(x, lazy=True))
        c2, p2 = ops.skip_and_pool(self.conv2(p1, lazy=True))
        c3, p3 = ops.skip_and_pool(self.conv3(p2, lazy=True))
        c4, p4 = ops.skip_and_pool(self.conv4(p3, lazy=True))
        center = self.dropout1(self.center(p4))
        up4 = self.up_concat4(c4, center)
        up3 = self.up_concat3(c3, up4)
        if heads is not None and up3.requires_grad:
            # backward: once up3's gradient is complete the aligners and the 48^3 / 96^3 decoder stages are done and the small deep levels
            # follow — the point from which an optimiser with gated update placement streams its big matrices (FusedSGD.flush_deferred)
            up3.register_hook(_open_update_gate)
        extra = heads([center, up4, up3]) if heads is not None else None
        up3_dec = up3
        if heads is not None and up3.requires_grad:
            up3_dec = up3.view_as(up3)      # (a view: its gradient is the upper decoder's share alone)
            up3_dec.register_hook(_open_decoder_gate)
        up2 = self.up_concat2(c2, up3_dec, lazy=True)
        up1 = self.up_concat1(c1, up2, lazy=True)
        drop = self.dropout2.p if (self.dropout2.training and self.dropout2.p > 0.0) else 0.0
        if isinstance(up1, ops.LazyAct):
            # final(dropout2(up1)) in one pass over the RAW output of up1's last convolution: InstanceNorm + ReLU, the mask, the 1x1x1 product
            logits = ops.conv1x1_lazy(up1, self.final.weight, self.final.bias, drop)
        elif drop > 0.0:
            # final(dropout2(up1)) with the mask applied inside the 1x1x1 convolution's passes (ops.dropout_conv1x1)
            logits = ops.dropout_conv1x1(up1, self.final.weight, self.final.bias, drop)
        else:
            logits = self.final(self.dropout2(up1))
        if heads is not None:
            return logits, [center, up4, up3], extra
        return logits, [center, up4, up3]


class unet_3D(UNet3DBackbone):  # noqa: N801 — reference class name
    def forward(self, inputs):
        return self.run_backbone(inputs)[0]

    @staticmethod
    def apply_argmax_softmax(pred):
        return torch.softmax(pred, dim=1)
