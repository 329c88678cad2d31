"""2-D U-Net on the HIP kernels — drop-in for the reference's ``UNet`` (/root/reference/code/networks/unet.py:305-322)
and the Encoder/Decoder it shares with ``UNet_icl`` (networks/unet_icl.py:39-193).  Same constructor, same 136-key
``state_dict`` (``encoder.in_conv.conv_conv.0.weight`` ...), BatchNorm2d + LeakyReLU(0.01) blocks, dropout
0.05..0.5, and the reference's actual up-sampling path: ``Decoder`` never forwards ``bilinear`` to ``UpBlock``, so every
UpBlock is conv1x1 + bilinear x2 with align_corners=True."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .layers import BatchNormAct, Conv2d, Dropout3, _Identity


class ConvBlock(nn.Module):
    """Sequential(Conv2d, BN, LeakyReLU, Dropout, Conv2d, BN, LeakyReLU) — unet_icl.py:39-57; BN+LeakyReLU are fused,
    indices 2 and 6 stay as no-ops so the key layout ('conv_conv.{0,1,4,5}.*') matches."""

    def __init__(self, cin, cout, dropout_p, device=None):
        super().__init__()
        self.conv_conv = nn.Sequential(
            Conv2d(cin, cout, 3, device=device, feeds_batch_norm=True), BatchNormAct(cout, 2, device), _Identity(),
            Dropout3(dropout_p),
            Conv2d(cout, cout, 3, device=device, feeds_batch_norm=True), BatchNormAct(cout, 2, device), _Identity())

    def forward(self, x):
        return self.conv_conv(x)


class _MaxPool2d(nn.Module):
    def forward(self, x):
        return ops.max_pool2d_2(x)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, dropout_p, device=None):
        super().__init__()
        self.maxpool_conv = nn.Sequential(_MaxPool2d(), ConvBlock(cin, cout, dropout_p, device))

    def forward(self, x):
        return self.maxpool_conv(x)


class UpBlock(nn.Module):
    """conv1x1 -> bilinear x2 (align_corners=True) -> cat([skip, up]) -> ConvBlock (unet_icl.py:75-96)."""

    def __init__(self, c1, c2, cout, dropout_p, device=None):
        super().__init__()
        self.conv1x1 = Conv2d(c1, c2, 1, device=device)
        self.up = _Identity()
        self.conv = ConvBlock(c2 * 2, cout, dropout_p, device)

    def forward(self, x1, x2):
        x1 = self.conv1x1(x1)
        x1 = ops.bilinear_resize(x1, (x1.shape[2] * 2, x1.shape[3] * 2), align_corners=True)
        return self.conv(torch.cat([x2, x1], dim=1))


class Encoder(nn.Module):
    def __init__(self, params, device=None):
        super().__init__()
        self.params = params
        ft, dp = params["feature_chns"], params["dropout"]
        assert len(ft) == 5
        self.in_conv = ConvBlock(params["in_chns"], ft[0], dp[0], device)
        self.down1 = DownBlock(ft[0], ft[1], dp[1], device)
        self.down2 = DownBlock(ft[1], ft[2], dp[2], device)
        self.down3 = DownBlock(ft[2], ft[3], dp[3], device)
        self.down4 = DownBlock(ft[3], ft[4], dp[4], device)

    def forward(self, x):
        x0 = self.in_conv(x)
        x1 = self.down1(x0)
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x4 = self.down4(x3)
        return [x0, x1, x2, x3, x4]


class Decoder(nn.Module):
    """Returns (logits, [x_1, x_2, x_3]) like unet_icl.py:176-193; the plain UNet keeps only the logits."""

    def __init__(self, params, device=None):
        super().__init__()
        self.params = params
        ft = params["feature_chns"]
        self.up1 = UpBlock(ft[4], ft[3], ft[3], 0.0, device)
        self.up2 = UpBlock(ft[3], ft[2], ft[2], 0.0, device)
        self.up3 = UpBlock(ft[2], ft[1], ft[1], 0.0, device)
        self.up4 = UpBlock(ft[1], ft[0], ft[0], 0.0, device)
        self.out_conv = Conv2d(ft[0], params["class_num"], 3, device=device)

    def forward(self, feature):
        x0, x1, x2, x3, x4 = feature
        x_1 = self.up1(x4, x3)
        x_2 = self.up2(x_1, x2)
        x_3 = self.up3(x_2, x1)
        x = self.up4(x_3, x0)
        return self.out_conv(x), [x_1, x_2, x_3]


def unet_params(in_chns, class_num):
    return {"in_chns": in_chns, "feature_chns": [16, 32, 64, 128, 256], "input_resolution": [16, 32, 64, 128, 256],
            "num_heads": (2, 4, 8), "depths": (2, 2, 2), "dropout": [0.05, 0.1, 0.2, 0.3, 0.5], "class_num": class_num,
            "bilinear": False, "acti_func": "relu"}


class UNet(nn.Module):
    def __init__(self, in_chns, class_num, device=None):
        super().__init__()
        params = unet_params(in_chns, class_num)
        self.encoder = Encoder(params, device)
        self.decoder = Decoder(params, device)

    def forward(self, x):
        return self.decoder(self.encoder(x))[0]
