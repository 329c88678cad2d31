"""Drop-in for the reference 2-D model factory (/root/reference/code/networks/net_factory.py:78-89):
``net_factory(net_type="unet", in_chns=1, class_num=3)`` -> module on the HIP device, or ``None`` for unknown types.
The Swin-UNet variants ("swinunet", "icl_swinunet") are outside the hot path."""
from __future__ import annotations

import torch

from .unet import UNet
from .unet_icl import UNet_icl


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("icl_amd.net_factory: no HIP device visible; the reference factory calls .cuda() "
                           "(net_factory.py:80,84) and so does this one — there is no CPU model")
    return torch.device("cuda", torch.cuda.current_device())


def net_factory(net_type="unet", in_chns=1, class_num=3):
    if net_type == "unet":
        return UNet(in_chns=in_chns, class_num=class_num, device=_device())
    if net_type == "icl_unet":
        return UNet_icl(in_chns=in_chns, class_num=class_num, device=_device())
    if net_type in ("swinunet", "icl_swinunet"):
        raise NotImplementedError("2D Swin-UNet(-ICL) is SURVEY.md §8 row f4, not part of the hot path")
    return None
