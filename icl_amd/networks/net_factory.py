"""Drop-in for the reference 2-D model factory (/root/reference/code/networks/net_factory.py:78-89):
``net_factory(net_type="unet", in_chns=1, class_num=3)`` -> module on the HIP device, or ``None`` for unknown types.
"swinunet" / "icl_swinunet" build the 2-D Swin-UNet with the configuration of
configs/swin_tiny_patch4_window7_224_lite.yaml (the reference reads it through yacs at import time, net_factory.py:75)."""
from __future__ import annotations

import torch

from ..optim import tag_model_parameters

from .unet import UNet
from .unet_icl import UNet_icl
from .vision_transformer import SwinUnet, default_config


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("icl_amd.net_factory: no HIP device visible; the reference factory calls .cuda() "
                           "(net_factory.py:80,84) and so does this one — there is no CPU model")
    return torch.device("cuda", torch.cuda.current_device())


def net_factory(net_type="unet", in_chns=1, class_num=3):
    if net_type == "unet":
        return tag_model_parameters(UNet(in_chns=in_chns, class_num=class_num, device=_device()))
    if net_type == "icl_unet":
        return tag_model_parameters(UNet_icl(in_chns=in_chns, class_num=class_num, device=_device()))
    if net_type in ("swinunet", "icl_swinunet"):     # net_factory.py:81-86: img_size [224, 224]
        return tag_model_parameters(SwinUnet(default_config(), img_size=224, num_classes=class_num, device=_device(), icl=net_type == "icl_swinunet"))
    return None
