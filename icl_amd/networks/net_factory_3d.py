"""Drop-in for the reference model factory (/root/reference/code/networks/net_factory_3d.py:39-68).

``net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2)`` returns a module already on the HIP device
in train mode, or ``None`` for an unknown type — same contract as the reference.  Differences that make
the unchanged trainers importable: no import of the four modules the reference
tree lacks, and ``parse_known_args`` instead of ``parse_args`` so the trainer's own flags do not abort.
"""
from __future__ import annotations

import argparse

import torch

from ..optim import tag_model_parameters
from .swinunetr import SwinUNETR
from .swinunetr_icl import SwinUNETR_icl
from .unet_3D import unet_3D
from .unet_3D_icl import unet_3D_icl

parser = argparse.ArgumentParser(add_help=False)
parser.add_argument("--roi_x", default=96, type=int)
parser.add_argument("--roi_y", default=96, type=int)
parser.add_argument("--roi_z", default=96, type=int)
parser.add_argument("--num_classes", type=int, default=2)
parser.add_argument("--feature_size", default=48, type=int)
parser.add_argument("--in_channels", default=1, type=int)
parser.add_argument("--dropout_path_rate", default=0.0, type=float)
parser.add_argument("--use_checkpoint", action="store_true")
args, _unknown = parser.parse_known_args()


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("icl_amd.net_factory_3d: no HIP device visible; the reference factory calls .cuda() "
                           "(net_factory_3d.py:41-43) and so does this one — there is no CPU model")
    return torch.device("cuda", torch.cuda.current_device())


def net_factory_3d(net_type="unet_3D", in_chns=1, class_num=2):
    if net_type == "unet_3D":
        return tag_model_parameters(unet_3D(n_classes=class_num, in_channels=in_chns, device=_device()))
    if net_type == "unet_3D_icl":
        return tag_model_parameters(unet_3D_icl(n_classes=class_num, in_channels=in_chns, device=_device()))
    if net_type in ("swinunetr", "swinunetr_icl"):   # net_factory_3d.py:44-63: shapes come from the parser, not the arguments
        cls = SwinUNETR if net_type == "swinunetr" else SwinUNETR_icl
        return tag_model_parameters(cls(img_size=(args.roi_x, args.roi_y, args.roi_z), in_channels=args.in_channels, out_channels=args.num_classes,
                   feature_size=args.feature_size, drop_rate=0.0, attn_drop_rate=0.0, dropout_path_rate=args.dropout_path_rate,
                   use_checkpoint=args.use_checkpoint, device=_device()))
    return None
