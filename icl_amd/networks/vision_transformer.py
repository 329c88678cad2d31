"""2-D Swin-UNet with the ICL aligner heads — drop-in for the reference's ``SwinUnet``
(/root/reference/code/networks/vision_transformer.py:32-108).

``SwinUnet(config, img_size=224, num_classes=...)``: ``config`` is the reference's yacs node (or any object with the same
attributes; ``default_config()`` reproduces configs/swin_tiny_patch4_window7_224_lite.yaml over networks/config.py).
``forward(x_lab, x_unlab=None, inference=False)``: grey images are repeated to three channels (:91,97-98); the aligners take the
decoder TOKENS of the 14^2 / 28^2 / 56^2 stages directly (:247-248), with fixed sizes (384, 192, 96) / (14, 28, 56) / heads
(24, 12, 6) (:55-58).  Both streams run as one batch (every backbone operator is per-token, per-window or per-sample).
"""
from __future__ import annotations

import copy
import types

import torch
import torch.nn as nn

from .. import ops
from .aligner import InherentConsistent
from .swinunet_icl import SwinTransformerSys


def default_config():
    ns = types.SimpleNamespace
    return ns(DATA=ns(IMG_SIZE=224),
              MODEL=ns(DROP_RATE=0.0, DROP_PATH_RATE=0.2, PRETRAIN_CKPT=None,
                       SWIN=ns(PATCH_SIZE=4, IN_CHANS=3, EMBED_DIM=96, DEPTHS=[2, 2, 2, 2], NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=7,
                               MLP_RATIO=4.0, QKV_BIAS=True, QK_SCALE=False, APE=False, PATCH_NORM=True)),
              TRAIN=ns(USE_CHECKPOINT=False))


class SwinUnet(nn.Module):
    def __init__(self, config=None, img_size=224, num_classes=21843, zero_head=False, vis=False, device=None, icl=True):
        super().__init__()
        config = config if config is not None else default_config()
        self.num_classes, self.config = num_classes, config
        sw = config.MODEL.SWIN
        self.swin_unet = SwinTransformerSys(img_size=config.DATA.IMG_SIZE, patch_size=sw.PATCH_SIZE, in_chans=sw.IN_CHANS,
                                            num_classes=num_classes, embed_dim=sw.EMBED_DIM, depths=sw.DEPTHS, num_heads=sw.NUM_HEADS,
                                            window_size=sw.WINDOW_SIZE, mlp_ratio=sw.MLP_RATIO, qkv_bias=sw.QKV_BIAS,
                                            qk_scale=sw.QK_SCALE or None, drop_rate=config.MODEL.DROP_RATE,
                                            drop_path_rate=config.MODEL.DROP_PATH_RATE, ape=sw.APE, patch_norm=sw.PATCH_NORM,
                                            use_checkpoint=config.TRAIN.USE_CHECKPOINT, device=device)
        self.icl = icl
        if icl:
            kw = dict(in_chans=(384, 192, 96), depths=(2, 2, 2), patch_size=sw.PATCH_SIZE, input_resolution=(14, 28, 56),
                      num_classes=num_classes, num_heads=(24, 12, 6), spatial_dims=2, device=device, tokenized_input=True)
            self.sspa = InherentConsistent(**kw)
            self.uscl = InherentConsistent(**kw)

    @staticmethod
    def _rgb(x):
        return x.repeat(1, 3, 1, 1) if x.shape[1] == 1 else x

    def forward(self, x_lab, x_unlab=None, inference=False):
        if inference or not self.icl:
            return self.swin_unet.run(self._rgb(x_lab))[0]
        bl = x_lab.shape[0]
        out, feats = self.swin_unet.run(torch.cat([self._rgb(x_lab), self._rgb(x_unlab)], 0))
        (maps_lab, qs_lab), (maps_con, _) = self.sspa.forward_labeled_pair(feats, bl)
        maps_unlab, _ = self.uscl([ops.split_batch(f, bl)[1] for f in feats], qs_lab, "unlabeled")
        out_lab, out_unlab = ops.split_batch(out, bl)
        return out_lab, out_unlab, maps_lab, maps_unlab, maps_con

    def load_from(self, config):
        """vision_transformer.py:110-146: ImageNet Swin-T checkpoint -> encoder, mirrored into the decoder stages."""
        path = config.MODEL.PRETRAIN_CKPT
        if path is None:
            print("none pretrain")
            return
        pretrained = torch.load(path, map_location=next(self.parameters()).device)
        if "model" not in pretrained:
            pretrained = {k[17:]: v for k, v in pretrained.items() if "output" not in k[17:]}
            self.swin_unet.load_state_dict(pretrained, strict=False)
            return
        pretrained = pretrained["model"]
        model_dict = self.swin_unet.state_dict()
        full = copy.deepcopy(pretrained)
        for k, v in pretrained.items():
            if "layers." in k:
                full["layers_up." + str(3 - int(k[7:8])) + k[8:]] = v
        for k in list(full.keys()):
            if k in model_dict and full[k].shape != model_dict[k].shape:
                del full[k]
        self.swin_unet.load_state_dict(full, strict=False)
