"""2-D U-Net with the ICL aligner heads — drop-in for the reference's ``UNet_icl``
(/root/reference/code/networks/unet_icl.py:196-252): ``forward(x_lab, x_unlab=None, inference=False)``.

Unlike the 3-D model the two streams are NOT batched together: the 2-D backbone uses BatchNorm with batch statistics,
so the labeled and the unlabeled half must see their own statistics exactly as in the reference (SURVEY.md §3.2,
Appendix A.10).  The two ``sspa`` calls still run in lock step (their BatchNorms are evaluated per input)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .aligner import InherentConsistent
from .unet import Decoder, Encoder, unet_params


class UNet_icl(nn.Module):  # noqa: N801 — reference class name
    def __init__(self, in_chns, class_num, device=None):
        super().__init__()
        params = unet_params(in_chns, class_num)
        self.encoder = Encoder(params, device)
        self.decoder = Decoder(params, device)
        ft, res = params["feature_chns"], params["input_resolution"]
        kw = dict(in_chans=(ft[3], ft[2], ft[1]), depths=params["depths"], patch_size=(2, 2),
                  input_resolution=(res[1], res[2], res[3]), num_classes=class_num, num_heads=params["num_heads"][::-1],
                  spatial_dims=2, device=device)
        self.sspa = InherentConsistent(**kw)
        self.uscl = InherentConsistent(**kw)

    def forward(self, x_lab, x_unlab=None, inference=False):
        output_lab, feats_lab = self.decoder(self.encoder(x_lab))
        if inference:
            return output_lab
        output_unlab, feats_unlab = self.decoder(self.encoder(x_unlab))
        both = [torch.cat([a, b], 0) for a, b in zip(feats_lab, feats_unlab)]
        (maps_lab, qs_lab), (maps_consis, _) = self.sspa.forward_labeled_pair(both, feats_lab[0].shape[0])
        maps_unlab, _ = self.uscl(feats_unlab, qs_lab, "unlabeled")
        return output_lab, output_unlab, maps_lab, maps_unlab, maps_consis
