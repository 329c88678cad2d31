"""SwinUNETR with the ICL aligner heads on the HIP kernels — drop-in for the reference's ``SwinUNETR_icl``
(/root/reference/code/networks/swinunetr_icl.py:30-357; BASELINE.json configs[3]).

``forward(x_lab, x_unlab=None, inference=False)``: truthy ``inference`` returns the labeled-stream logits; otherwise the
5-tuple ``(logits_lab, logits_unlab, feat_Maps_lab, feat_Maps_unlab, feat_Maps_consis)``.  The aligners receive the four
decoder maps ``[dec3, dec2, dec1, dec0]`` and use the first three (``depths[:3]``), with heads ``num_heads[::-1][:3]`` and
the learnable class query named ``guide_Q`` (:233-254, :403).
"""
from __future__ import annotations

import torch

from .. import ops
from .aligner import InherentConsistent
from .swinunetr import SwinUNETRBackbone


class SwinUNETR_icl(SwinUNETRBackbone):  # noqa: N801 — reference class name
    def __init__(self, img_size, in_channels, out_channels, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24),
                 feature_size=24, norm_name="instance", drop_rate=0.0, attn_drop_rate=0.0, dropout_path_rate=0.0,
                 normalize=True, use_checkpoint=False, spatial_dims=3, device=None):
        super().__init__(img_size, in_channels, out_channels, depths, num_heads, feature_size, norm_name, drop_rate,
                         attn_drop_rate, dropout_path_rate, normalize, use_checkpoint, spatial_dims, device)
        f = feature_size
        ori = self.img_size[0]
        kw = dict(in_chans=(8 * f, 4 * f, 2 * f), depths=tuple(depths[:3]), patch_size=(2, 2, 2),
                  input_resolution=(ori // 16, ori // 8, ori // 4), num_classes=out_channels,
                  num_heads=tuple(num_heads[::-1][:3]), device=device, query_name="guide_Q")
        self.sspa = InherentConsistent(**kw)
        self.uscl = InherentConsistent(**kw)

    def forward(self, x_lab, x_unlab=None, inference=False):
        if inference:
            return self.run_backbone(x_lab)[0]
        # Both streams share the weights and every backbone operator is per-sample (LayerNorm per token, attention per
        # window, InstanceNorm per sample), so they run as one batch — same per-sample results as the reference's two
        # passes (:313-347) with half the launches and one weight-gradient product per layer.
        bl = x_lab.shape[0]

        def heads(feats):
            # the aligners need only dec3..dec1: they run on a second stream next to the 48^3 / 96^3 decoder stages
            with ops.SideStream(feats) as side:
                (maps_lab, qs_lab), (maps_con, _) = self.sspa.forward_labeled_pair(feats, bl)
                maps_unlab, _ = self.uscl([ops.split_batch(t, bl)[1] for t in feats], qs_lab, "unlabeled")
            return side, maps_lab, maps_unlab, maps_con

        logits, _, (side, maps_lab, maps_unlab, maps_con) = self.run_backbone(ops.cat_batch(x_lab, x_unlab), heads)
        side.join(maps_lab + maps_unlab + maps_con)
        logits_lab, logits_unlab = ops.split_batch(logits, bl)
        return logits_lab, logits_unlab, maps_lab, maps_unlab, maps_con
