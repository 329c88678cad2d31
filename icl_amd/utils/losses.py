"""Loss functions of the ICL trainers with the reference's names and signatures
(/root/reference/code/utils/losses.py: DiceLoss :195-231, AuxLoss3D :254-271, PseudoSoftLoss3D :287-299,
softmax_mse_loss :68-90, softmax_dice_loss :42-59, dice_loss1 :22-30).  Every function returns a 0-dim fp32
tensor attached to autograd.

Differences from the reference that do not change results: no per-class ``.item()`` host syncs
(losses.py:229 forces nc device→host round trips per call), and the trilinear resize runs on the HIP kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.modules.loss import CrossEntropyLoss

from .. import ops


def dice_loss1(score, target):
    """losses.py:22-30 — plain sums in the denominator."""
    target = target.float()
    smooth = 1e-5
    intersect = torch.sum(score * target)
    return 1 - (2 * intersect + smooth) / (torch.sum(score) + torch.sum(target) + smooth)


def softmax_dice_loss(input_logits, target_logits):
    """losses.py:42-59."""
    assert input_logits.size() == target_logits.size()
    return ops.soft_dice_loss(input_logits, target_logits)


def _mean_of(leaves):
    """Mean of a list of 0-dim loss terms in one launch (ops.combine_scalars) instead of a chain of one-element adds and a divide."""
    if len(leaves) == 1:
        return leaves[0]
    return ops.combine_scalars(leaves, [[1.0 / len(leaves)] * len(leaves)])[0]


def softmax_mse_leaves(input_logits, target_logits):
    """The per-scale terms of softmax_mse_loss (each enters the loss with weight 1 / len)."""
    return [ops.softmax_mse(a, b.detach()) for a, b in zip(input_logits, target_logits)]


def softmax_mse_loss(input_logits, target_logits, sigmoid=False):
    """losses.py:68-90: lists of per-scale maps; targets detached; mean over scales."""
    if sigmoid:
        raise NotImplementedError("the ICL trainers never pass sigmoid=True")
    return _mean_of(softmax_mse_leaves(input_logits, target_logits))


class DiceLoss(nn.Module):
    """losses.py:195-231.  ``target`` is [B,1,...] class indices; one-hot by equality."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        assert inputs.shape[1] == self.n_classes and inputs.shape[0] == target.shape[0] \
            and inputs.shape[2:] == target.shape[2:], \
            "predict {} & target {} shape do not match".format(inputs.size(), target.size())
        return ops.dice_loss(inputs, target[:, 0], self.n_classes, softmax, weight)


def _resize(t, size):
    return ops.trilinear_resize(t.float(), size)


def _resize2d(t, size):
    return ops.bilinear_resize(t.float(), size)


class AuxLoss(nn.Module):
    """2-D twin of AuxLoss3D (losses.py:233-251): bilinear resize to ``resize``, CE + Dice(softmax=True) per map."""

    def __init__(self, n_classes, resize=(224, 224)):
        super().__init__()
        self.n_classes = n_classes
        self.ce_loss = CrossEntropyLoss()
        self.dice_loss = DiceLoss(n_classes)
        self.resize = tuple(resize)

    def leaves(self, feat_maps, labels):
        """[ce_0, dice_0, ce_1, dice_1, ...]: the loss is their sum / len(feat_maps)."""
        out = []
        for fm in feat_maps:
            out += list(ops.cross_entropy_dice_parts(_resize2d(fm, self.resize), labels.long(), self.n_classes))
        return out

    def forward(self, feat_maps, labels):
        lv = self.leaves(feat_maps, labels)
        return ops.combine_scalars(lv, [[1.0 / len(feat_maps)] * len(lv)])[0]


class PseudoSoftLoss(nn.Module):
    """2-D twin of PseudoSoftLoss3D (losses.py:273-285)."""

    def __init__(self, n_classes, resize=(224, 224)):
        super().__init__()
        self.resize = tuple(resize)

    def leaves(self, feat_maps, predicts):
        tgt = predicts.detach()
        return [softmax_dice_loss(_resize2d(fm, self.resize), tgt) for fm in feat_maps]

    def forward(self, feat_maps, predicts):
        return _mean_of(self.leaves(feat_maps, predicts))


class AuxLoss3D(nn.Module):
    """losses.py:254-271 (resize hard-coded to 96^3 there; exposed as an argument with that default)."""

    def __init__(self, n_classes, resize=(96, 96, 96)):
        super().__init__()
        self.n_classes = n_classes
        self.ce_loss = CrossEntropyLoss()
        self.dice_loss = DiceLoss(n_classes)
        self.resize = tuple(resize)

    def leaves(self, feat_maps, labels):
        """[ce_0, dice_0, ce_1, dice_1, ...] (CE + Dice(softmax=True) per map): the loss is their sum / len(feat_maps)."""
        out = []
        for fm in feat_maps:
            out += list(ops.cross_entropy_dice_parts(_resize(fm, self.resize), labels.long(), self.n_classes))
        return out

    def terms(self, feat_maps, labels):
        """The same terms as (logits, target, mode) triples for ops.fused_losses (all loss terms of a step in one launch per pass)."""
        return [(_resize(fm, self.resize), labels, 1) for fm in feat_maps]

    def forward(self, feat_maps, labels):
        lv = self.leaves(feat_maps, labels)
        return ops.combine_scalars(lv, [[1.0 / len(feat_maps)] * len(lv)])[0]


class PseudoSoftLoss3D(nn.Module):
    """losses.py:287-299: soft Dice against the detached unlabeled prediction."""

    def __init__(self, n_classes, resize=(96, 96, 96)):
        super().__init__()
        self.resize = tuple(resize)

    def leaves(self, feat_maps, predicts):
        tgt = predicts.detach()
        return [softmax_dice_loss(_resize(fm, self.resize), tgt) for fm in feat_maps]

    def terms(self, feat_maps, predicts):
        tgt = predicts.detach()
        return [(_resize(fm, self.resize), tgt, 2) for fm in feat_maps]

    def forward(self, feat_maps, predicts):
        return _mean_of(self.leaves(feat_maps, predicts))
