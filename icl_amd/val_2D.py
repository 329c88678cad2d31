"""Slice-wise validation of the 2-D trainers — counterpart of the reference's ``val_2D.py``
(/root/reference/code/val_2D.py:10-54,109-131; called at train_inherent_consistent_unet_2D.py / ..._swinunet_2D.py).

``test_single_volume_ours(image, label, net, writer, iter_num, classes, patch_size)`` keeps the reference procedure — every slice
is resized to ``patch_size`` with ``scipy.ndimage.zoom(order=0)``, pushed through ``net(x, inference=True)``, arg-maxed and resized
back — but pushes ``slices_per_batch`` slices per forward (eval-mode networks are per-sample: BatchNorm uses its running statistics,
so batching does not change any slice's logits; the reference runs one slice per forward).  The nearest-neighbour resize stays
scipy's, so the index mapping is the reference's bit for bit.  Metrics: Dice + HD95 with the reference's empty-mask conventions
(``icl_amd.val_3D.cal_metric``; HD95 parity unpinned — MedPy is not installed in the build image).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.ndimage import zoom

from .val_3D import binary_dice, binary_hd95


def calculate_metric_percase(pred, gt):
    """val_2D.py:10-22 (binarises its arguments like the reference)."""
    pred = np.asarray(pred) > 0
    gt = np.asarray(gt) > 0
    if pred.sum() > 0 and gt.sum() > 0:
        return binary_dice(pred, gt), binary_hd95(pred, gt)
    if pred.sum() > 0 and gt.sum() == 0:
        return 0, 373.128664
    if pred.sum() == 0 and gt.sum() > 0:
        return 0, 373.128664
    return 1, 0


@torch.no_grad()
def _predict_volume(image, net, patch_size, inference_kw, slices_per_batch):
    dev = next(net.parameters()).device
    net.eval()
    n, x, y = image.shape
    prediction = np.zeros((n, x, y), dtype=np.int64)
    resized = np.stack([zoom(image[i], (patch_size[0] / x, patch_size[1] / y), order=0) for i in range(n)])
    for lo in range(0, n, slices_per_batch):
        batch = torch.from_numpy(resized[lo:lo + slices_per_batch]).unsqueeze(1).float().to(dev)
        out = torch.argmax(torch.softmax(net(batch, **inference_kw), dim=1), dim=1).cpu().numpy()
        for j in range(out.shape[0]):
            prediction[lo + j] = zoom(out[j], (x / patch_size[0], y / patch_size[1]), order=0)
    return prediction


def test_single_volume_ours(image, label, net, writer=None, iter_num=0, classes=4, patch_size=(224, 224), slices_per_batch=16):
    """val_2D.py:35-54: ``image`` / ``label`` are [1, slices, H, W] tensors (one case of the validation loader)."""
    image = image.squeeze(0).cpu().detach().numpy()
    label = label.squeeze(0).cpu().detach().numpy()
    prediction = _predict_volume(image, net, patch_size, {"inference": True}, slices_per_batch)
    return [calculate_metric_percase(prediction == i, label == i) for i in range(1, classes)]


def test_single_volume(image, label, net, classes, patch_size=(256, 256), slices_per_batch=16):
    """val_2D.py:109-131: the plain networks (``net(input)``)."""
    image = image.squeeze(0).cpu().detach().numpy()
    label = label.squeeze(0).cpu().detach().numpy()
    prediction = _predict_volume(image, net, patch_size, {}, slices_per_batch)
    return [calculate_metric_percase(prediction == i, label == i) for i in range(1, classes)]
