"""Sliding-window validation on the device — counterpart of the reference's ``val_3D.py``
(/root/reference/code/val_3D.py:15-118).

``test_single_case_base`` keeps the reference's window grid (``ceil((dim - patch) / stride) + 1`` positions per axis, the
last one clamped to ``dim - patch``, zero padding of small volumes :18-41,43-55) but keeps the volume, the score map and
the count map in HBM and pushes several windows per forward (the backbone is per-sample, so batching windows does not
change any window's logits); the reference moves every window host->device->host and accumulates in numpy (:57-73).
``cal_metric`` reproduces the empty-mask conventions (:85-97).  Dice is computed here; HD95 follows MedPy 0.4.0's
published definition (``medpy.metric.binary.hd95``: 95th percentile of the symmetric surface distances, surfaces from a
connectivity-1 erosion, distances from ``scipy.ndimage.distance_transform_edt``) — MedPy itself is not installed in the
build image, so the HD95 half is "parity unpinned" (DESIGN.md §2).
"""
from __future__ import annotations

import math

import numpy as np
import torch


@torch.no_grad()
def test_single_case_base(net, net_type, image, stride_xy, stride_z, patch_size, num_classes=1, windows_per_batch=4,
                          return_score=False):
    """image: numpy [W,H,D] (the reference's h5 layout).  Returns the label map (numpy int64 [W,H,D])."""
    dev = next(net.parameters()).device
    w, h, d = image.shape
    pads = []
    for dim, p in zip((w, h, d), patch_size):
        tot = max(p - dim, 0)
        pads.append((tot // 2, tot - tot // 2))
    add_pad = any(a + b > 0 for a, b in pads)
    vol = torch.as_tensor(np.ascontiguousarray(image), dtype=torch.float32, device=dev)
    if add_pad:
        vol = torch.nn.functional.pad(vol, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    ww, hh, dd = vol.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score = torch.zeros((num_classes, ww, hh, dd), dtype=torch.float32, device=dev)
    cnt = torch.zeros((ww, hh, dd), dtype=torch.float32, device=dev)
    starts = []
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                zs = min(stride_z * z, dd - patch_size[2])
                starts.append((xs, ys, zs))
    px, py, pz = patch_size
    icl = net_type in ("swinunetr_icl", "unet_3D_icl")
    for i in range(0, len(starts), windows_per_batch):
        chunk = starts[i:i + windows_per_batch]
        batch = torch.stack([vol[xs:xs + px, ys:ys + py, zs:zs + pz] for xs, ys, zs in chunk]).unsqueeze(1).contiguous()
        logits = net(batch, inference=True) if icl else net(batch)
        prob = torch.softmax(logits, dim=1)
        for j, (xs, ys, zs) in enumerate(chunk):
            score[:, xs:xs + px, ys:ys + py, zs:zs + pz] += prob[j]
            cnt[xs:xs + px, ys:ys + py, zs:zs + pz] += 1
    score = score / cnt.unsqueeze(0)
    label_map = torch.argmax(score, dim=0)
    if add_pad:
        sl = tuple(slice(a, a + n) for (a, _), n in zip(pads, (w, h, d)))
        label_map = label_map[sl]
        score = score[(slice(None),) + sl]
    if return_score:
        return label_map.cpu().numpy(), score.cpu().numpy()
    return label_map.cpu().numpy()


def binary_dice(pred, gt) -> float:
    """medpy.metric.binary.dc: 2|A&B| / (|A|+|B|) (0 when both are empty)."""
    pred = np.asarray(pred).astype(bool)
    gt = np.asarray(gt).astype(bool)
    inter = np.count_nonzero(pred & gt)
    size = np.count_nonzero(pred) + np.count_nonzero(gt)
    return 2.0 * inter / float(size) if size else 0.0


def _surface_distances(result, reference):
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    result = np.atleast_1d(result.astype(bool))
    reference = np.atleast_1d(reference.astype(bool))
    fp = generate_binary_structure(result.ndim, 1)
    rb = result ^ binary_erosion(result, structure=fp, iterations=1)
    fb = reference ^ binary_erosion(reference, structure=fp, iterations=1)
    dt = distance_transform_edt(~fb)
    return dt[rb]


def binary_hd95(pred, gt) -> float:
    """medpy.metric.binary.hd95 (MedPy 0.4.0): 95th percentile of both directed surface-distance sets."""
    a = _surface_distances(np.asarray(pred), np.asarray(gt))
    b = _surface_distances(np.asarray(gt), np.asarray(pred))
    return float(np.percentile(np.hstack((a, b)), 95))


def cal_metric(gt, pred):
    """val_3D.py:85-97, including the hard-coded fall-backs for empty masks."""
    pred = np.asarray(pred) > 0
    gt = np.asarray(gt) > 0
    if pred.sum() > 0 and gt.sum() > 0:
        return binary_dice(pred, gt), binary_hd95(pred, gt)
    if pred.sum() > 0 and gt.sum() == 0:
        return 0, 373.128664
    if pred.sum() == 0 and gt.sum() > 0:
        return 0, 373.128664
    return 1, 0


def test_all_case_base(net, net_type, base_dir, test_list="full_test.list", num_classes=4, patch_size=(48, 160, 160),
                       stride_xy=32, stride_z=24, cases=None):
    """val_3D.py:100-118.  ``cases`` (iterable of (image, label) numpy pairs) replaces the h5 files when given; otherwise
    the reference's ``{base_dir}/data/{id}.h5`` layout is read with h5py (not installed in the build image)."""
    if cases is None:
        import h5py  # noqa: F401 — same dependency as the reference
        with open(base_dir + "/{}".format(test_list), "r") as f:
            ids = [ln.strip().split(",")[0] for ln in f if ln.strip()]

        def _gen():
            for i in ids:
                with h5py.File(base_dir + "/data/{}.h5".format(i), "r") as h5f:
                    yield h5f["image"][:], h5f["label"][:]
        cases = _gen()
    metric_cal = [[] for _ in range(num_classes - 1)]
    for image, label in cases:
        prediction = test_single_case_base(net, net_type, image, stride_xy, stride_z, patch_size, num_classes=num_classes)
        for i in range(1, num_classes):
            metric_cal[i - 1].append(cal_metric(label == i, prediction == i))
    return metric_cal


# --------------------------------------------------------------------------------------------------------------------
# AMOS validation: MONAI-style sliding window (val_3D.py:120-137)
# --------------------------------------------------------------------------------------------------------------------

def _scan_starts(size: int, roi: int, overlap: float):
    """Window origins along one axis as MONAI 1.0.1 ``sliding_window_inference`` places them (``_get_scan_interval`` +
    ``dense_patch_slices``): stride ``int(roi * (1 - overlap))`` (the whole axis when it equals the window), ``ceil((size - roi) /
    stride) + 1`` windows, the last ones shifted back inside the volume."""
    if size == roi:
        return [0]
    interval = int(roi * (1 - overlap))
    interval = interval if interval > 0 else 1
    num = int(math.ceil(float(size - roi) / interval)) + 1
    return [i * interval - max(i * interval + roi - size, 0) for i in range(num)]


@torch.no_grad()
def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap: float = 0.25, **kwargs):
    """Restatement of ``monai.inferers.sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25,
    mode="constant")`` as the reference calls it (val_3D.py:130-132): symmetric zero padding of volumes smaller than the window,
    the window grid of ``_scan_starts``, ``sw_batch_size`` windows per forward, predictor OUTPUTS (logits) averaged with constant
    weights over overlapping windows, padding cropped off again.  Everything stays on the device of ``inputs``.  MONAI is not
    installed in the build image: parity unpinned."""
    b = inputs.shape[0]
    size = list(inputs.shape[2:])
    roi = [int(r) for r in roi_size]
    pads = []
    for s, r in zip(size, roi):
        diff = max(r - s, 0)
        pads.append((diff // 2, diff - diff // 2))
    if any(a + c for a, c in pads):
        flat = [v for a, c in reversed(pads) for v in (a, c)]
        inputs = torch.nn.functional.pad(inputs, flat)
    psize = list(inputs.shape[2:])
    grids = [_scan_starts(s, r, overlap) for s, r in zip(psize, roi)]
    starts = [(bi, z, y, x) for bi in range(b) for z in grids[0] for y in grids[1] for x in grids[2]]
    out = cnt = None
    for i in range(0, len(starts), sw_batch_size):
        chunk = starts[i:i + sw_batch_size]
        win = torch.stack([inputs[bi, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] for bi, z, y, x in chunk]).contiguous()
        pred = predictor(win, **kwargs)
        if out is None:
            out = torch.zeros((b, pred.shape[1], *psize), dtype=torch.float32, device=inputs.device)
            cnt = torch.zeros((b, 1, *psize), dtype=torch.float32, device=inputs.device)
        for j, (bi, z, y, x) in enumerate(chunk):
            out[bi, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += pred[j].float()
            cnt[bi, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += 1
    out = out / cnt
    sl = tuple(slice(a, a + s) for (a, _), s in zip(pads, size))
    return out[(slice(None), slice(None)) + sl]


def test_all_case_amos(net, net_type, val_loader, num_classes=4):
    """val_3D.py:120-137: every validation volume through ``sliding_window_inference(image, (96, 96, 96), 4, net[, inference=True])``,
    argmax over the averaged logits, ``cal_metric`` per foreground class.  The reference wraps the inference in fp16 autocast; the
    HIP kernels compute in fp32 (a strictly more accurate evaluation of the same network)."""
    metric_cal = [[] for _ in range(num_classes - 1)]
    net.eval()
    dev = next(net.parameters()).device
    for batch in val_loader:
        image, label = batch["image"].to(dev), batch["label"].squeeze(0)
        kw = {"inference": True} if net_type in ("unet_3D_icl", "swinunetr_icl") else {}
        logits = sliding_window_inference(image, (96, 96, 96), 4, net, **kw)
        prediction = torch.argmax(logits, dim=1).cpu().numpy()
        label = np.asarray(label.cpu() if torch.is_tensor(label) else label)
        for i in range(1, num_classes):
            metric_cal[i - 1].append(cal_metric(label == i, prediction == i))
    return metric_cal
