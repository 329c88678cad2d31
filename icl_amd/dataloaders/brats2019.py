"""Data feed of the 3-D ICL trainers, MI355X-first.

What the reference does (``/root/reference/code/dataloaders/brats2019.py``, used at
``train_inherent_consistent_unet_3D_BraTS.py:66-83``): an h5 volume per index, ``RandomRotFlip -> RandomCrop(96^3) -> ToTensor``
in four host worker processes, batches of labeled + unlabeled indices from ``TwoStreamBatchSampler``.  At >100 volumes/s per GPU a
numpy loader is the bottleneck, and the whole training set is 9 GB of fp32 against 288 GB of HBM, so here

* ``DeviceVolumeStore`` keeps every training volume resident on the GPU (fp32 image + uint8 label),
* ``OnDeviceAugment`` draws the augmentation parameters on the host with the reference's numpy call sequence
  (``:136-139`` then ``:116-118``) and produces the whole batch — rot90, flip, zero padding, crop, ``ToTensor`` casts — with ONE
  gather kernel (``csrc/kernels/datafeed.h``),
* ``TwoStreamBatchSampler`` yields the same index batches as the reference class for the same numpy seed (``:191-237``) and can
  shard every global batch over data-parallel ranks,
* ``BraTS2019`` reads the reference's file layout (``train.txt`` / ``val.txt`` + ``data/<case>.h5``) for callers that want the
  host path (needs h5py, which the reference requires too).
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from .. import _lib


# --------------------------------------------------------------------------------------------------------------------
# index streams
# --------------------------------------------------------------------------------------------------------------------

class _ShuffledStream:
    """Index source that hands out ``count`` indices at a time from successive ``np.random.permutation`` draws.  A new
    permutation is drawn only at the moment the previous one is used up (possibly in the middle of a request), which is when
    the reference's chained generator draws it — the numpy generator is shared with everything else in the process, so the
    moment of every draw is part of the contract."""

    def __init__(self, indices: Sequence[int], endless: bool):
        self.indices = indices
        self.endless = endless
        self.pool = np.empty(0, dtype=np.int64)
        self.pos = 0
        self.drawn = False

    def _refill(self) -> bool:
        if self.drawn and not self.endless:
            return False
        self.pool = np.random.permutation(self.indices)
        self.pos = 0
        self.drawn = True
        return True

    def take(self, count: int) -> Optional[tuple]:
        got = []
        while len(got) < count:
            if self.pos >= len(self.pool) and not self._refill():
                return None            # a single pass ran out: the incomplete tail is dropped
            got.append(self.pool[self.pos])
            self.pos += 1
        return tuple(got)


class TwoStreamBatchSampler(Sampler):
    """Batches of ``batch_size - secondary_batch_size`` primary (labeled) indices followed by ``secondary_batch_size`` secondary
    (unlabeled) ones.  One epoch is a single shuffled pass over the primary indices; the secondary indices come from an endless
    sequence of shuffles (reference: ``dataloaders/brats2019.py:191-237``).  With ``world_size > 1`` every rank draws the same
    global batches (same numpy seed) and keeps elements ``rank::world_size`` of each half, so W ranks together consume exactly
    the batches of one reference process."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size, rank: int = 0, world_size: int = 1):
        self.primary_indices = primary_indices
        self.secondary_indices = secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        self.rank, self.world_size = rank, world_size
        if not (len(primary_indices) >= self.primary_batch_size > 0):
            raise AssertionError("need at least one full primary batch")
        if not (len(secondary_indices) >= self.secondary_batch_size > 0):
            raise AssertionError("need at least one full secondary batch")
        if self.primary_batch_size % world_size or self.secondary_batch_size % world_size:
            raise AssertionError("the global labeled / unlabeled batch must split evenly over the ranks")

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size

    def __iter__(self):
        primary = _ShuffledStream(self.primary_indices, endless=False)
        primary._refill()                                   # the epoch's primary shuffle is drawn here, when iteration is set up ...
        secondary = _ShuffledStream(self.secondary_indices, endless=True)   # ... the first secondary one with the first batch
        return self._batches(primary, secondary)

    def _batches(self, primary, secondary):
        r, w = self.rank, self.world_size
        while True:
            lab = primary.take(self.primary_batch_size)
            if lab is None:
                return
            unl = secondary.take(self.secondary_batch_size)
            yield lab[r::w] + unl[r::w]


# --------------------------------------------------------------------------------------------------------------------
# host dataset (reference file layout)
# --------------------------------------------------------------------------------------------------------------------

class BraTS2019(Dataset):
    """``<base_dir>/train.txt`` | ``val.txt`` list the cases, ``<base_dir>/data/<case>.h5`` holds ``image`` (fp32 [W,H,D]) and
    ``label`` (reference: ``dataloaders/brats2019.py:11-46``).  ``split='test'`` reads ``val.txt`` as the reference does."""

    def __init__(self, base_dir=None, split="train", num=None, transform=None):
        self._base_dir = base_dir
        self.transform = transform
        listing = {"train": "train.txt", "test": "val.txt"}[split]
        with open(os.path.join(base_dir, listing)) as f:
            self.image_list = [ln.strip().split(",")[0] for ln in f if ln.strip()]
        if num is not None:
            self.image_list = self.image_list[:num]

    def __len__(self):
        return len(self.image_list)

    def read_case(self, idx) -> Tuple[np.ndarray, np.ndarray]:
        import h5py   # the reference's reader; not installed everywhere, only needed for real data
        with h5py.File(os.path.join(self._base_dir, "data", f"{self.image_list[idx]}.h5"), "r") as h5f:
            return h5f["image"][:], h5f["label"][:].astype(np.uint8)

    def __getitem__(self, idx):
        image, label = self.read_case(idx)
        sample = {"image": image, "label": label}
        return self.transform(sample) if self.transform else sample


# --------------------------------------------------------------------------------------------------------------------
# device-resident volumes + fused augmentation
# --------------------------------------------------------------------------------------------------------------------

class DeviceVolumeStore:
    """All volumes of a split resident in HBM: ``images[i]`` fp32 [W,H,D], ``labels[i]`` uint8 [W,H,D] (the h5 dtypes)."""

    def __init__(self, volumes: Sequence[Tuple[np.ndarray, np.ndarray]], device):
        self.device = torch.device(device)
        self.images: List[torch.Tensor] = []
        self.labels: List[torch.Tensor] = []
        for image, label in volumes:
            if image.shape != label.shape or image.ndim != 3:
                raise ValueError("a volume is an (image [W,H,D], label [W,H,D]) pair")
            self.images.append(torch.as_tensor(np.ascontiguousarray(image, dtype=np.float32)).to(self.device))
            self.labels.append(torch.as_tensor(np.ascontiguousarray(label, dtype=np.uint8)).to(self.device))

    @classmethod
    def from_dataset(cls, dataset: BraTS2019, device):
        return cls([dataset.read_case(i) for i in range(len(dataset))], device)

    def __len__(self):
        return len(self.images)


def draw_rotflip_crop(shape: Sequence[int], output_size: Sequence[int]) -> List[int]:
    """The random draws of ``RandomRotFlip()`` followed by ``RandomCrop(output_size)`` for one volume of ``shape``, in the
    reference's order (``brats2019.py:136-139``: ``k = randint(0, 4)``, ``axis = randint(0, 2)``; ``:97-118``: padding by
    ``max((out - dim) // 2 + 3, 0)`` on every axis if ANY axis is not larger than the patch, then ``randint(0, dim - out)`` per
    axis).  Returns the 11 integers of one ``icl_crop_rotflip`` parameter row."""
    n0, n1, n2 = (int(v) for v in shape)
    k = int(np.random.randint(0, 4))
    axis = int(np.random.randint(0, 2))
    dims = [n1, n0, n2] if k & 1 else [n0, n1, n2]          # rot90 on axes (0, 1) swaps them for odd k
    pads = [0, 0, 0]
    if any(dims[a] <= output_size[a] for a in range(3)):
        pads = [max((output_size[a] - dims[a]) // 2 + 3, 0) for a in range(3)]
    origin = [int(np.random.randint(0, dims[a] + 2 * pads[a] - output_size[a])) for a in range(3)]
    return [n0, n1, n2, k, axis, *pads, *origin]


class OnDeviceAugment:
    """``Compose([RandomRotFlip(), RandomCrop(patch), ToTensor()])`` + batch collation on the GPU: ``batch(indices)`` returns
    ``{'image': [B,1,*patch] fp32, 'label': [B,*patch] int64}`` like the reference DataLoader's ``sampled_batch``."""

    def __init__(self, store: DeviceVolumeStore, patch_size: Sequence[int] = (96, 96, 96)):
        self.store = store
        self.patch = tuple(int(v) for v in patch_size)

    def batch(self, indices: Sequence[int], params: Optional[Sequence[Sequence[int]]] = None):
        st = self.store
        n = len(indices)
        if params is None:
            params = [draw_rotflip_crop(st.images[i].shape, self.patch) for i in indices]
        dev = st.device
        image = torch.empty((n, 1) + self.patch, dtype=torch.float32, device=dev)
        label = torch.empty((n,) + self.patch, dtype=torch.int64, device=dev)
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        for lo in range(0, n, 16):                        # icl_crop_rotflip takes up to 16 samples per launch
            chunk = list(indices[lo:lo + 16])
            m = len(chunk)
            ptrs = ctypes.c_void_p * m
            flat = (ctypes.c_int32 * (11 * m))(*[int(v) for row in params[lo:lo + m] for v in row])
            _lib.check(L.icl_crop_rotflip(ptrs(*[st.images[i].data_ptr() for i in chunk]), ptrs(*[st.labels[i].data_ptr() for i in chunk]),
                                          flat, m, image[lo:].data_ptr(), label[lo:].data_ptr(), *self.patch, stream), "crop_rotflip")
        return {"image": image, "label": label}


# --------------------------------------------------------------------------------------------------------------------
# host transforms (the reference's numpy path, for callers that keep its DataLoader: same sample dicts, same random draws)
# --------------------------------------------------------------------------------------------------------------------

def _pad_to_patch(arrays, output_size):
    """Zero-pad every array on both sides of each axis by ``max((out - dim) // 2 + 3, 0)`` when ANY axis of the volume does not
    exceed the patch (brats2019.py:53-62,97-109)."""
    shape = arrays[0].shape
    if all(shape[a] > output_size[a] for a in range(3)):
        return arrays
    widths = [(max((output_size[a] - shape[a]) // 2 + 3, 0),) * 2 for a in range(3)]
    return [np.pad(v, widths, mode="constant", constant_values=0) for v in arrays]


def _window(arrays, origin, output_size):
    sl = tuple(slice(origin[a], origin[a] + output_size[a]) for a in range(3))
    return [v[sl] for v in arrays]


class RandomRotFlip:
    """k = randint(0, 4) quarter turns in the (0, 1) plane, then a flip along axis randint(0, 2) (brats2019.py:128-144)."""

    def __call__(self, sample):
        k = np.random.randint(0, 4)
        axis = np.random.randint(0, 2)
        out = dict(sample)
        for key in ("image", "label"):
            out[key] = np.flip(np.rot90(sample[key], k), axis=axis).copy()
        return out


class RandomCrop:
    """Patch at a uniformly drawn origin, after padding small volumes (brats2019.py:80-127; the ``sdf`` channel is carried along
    when ``with_sdf``)."""

    def __init__(self, output_size, with_sdf=False):
        self.output_size = output_size
        self.with_sdf = with_sdf

    def __call__(self, sample):
        keys = ["image", "label"] + (["sdf"] if self.with_sdf else [])
        arrays = _pad_to_patch([sample[k] for k in keys], self.output_size)
        origin = [np.random.randint(0, arrays[0].shape[a] - self.output_size[a]) for a in range(3)]
        return dict(zip(keys, _window(arrays, origin, self.output_size)))


class CenterCrop:
    """Centred patch, same padding rule (brats2019.py:49-77)."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        arrays = _pad_to_patch([sample["image"], sample["label"]], self.output_size)
        origin = [int(round((arrays[0].shape[a] - self.output_size[a]) / 2.)) for a in range(3)]
        image, label = _window(arrays, origin, self.output_size)
        return {"image": image, "label": label}


class ToTensor:
    """image -> fp32 tensor [1,W,H,D], label -> int64 tensor (brats2019.py:177-189)."""

    def __call__(self, sample):
        image = np.ascontiguousarray(sample["image"], dtype=np.float32)[None]
        out = {"image": torch.from_numpy(image), "label": torch.from_numpy(np.ascontiguousarray(sample["label"])).long()}
        if "onehot_label" in sample:
            out["onehot_label"] = torch.from_numpy(sample["onehot_label"]).long()
        return out
