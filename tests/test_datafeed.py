"""Data feed against vectors generated from the real reference (tests/golden/make_golden_datafeed.py):
the two-stream index batches for fixed numpy seeds and the fused on-device RandomRotFlip -> RandomCrop -> ToTensor kernel
(csrc/kernels/datafeed.h, here on the CPU emulation of the kernel sources; the GPU run is in test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))
sys.path.insert(0, os.path.join(HERE, "golden"))
from build_emu import build_emu  # noqa: E402
from make_golden_datafeed import BIG_CASES, CASES, case_volume, digest  # noqa: E402

from icl_amd import _lib  # noqa: E402
from icl_amd.dataloaders.brats2019 import (DeviceVolumeStore, OnDeviceAugment, TwoStreamBatchSampler,  # noqa: E402
                                           draw_rotflip_crop)

GOLD = np.load(os.path.join(HERE, "golden", "datafeed.npz"))
GOLD_BIG = np.load(os.path.join(HERE, "golden", "datafeed_big.npz"))


@pytest.fixture(scope="module")
def emu_library():
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    yield
    _lib._use_library_for_tests(None)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_two_stream_batches_equal_the_reference_stream(tag):
    nlab, ntot, bs, sbs, seed, epochs = (int(v) for v in GOLD[f"sampler.{tag}.cfg"])
    np.random.seed(seed)
    s = TwoStreamBatchSampler(list(range(nlab)), list(range(nlab, ntot)), bs, sbs)
    rows = []
    for _ in range(epochs):
        rows += [list(map(int, b)) for b in s]
    assert np.array_equal(np.array(rows, dtype=np.int64), GOLD[f"sampler.{tag}.batches"])
    assert len(s) == nlab // (bs - sbs)


def test_rank_shards_partition_every_global_batch():
    np.random.seed(3)
    glob = list(TwoStreamBatchSampler(list(range(16)), list(range(16, 64)), 8, 4))
    shards = []
    for r in range(2):
        np.random.seed(3)
        shards.append(list(TwoStreamBatchSampler(list(range(16)), list(range(16, 64)), 8, 4, rank=r, world_size=2)))
    for g, a, b in zip(glob, *shards):
        assert g[:4] == tuple(v for pair in zip(a[:2], b[:2]) for v in pair)       # labeled half, rank-strided
        assert g[4:] == tuple(v for pair in zip(a[2:], b[2:]) for v in pair)       # unlabeled half
    with pytest.raises(AssertionError):
        TwoStreamBatchSampler(list(range(16)), list(range(16, 64)), 6, 3, rank=0, world_size=2)


def check_augment_cases(device):
    vols = [case_volume(shape, seed) for shape, _, seed in CASES]
    store = DeviceVolumeStore(vols, device)
    for n, (shape, patch, seed) in enumerate(CASES):
        np.random.seed(seed)
        out = OnDeviceAugment(store, patch).batch([n])
        assert out["image"].shape == (1, 1) + tuple(patch) and out["label"].dtype == torch.int64
        assert np.array_equal(out["image"][0].cpu().numpy(), GOLD[f"aug.{n}.image"]), (n, shape, patch)
        assert np.array_equal(out["label"][0].cpu().numpy(), GOLD[f"aug.{n}.label"]), (n, shape, patch)
    # one launch for a whole batch, parameters drawn per sample in batch order
    idx = [5, 6, 7, 8]
    np.random.seed(77)
    params = [draw_rotflip_crop(CASES[i][0], (8, 8, 8)) for i in idx]
    np.random.seed(77)
    both = OnDeviceAugment(store, (8, 8, 8)).batch(idx)
    for j, i in enumerate(idx):
        one = OnDeviceAugment(store, (8, 8, 8)).batch([i], params=[params[j]])
        assert torch.equal(both["image"][j], one["image"][0]) and torch.equal(both["label"][j], one["label"][0])


def test_fused_rotflip_crop_kernel_equals_the_reference_transforms(emu_library):
    check_augment_cases("cpu")


def check_big_cases(make_batch):
    """Real BraTS2019 extent, 240 x 240 x 155 (and a transposed one) -> 96^3: CRC-32 of the output bytes, their sum and a stride-4
    subsample against the reference's outputs.  ``make_batch(image, label, patch, seed)`` -> (image [1, *patch] fp32, label int64)."""
    for n, (shape, patch, seed) in enumerate(BIG_CASES):
        image, label = case_volume(shape, seed)
        img, lab = make_batch(image, label, patch, seed)
        for key, a in (("image", img), ("label", lab)):
            crc, total, sub = digest(np.asarray(a))
            assert np.array_equal(sub, GOLD_BIG[f"big.{n}.{key}.sub"]), (n, key)
            assert int(crc[0]) == int(GOLD_BIG[f"big.{n}.{key}.crc"][0]) and float(total[0]) == float(GOLD_BIG[f"big.{n}.{key}.sum"][0]), (n, key)


def test_host_transforms_at_the_real_brats_extent():
    """The host path (`RandomRotFlip -> RandomCrop -> ToTensor`, the classes a caller of the reference's DataLoader keeps) on
    240 x 240 x 155 volumes."""
    from icl_amd.dataloaders.brats2019 import RandomCrop, RandomRotFlip, ToTensor

    def host(image, label, patch, seed):
        np.random.seed(seed)
        s = ToTensor()(RandomCrop(patch)(RandomRotFlip()({"image": image, "label": label})))
        return s["image"].numpy(), s["label"].numpy()
    check_big_cases(host)


def device_big_batch(device):
    def run(image, label, patch, seed):
        store = DeviceVolumeStore([(image, label)], device)
        np.random.seed(seed)
        out = OnDeviceAugment(store, patch).batch([0])
        return out["image"][0].cpu().numpy(), out["label"][0].cpu().numpy()
    return run


@pytest.mark.slow
def test_fused_kernel_at_the_real_brats_extent(emu_library):
    """The fused gather kernel on a 240 x 240 x 155 volume (one case on the CPU emulation: 884,736 output voxels through fibers)."""
    image, label = case_volume(*[BIG_CASES[0][i] for i in (0, 2)])
    img, lab = device_big_batch("cpu")(image, label, BIG_CASES[0][1], BIG_CASES[0][2])
    crc, _, _ = digest(img)
    assert int(crc[0]) == int(GOLD_BIG["big.0.image.crc"][0])
    assert int(digest(lab)[0][0]) == int(GOLD_BIG["big.0.label.crc"][0])
