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
from make_golden_datafeed import CASES, case_volume  # noqa: E402

from icl_amd import _lib  # noqa: E402
from icl_amd.dataloaders.brats2019 import (DeviceVolumeStore, OnDeviceAugment, TwoStreamBatchSampler,  # noqa: E402
                                           draw_rotflip_crop)

GOLD = np.load(os.path.join(HERE, "golden", "datafeed.npz"))


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
