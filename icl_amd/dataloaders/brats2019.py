"""Two-stream batch sampler of the ICL trainers — same semantics as the reference's ``TwoStreamBatchSampler``
(/root/reference/code/dataloaders/brats2019.py:191-237): an epoch is one pass over the primary (labeled) indices in a
fresh ``np.random.permutation``; secondary (unlabeled) indices are drawn from an endless chain of permutations; each
batch is ``primary_batch + secondary_batch``.  ``rank``/``world_size`` add the data-parallel sharding the build needs
(SURVEY.md §8e): every rank draws the SAME global permutations (same numpy seed) and keeps the rank-strided slice of
each global batch, so W ranks together consume exactly the batches one reference process would."""
from __future__ import annotations

import itertools

import numpy as np
from torch.utils.data.sampler import Sampler


def iterate_once(iterable):
    return np.random.permutation(iterable)


def iterate_eternally(indices):
    def infinite_shuffles():
        while True:
            yield np.random.permutation(indices)
    return itertools.chain.from_iterable(infinite_shuffles())


def grouper(iterable, n):
    "grouper('ABCDEFG', 3) --> ABC DEF (incomplete tail dropped)"
    args = [iter(iterable)] * n
    return zip(*args)


class TwoStreamBatchSampler(Sampler):
    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size, rank=0, world_size=1):
        self.primary_indices = primary_indices
        self.secondary_indices = secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        self.rank, self.world_size = rank, world_size
        assert len(self.primary_indices) >= self.primary_batch_size > 0
        assert len(self.secondary_indices) >= self.secondary_batch_size > 0
        assert self.primary_batch_size % world_size == 0 and self.secondary_batch_size % world_size == 0, \
            "the global labeled / unlabeled batch must split evenly over the ranks"

    def __iter__(self):
        primary_iter = iterate_once(self.primary_indices)
        secondary_iter = iterate_eternally(self.secondary_indices)
        r, w = self.rank, self.world_size
        return (
            tuple(primary_batch[r::w]) + tuple(secondary_batch[r::w])
            for (primary_batch, secondary_batch)
            in zip(grouper(primary_iter, self.primary_batch_size),
                   grouper(secondary_iter, self.secondary_batch_size))
        )

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size
