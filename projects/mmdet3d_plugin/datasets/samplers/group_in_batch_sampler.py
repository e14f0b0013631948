"""``GroupInBatchSampler`` (reference datasets/samplers/group_in_batch_sampler.py:48-178): every slot of the global
batch (rank * batch_size + local index) walks its own driving sequence ("group") frame by frame; when a sequence ends the
slot takes the next entry of ONE seeded infinite permutation stream of the groups, strided by the global batch size, so no
two slots of any rank hold the same sequence.  Each yielded item carries the sequence's aug_config.  Host logic only.

Same draws as the reference: torch.randperm from a torch.Generator seeded with the (rank-0) seed for the group order,
the numpy global RNG for frame skipping / sequence reversal, ``dataset.get_augmentation()`` per sequence (or per frame
when the dataset does not keep the augmentation consistent)."""
import itertools

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler

__all__ = ["GroupInBatchSampler", "sync_random_seed"]


def _dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sync_random_seed(seed=None, device="cuda"):
    """Rank 0's seed on every rank (one int32 broadcast when world_size > 1)."""
    if seed is None:
        seed = np.random.randint(2**31)
    assert isinstance(seed, int)
    rank, world = _dist_info()
    if world == 1:
        return seed
    if dist.get_backend() == "gloo":
        device = "cpu"
    value = torch.tensor(seed if rank == 0 else 0, dtype=torch.int32, device=device)
    dist.broadcast(value, src=0)
    return value.item()


class GroupInBatchSampler(Sampler):
    def __init__(self, dataset, batch_size=1, world_size=None, rank=None, seed=0, skip_prob=0., sequence_flip_prob=0.):
        _rank, _world = _dist_info()
        self.dataset = dataset
        self.batch_size = batch_size
        self.world_size = _world if world_size is None else world_size
        self.rank = _rank if rank is None else rank
        self.seed = sync_random_seed(seed)
        self.size = len(dataset)
        assert hasattr(dataset, "flag")
        self.flag = np.asarray(dataset.flag)
        self.group_sizes = np.bincount(self.flag)
        self.groups_num = len(self.group_sizes)
        self.global_batch_size = batch_size * self.world_size
        assert self.groups_num >= self.global_batch_size
        order = np.argsort(self.flag, kind="stable")
        ends = np.cumsum(self.group_sizes)
        self.group_idx_to_sample_idxs = {g: order[ends[g] - self.group_sizes[g]:ends[g]].tolist()
                                         for g in range(self.groups_num)}
        self._streams = [self._slot_stream(self.rank * batch_size + i) for i in range(batch_size)]
        self._frames = [[] for _ in range(batch_size)]
        self._aug = [None] * batch_size
        self.skip_prob = skip_prob
        self.sequence_flip_prob = sequence_flip_prob

    def _group_stream(self):
        g = torch.Generator()
        g.manual_seed(self.seed)
        while True:
            yield from torch.randperm(self.groups_num, generator=g).tolist()

    def _slot_stream(self, global_slot):
        return itertools.islice(self._group_stream(), global_slot, None, self.global_batch_size)

    def __iter__(self):
        while True:
            batch = []
            for slot in range(self.batch_size):
                frames = self._frames[slot]
                skip = np.random.uniform() < self.skip_prob and len(frames) > 1
                if not frames:
                    frames = list(self.group_idx_to_sample_idxs[next(self._streams[slot])])
                    if np.random.uniform() < self.sequence_flip_prob:
                        frames.reverse()
                    self._frames[slot] = frames
                    if self.dataset.keep_consistent_seq_aug:
                        self._aug[slot] = self.dataset.get_augmentation()
                if not self.dataset.keep_consistent_seq_aug:
                    self._aug[slot] = self.dataset.get_augmentation()
                if skip:
                    frames.pop(0)
                batch.append(dict(idx=frames.pop(0), aug_config=self._aug[slot]))
            yield batch

    def __len__(self):
        return self.size

    def set_epoch(self, epoch):
        self.epoch = epoch
