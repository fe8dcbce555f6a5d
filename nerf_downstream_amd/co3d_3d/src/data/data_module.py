"""Data loaders (counterpart of the reference's co3d_3d/src/data/data_module.py:12-98).

Per-rank batch = `batch_size` (reference semantics: the DataLoader is built with
batch_size=self.batch_size and Lightning adds a DistributedSampler), so the global batch is
batch_size x world."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Sampler
from torch.utils.data.distributed import DistributedSampler

from .datasets import get_dataset
from .utils import collate_mink


class LengthBalancedDistributedSampler(Sampler):
    """DistributedSampler with length-aware placement (variable voxel counts per scene are the main
    data-parallel scaling hazard -- every step waits for the rank with the heaviest batch).

    Each global step still consumes the SAME set of `world x batch` samples a `DistributedSampler` with this seed would
    hand out (consecutive chunks of the epoch's permutation), so the averaged gradient of a step is unchanged; only
    which rank gets which scene differs: within a chunk, scenes are dealt longest first to the rank with the
    smallest voxel total that still has a free slot (LPT), which bounds a step's imbalance by one scene instead of
    letting it grow with the batch size.  Every rank computes the same assignment from the shared seed."""

    def __init__(self, lengths, batch_size, world_size, rank, seed=0):
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.batch_size, self.world, self.rank, self.seed, self.epoch = int(batch_size), int(world_size), int(rank), int(seed), 0
        self.chunk = self.batch_size * self.world
        self.num_chunks = len(self.lengths) // self.chunk  # drop_last, like the training loader

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_chunks * self.batch_size

    @staticmethod
    def assign(lengths, world, batch_size):
        """LPT with a slot limit: positions of `lengths` (one chunk) per rank, each list in dealing order."""
        order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
        load, slots = [0] * world, [[] for _ in range(world)]
        for i in order:
            r = min((q for q in range(world) if len(slots[q]) < batch_size), key=lambda q: (load[q], q))
            slots[r].append(int(i))
            load[r] += int(lengths[i])
        return slots

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        perm = torch.randperm(len(self.lengths), generator=g).numpy()
        for c in range(self.num_chunks):
            idx = perm[c * self.chunk : (c + 1) * self.chunk]
            mine = self.assign(self.lengths[idx], self.world, self.batch_size)[self.rank]
            yield from (int(idx[j]) for j in mine)


class EpochSampler(Sampler):
    """Single-rank shuffling whose order is a function of (seed, epoch) alone -- what DistributedSampler gives the
    multi-rank runs -- so a run resumed inside epoch e draws epoch e's permutation again, and `skip(k)` drops the k
    samples trained on before the checkpoint at the INDEX level (no scene is loaded to be thrown away)."""

    def __init__(self, n, seed=0):
        self.n, self.seed, self.epoch, self._skip = int(n), int(seed), 0, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def skip(self, samples):
        self._skip = int(samples)  # for the next iteration only

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        perm = torch.randperm(self.n, generator=g).tolist()
        k, self._skip = self._skip, 0
        return iter(perm[k:])


class DataModule:
    def __init__(self, train_phase="train", val_phase="val", test_phase="test", batch_size=12, val_batch_size=6,
                 train_num_workers=4, val_num_workers=2, collate_func_name="collate_mink", world_size=1, rank=0, seed=0):
        if collate_func_name != "collate_mink":
            raise ValueError(f"{collate_func_name} is not supported on the sparse-voxel path.")
        self.__dict__.update({k: v for k, v in locals().items() if k != "self"})
        self.collate_fn = collate_mink

    def get_dataset(self, phase="train"):
        return get_dataset()(phase=phase)

    def _loader(self, ds, batch_size, workers, shuffle):
        sampler = None
        if self.world_size > 1:
            lengths = ds.sample_lengths() if shuffle and hasattr(ds, "sample_lengths") else None
            if lengths is not None and len(lengths) >= batch_size * self.world_size:
                sampler = LengthBalancedDistributedSampler(lengths, batch_size, self.world_size, self.rank, seed=self.seed)
            else:
                sampler = DistributedSampler(ds, self.world_size, self.rank, shuffle=shuffle, seed=self.seed, drop_last=shuffle)
        elif shuffle:
            sampler = EpochSampler(len(ds), seed=self.seed)
        return DataLoader(ds, batch_size=batch_size, num_workers=workers, collate_fn=self.collate_fn,
                          shuffle=False, sampler=sampler, pin_memory=False,
                          persistent_workers=workers > 0, drop_last=shuffle)

    def train_dataloader(self):
        if not hasattr(self, "train_dataset"):
            self.train_dataset = self.get_dataset(self.train_phase)
        workers = min(max(self.batch_size // self.world_size, 2), self.train_num_workers) if self.train_num_workers else 0
        return self._loader(self.train_dataset, self.batch_size, workers, True)

    def val_dataloader(self):
        if not hasattr(self, "val_dataset"):
            self.val_dataset = self.get_dataset(self.val_phase)
        return self._loader(self.val_dataset, self.val_batch_size, min(self.val_batch_size, self.val_num_workers), False)
