"""Data loaders (counterpart of the reference's co3d_3d/src/data/data_module.py:12-98).

Per-rank batch = `batch_size` (reference semantics: the DataLoader is built with
batch_size=self.batch_size and Lightning adds a DistributedSampler), so the global batch is
batch_size x world."""
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .datasets import get_dataset
from .utils import collate_mink


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
            sampler = DistributedSampler(ds, self.world_size, self.rank, shuffle=shuffle, seed=self.seed, drop_last=shuffle)
        g = torch.Generator()
        g.manual_seed(self.seed)
        return DataLoader(ds, batch_size=batch_size, num_workers=workers, collate_fn=self.collate_fn,
                          shuffle=shuffle and sampler is None, sampler=sampler, pin_memory=False,
                          persistent_workers=workers > 0, drop_last=shuffle, generator=g)

    def train_dataloader(self):
        if not hasattr(self, "train_dataset"):
            self.train_dataset = self.get_dataset(self.train_phase)
        workers = min(max(self.batch_size // self.world_size, 2), self.train_num_workers) if self.train_num_workers else 0
        return self._loader(self.train_dataset, self.batch_size, workers, True)

    def val_dataloader(self):
        if not hasattr(self, "val_dataset"):
            self.val_dataset = self.get_dataset(self.val_phase)
        return self._loader(self.val_dataset, self.val_batch_size, min(self.val_batch_size, self.val_num_workers), False)
