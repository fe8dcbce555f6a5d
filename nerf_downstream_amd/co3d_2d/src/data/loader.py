"""Data of the 2-D baseline (counterpart of the reference's co3d_2d/src/data/loader.py:232-274): batches
{"images": float [B,3,224,224], "labels": int64 [B]}.  The reference reads rendered PeRFception / CO3D frames from
disk; there is no dataset here, so `SyntheticRenders` draws deterministic 224^2 "renders" -- a class-dependent blob on
a textured background, normalised like ImageNet inputs -- with the same sample dict."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from nerf_downstream_amd import gin_lite as gin


@gin.configurable
class SyntheticRenders(Dataset):
    def __init__(self, phase="train", num_samples=512, num_classes=51, size=224):
        self.split = 0 if phase == "train" else 1
        self.num_samples = num_samples if self.split == 0 else max(1, num_samples // 4)
        self.num_classes, self.size = num_classes, size

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        rng = np.random.default_rng(2_000_003 * self.split + idx)
        label = idx % self.num_classes
        crng = np.random.default_rng(4242 + label)
        s = self.size
        yy, xx = np.mgrid[0:s, 0:s].astype(np.float32) / s
        cx, cy, r = 0.3 + 0.4 * crng.random(), 0.3 + 0.4 * crng.random(), 0.12 + 0.2 * crng.random()
        colour = crng.random(3).astype(np.float32)
        blob = np.exp(-(((xx - cx - 0.05 * rng.standard_normal()) ** 2 + (yy - cy - 0.05 * rng.standard_normal()) ** 2) / (2 * r * r)))
        freq = 2 + 6 * crng.random(2)
        tex = 0.5 + 0.5 * np.sin(2 * np.pi * (freq[0] * xx + freq[1] * yy) + rng.random() * 6.28)
        img = colour[:, None, None] * blob[None] + 0.25 * tex[None] + 0.1 * rng.standard_normal((3, s, s)).astype(np.float32)
        img = (img - np.array([0.485, 0.456, 0.406], np.float32)[:, None, None]) / np.array([0.229, 0.224, 0.225], np.float32)[:, None, None]
        return {"images": torch.from_numpy(img.astype(np.float32)), "labels": torch.tensor(label, dtype=torch.int64)}


@gin.configurable
class DataModule:
    def __init__(self, num_workers=16, batch_size=32, chunks=32, train_co3d=True, eval_co3d=True, num_samples=512, size=224):
        self.num_workers, self.batch_size, self.chunks = num_workers, batch_size, chunks
        self.train_co3d, self.eval_co3d, self.num_samples, self.size = train_co3d, eval_co3d, num_samples, size

    def _loader(self, phase, batch_size, shuffle):
        ds = SyntheticRenders(phase, num_samples=self.num_samples, size=self.size)
        g = torch.Generator()
        g.manual_seed(0)
        return DataLoader(ds, batch_size=batch_size, num_workers=self.num_workers, shuffle=shuffle, drop_last=shuffle,
                          persistent_workers=self.num_workers > 0, generator=g)

    def train_dataloader(self):
        return self._loader("train", self.batch_size, True)

    def val_dataloader(self):
        return self._loader("val", self.chunks, False)

    def test_dataloader(self):
        return self._loader("test", self.chunks, False)
