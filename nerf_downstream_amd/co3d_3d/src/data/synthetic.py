"""Deterministic synthetic plenoxel voxel grids of the co3d_3d default shape.

`SparseVoxelDataset` returns the same sample dict as the reference's
`Co3DDatasetBase.__getitem__` (co3d_3d/src/data/co3d.py:231-242):
    {"coordinates": f32 [N,3] integer-valued (x,y,z), "features": f32 [N,C],
     "xyzs": f32 [N,3], "labels": np.int64 [1]}
with raw features laid out [xyz(3) | density(1) | sh(27)] and the selected columns named by
`features` exactly as in co3d.py:205-229 ("density", "sh", "xyzs", "ones").

Recipe: rng = default_rng(1_000_003*split + index); 128^3 grid
(`data.npz` resolution, co3d.py:171); occupancy = shell |r-1| < 0.036 of an ellipsoid centred
at 64 with semi-axes (44,36,40) (odd classes of the 2-class set-up use a box shell), 10 %
voxel drop-out -> ~51.7 k voxels; density ~ LogNormal(0,1); SH ~ N(mu_class, 0.5^2).
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from nerf_downstream_amd import gin_lite as gin

SPLIT_ID = {"train": 0, "val": 1, "test": 1, "trainval": 0}


def _class_sh_mean(label, num_classes):
    rng = np.random.default_rng(977 + 13 * label)
    return rng.normal(0.0, 0.6, 27).astype(np.float32)


def make_scene(index, split=0, num_classes=51, grid=128, box_for_odd=False, class_sep=1.0, scene_sigma=0.0):
    """`class_sep` scales the class-dependent SH mean and `scene_sigma` adds a per-scene offset N(0, scene_sigma^2) to
    it (one draw per scene and SH channel, from a generator of its own so the other draws do not move): with a small
    separation and a comparable scene offset the classes overlap and top-1 cannot saturate -- the fixed-split parity
    test (tests/test_gpu_parity_full.py) needs an accuracy that can actually differ between two implementations."""
    rng = np.random.default_rng(1_000_003 * split + index)
    label = index % num_classes
    g = np.arange(grid, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    s = grid / 128.0
    # class-dependent mild anisotropy keeps N within a few % of the nominal 51.7 k
    wob = 1.0 + 0.04 * np.sin(1.7 * label + np.array([0.0, 2.1, 4.2]))
    ax = np.array([44.0, 36.0, 40.0]) * s * wob
    c = grid / 2.0
    if box_for_odd and label % 2 == 1:
        r = np.maximum.reduce([np.abs(x - c) / ax[0], np.abs(y - c) / ax[1], np.abs(z - c) / ax[2]])
        occ = np.abs(r - 0.8) < 0.042
    else:
        r = np.sqrt(((x - c) / ax[0]) ** 2 + ((y - c) / ax[1]) ** 2 + ((z - c) / ax[2]) ** 2)
        occ = np.abs(r - 1.0) < 0.036 * (128.0 / grid if grid < 128 else 1.0)
    occ &= rng.random(occ.shape) >= 0.10
    xyz = np.stack(np.nonzero(occ), 1).astype(np.float32)  # sorted by (x,y,z) like `links`
    n = xyz.shape[0]
    density = rng.lognormal(0.0, 1.0, (n, 1)).astype(np.float32)
    mean = _class_sh_mean(label, num_classes) * np.float32(class_sep)
    if scene_sigma:
        mean = mean + np.random.default_rng(5_000_011 * (split + 1) + index).normal(0.0, scene_sigma, 27).astype(np.float32)
    sh = (mean + rng.normal(0.0, 0.5, (n, 27))).astype(np.float32)
    return xyz, density, sh, label


@gin.configurable
class SparseVoxelDataset(Dataset):
    def __init__(self, phase="train", num_samples=512, num_classes=51, grid=128, features=("density", "sh"),
                 box_for_odd=False, train_transformations=(), class_sep=1.0, scene_sigma=0.0):
        """`train_transformations`: names of data/transforms.py classes, drawn per scene for the training split
        and applied on the GPU (like Co3DDatasetBase)."""
        from . import transforms

        self.phase, self.split = phase, SPLIT_ID.get(phase, 2)
        names = list(train_transformations) if self.split == 0 else []
        self.transformations = transforms.Compose([getattr(transforms, t)() for t in names]) if names else None
        self.num_samples = num_samples if self.split == 0 else max(1, num_samples // 4)
        self.num_classes, self.grid, self.features, self.box_for_odd = num_classes, grid, list(features), box_for_odd
        self.NUM_CLASSES = num_classes
        self.class_sep, self.scene_sigma = float(class_sep), float(scene_sigma)

    def __len__(self):
        return self.num_samples

    def __getitem__(self, index):
        xyz, density, sh, label = make_scene(index, self.split, self.num_classes, self.grid, self.box_for_odd,
                                             self.class_sep, self.scene_sigma)
        coordinates = torch.from_numpy(xyz)
        # per-point mean over (x,y,z), as the reference does (co3d.py:211)
        xyzs = coordinates - coordinates.mean(dim=1, keepdim=True)
        xyzs = xyzs / torch.linalg.norm(xyzs, dim=1).max()
        cols = {"xyzs": xyzs, "density": torch.from_numpy(density), "sh": torch.from_numpy(sh)}
        cols["ones"] = torch.ones_like(cols["density"])
        feats = torch.cat([cols[f] for f in self.features], dim=1).float()
        sample = {"coordinates": coordinates, "features": feats, "xyzs": xyzs, "labels": np.array([label])}
        if self.transformations is not None:
            params, stream = self.transformations.sample()
            sample.update(aug_params=torch.from_numpy(params), aug_stream=stream, feature_names=tuple(self.features))
        return sample


@gin.configurable
class SparseVoxelSegDataset(SparseVoxelDataset):
    """Per-voxel labels for the segmentation family (Res16UNet; the reference trains it on PeRFception-ScanNet,
    data/scannet.py, which is not reproduced here): label = octant of the voxel about the grid centre, rotated by
    the scene's class so the features matter, `ignore_ratio` of the voxels carry `ignore_label` (ScanNet's
    unlabelled points).  Sample dict as the reference segmentation datasets: "labels" is int64 [N]."""

    def __init__(self, phase="train", num_samples=64, num_classes=8, grid=64, features=("density", "sh"),
                 ignore_label=255, ignore_ratio=0.05, train_transformations=()):
        super().__init__(phase, num_samples, num_classes, grid, features, False, train_transformations)
        self.ignore_label, self.ignore_ratio = ignore_label, ignore_ratio

    def __getitem__(self, index):
        if self.transformations is not None:
            raise NotImplementedError("augmentation with dropout would need the per-voxel labels compacted alongside")
        sample = super().__getitem__(index)
        xyz = sample["coordinates"].numpy()
        scene_class = int(sample["labels"][0])
        oct_ = ((xyz[:, 0] >= self.grid / 2).astype(np.int64) + 2 * (xyz[:, 1] >= self.grid / 2) + 4 * (xyz[:, 2] >= self.grid / 2))
        labels = (oct_ + scene_class) % min(self.num_classes, 8)
        rng = np.random.default_rng(7_000_003 * self.split + index)
        labels[rng.random(len(labels)) < self.ignore_ratio] = self.ignore_label
        sample["labels"] = labels.astype(np.int64)
        return sample
