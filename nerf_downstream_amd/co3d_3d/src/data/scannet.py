"""PeRFception-ScanNet plenoxel dataset for the segmentation family (counterpart of the reference's
`PlenoxelScannetDataset`, co3d_3d/src/data/scannet.py:450-660; Res16UNet is trained on it by
configs/scannet_plenoxel.gin).

Scene directory `<data_root>/plenoxel_torch_<scene>/data.npz`:
    links int [N] flat index into the `reso` grid, density f32 [N,1], sh uint8 [N,27] (* sh_scale + sh_min),
    reso int [3], labels int [N] (ScanNet-40 ids), dists f32 [N] (distance of the voxel to the labelled mesh)
`<dirname(data_root)>/split/scannet_256_{train,val}.txt` list the scenes, `split/scene_scales.data` (pickle) holds the
scale of every scene.  A sample, as in the reference (:585-653):

* voxels farther than `valid_thres` from the mesh get `void_label`; with `ignore_thres` the ones beyond it are dropped;
* `downsample_mode = 1` keeps the voxels whose grid coordinates are multiples of `downsample_stride`;
* coordinates = ((grid / reso) * 2 - 1) / scene_scale / voxel_size  -- metric voxels of `voxel_size`, NOT integers:
  `TensorField.sparse()` floors them and averages the features that share a voxel, `out.slice(field)` carries the
  prediction back to every input row (models/mink/res16unet.py);
* features selected by name from [dists | density (max-normalised when more than one feature) | sh | ones];
* labels mapped from the 40 raw ids to the 20 evaluated classes, everything else to `ignore_label`.

The reference's CPU augmentations for this data set (RandomCrop, ElasticDistortion, ...) are outside this path's scope;
only an empty transformation list is accepted."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.safe_load import load_plain_pickle

CLASS_LABELS = ("wall", "floor", "cabinet", "bed", "chair", "sofa", "table", "door", "window", "bookshelf", "picture", "counter",
                "desk", "curtain", "refrigerator", "shower curtain", "toilet", "sink", "bathtub", "otherfurniture")
VALID_CLASS_IDS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39)


@gin.configurable
class PlenoxelScannetDataset(Dataset):
    NUM_LABELS = 41  # raw ids; mapped onto the 20 evaluated classes
    IGNORE_LABELS = tuple(set(range(NUM_LABELS)) - set(VALID_CLASS_IDS))
    DATA_PATH_FILE = {"train": "scannet_256_train.txt", "val": "scannet_256_val.txt", "test": "scannet_256_val.txt"}
    CLASS_LABELS = CLASS_LABELS
    VALID_CLASS_IDS = VALID_CLASS_IDS

    def __init__(self, phase, data_root="co3d_3d/datasets/co3d", train_transformations=(), eval_transformations=(),
                 downsample_mode=1, downsample_stride=2, voxel_size=0.02, num_points=-1, features=("sh",), ignore_label=-100,
                 void_label=None, valid_thres=0.05, ignore_thres=None):
        phase = "test" if phase in ("val", "test") else "train"
        names = list(train_transformations if phase == "train" else eval_transformations)
        if names:
            raise NotImplementedError(f"augmentations {names} of the ScanNet recipe run on the CPU in the reference and are "
                                      "outside the scope of this path: pass an empty transformation list")
        if downsample_mode != 1:
            raise NotImplementedError("downsample_mode 0 (average pooling in the loader) is not implemented; mode 1 sub-samples")
        self.phase, self.data_root, self.features = phase, data_root, list(features)
        self.voxel_size, self.ignore_label = voxel_size, ignore_label
        self.void_label = void_label if void_label is not None else ignore_label
        self.valid_thres, self.ignore_thres, self.downsample_stride = valid_thres, ignore_thres, downsample_stride
        split = os.path.join(os.path.dirname(self.data_root), "split")
        with open(os.path.join(split, self.DATA_PATH_FILE[phase])) as f:
            self.files = [line.strip("\n") for line in f if line.strip() and not line.startswith("#")]
        label_map, n_used = {}, 0
        for raw in range(self.NUM_LABELS):
            if raw in self.IGNORE_LABELS:
                label_map[raw] = ignore_label
            else:
                label_map[raw] = n_used
                n_used += 1
        label_map[ignore_label] = ignore_label
        if void_label is not None and void_label != ignore_label:
            label_map[void_label] = n_used
        self.label_map = label_map
        self._lut = np.full(self.NUM_LABELS, ignore_label, dtype=np.int64)
        for raw, v in label_map.items():
            if 0 <= raw < self.NUM_LABELS:
                self._lut[raw] = v
        self.scene_scales = load_plain_pickle(os.path.join(split, "scene_scales.data"))  # {scene: scale}: plain data only
        self.NUM_CLASSES = len(self.CLASS_LABELS)

    def load_data(self, inst_id):
        z = np.load(os.path.join(self.data_root, f"plenoxel_torch_{inst_id}", "data.npz"))
        links = z["links"].astype(np.int64)
        density = z["density"].astype(np.float32).reshape(-1, 1)
        sh = z["sh"].astype(np.float32) * z["sh_scale"] + z["sh_min"]
        labels = z["labels"].astype(np.int64).reshape(-1, 1).copy()
        dists = z["dists"].astype(np.float32).reshape(-1, 1)
        labels[dists > self.valid_thres] = self.void_label
        if self.ignore_thres is not None and self.ignore_thres > 0:
            valid = (dists < self.ignore_thres).reshape(-1)
            links, sh, density, labels = links[valid], sh[valid], density[valid], labels[valid]
            # (the reference keeps `dists` unfiltered here, scannet.py:577-583, which only works when nothing is dropped)
            dists = dists[valid]
        return links, density, sh.reshape(len(links), -1).astype(np.float32), np.asarray(z["reso"]).astype(np.int64), labels, dists

    def __getitem__(self, index):
        inst_id = self.files[index]
        links, density, sh, reso, labels, dists = self.load_data(inst_id)
        grid = np.stack([links // (reso[1] * reso[2]), links % (reso[1] * reso[2]) // reso[2], links % reso[2]], 1).astype(np.float32)
        if len(self.features) > 1:
            density = density / (np.abs(density).max() + 1e-5)
        sel = (grid % self.downsample_stride == 0).all(axis=1)  # downsample_mode 1
        grid, dists, density, sh, labels = grid[sel], dists[sel], density[sel], sh[sel], labels[sel]
        xyzs = ((grid / reso.astype(np.float32) * 2 - 1.0) / np.float32(self.scene_scales[inst_id]) / np.float32(self.voxel_size)).astype(np.float32)
        cols = {"xyzs": xyzs, "dists": dists, "density": density.astype(np.float32), "sh": sh, "ones": np.ones_like(density, dtype=np.float32)}
        features = np.concatenate([cols[f] for f in self.features], axis=1).astype(np.float32)
        raw = labels.reshape(-1)
        mapped = np.where((raw >= 0) & (raw < self.NUM_LABELS), self._lut[np.clip(raw, 0, self.NUM_LABELS - 1)],
                          np.where(raw == self.void_label, self.label_map.get(self.void_label, self.ignore_label), self.ignore_label))
        return {"coordinates": torch.from_numpy(xyzs), "features": torch.from_numpy(features), "xyzs": torch.from_numpy(xyzs),
                "labels": mapped.astype(np.int64), "dists": dists.reshape(-1, 1), "metadata": {"file": inst_id}}

    def __len__(self):
        return len(self.files)

    def __repr__(self):
        return f"{self.__class__.__name__}(phase={self.phase}, length={len(self)})"
