"""Dataset registry (counterpart of the reference's co3d_3d/src/data/datasets.py:20-22)."""
from nerf_downstream_amd import gin_lite as gin

from .co3d import Co3D10pDataset, Co3DDataset
from .scannet import PlenoxelScannetDataset
from .synthetic import SparseVoxelDataset, SparseVoxelSegDataset

DATASETS = {c.__name__: c for c in (Co3DDataset, Co3D10pDataset, PlenoxelScannetDataset, SparseVoxelDataset, SparseVoxelSegDataset)}


@gin.configurable
def get_dataset(dataset_name: str):
    if dataset_name not in DATASETS:
        raise KeyError(f"dataset {dataset_name!r} not available; choose from {sorted(DATASETS)}")
    return DATASETS[dataset_name]
