"""Batch collation (counterpart of the reference's co3d_3d/src/data/utils.py:25-50)."""
import numpy as np
import torch

from nerf_downstream_amd.minkowski import utils as me_utils


def collate_mink(list_data):
    """list of sample dicts -> {"coordinates": f32 [sumN,4] (batch,x,y,z), "features": f32 [sumN,C],
    "labels": int64 [B]}.  Runs in DataLoader workers: CPU only."""
    coords, feats = me_utils.sparse_collate(
        [d["coordinates"] for d in list_data], [d["features"] for d in list_data], dtype=torch.float32
    )
    package = {"coordinates": coords, "features": feats}
    for key in list_data[0].keys():
        if "label" in key or "instance" in key or "dists" in key:
            package[key] = torch.from_numpy(np.concatenate([np.asarray(d[key]) for d in list_data]))
    for key in ("metadata", "dataset", "colors"):
        if key in list_data[0]:
            package[key] = [d[key] for d in list_data]
    return package
