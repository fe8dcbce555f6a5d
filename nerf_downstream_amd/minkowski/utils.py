"""ME.utils subset.  Runs in DataLoader worker processes (reference data_module.py:54-65), so it
is pure-CPU torch and must never touch HIP."""
import numpy as np
import torch


def _t(a):
    return torch.from_numpy(a) if isinstance(a, np.ndarray) else a


def batched_coordinates(coords, dtype=torch.int32, device=None):
    n = [int(c.shape[0]) for c in coords]
    D = int(coords[0].shape[1])
    out = torch.zeros(sum(n), D + 1, dtype=dtype, device=device)
    s = 0
    for j, c in enumerate(coords):
        out[s : s + n[j], 1:] = _t(c).to(dtype)
        out[s : s + n[j], 0] = j
        s += n[j]
    return out


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """A11 (reference data/utils.py:25-30): concatenate per-sample coordinates with the batch
    index in column 0 and concatenate the features; labels (if given) are concatenated too."""
    bcoords = batched_coordinates(coords, dtype=dtype, device=device)
    bfeats = torch.cat([_t(f) for f in feats], 0)
    if labels is None:
        return bcoords, bfeats
    return bcoords, bfeats, torch.cat([_t(l) for l in labels], 0)
