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


def decode_plenoxel_batch(batch, reso=(128, 128, 128)):
    """Compact PeRFception `data.npz` batch (device tensors: links, density, sh_q, scene_offsets,
    sh_scale, sh_min + `feature_names`) -> (coordinates int32 [N,4], features f32 [N,C]) with one
    HIP kernel (`mink_decode_plenoxel`; reference co3d.py:160-166,196-229)."""
    import torch

    from .._lib import check, lib

    links = batch["links"]
    if not links.is_cuda:
        raise RuntimeError("decode_plenoxel_batch runs on the GPU: move the batch to cuda first")
    names = list(batch["feature_names"])
    width = {"density": 1, "sh": 27, "ones": 1}
    col, C = {"density": -1, "sh": -1, "ones": -1}, 0
    for f in names:
        if f not in width or col[f] >= 0:
            raise ValueError(f"feature {f!r} cannot be decoded on the GPU (supported once each: density, sh, ones)")
        col[f] = C
        C += width[f]
    n = links.shape[0]
    coords = torch.empty(n, 4, dtype=torch.int32, device=links.device)
    feats = torch.empty(n, C, dtype=torch.float32, device=links.device)
    stream = torch._C._cuda_getCurrentRawStream(links.device.index)
    check(
        lib().mink_decode_plenoxel(
            links.data_ptr(), batch["density"].data_ptr(), batch["sh_q"].data_ptr(), batch["scene_offsets"].data_ptr(),
            batch["scene_offsets"].numel() - 1, batch["sh_scale"].data_ptr(), batch["sh_min"].data_ptr(), n, reso[1], reso[2],
            col["density"], col["sh"], col["ones"], C, coords.data_ptr(), feats.data_ptr(), C, stream,
        )
    )
    return coords, feats
