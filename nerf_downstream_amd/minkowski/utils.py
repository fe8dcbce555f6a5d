"""ME.utils subset.  Runs in DataLoader worker processes (reference data_module.py:54-65), so it
is pure-CPU torch and must never touch HIP."""
import ctypes

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


def decode_plenoxel_batch(batch, reso=None):
    """Compact PeRFception `data.npz` batch (device tensors: links, density, sh_q, scene_offsets,
    sh_scale, sh_min + `feature_names`) -> (coordinates int32 [N,4], features f32 [N,C]) with one
    HIP kernel (`mink_decode_plenoxel`; reference co3d.py:160-166,196-229)."""
    import torch

    from .._lib import check, lib

    links = batch["links"]
    if not links.is_cuda:
        raise RuntimeError("decode_plenoxel_batch runs on the GPU: move the batch to cuda first")
    reso = tuple(reso or batch.get("reso") or (128, 128, 128))  # data.npz: 128^3; last.ckpt scenes: 256^3
    names = list(batch["feature_names"])
    width = {"density": 1, "sh": 27, "ones": 1, "xyzs": 3}
    col, C = {"density": -1, "sh": -1, "ones": -1, "xyzs": -1}, 0
    for f in names:
        if f not in width or col[f] >= 0:
            raise ValueError(f"feature {f!r} cannot be decoded on the GPU (supported once each: xyzs, density, sh, ones)")
        col[f] = C
        C += width[f]
    n = links.shape[0]
    coords = torch.empty(n, 4, dtype=torch.int32, device=links.device)
    feats = torch.empty(n, C, dtype=torch.float32, device=links.device)
    n_scenes = batch["scene_offsets"].numel() - 1
    scratch = torch.empty(n_scenes, dtype=torch.float32, device=links.device) if col["xyzs"] >= 0 else None  # per-scene max norm
    stream = torch._C._cuda_getCurrentRawStream(links.device.index)
    check(
        lib().mink_decode_plenoxel(
            links.data_ptr(), batch["density"].data_ptr(), batch["sh_q"].data_ptr(), batch["scene_offsets"].data_ptr(),
            batch["scene_offsets"].numel() - 1, batch["sh_scale"].data_ptr(), batch["sh_min"].data_ptr(), n, reso[1], reso[2],
            col["density"], col["sh"], col["ones"], col["xyzs"], None if scratch is None else scratch.data_ptr(), C,
            coords.data_ptr(), feats.data_ptr(), C, stream,
        )
    )
    return coords, feats


def augment_batch(coords, feats, scene_offsets, params, streams, seed, raw_cols, count_async=False):
    """Apply the drawn per-scene augmentation programs to a whole batch with `mink_augment_scenes`
    (reference transforms.py, applied per scene on the CPU at co3d.py:216-219).

    coords int32/f32 [N,4] (batch,x,y,z) sorted by batch, feats f32 [N,C], scene_offsets int32 [S+1],
    params f32 [S, MINK_AUG_PARAMS], streams int32 [S] (uint32 bits) -- all on the device; `raw_cols`
    a host list (transforms.raw_columns).  Returns float coordinates and features of the surviving
    voxels.  The survivor count is read back (one small synchronisation of the current stream) unless
    `params` is a host tensor whose DROPOUT column is all zero (then every voxel survives).
    `count_async=True` never blocks: it returns (coords, feats, pending) with full-length buffers and
    pending = None or (pinned int32 count, event recorded after its copy) for the caller to slice later."""
    import torch

    from .._lib import check, lib

    if not coords.is_cuda:
        raise RuntimeError("augment_batch runs on the GPU: move the batch to cuda first")
    n, C = coords.shape[0], feats.shape[1]
    dev = coords.device
    if coords.dtype not in (torch.int32, torch.float32) or feats.dtype != torch.float32:
        raise TypeError("augment_batch: coordinates int32 or float32, features float32")
    coords, feats = coords.contiguous(), feats.contiguous()
    host_params = params if not params.is_cuda else None
    params = params.to(dev, torch.float32, non_blocking=True).contiguous()
    out_c = torch.empty(n, 4, dtype=torch.float32, device=dev)
    out_f = torch.empty(n, C, dtype=torch.float32, device=dev)
    kept = torch.empty(1, dtype=torch.int32, device=dev)
    n_scenes = scene_offsets.numel() - 1
    offs = scene_offsets.to(dev, torch.int32, non_blocking=True).contiguous()
    strm = streams.to(dev, torch.int32, non_blocking=True).contiguous()
    ws = torch.empty(max(1, lib().mink_augment_workspace_bytes(n, n_scenes)), dtype=torch.uint8, device=dev)
    cols = (ctypes.c_int32 * C)(*[int(c) for c in raw_cols])
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    check(
        lib().mink_augment_scenes(
            coords.data_ptr(), int(coords.dtype == torch.int32), feats.data_ptr(), C, C, n,
            offs.data_ptr(), n_scenes, params.data_ptr(), strm.data_ptr(),
            int(seed) & (2 ** 64 - 1), ctypes.cast(cols, ctypes.c_void_p), out_c.data_ptr(), out_f.data_ptr(), C,
            kept.data_ptr(), ws.data_ptr(), ws.numel(), stream,
        )
    )
    # column 37 = MINK_AUG_DROPOUT: without dropout every voxel survives and the count is known on the host
    if host_params is not None and not bool((host_params[:, 37] != 0).any()):
        return (out_c, out_f, None) if count_async else (out_c, out_f)
    if count_async:
        count = torch.empty(1, dtype=torch.int32, pin_memory=True)
        count.copy_(kept, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return out_c, out_f, (count, ev)
    k = int(kept.item())
    return out_c[:k], out_f[:k]


def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ for convolution kernels laid out (K, Cin, Cout): fan_in = K * Cin, fan_out = K * Cout."""
    import math

    k = tensor.shape[0] if tensor.dim() == 3 else 1
    fan = k * (tensor.shape[-2] if mode == "fan_in" else tensor.shape[-1])
    std = torch.nn.init.calculate_gain(nonlinearity, a) / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
