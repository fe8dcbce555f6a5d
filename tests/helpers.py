"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np


def shell_scene(seed, grid=32, cin=28, drop=0.1, batch=0, negative=False):
    """Small plenoxel-like scene: ellipsoid shell in a grid^3 volume, integer coords."""
    rng = np.random.default_rng(seed)
    g = np.arange(grid)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    c = (grid - 1) / 2.0
    r = np.sqrt(((x - c) / (0.34 * grid)) ** 2 + ((y - c) / (0.28 * grid)) ** 2 + ((z - c) / (0.31 * grid)) ** 2)
    occ = np.abs(r - 1.0) < 0.12
    occ &= rng.random(occ.shape) > drop
    xyz = np.stack(np.nonzero(occ), 1).astype(np.float32)
    if negative:
        xyz -= grid // 2
    feats = rng.standard_normal((xyz.shape[0], cin)).astype(np.float32)
    return xyz, feats


def batch_scenes(seeds, **kw):
    import torch

    cs, fs = [], []
    for j, s in enumerate(seeds):
        c, f = shell_scene(s, **kw)
        cs.append(np.concatenate([np.full((c.shape[0], 1), j, np.float32), c], 1))
        fs.append(f)
    return torch.from_numpy(np.concatenate(cs)), torch.from_numpy(np.concatenate(fs))


def trunk_node(out):
    """The native-trunk autograd node (minkowski/trunk.py: TrunkFunction) in the graph of `out`, or None when the forward
    pass went module by module.  (`model._trunk_plan` only says the model COULD take the native trunk: a stem too small for
    the streaming weight-gradient kernel -- fewer than ~44 k voxels -- takes the module path.)"""
    seen, stack = set(), [out.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        if hasattr(fn, "saved") and hasattr(fn, "plan"):
            return fn
        stack += [f for f, _ in fn.next_functions]
    return None
