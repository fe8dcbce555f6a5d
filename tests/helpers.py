"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import os

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


def host_threads(cap=16):
    """Cores this process may really use: the scheduler affinity mask capped by the cgroup CPU quota -- never
    os.cpu_count(), which on a shared GPU box reports the machine (128+) while the cgroup grants a fraction: oversubscribed
    BLAS / OpenMP threads made the oracle 12x slower there (BENCH_r03.json cpu_baseline: 128 threads 39.6 k voxels/s, 12
    threads 486.7 k)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(cap, n))
