"""Kernel-map construction in isolation (GPU box): the stem's 825 k x 27 table and the whole set of a ResNet14 pass,
through the per-voxel hash map (mink_kernel_map) and through the block index (mink_kernel_map_batch)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import make_batches
from nerf_downstream_amd import minkowski as ME

dev = torch.device("cuda", 0)
b = make_batches(1, 16, 0, 51, 128, 28)[0]
coords = b["coordinates"].to(dev)
tf = ME.TensorField(coordinates=coords, features=torch.zeros(coords.shape[0], 4, device=dev))
m = tf.coordinate_manager
keys = {1: ME.CoordinateMapKey(1)}
for ts in (2, 4, 8, 16, 32):
    keys[ts] = m.stride(keys[ts // 2], 2)
OPS = [("ktable", 1, 1, 3, 1, False), ("ktable", 1, 2, 2, 1, False)]
for ts in (2, 4, 8, 16):
    OPS += [("ktable", ts, 2 * ts, 3, 1, True), ("ktable", ts, 2 * ts, 1, 1, True), ("ktable", 2 * ts, 2 * ts, 3, 1, False)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def lazy_stem():
    m.tables.pop((1, 1, 3, 1), None)
    m.kernel_table(keys[1], keys[1], 3, 1)


def batched(ops):
    def run():
        m.tables.clear()
        m._build_tables_batched(ops)
    return run


print(f"stem table {m.levels[1].n} x 27: per-voxel hash {timeit(lazy_stem):.0f} us, block index (incl. building the index) {timeit(batched(OPS[:1])):.0f} us")
print(f"all {len(OPS)} tables of a ResNet14 pass through the block index: {timeit(batched(OPS)):.0f} us")
