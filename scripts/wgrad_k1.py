"""GPU box: weight gradient of the 1x1x1 stride-2 shortcut convolutions (K = 1) in isolation."""
import sys
import torch
sys.path.insert(0, ".")
from bench import make_batches
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd.minkowski import functional as Fn
from nerf_downstream_amd._lib import lib

dev = torch.device("cuda", 0)
b = make_batches(1, 16, 0, 51, 128, 28)[0]
tf = ME.TensorField(coordinates=b["coordinates"].to(dev), features=b["features"].to(dev))
x = tf.sparse()
m = x.coordinate_manager

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

keys = {1: ME.CoordinateMapKey(1)}
for ts in (2, 4, 8, 16, 32):
    keys[ts] = m.stride(keys[ts // 2], 2)
for ts, cin, cout in ((2, 64, 64), (4, 64, 128), (8, 128, 256), (16, 256, 512)):
    nbr, _ = m.kernel_table(keys[ts], keys[ts * 2], 1, 1)
    n_in, n_out = m.size(keys[ts]), nbr.shape[0]
    xin = torch.randn(n_in, cin, device=dev)
    gy = torch.randn(n_out, cout, device=dev)
    for force in (0, ):
        t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (1, cin, cout)))
        has = (nbr[:, 0] >= 0)  # a coarse voxel only pairs with the fine voxel at exactly its coordinate
        ref = (xin[nbr[:, 0].clamp_min(0).long()] * has[:, None]).double().t() @ gy.double()
        got = Fn.conv_wgrad(xin, gy, nbr, (1, cin, cout))[0].double()
        print(f"ts={ts} K=1 n_in={n_in} n_out={n_out} {cin}->{cout}: wgrad {t:.1f} us ({2e-6*n_out*cin*cout/t:.2f} TF/s), rel err {float((got-ref).abs().max()/ref.abs().max()):.1e}")
    t = timeit(lambda: (xin[nbr[:, 0].clamp_min(0).long()] * has[:, None]).t() @ gy)
    print(f"      torch gather + GEMM: {t:.1f} us")
