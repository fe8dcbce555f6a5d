"""GPU box: data-gradient gather-GEMM reading W[k]^T in place (w_transposed) vs from a materialised transpose."""
import sys
import torch
sys.path.insert(0, ".")
from bench import make_batches
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd.minkowski import functional as Fn

dev = torch.device("cuda", 0)
b = make_batches(1, 16, 0, 51, 128, 28)[0]
tf = ME.TensorField(coordinates=b["coordinates"].to(dev), features=b["features"].to(dev))
x = tf.sparse()
m = x.coordinate_manager

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

keys = {1: ME.CoordinateMapKey(1)}
for ts in (2, 4, 8, 16):
    keys[ts] = m.stride(keys[ts // 2], 2)
for ts, cin, cout in ((1, 96, 96), (1, 128, 96), (2, 64, 64), (4, 128, 128), (8, 256, 256), (16, 512, 512)):
    nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
    n = nbr.shape[0]
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    gy = torch.randn(n, cout, device=dev)
    xin = torch.randn(n, cin, device=dev)
    t_f = timeit(lambda: Fn.gather_gemm(xin, w, nbr, cout))
    t_a = timeit(lambda: Fn.gather_gemm(gy, w, nbr, cin, w_transposed=True, flip_k=True))
    wt = w.transpose(1, 2).contiguous()
    t_b = timeit(lambda: Fn.gather_gemm(gy, wt, nbr, cin, flip_k=True))
    t_t = timeit(lambda: w.transpose(1, 2).contiguous())
    a = Fn.gather_gemm(gy, w, nbr, cin, w_transposed=True, flip_k=True)
    bb = Fn.gather_gemm(gy, wt, nbr, cin, flip_k=True)
    print(f"ts={ts} n={n} {cin}->{cout}: fwd {t_f:.1f} us | dgrad in-place W^T {t_a:.1f} us | materialised {t_b:.1f} us (+{t_t:.1f} us transpose) | max diff {float((a-bb).abs().max()):.2e}")
