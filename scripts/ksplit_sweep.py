"""Split-K factor sweep of the mid-layer forward / data-gradient convolutions in one math mode (run on the GPU box).
usage: python scripts/ksplit_sweep.py [fp32|bf16] [reps] [scenes per batch = 16]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import make_batches
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd._lib import lib
from nerf_downstream_amd.minkowski import functional as Fn

math = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
b = make_batches(1, B, 0, 51, 128, 28)[0]
x0 = ME.TensorField(coordinates=b["coordinates"].to(dev), features=b["features"].to(dev)).sparse()
m = x0.coordinate_manager
ME.set_conv_math(math)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def layer(name, ts_in, ts_out, cin, cout):
    kin, kout = ME.CoordinateMapKey(ts_in), ME.CoordinateMapKey(ts_out)
    if ts_out != ts_in:
        m.stride(kin, 2)
    nbr, nbr_t = m.kernel_table(kin, kout, 3, 1, transposed=(ts_in != ts_out))
    n_in, n_out = m.levels[ts_in].n, nbr.shape[0]
    x = torch.randn(n_in, cin, device=dev); gy = torch.randn(n_out, cout, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    perm = m.class_perm(kin) if ts_in != ts_out else None
    for kind in ("fwd", "dgrad"):
        plan = lib().mink_conv_plan(n_out if kind == "fwd" else (perm.numel() if perm is not None else n_out), 27,
                                    cin if kind == "fwd" else cout, cout if kind == "fwd" else cin, int(kind == "dgrad" and perm is not None))
        res = []
        for ks in (1, 2, 3, 4, 5, 7, 9, 14, 27):
            Fn._FORCE_KSPLIT = ks
            try:
                if kind == "fwd":
                    t = timeit(lambda: Fn.gather_gemm(x, w, nbr, cout, stats=True))
                elif perm is None:
                    t = timeit(lambda: Fn.gather_gemm(gy, w, nbr, cin, w_transposed=True, flip_k=True))
                else:
                    t = timeit(lambda: Fn.gather_gemm(gy, w, nbr_t, cin, w_transposed=True, row_perm=perm))
                res.append(f"{ks}:{t:6.1f}")
            except Exception as e:  # a split the kernel of this shape does not take
                res.append(f"{ks}:   n/a")
            finally:
                Fn._FORCE_KSPLIT = 0
        print(f"{name:6s} {kind:5s} plan {plan:2d} | " + "  ".join(res))


m.stride(ME.CoordinateMapKey(1), 2)
layer("l1.c1", 2, 4, 64, 64); layer("l1.c2", 4, 4, 64, 64)
layer("l2.c1", 4, 8, 64, 128); layer("l2.c2", 8, 8, 128, 128)
layer("l3.c1", 8, 16, 128, 256); layer("l3.c2", 16, 16, 256, 256)
layer("l4.c1", 16, 32, 256, 512); layer("l4.c2", 32, 32, 512, 512)
