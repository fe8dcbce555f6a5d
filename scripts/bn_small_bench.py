"""Times the one-launch batch norm of few-row layers (mink_bn_small_fwd / _bwd) at the shapes of the deep stages, for both
channel widths per workgroup.  Slabs are written by another kernel first (a copy), so they come from memory / other XCDs' L2
as they do behind the convolution.  usage: python scripts/bn_small_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from nerf_downstream_amd._lib import check, lib

L = lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for n, C, nslab in [(532, 256, 14), (128, 512, 14), (532, 256, 5), (1000, 128, 7), (512, 512, 14), (128, 512, 2)]:
    src = torch.randn(nslab, n, C, device=dev)
    y, out = torch.empty(n, C, device=dev), torch.empty(n, C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, invstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dx, dg, db = torch.empty(n, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    res = {}
    for width in (16, 8):
        L.mink_bn_set_small(width)
        for kind in ("fwd", "bwd"):
            ts = []
            for rep in range(30):
                slabs = src.clone()  # fresh lines, written by a kernel whose workgroups sit on every XCD
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if kind == "fwd":
                    check(L.mink_bn_small_fwd(slabs.data_ptr(), nslab, n, C, y.data_ptr(), 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), None, 1,
                                              out.data_ptr(), mean.data_ptr(), invstd.data_ptr(), None, None, st))
                else:
                    check(L.mink_bn_small_bwd(slabs.data_ptr(), nslab, None, dx.data_ptr(), y.data_ptr(), out.data_ptr(), n, C, mean.data_ptr(),
                                              invstd.data_ptr(), gamma.data_ptr(), 1, dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), st))
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            res[(width, kind)] = ts[len(ts) // 2]
    L.mink_bn_set_small(1)
    print(f"n={n} C={C} nslab={nslab}: " + "  ".join(f"{k[1]}/{k[0]}ch {v:.1f} us" for k, v in res.items()), flush=True)
