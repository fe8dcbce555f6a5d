"""mink_sgd_step alone (GPU box): bytes moved / time at ResNet14 and ResNet34 sizes.  usage: python scripts/sgd_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from nerf_downstream_amd._lib import check, lib

L = lib()
dev = torch.device("cuda", 0)
for n in (2_800_000, 21_300_000, 85_000_000):
    n = n // 4 * 4
    w, g, m = (torch.randn(n, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    for zero in (1, 0):
        for _ in range(3):
            check(L.mink_sgd_step(w.data_ptr(), g.data_ptr(), m.data_ptr(), n, 0.01, 0.9, 1e-4, zero, st))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            check(L.mink_sgd_step(w.data_ptr(), g.data_ptr(), m.data_ptr(), n, 0.01, 0.9, 1e-4, zero, st))
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        by = n * 4 * (5 + zero)
        print(f"n = {n / 1e6:5.1f} M, clear gradients {zero}: {us:7.1f} us, {by / us / 1e6:.2f} TB/s")
