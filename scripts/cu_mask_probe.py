"""What a CU-subset stream (hipExtStreamCreateWithCUMask) costs on this stack: small-launch rate and large-kernel
throughput on streams confined to 256 / 128 / 64 / 32 compute units, alone and beside a busy default stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_downstream_amd.minkowski import functional as Fn

dev = torch.device("cuda", 0)
big = torch.randn(64 << 20, device=dev)
small = torch.randn(4096, device=dev)
a = torch.randn(8192, 8192, device=dev)


def timed(stream, fn, reps):
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for spec in ("", "256", "128", "64", "32", "128:64"):
    os.environ["MINK_CUS_PROBE"] = spec
    st = Fn.new_stream(dev, "probe")
    t_small = timed(st, lambda: small.mul_(1.0001), 400)
    t_big = timed(st, lambda: big.mul_(1.0001), 20)
    t_mm = timed(st, lambda: torch.mm(a, a), 3)
    # beside a busy default stream
    torch.cuda.synchronize()
    for _ in range(6):
        torch.mm(a, a)
    t_small_busy = timed(st, lambda: small.mul_(1.0001), 400)
    torch.cuda.synchronize()
    print(f"CUs {spec or 'all (plain stream)':>18s}: tiny launch {t_small:6.1f} us   256 MB scale {t_big:7.1f} us   8192^3 fp32 mm {t_mm / 1e3:7.2f} ms   "
          f"tiny launch beside a busy default stream {t_small_busy:6.1f} us")
