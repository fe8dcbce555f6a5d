"""Micro-benchmark of the convolution kernels on one BASELINE-shape batch (run on the GPU box).
usage: python scripts/kbench.py [stem|l1|l4|all] [reps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import make_batches
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd.minkowski import functional as Fn

which = sys.argv[1] if len(sys.argv) > 1 else "stem"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if len(sys.argv) > 3:
    ME.set_conv_math(sys.argv[3])
    print("conv math:", sys.argv[3])
dev = torch.device("cuda", 0)
b = make_batches(1, 16, 0, 51, 128, 28)[0]
tf = ME.TensorField(coordinates=b["coordinates"].to(dev), features=b["features"].to(dev))
x = tf.sparse()
m = x.coordinate_manager
torch.manual_seed(0)

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def bench_layer(name, xin, in_key, out_key, ks, cin, cout, stride):
    nbr, nbr_t = m.kernel_table(in_key, out_key, ks, 1, transposed=(stride != 1))
    pairs = int((nbr >= 0).sum())
    w = torch.randn(ks ** 3, cin, cout, device=dev) * 0.05
    gy = torch.randn(nbr.shape[0], cout, device=dev)
    fl = 2.0 * pairs * cin * cout
    t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, cout), reps)
    print(f"{name:10s} fwd   n_out={nbr.shape[0]:7d} pairs={pairs:9d} {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s")
    if stride == 1:
        wt = w.transpose(1, 2).contiguous()
        t = timeit(lambda: Fn.gather_gemm(gy, wt, nbr, cin, flip_k=True), reps)
    else:
        perm = m.class_perm(in_key) if stride == 2 else None
        wt = w.transpose(1, 2).contiguous()
        t = timeit(lambda: Fn.gather_gemm(gy, wt, nbr_t, cin, row_perm=perm), reps)
    print(f"{name:10s} dgrad n_in ={xin.shape[0]:7d}                 {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s (materialised W^T)")
    if stride == 1:
        t = timeit(lambda: Fn.gather_gemm(gy, w, nbr, cin, w_transposed=True, flip_k=True), reps)
    else:
        t = timeit(lambda: Fn.gather_gemm(gy, w, nbr_t, cin, w_transposed=True, row_perm=perm), reps)
    print(f"{name:10s} dgrad (W read transposed in place)   {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s")
    t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, w.shape), reps)
    print(f"{name:10s} wgrad                                {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s")

k1 = ME.CoordinateMapKey(1)
if which == "stagger":
    from nerf_downstream_amd._lib import lib
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    w = torch.randn(27, 28, 64, device=dev) * 0.05
    xin = x.F.contiguous()
    for st in (0, 512, 0, 512, 0, 512):  # bit 9: without the flattened-K stem path
        lib().mink_conv_set_stagger(st)
        t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, 64), reps)
        print(f"stagger {st}: stem fwd {t*1e3:8.1f} us")
if which == "wablate":
    from nerf_downstream_amd._lib import lib
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    xin = x.F.contiguous()
    gy = torch.randn(nbr.shape[0], 64, device=dev)
    for st in (0, 1024, 0, 1024):
        lib().mink_conv_set_stagger(st)
        t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64)), reps)
        print(f"ablate {st}: stem wgrad {t*1e3:8.1f} us")
    lib().mink_conv_set_stagger(0)
    a = Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64))
    lib().mink_conv_set_stagger(1024)
    b = Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64))
    lib().mink_conv_set_stagger(0)
    print("stream vs tiled wgrad: max |diff|", float((a - b).abs().max()), "max |ref|", float(b.abs().max()))
if which == "wxcd":  # streaming stem wgrad: groups of a row split on one XCD (default) against plain workgroup order (bit 29)
    from nerf_downstream_amd._lib import lib
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    xin = x.F.contiguous()
    gy = torch.randn(nbr.shape[0], 64, device=dev)
    for st in (0, 1 << 29, 0, 1 << 29):
        lib().mink_conv_set_stagger(st)
        t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64)), reps)
        print(f"plain order {st >> 29}: stem wgrad {t*1e3:8.1f} us")
    lib().mink_conv_set_stagger(0)
    a = Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64))
    lib().mink_conv_set_stagger(1 << 29)
    b = Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64))
    lib().mink_conv_set_stagger(0)
    print("bitwise equal:", bool(torch.equal(a, b)))
if which == "decode":
    import numpy as np
    n = x.F.shape[0]
    rng = np.random.default_rng(0)
    batch = {
        "links": torch.from_numpy(rng.integers(0, 128 ** 3, n).astype(np.int32)).to(dev),
        "density": torch.rand(n, device=dev), "sh_q": torch.randint(0, 256, (n, 27), dtype=torch.uint8, device=dev),
        "scene_offsets": torch.arange(0, n + 1, n // 16, dtype=torch.int32, device=dev)[:17].contiguous(),
        "sh_scale": torch.rand(16, 27, device=dev), "sh_min": torch.rand(16, 27, device=dev), "feature_names": ("density", "sh"),
    }
    batch["scene_offsets"][-1] = n
    t = timeit(lambda: ME.utils.decode_plenoxel_batch(batch), reps)
    nbytes = n * (4 + 4 + 27 + 16 + 112)
    print(f"decode_plenoxel n={n}: {t*1e3:.1f} us, {nbytes/1e6:.1f} MB algorithmic -> {nbytes/t/1e6:.0f} GB/s ({nbytes/t/1e6/8000*100:.1f} % of 8 TB/s)")
if which == "augment":  # the full co3d_aug3 recipe on a whole batch (dropout, flip bounds, jitter, SH noise)
    import random

    import numpy as np
    from nerf_downstream_amd.co3d_3d.src.data import transforms as T
    n = x.F.shape[0]
    random.seed(0), np.random.seed(0)
    stages = [("linear", T.rotation_matrix([0.01, 1, 0.02], 1.0)), ("dropout", 0.2), ("flip", (0, 2)), ("translate", [0.1, -0.1, 0.05]),
              ("jitter", 1.0), ("linear", np.eye(3) * 1.2), ("feature_jitter", 0.01, 4, 27)]
    params = torch.from_numpy(np.stack([T.compile_program(stages)] * 16))
    offs = torch.arange(0, n + 1, n // 16, dtype=torch.int32)[:17].contiguous()
    offs[-1] = n
    streams = torch.arange(16, dtype=torch.int32)
    coords = x.C.contiguous()
    feats = torch.randn(n, 28, device=dev)
    raw = T.raw_columns(["density", "sh"])
    for label, pr in (("aug3 (with dropout: +1 read-back)", params), ("aug3 without dropout", params.clone())):
        if "without" in label:
            pr[:, T.AUG["DROPOUT"]] = 0
        t = timeit(lambda: ME.utils.augment_batch(coords, feats, offs, pr, streams, 1234, raw), reps)
        kept = ME.utils.augment_batch(coords, feats, offs, pr, streams, 1234, raw)[0].shape[0]
        nbytes = n * (16 + 112) + kept * (16 + 112)
        print(f"augment n={n} kept={kept} {label}: {t*1e3:.1f} us, {nbytes/1e6:.1f} MB algorithmic -> {nbytes/t/1e6:.0f} GB/s ({nbytes/t/1e6/8000*100:.1f} % of 8 TB/s)")
if which == "unet":  # Res16UNet forward + backward + SGD on the B=16 synthetic batch (per-voxel labels)
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    import torch.nn.functional as F
    import os
    for name in os.environ.get("KB_UNET", "Res16UNet14A,Res16UNet34C").split(","):
        torch.manual_seed(0)
        net = get_model(name, 28, 20).to(dev).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, fused=True)
        batch = {"coordinates": b["coordinates"].to(dev), "features": b["features"].to(dev)}
        labels = torch.randint(0, 20, (batch["coordinates"].shape[0],), device=dev)
        state = {"tf": net.process_input(batch)}
        def step():
            tf = state["tf"]
            nxt = net.process_input(batch, defer=True)
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(net(tf), labels)
            loss.backward()
            state["tf"] = net.finish_input(nxt)
            opt.step()
        for _ in range(3): step()
        t = timeit(step, reps)
        n = batch["coordinates"].shape[0]
        print(f"{name}: {t:.2f} ms/step fwd+bwd+SGD, {n} voxels -> {n/t/1e3:.1f} M voxels/s")
if which == "ksweep":
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[ts * 2]
        for name, ik, ok, ci, stride in ((f"l@{ts}.c1", keys[ts], keys[ts * 2], cin, 2), (f"l@{ts*2}.c2", keys[ts * 2], keys[ts * 2], cout, 1)):
            nbr, nbr_t = m.kernel_table(ik, ok, 3, 1, transposed=(stride != 1))
            xin = torch.randn(m.levels[ts if stride == 2 else ts * 2].n, ci, device=dev)
            w = torch.randn(27, ci, cout, device=dev) * 0.05
            res = []
            for zs in sorted({-(-27 // kper) for kper in range(1, 28)}):
                Fn._FORCE_KSPLIT = zs
                t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, cout), reps)
                res.append((zs, t * 1e3))
            Fn._FORCE_KSPLIT = 0
            t0 = timeit(lambda: Fn.gather_gemm(xin, w, nbr, cout), reps) * 1e3
            print(f"{name} fwd n_out={nbr.shape[0]} {ci}->{cout} planner={t0:.1f}us | " + " ".join(f"{z}:{t:.0f}" for z, t in res))
            if stride == 2:
                gy = torch.randn(nbr.shape[0], cout, device=dev)
                wt = w.transpose(1, 2).contiguous()
                perm = m.class_perm(ik)
                res = []
                for zs in sorted({-(-27 // kper) for kper in range(1, 28)}):
                    Fn._FORCE_KSPLIT = zs
                    t = timeit(lambda: Fn.gather_gemm(gy, wt, nbr_t, ci, row_perm=perm), reps)
                    res.append((zs, t * 1e3))
                Fn._FORCE_KSPLIT = 0
                t0 = timeit(lambda: Fn.gather_gemm(gy, wt, nbr_t, ci, row_perm=perm), reps) * 1e3
                print(f"{name} dgrad n_in={xin.shape[0]} {cout}->{ci} planner={t0:.1f}us | " + " ".join(f"{z}:{t:.0f}" for z, t in res))
if which == "compact":  # mid layers: row-compacted kernel (default) against gather_gemm2 (set_stagger bit 30)
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    tot = {0: 0.0, 1: 0.0}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[ts * 2]
        for name, ik, ok, ci, stride in ((f"l@{ts}.c1", keys[ts], keys[ts * 2], cin, 2), (f"l@{ts*2}.c2", keys[ts * 2], keys[ts * 2], cout, 1)):
            nbr, _ = m.kernel_table(ik, ok, 3, 1)
            fill = float((nbr >= 0).float().mean())
            xin = torch.randn(m.levels[ts if stride == 2 else ts * 2].n, ci, device=dev)
            w = torch.randn(27, ci, cout, device=dev) * 0.05
            cases = [("fwd", lambda: Fn.gather_gemm(xin, w, nbr, cout))]
            if stride == 1:
                gy = torch.randn(nbr.shape[0], cout, device=dev)
                cases.append(("dgrad", lambda: Fn.gather_gemm(gy, w, nbr, ci, w_transposed=True, flip_k=True)))
            for cname, fn in cases:
                out, tt = {}, {}
                for off in (1, 0, 1, 0):
                    lib().mink_conv_set_stagger(off << 30)
                    Fn._PLAN_CACHE.clear()
                    tt[off] = timeit(fn, reps) * 1e3
                    out[off] = fn()
                lib().mink_conv_set_stagger(0)
                err = float((out[0] - out[1]).abs().max() / out[1].abs().max())
                tot[0] += tt[0]; tot[1] += tt[1]
                print(f"{name} {cname:5s} rows={nbr.shape[0]:6d} {ci}->{cout} fill={fill:.2f}: compact {tt[0]:7.1f} us, dense {tt[1]:7.1f} us, rel diff {err:.2e}")
    print(f"sum: compact {tot[0]:.1f} us, dense {tot[1]:.1f} us")
if which == "cablate":  # l1.conv2 / l4.conv2 forward on the compact and the dense kernel (the PMC passes of scripts/pmc_kbench.sh run this)
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    for ts, c in ((4, 64), (32, 512)):
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        xin = torch.randn(nbr.shape[0], c, device=dev)
        w = torch.randn(27, c, c, device=dev) * 0.05
        for bits, label in ((0, "compact kernel"), (1 << 30, "dense kernel")):
            lib().mink_conv_set_stagger(bits)
            Fn._PLAN_CACHE.clear()
            t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, c), reps) * 1e3
            print(f"ts={ts} rows={nbr.shape[0]} {c}->{c}: {label:18s} {t:7.1f} us")
        lib().mink_conv_set_stagger(0)
if which == "wsweep":
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[ts * 2]
        for name, ik, ok, ci, stride in ((f"l@{ts}.c1", keys[ts], keys[ts * 2], cin, 2), (f"l@{ts*2}.c2", keys[ts * 2], keys[ts * 2], cout, 1)):
            nbr, _ = m.kernel_table(ik, ok, 3, 1)
            xin = torch.randn(m.levels[ts if stride == 2 else ts * 2].n, ci, device=dev)
            gy = torch.randn(nbr.shape[0], cout, device=dev)
            lib().mink_conv_set_stagger(0)
            t0 = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, ci, cout)), reps) * 1e3
            out = []
            for gcode, G in ((1, 1), (2, 3), (3, 9)):
                for z in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
                    if z > -(-nbr.shape[0] // 128):
                        continue
                    lib().mink_conv_set_stagger((gcode << 12) | (z << 16))
                    Fn._PLAN_CACHE.clear()  # (the slab workspace is sized from the cached plan: a forced split needs its own)
                    t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, ci, cout)), reps) * 1e3
                    out.append((t, G, z))
            lib().mink_conv_set_stagger(0)
            Fn._PLAN_CACHE.clear()
            out.sort()
            print(f"{name} wgrad n_out={nbr.shape[0]} {ci}->{cout} planner={t0:.1f}us best: " + " ".join(f"G{G}z{z}:{t:.0f}" for t, G, z in out[:6]))
if which == "sparsity":
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    v = nbr >= 0
    n = v.shape[0] // 2 * 2
    print("rows", v.shape[0], "density", float(v.float().mean()))
    for grp in (2, 4, 8, 16, 32):
        nn = v.shape[0] // grp * grp
        anyv = v[:nn].view(-1, grp, 27).any(dim=1)
        print(f"group of {grp} consecutive rows: fraction of (group, offset) with any neighbour {float(anyv.float().mean()):.3f}")
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    for ts in (2, 4, 8, 16):
        for name, ik, ok in ((f"{ts}->{ts*2} (stride 2)", keys[ts], keys[ts * 2]), (f"{ts*2}->{ts*2}", keys[ts * 2], keys[ts * 2])):
            nb, _ = m.kernel_table(ik, ok, 3, 1)
            vv = nb >= 0
            line = f"level {name}: rows {vv.shape[0]} density {float(vv.float().mean()):.3f}"
            for grp in (32, 128):
                nn = vv.shape[0] // grp * grp
                if nn:
                    line += f" | any over {grp} rows {float(vv[:nn].view(-1, grp, 27).any(dim=1).float().mean()):.3f}"
            print(line)
    c = m.levels[1].coords if hasattr(m.levels[1], "coords") else None
    if c is not None:
        print("first coords", c[:12].tolist())
if which == "pad":
    from nerf_downstream_amd._lib import lib
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    gy = torch.randn(nbr.shape[0], 64, device=dev)
    w = torch.randn(27, 28, 64, device=dev) * 0.05
    xc = x.F.contiguous()
    xp = torch.zeros(xc.shape[0], 32, device=dev)
    xp[:, :28] = xc
    xp = xp[:, :28]
    for name, xin in (("ldx=28", xc), ("ldx=32", xp)):
        for st in (1024, 0, 2048):
            lib().mink_conv_set_stagger(st)
            t = timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64)), reps)
            print(f"{name} mode {st}: stem wgrad {t*1e3:8.1f} us")
        lib().mink_conv_set_stagger(0)
        t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, 64), reps)
        print(f"{name}: stem fwd {t*1e3:8.1f} us")
    ya = Fn.gather_gemm(xc, w, nbr, 64); yb = Fn.gather_gemm(xp, w, nbr, 64)
    print("fwd diff", float((ya - yb).abs().max()))
if which == "perm":  # data gradient of the strided convolutions: row-compacted kernel over class-permuted rows against gather_gemm2 (bit 31)
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[ts * 2]
        nbr, nbr_t = m.kernel_table(keys[ts], keys[ts * 2], 3, 1, transposed=True)
        perm = m.class_perm(keys[ts])
        pairs = int((nbr >= 0).sum())
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        gy = torch.randn(nbr.shape[0], cout, device=dev)
        fl = 2.0 * pairs * cin * cout
        outs = []
        for bit in (0, -(1 << 31), 0, -(1 << 31)):
            lib().mink_conv_set_stagger(bit)
            Fn._PLAN_CACHE.clear()
            t = timeit(lambda: Fn.gather_gemm(gy, w, nbr_t, cin, w_transposed=True, row_perm=perm), reps)
            outs.append(Fn.gather_gemm(gy, w, nbr_t, cin, w_transposed=True, row_perm=perm))
            ks = Fn._plan_ksplit(lib(), perm.numel(), 27, cout, cin, 1)
            print(f"l@{ts}.c1 dgrad rows={nbr_t.shape[0]:7d} {cout}->{cin} {'dense ' if bit else 'compact'} split={ks:2d} {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s")
        lib().mink_conv_set_stagger(0)
        Fn._PLAN_CACHE.clear()
        print("   max |compact - dense|", float((outs[0] - outs[1]).abs().max()), "max |dense|", float(outs[1].abs().max()),
              "bitwise repeatable:", bool(torch.equal(outs[0], outs[2])))
if which == "ctrace":  # round 6: where a mid-layer forward launch's time goes, workgroup by workgroup (mink_conv_trace)
    import ctypes
    import numpy as np
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (4, 8, 16, 32):
        c = chans[ts]
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        xin = torch.randn(nbr.shape[0], c, device=dev)
        w = torch.randn(27, c, c, device=dev) * 0.05
        fn = lambda: Fn.gather_gemm(xin, w, nbr, c)
        t_plain = timeit(fn, reps) * 1e3
        cap = 8192
        buf = torch.zeros(cap * 5, dtype=torch.int64, device=dev)
        lib().mink_conv_trace(ctypes.c_void_p(buf.data_ptr()), cap)
        t_traced = timeit(fn, reps) * 1e3
        buf.zero_(); torch.cuda.synchronize()
        fn(); torch.cuda.synchronize()
        lib().mink_conv_trace(None, 0)
        t = buf.cpu().numpy().reshape(cap, 5)
        t = t[(t[:, 3] != 0) & (t[:, 0] != 0)]
        t0, t1, t2, t3 = (t[:, i].astype(np.float64) for i in range(4))
        hw, xcc = t[:, 4] & 0xFFFFFFFF, (t[:, 4] >> 32) & 0xF
        cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
        for cid in np.unique(cu):  # (the clock counters are not aligned across the chip: every CU from its own first start)
            sel = cu == cid
            base = t0[sel].min()
            for a in (t0, t1, t2, t3):
                a[sel] -= base
        span = t3.max() - t0.min()
        pro, loop, epi = t1 - t0, t2 - t1, t3 - t2
        ncu = len(np.unique(cu))
        q = lambda a: " / ".join(f"{np.percentile(a, p) / span * 100:5.1f}" for p in (10, 50, 90))
        # per CU: the share of the launch's span during which at least one / on average how many of its workgroups are inside their item loop
        any_loop, avg_loop, first_start, last_end = [], [], [], []
        for cid in np.unique(cu):
            sel = cu == cid
            ev = sorted([(a, 1) for a in t1[sel]] + [(b, -1) for b in t2[sel]])
            depth, last, busy, area = 0, t0.min(), 0.0, 0.0
            for tt, d in ev:
                if depth > 0:
                    busy += tt - last
                area += depth * (tt - last)
                depth, last = depth + d, tt
            any_loop.append(busy / span), avg_loop.append(area / span)
            first_start.append((t1[sel].min() - t0.min()) / span), last_end.append((t3.max() - t2[sel].max()) / span)
        print(f"l@{ts}.c2 fwd rows={nbr.shape[0]} {c}->{c}: {len(t)} workgroups on {ncu} CUs ({len(t) / ncu:.2f} per CU), conv + reduce {t_plain:.1f} us "
              f"({t_traced:.1f} traced); clock span of the conv launch {span:.0f} ticks")
        print(f"    per workgroup, % of the launch's span (10th / median / 90th percentile): prologue {q(pro)}   item loop {q(loop)}   epilogue + store drain {q(epi)}")
        print(f"    per CU: some workgroup inside its item loop {np.mean(any_loop) * 100:.1f} % of the span, workgroups inside their loops on average "
              f"{np.mean(avg_loop):.2f} (of 4 slots); first loop starts {np.mean(first_start) * 100:.1f} % in, last loop ends {np.mean(last_end) * 100:.1f} % before the end")
        print(f"    sum of all (prologue, loop, epilogue) / (4 slots x CUs x span): {pro.sum() / (4 * ncu * span):.3f} {loop.sum() / (4 * ncu * span):.3f} {epi.sum() / (4 * ncu * span):.3f}"
              f" -- the rest of the slot time has no workgroup in it; span {span:.0f} ticks = {span / 2.4e3:.1f} us at 2.4 GHz; median workgroup {np.median(t3 - t0) / 2.4e3:.1f} us")
        # how the launch fills: resident workgroups per CU at 5 % steps of the span
        grid_t = np.linspace(0, span, 21)
        res = [(np.sum((t0 <= g) & (t3 > g)) / ncu) for g in grid_t]
        print("    resident workgroups per CU at 0, 5, ... 100 % of the span: " + " ".join(f"{r:.1f}" for r in res))
        starts = np.sort(t0)
        print(f"    workgroup starts: the first {min(1024, len(starts))} within {starts[min(1024, len(starts)) - 1] / 2.4e3:.1f} us; the last start at {starts[-1] / 2.4e3:.1f} us")
if which == "sk":  # round 6: stream-K launches of the stride-1 mid-layer kernel (mink_conv_set_pipeline bit 3) against the (row tile, slice) grid
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    tot = {0: 0.0, 8: 0.0}
    for ts in (4, 8, 16, 32):
        c = chans[ts]
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        w = torch.randn(27, c, c, device=dev) * 0.05
        x2 = torch.randn(nbr.shape[0], c, device=dev)
        gy = torch.randn(nbr.shape[0], c, device=dev)
        ref = None
        for name, fn in ((f"l@{ts}.c2 fwd   rows={nbr.shape[0]:6d} {c}->{c}", lambda: Fn.gather_gemm(x2, w, nbr, c)),
                         (f"l@{ts}.c2 dgrad rows={nbr.shape[0]:6d} {c}->{c}", lambda: Fn.gather_gemm(gy, w, nbr, c, w_transposed=True, flip_k=True))):
            tt, out, ks = {}, {}, {}
            for mode in (0, 8, 0, 8):
                lib().mink_conv_set_pipeline(mode)
                Fn._PLAN_CACHE.clear()
                t = timeit(fn, reps) * 1e3
                tt[mode] = min(tt.get(mode, 1e9), t)
                o = fn()
                if mode in out:
                    assert torch.equal(o, out[mode]), "not repeatable"
                out[mode] = o
                ks[mode] = Fn._plan_ksplit(lib(), nbr.shape[0], 27, c, c, 0)
            lib().mink_conv_set_pipeline(0)
            Fn._PLAN_CACHE.clear()
            err = float((out[0] - out[8]).abs().max() / out[0].abs().max())
            tot[0] += tt[0]; tot[8] += tt[8]
            print(f"{name}: grid {tt[0]:7.1f} us ({ks[0]} slabs), stream-K {tt[8]:7.1f} us ({ks[8]} slabs) ({(tt[8] / tt[0] - 1) * 100:+5.1f} %), max rel diff {err:.1e}")
    print(f"sum: grid {tot[0]:.1f} us, stream-K {tot[8]:.1f} us")
if which == "p3":  # round 6: the three-stage form of compact_gemm_kernel (mink_conv_set_pipeline) against the two-stage one, bit for bit
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    tot = {0: 0.0, 3: 0.0, 4: 0.0}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[ts * 2]
        nbr1, nbr1_t = m.kernel_table(keys[ts], keys[ts * 2], 3, 1, transposed=True)
        nbr2, _ = m.kernel_table(keys[ts * 2], keys[ts * 2], 3, 1)
        perm = m.class_perm(keys[ts])
        w1 = torch.randn(27, cin, cout, device=dev) * 0.05
        w2 = torch.randn(27, cout, cout, device=dev) * 0.05
        x2 = torch.randn(nbr2.shape[0], cout, device=dev)
        gy = torch.randn(nbr2.shape[0], cout, device=dev)
        cases = [
            (f"l@{ts*2}.c2 fwd        rows={nbr2.shape[0]:6d} {cout}->{cout}", lambda: Fn.gather_gemm(x2, w2, nbr2, cout)),
            (f"l@{ts*2}.c2 dgrad      rows={nbr2.shape[0]:6d} {cout}->{cout}", lambda: Fn.gather_gemm(gy, w2, nbr2, cout, w_transposed=True, flip_k=True)),
            (f"l@{ts}.c1 dgrad(perm) rows={nbr1_t.shape[0]:6d} {cout}->{cin}", lambda: Fn.gather_gemm(gy, w1, nbr1_t, cin, w_transposed=True, row_perm=perm)),
        ]
        for name, fn in cases:
            tt, out = {}, {}
            for mode in (0, 3, 4, 0, 3, 4):  # (4: the two-stage form held to three workgroups per CU -- occupancy alone)
                lib().mink_conv_set_pipeline(mode)
                t = timeit(fn, reps) * 1e3
                tt[mode] = min(tt.get(mode, 1e9), t)
                out[mode] = fn()
            lib().mink_conv_set_pipeline(0)
            tot[0] += tt[0]; tot[3] += tt[3]; tot[4] += tt[4]
            print(f"{name}: two-stage {tt[0]:7.1f} us, three-stage {tt[3]:7.1f} us ({(tt[3] / tt[0] - 1) * 100:+5.1f} %), two-stage at 3 WG/CU {tt[4]:7.1f} us, bitwise equal: {bool(torch.equal(out[0], out[3]))}")
    print(f"sum: two-stage {tot[0]:.1f} us, three-stage {tot[3]:.1f} us, two-stage at three workgroups per CU {tot[4]:.1f} us")
if which in ("stem", "all"):
    bench_layer("stem", x.F.contiguous(), k1, k1, 3, 28, 64, 1)
if which in ("l1", "l4", "all"):
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (2, 4, 8, 16):
        if which == "l1" and ts > 2: break
        if which == "l4" and ts < 16: continue
        cin, cout = chans[ts], chans[ts * 2]
        xin = torch.randn(m.levels[ts].n, cin, device=dev)
        bench_layer(f"l@{ts}.c1", xin, keys[ts], keys[ts * 2], 3, cin, cout, 2)
        xin2 = torch.randn(m.levels[ts * 2].n, cout, device=dev)
        bench_layer(f"l@{ts*2}.c2", xin2, keys[ts * 2], keys[ts * 2], 3, cout, cout, 1)
if which == "wmid":  # mid-layer weight gradient (tiled kernel): full, without the MFMA loop (128), without the x gathers (64), without both
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (4, 8, 32):
        c = chans[ts]
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        xin = torch.randn(nbr.shape[0], c, device=dev)
        gy = torch.randn(nbr.shape[0], c, device=dev)
        res = []
        for ab in (0, 128, 64, 192):
            lib().mink_conv_set_stagger(ab)
            res.append((ab, timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, c, c)), reps) * 1e3))
        lib().mink_conv_set_stagger(0)
        print(f"l@{ts}.c2 wgrad rows={nbr.shape[0]} {c}->{c}: " + "  ".join(f"ablate {a}: {t:.1f} us" for a, t in res))
if which == "cab":  # compact kernel: full / every gather reads row 0 (16) / no matrix work (32) / both (48): what an item's time is made of
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (4, 8, 16, 32):
        c = chans[ts]
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        xin = torch.randn(nbr.shape[0], c, device=dev)
        w = torch.randn(27, c, c, device=dev) * 0.05
        res = []
        for ab in (0, 16, 32, 48, 0, 32 + 64, 32 + 128, 32 + 64 + 128, 8, 8 + 16, 8 + 16 + 32, 4):
            lib().mink_conv_set_stagger(ab)
            res.append(timeit(lambda: Fn.gather_gemm(xin, w, nbr, c), reps) * 1e3)
        lib().mink_conv_set_stagger(0)
        print(f"l@{ts}.c2 fwd rows={nbr.shape[0]} {c}->{c}: full {res[0]:.1f} / {res[4]:.1f} us, gathers from row 0 {res[1]:.1f}, no MFMA {res[2]:.1f}, neither {res[3]:.1f}; no MFMA + no scatter {res[5]:.1f}, + one block of LDS operands {res[6]:.1f}, + both {res[7]:.1f}; one weight block {res[8]:.1f}, + gathers from row 0 {res[9]:.1f}, + no MFMA {res[10]:.1f}; NO ITEMS (prologue + epilogue + split-K reduce) {res[11]:.1f}")

if which == "cabp":  # the class-permuted data gradient of the stride-2 convolutions: full / no matrix work / no items at all
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (2, 4, 8, 16):
        cin, cout = chans[ts], chans[2 * ts]
        nbr, nbr_t = m.kernel_table(keys[ts], keys[2 * ts], 3, 1, transposed=True)
        perm = m.class_perm(keys[ts])
        gy = torch.randn(nbr.shape[0], cout, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        res = []
        for ab in (0, 32, 32 + 64, 16, 8 + 16, 4, 0):
            lib().mink_conv_set_stagger(ab)
            res.append(timeit(lambda: Fn.gather_gemm(gy, w, nbr_t, cin, w_transposed=True, row_perm=perm), reps) * 1e3)
        lib().mink_conv_set_stagger(0)
        pairs = int((nbr >= 0).sum())
        print(f"l@{2 * ts}.c1 dgrad rows={nbr_t.shape[0]} (perm {perm.numel()}) {cout}->{cin}, {pairs} pairs = {2e-9 * pairs * cin * cout:.2f} GFLOP: full {res[0]:.1f} / {res[6]:.1f} us, "
              f"no MFMA {res[1]:.1f}, + no scatter {res[2]:.1f}; gathers from row 0 {res[3]:.1f}, + one weight block {res[4]:.1f}; NO ITEMS {res[5]:.1f}")

if which == "stemc":  # zero skipping in the stem, MEASURED (round-4 review, item 6): the row-compacted kernel on the stem's own shape --
    # 32-channel padded rows, every live offset of a 64-row tile in rounds of nine (the class-permuted form with an identity
    # permutation: un-split, direct epilogue) -- against the dense flattened-K kernel that runs in the step
    from nerf_downstream_amd._lib import lib
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    xin = x.F.contiguous()
    w = torch.randn(27, 28, 64, device=dev) * 0.05
    x32 = torch.nn.functional.pad(xin, (0, 4)).contiguous()
    w32 = torch.nn.functional.pad(w, (0, 0, 0, 4)).contiguous()
    perm = torch.arange(nbr.shape[0], device=dev, dtype=torch.int32)
    pairs = int((nbr >= 0).sum())
    fl = 2.0 * pairs * 28 * 64
    t = timeit(lambda: Fn.gather_gemm(xin, w, nbr, 64), reps)
    y0 = Fn.gather_gemm(xin, w, nbr, 64)
    print(f"stem fwd dense (flattened K, in the step): {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s useful; table fill {pairs / nbr.numel():.3f}")
    lib().mink_conv_set_stagger(256)
    Fn._PLAN_CACHE.clear()
    Fn._FORCE_KSPLIT = 1
    try:
        t = timeit(lambda: Fn.gather_gemm(x32, w32, nbr, 64, row_perm=perm), reps)
        y1 = Fn.gather_gemm(x32, w32, nbr, 64, row_perm=perm)
    finally:
        lib().mink_conv_set_stagger(0)
        Fn._FORCE_KSPLIT = 0
        Fn._PLAN_CACHE.clear()
    print(f"stem fwd row-compacted per offset (32-channel rows, 16-row blocks, C tile in LDS): {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s useful; "
          f"max |difference| {float((y1 - y0).abs().max()):.2e} of {float(y0.abs().max()):.2e}")


if which == "wxcdmid":  # tiled weight gradient of layer 1 / 2: the workgroups of a row split on one XCD (default) against plain launch order (bit 29)
    from nerf_downstream_amd._lib import lib
    keys = {1: k1}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = m.stride(keys[ts // 2], 2)
    chans = {2: 64, 4: 64, 8: 128, 16: 256, 32: 512}
    for ts in (4, 8):
        c = chans[ts]
        nbr, _ = m.kernel_table(keys[ts], keys[ts], 3, 1)
        xin = torch.randn(nbr.shape[0], c, device=dev)
        gy = torch.randn(nbr.shape[0], c, device=dev)
        res, outs = [], []
        for st in (0, 1 << 29, 0, 1 << 29):
            lib().mink_conv_set_stagger(st)
            res.append(timeit(lambda: Fn.conv_wgrad(xin, gy, nbr, (27, c, c)), reps) * 1e3)
            outs.append(Fn.conv_wgrad(xin, gy, nbr, (27, c, c)))
        lib().mink_conv_set_stagger(0)
        print(f"l@{ts}.c2 wgrad rows={nbr.shape[0]} {c}->{c}: splits on one XCD {res[0]:.1f} / {res[2]:.1f} us, plain order {res[1]:.1f} / {res[3]:.1f} us; bitwise equal: {bool(torch.equal(outs[0], outs[1]))}")
