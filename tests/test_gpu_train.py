"""-m gpu: short training runs on the HIP backend vs the CPU oracle with identical init, data
order and recipe: loss trajectory and top-1 on the fixed synthetic split must agree
(north_star: "top-1 on a fixed synthetic split matching reference +-0.1%"), and 2-rank data
parallelism on the card (gloo transport, both ranks on cuda:0) must match the rank average."""
import copy
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from nerf_downstream_amd import gin_lite as gin

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "nerf_downstream_amd", "co3d_3d", "configs")


def _run(tmp, ME, steps):
    from nerf_downstream_amd.co3d_3d.train import train

    gin.clear_config()
    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin", f"{CFG}/synthetic_cls.gin"],
        ["train.gpus=1", f"train.max_steps={steps}", f"train.val_every_n_steps={steps}", "train.log_every_n_steps=1",
         "SparseVoxelDataset.grid=32", "SparseVoxelDataset.num_samples=64", "SparseVoxelDataset.num_classes=4",
         "get_model.out_channel=4", "train.batch_size=8", "train.val_batch_size=8", "train.lr=0.003",
         "train.train_num_workers=0", "train.val_num_workers=0"],
    )
    res = train(save_path=str(tmp), resume_training=False, run_name="r", run_name_postfix=None, ME=ME, seed=11)
    gin.clear_config()
    losses = [h["train/loss"] for h in res["history"] if "train/loss" in h]
    val = [h for h in res["history"] if "val/acc1" in h][-1]
    return np.array(losses), val


def test_training_matches_oracle(tmp_path, oracle_maps):
    from oracle import me_cpu as OME

    steps = 16
    lg, vg = _run(tmp_path / "hip", None, steps)
    lc, vc = _run(tmp_path / "cpu", OME, steps)
    assert len(lg) == len(lc) == steps
    # same trajectory from the same init; fp32 summation-order differences amplify slowly with SGD steps
    assert np.allclose(lg[:4], lc[:4], atol=5e-3), (lg[:4], lc[:4])
    assert np.allclose(lg, lc, atol=8e-2), np.abs(lg - lc).max()
    assert lg[-4:].mean() < lg[:4].mean()  # it learns
    assert abs(vg["val/acc1"] - vc["val/acc1"]) <= 100.0 / 16 + 1e-6  # 16 validation samples: at most one flips
    assert abs(vg["val/loss"] - vc["val/loss"]) < 8e-2


def _schedule(model, multi, lazy_fork=False):
    """multi=True: every overlap the training loops use (prepare-ahead on its stream, weight
    gradients on the side stream, shortcut branch on its own stream) with each auxiliary stream
    DELAYED by ~1 ms per use (Fn._SKEW), so a missing cross-stream dependency changes the result
    instead of passing by luck.  multi=False: everything on one stream."""
    from nerf_downstream_amd.minkowski import functional as Fn

    Fn._SKEW = 2_000_000 if multi else 0
    Fn.set_wgrad_overlap(multi)
    Fn.set_branch_fork(True)  # (a data-parallel reducer switches both off by default; the tests force them)
    model.prepare_ahead = multi and not lazy_fork
    for m in model.modules():
        if hasattr(m, "_fork"):
            m._fork = multi
            m._fork_unprepared = lazy_fork  # maps built on demand from whichever stream asks first


def _train_steps(model, batches, labels, steps, reducer=None):
    """The step loop of bench.py / train.py (two-phase prepare-ahead, fused SGD)."""
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
    tf = model.process_input(batches[0])
    for i in range(steps):
        nxt = model.process_input(batches[(i + 1) % len(batches)], defer=True)
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        F.cross_entropy(model(tf), labels[i % len(batches)]).backward()
        if os.environ.get("MINK_TEST_TRACE"):
            torch.cuda.synchronize()
            print(f"[trace] step {i} backward done (reducer={reducer is not None})", flush=True)
        tf = model.finish_input(nxt)
        if reducer is not None:
            reducer.finish()
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()


@pytest.mark.parametrize("grid,sink,math", [(32, False, "fp32"), (64, False, "fp32"), (64, True, "fp32"), (64, True, "bf16s")])
def test_multi_stream_schedule_is_bitwise_single_stream(grid, sink, math):
    """Five training steps with every stream overlap on and the auxiliary streams skewed ==
    the same five steps on a single stream, bit for bit (all kernels are deterministic).  32^3 scenes take the
    module-by-module path, 64^3 scenes (~63 k voxels per batch) the native trunk -- asserted; `sink`: the flat gradient
    buffer of a one-rank reducer as the gradient sink of the backward kernels, which is how bench.py and train.py run;
    "bf16s": bf16 matrix math with bf16 storage of the full-resolution stage, whose bf16 copy of the input rows is made
    AHEAD on the (skewed) prepare stream."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import batch_scenes, trunk_node

    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.minkowski import functional as Fn
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    dev = torch.device("cuda", 0)
    batches, labels = [], []
    for j in range(2):
        coords, feats = batch_scenes([70 + 3 * j, 71 + 3 * j, 72 + 3 * j], grid=grid, cin=28)
        batches.append({"coordinates": coords.to(dev), "features": feats.to(dev)})
        labels.append(torch.tensor([j, 1 + j, 2 + j], device=dev))
    out = {}
    from nerf_downstream_amd import minkowski as ME

    old_math, old_storage = ME.set_conv_math("bf16" if math == "bf16s" else math), ME.set_conv_storage("bf16" if math == "bf16s" else "fp32")
    try:
        # "multi-side": the data-parallel schedule of the native trunk -- the shortcut branch on the weight-gradient stream
        # "multi-gated": the map builds held back to a point of the compute stream's latest pass (MINK_PREPARE_GATE=f0)
        for mode, (multi, lazy_fork) in {"single": (False, False), "multi": (True, False), "multi-lazy": (True, True),
                                          "multi-side": (True, False), "multi-gated": (True, False)}.items():
            torch.manual_seed(5)
            m = get_model("ResNet14", 28, 5).to(dev)
            reducer = BucketedGradAllReduce(m) if sink else None
            _schedule(m, multi, lazy_fork)
            Fn.set_trunk_branch_on_side(mode == "multi-side")
            Fn._PREPARE_GATE = (0, 0) if mode == "multi-gated" else None
            Fn._PREPARE_GATE_EVENT.clear()
            if mode == "multi-side":
                Fn.set_branch_fork(False)  # (as a data-parallel reducer leaves it)
                assert Fn.trunk_branch_mode() == ("side" if math == "fp32" else None)  # (bf16 math: no branch at all under data parallelism)
            node = trunk_node(m(m.process_input(batches[0])))
            native = node is not None
            assert native == (grid >= 64), (grid, native)
            if math == "bf16s":
                assert node.saved[0][7] is True  # bf16 storage taken
            out[mode] = _train_steps(m, batches, labels, 5, reducer)
            if mode == "multi-gated":
                assert bool(Fn._PREPARE_GATE_EVENT) == native  # (the gate point is a stage of the native trunk)
            Fn.set_grad_sink(None)
    finally:
        Fn._SKEW = 0
        Fn._PREPARE_GATE = None
        Fn._PREPARE_GATE_EVENT.clear()
        Fn.set_wgrad_overlap(True)
        Fn.set_grad_sink(None)
        Fn.set_trunk_branch_on_side(False)
        Fn.set_branch_fork(True)
        ME.set_conv_math(old_math), ME.set_conv_storage(old_storage)
    assert torch.isfinite(out["multi"]).all()
    assert torch.equal(out["single"], out["multi"])
    assert torch.equal(out["single"], out["multi-lazy"])
    assert torch.equal(out["single"], out["multi-side"])
    assert torch.equal(out["single"], out["multi-gated"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out):
    import faulthandler

    faulthandler.enable()  # a GPU fault aborts the process: show where the host was
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import batch_scenes

    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    m = get_model("ResNet14", 28, 5).to(dev)
    red = BucketedGradAllReduce(m, bucket_bytes=8 << 20)
    coords, feats = batch_scenes([50 + 2 * rank, 51 + 2 * rank], grid=24, cin=28)
    labels = torch.tensor([rank, 3 - rank], device=dev)
    red.zero_grad()
    F.cross_entropy(m(m.process_input({"coordinates": coords.to(dev), "features": feats.to(dev)})), labels).backward()
    # gradient-sink mode defers complete buckets to the flush points of the backward pass: by its end everything but
    # the stem's own little bucket (complete only with the last kernel of backward) must be on its way
    launched = all(red._launched[:-1]) and red.defer and len(red.buckets) >= 3
    tail_bytes = 4 * (red.buckets[-1][1] - red.buckets[-1][0])
    red.finish()
    torch.cuda.synchronize()
    torch.save({"g": red.gradients().cpu(), "launched": launched, "tail_bytes": tail_bytes}, f"{out}/r{rank}.pt")
    # imbalanced batches: rank 1's is large enough for the native trunk (which reports a stage's gradients in registration
    # order after the stage), rank 0's takes the module path (autograd order): both must issue the same sequence of
    # collectives (parallel.BucketedGradAllReduce._drain) and end with the same averaged gradients
    from helpers import trunk_node

    c2, f2 = batch_scenes([60, 61] if rank == 0 else list(range(70, 82)), grid=24 if rank == 0 else 48, cin=28)
    l2 = (torch.arange(c2[:, 0].max().int().item() + 1) % 5).to(dev)
    del red.launch_log[:]
    red.zero_grad()
    o2 = m(m.process_input({"coordinates": c2.to(dev), "features": f2.to(dev)}))
    F.cross_entropy(o2, l2).backward()
    red.finish()
    torch.cuda.synchronize()
    torch.save({"g": red.gradients().cpu(), "log": list(red.launch_log), "trunk": trunk_node(o2) is not None, "n": len(red.buckets),
                "rows": int(c2.shape[0])}, f"{out}/imb{rank}.pt")
    # single-rank reference gradients of this rank's batch (fresh model, same seed)
    torch.manual_seed(3)
    m2 = get_model("ResNet14", 28, 5).to(dev)
    F.cross_entropy(m2(m2.process_input({"coordinates": coords.to(dev), "features": feats.to(dev)})), labels).backward()
    torch.save(torch.cat([p.grad.flatten() for p in list(m2.parameters())[::-1]]).cpu(), f"{out}/local{rank}.pt")
    # four data-parallel training steps: every stream overlap on + skewed auxiliary streams must
    # give the single-stream parameters bit for bit (the 2-rank sum is order independent)
    batches = [{"coordinates": coords.to(dev), "features": feats.to(dev)}]
    params = {}
    for multi in (False, True):
        torch.manual_seed(4)
        m3 = get_model("ResNet14", 28, 5).to(dev)
        red3 = BucketedGradAllReduce(m3, bucket_bytes=8 << 20)  # (switches the overlaps off by default ...)
        _schedule(m3, multi)                                     # (... the multi run forces them back on)
        params[multi] = _train_steps(m3, batches, [labels], 4, reducer=red3)
    torch.save({"equal": bool(torch.equal(params[False], params[True])), "finite": bool(torch.isfinite(params[True]).all())},
               f"{out}/sched{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.long
@pytest.mark.timeout(60)
def test_data_parallel_two_ranks_on_card(tmp_path):
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["launched"] and r1["launched"]  # buckets were reduced from inside the backward pass (deferred launches on)
    assert r0["tail_bytes"] < (1 << 20)  # the bucket nothing can overlap with is the stem alone
    assert torch.equal(r0["g"], r1["g"])
    ref = 0.5 * (torch.load(tmp_path / "local0.pt") + torch.load(tmp_path / "local1.pt"))
    assert torch.allclose(r0["g"], ref, atol=1e-5 * float(ref.abs().max()), rtol=1e-4)  # plumbing: the HIP path's own single-rank gradients
    # the arithmetic, against the ORACLE: mean over the two ranks of the CPU oracle's gradients of each rank's batch (what DDP
    # computes in the reference, co3d_3d/train.py:174-186), same weights (seed 3), same flat order (reverse registration)
    from helpers import batch_scenes
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    og = []
    for rank in (0, 1):
        torch.manual_seed(3)
        om = get_model("ResNet14", 28, 5, ME=OME)
        coords, feats = batch_scenes([50 + 2 * rank, 51 + 2 * rank], grid=24, cin=28)
        F.cross_entropy(om(om.process_input({"coordinates": coords, "features": feats})), torch.tensor([rank, 3 - rank])).backward()
        og.append(torch.cat([p.grad.flatten() for p in list(om.parameters())[::-1]]))
    oref = 0.5 * (og[0] + og[1])
    assert r0["g"].shape == oref.shape
    rel = float((r0["g"].double() - oref.double()).norm() / oref.double().norm())
    print(f"2-rank averaged gradient on the card vs the oracle's average: relative L2 {rel:.2e}")
    assert rel < 1e-3, rel
    for r in (0, 1):
        sched = torch.load(tmp_path / f"sched{r}.pt")
        assert sched["finite"] and sched["equal"], (r, sched)
    i0, i1 = torch.load(tmp_path / "imb0.pt"), torch.load(tmp_path / "imb1.pt")
    assert i1["trunk"] and not i0["trunk"], (i0["rows"], i1["rows"])  # the two ranks really took different paths
    assert i0["log"] == i1["log"] and [b for b, _, _ in i0["log"]] == list(range(i0["n"]))
    assert torch.equal(i0["g"], i1["g"]) and bool(torch.isfinite(i0["g"]).all())


def _syncbn_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_downstream_amd import minkowski as ME

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    n, C = 1500, 64
    x_all = torch.randn(2 * n + 77, C, generator=g) * 1.5 + 0.3
    w_all = torch.randn(2 * n + 77, C, generator=g)
    sl = slice(0, n) if rank == 0 else slice(n, 2 * n + 77)  # uneven shards
    x = x_all[sl].to(dev).requires_grad_(True)
    coords = torch.zeros(x.shape[0], 4)
    coords[:, 1] = torch.arange(x.shape[0])
    m = ME.TensorField(coordinates=coords.to(dev), features=x.detach()).coordinate_manager
    model = torch.nn.Sequential(ME.MinkowskiBatchNorm(C)).to(dev)
    with torch.no_grad():
        model[0].bn.weight.copy_(torch.linspace(0.5, 1.5, C)), model[0].bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
    model = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    assert isinstance(model[0], ME.MinkowskiSyncBatchNorm)
    y = model[0](ME.SparseTensor(x, ME.CoordinateMapKey(1), m), relu=True).F
    (y * w_all[sl].to(dev)).sum().backward()
    torch.cuda.synchronize()
    torch.save({"y": y.detach().cpu(), "dx": x.grad.cpu(), "dw": model[0].bn.weight.grad.cpu(), "db": model[0].bn.bias.grad.cpu(),
                "rm": model[0].bn.running_mean.cpu(), "rv": model[0].bn.running_var.cpu()}, f"{out}/s{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.long
@pytest.mark.timeout(60)
def test_sync_batch_norm_two_ranks(tmp_path):
    """MinkowskiSyncBatchNorm over 2 ranks == BatchNorm1d over the concatenated rows."""
    mp.spawn(_syncbn_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    g = torch.Generator().manual_seed(0)
    n, C = 1500, 64
    x = (torch.randn(2 * n + 77, C, generator=g) * 1.5 + 0.3).requires_grad_(True)
    w = torch.randn(2 * n + 77, C, generator=g)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, C)), bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
    y = torch.relu(bn(x))
    (y * w).sum().backward()
    assert torch.allclose(torch.cat([r0["y"], r1["y"]]), y, atol=1e-5, rtol=1e-5)
    assert torch.allclose(torch.cat([r0["dx"], r1["dx"]]), x.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(r0["dw"] + r1["dw"], bn.weight.grad, atol=2e-3, rtol=1e-4)  # local sums add up
    assert torch.allclose(r0["db"] + r1["db"], bn.bias.grad, atol=2e-3, rtol=1e-4)
    assert torch.allclose(r0["rm"], bn.running_mean, atol=1e-6) and torch.allclose(r1["rv"], bn.running_var, atol=1e-5)


@pytest.mark.long
@pytest.mark.timeout(60)
def test_train_cli_with_dataloader_workers(tmp_path):
    """The CLI entry point end to end on the GPU: gin files, DataLoader worker processes running
    collate_mink (CPU-only, forked after HIP is initialised in the parent), checkpoints."""
    import subprocess

    cmd = [sys.executable, "-m", "nerf_downstream_amd.co3d_3d.train", "--ginc", f"{CFG}/co3d_cls.gin", "--ginc", f"{CFG}/resnet14.gin",
           "--ginc", f"{CFG}/synthetic_cls.gin", "--save_path", str(tmp_path), "--run_name", "cli", "--gpus", "1", "--seed", "5",
           "--ginb", "train.max_steps=6", "--ginb", "train.val_every_n_steps=6", "--ginb", "train.log_every_n_steps=2",
           "--ginb", "SparseVoxelDataset.grid=32", "--ginb", "SparseVoxelDataset.num_samples=32", "--ginb", "train.batch_size=4",
           "--ginb", "train.val_batch_size=4", "--ginb", "train.train_num_workers=2", "--ginb", "train.val_num_workers=2", "--ginb", "train.lr=0.01"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "val/acc1" in r.stdout
    assert (tmp_path / "cli" / "last.ckpt").exists() and (tmp_path / "cli" / "metrics.csv").exists()


def _write_co3d_scenes(root, n_scenes=16):
    """A tiny PeRFception-CO3D tree in the reference's on-disk format (scripts/preprocess.py:30-57)."""
    rng = np.random.default_rng(3)
    (root / "filelist").mkdir(parents=True)
    lines = []
    for j in range(n_scenes):
        cls = ("cup", "apple")[j % 2]
        g = np.arange(32)
        x, y, z = np.meshgrid(g, g, g, indexing="ij")
        r = np.sqrt(((x - 15.5) / (9 + 3 * (j % 2))) ** 2 + ((y - 15.5) / 8) ** 2 + ((z - 15.5) / 10) ** 2)
        occ = (np.abs(r - 1.0) < 0.15) & (rng.random(r.shape) > 0.1)
        xyz = np.stack(np.nonzero(occ), 1)
        links = (xyz[:, 0] * 128 * 128 + xyz[:, 1] * 128 + xyz[:, 2]).astype(np.int32)
        scene = root / "data" / f"plenoxel_co3d_s{j}"
        scene.mkdir(parents=True)
        np.savez(scene / "data.npz", links=links, density=rng.random((len(links), 1)).astype(np.float32),
                 sh=rng.integers(0, 256, (len(links), 27)).astype(np.uint8), sh_min=np.float32(-1.0 - 0.5 * (j % 2)),
                 sh_scale=np.float32(0.008), reso=[[128] * 3, [256] * 3])
        lines.append(f"{cls} s{j}")
    for phase in ("train", "test"):
        (root / "filelist" / f"{phase}.txt").write_text("\n".join(lines) + "\n")


@pytest.mark.long
@pytest.mark.timeout(120)
def test_train_on_co3d_format_compact_equals_decoded(tmp_path, monkeypatch):
    """train() on a tiny tree in the reference's on-disk format, DataLoader workers included: the
    compact path (GPU-side decode) and the ordinary path (CPU decode) give the same loss history -- and so do the three ways a
    batch reaches the device (round 6): pageable copies on the compute stream, pinned staging + a copy stream, and the direct
    reader that fills the pinned buffers from the scene files without a DataLoader."""
    from nerf_downstream_amd.co3d_3d.train import train

    _write_co3d_scenes(tmp_path)
    monkeypatch.chdir(tmp_path)
    hist = {}
    # (compact form, MINK_STAGING, MINK_DIRECT_LOADER): the reference-style loop (pageable copies on the compute stream), the DataLoader
    # through pinned staging + copy stream, and the direct reader of the scene files into the pinned buffers (data/staging.py)
    for compact, staging, direct in ((False, "0", "0"), (False, "1", "0"), (True, "1", "0"), (True, "1", "1"), (True, "0", "0")):
        monkeypatch.setenv("MINK_STAGING", staging), monkeypatch.setenv("MINK_DIRECT_LOADER", direct)
        gin.clear_config()
        gin.parse_config_files_and_bindings(
            [f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin"],
            ["train.gpus=1", "train.max_steps=6", "train.val_every_n_steps=6", "train.log_every_n_steps=1", "train.loggers=['csv']",
             f"Co3DDatasetBase.data_root='{tmp_path}/data'", "Co3DDatasetBase.features=['density','sh']",
             f"Co3DDatasetBase.compact={compact}", "get_model.in_channel=28", "train.batch_size=4", "train.val_batch_size=4",
             "train.lr=0.003", "train.train_num_workers=2", "train.val_num_workers=0"],
        )
        res = train(save_path=str(tmp_path / f"run{int(compact)}{staging}{direct}"), resume_training=False, run_name="r", run_name_postfix=None, seed=9)
        gin.clear_config()
        hist[(compact, staging, direct)] = [h["train/loss"] for h in res["history"] if "train/loss" in h]
        assert [h for h in res["history"] if "val/acc1" in h]
    first = hist[(False, "0", "0")]
    assert len(first) == 6 and all(h == first for h in hist.values()), hist


def test_sgd_inside_the_backward_call_is_the_step_behind_it():
    """parallel.FlatSGD(in_backward=True) (one rank, round 6): the update of every bucket of parameters the native trunk completes is
    launched from inside the backward call on the weight-gradient stream, the rest by step() -- the same kernel on slices of the same
    flat buffers.  Five training steps at 64^3 (the native trunk -- asserted -- with the weight-gradient and shortcut streams on and
    the next batch's maps prepared ahead, as bench.py runs) must leave parameters AND momentum bit for bit where the one-kernel
    step behind the backward pass leaves them; several buckets, so that some are updated inside the call and the stem's by step();
    the gradient buffer is cleared either way."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import batch_scenes, trunk_node

    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.minkowski import functional as Fn
    from nerf_downstream_amd.parallel import BucketedGradAllReduce, FlatSGD

    dev = torch.device("cuda", 0)
    batches, labels = [], []
    for j in range(2):
        coords, feats = batch_scenes([90 + 3 * j, 91 + 3 * j, 92 + 3 * j], grid=64, cin=28)
        batches.append({"coordinates": coords.to(dev), "features": feats.to(dev)})
        labels.append(torch.tensor([j, 1 + j, 2 + j], device=dev))
    out = {}
    try:
        for mode in (False, True):
            torch.manual_seed(9)
            m = get_model("ResNet14", 28, 5).to(dev)
            red = BucketedGradAllReduce(m, bucket_bytes=8 << 20)
            assert len(red.buckets) >= 4
            opt = FlatSGD(red, lr=0.05, momentum=0.9, weight_decay=1e-3, in_backward=mode)
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
            inside = []
            if mode:
                real = opt._bucket_ready
                red.set_bucket_callback(lambda b, s_, e_, st: (inside.append(b), real(b, s_, e_, st)))
            tf = m.process_input(batches[0])
            for i in range(5):
                nxt = m.process_input(batches[(i + 1) % 2], defer=True)
                red.zero_grad()
                o = m(tf)
                assert trunk_node(o) is not None
                F.cross_entropy(o, labels[i % 2]).backward()
                tf = m.finish_input(nxt)
                red.finish()
                opt.step()
                sched.step()
            torch.cuda.synchronize()
            assert red.cleared and float(red.flat.abs().max()) == 0.0
            if mode:
                assert len(inside) == 5 * (len(red.buckets) - 1), (inside, len(red.buckets))  # every bucket but the stem's, every step
            out[mode] = (opt.flat_w.clone(), opt.flat_m.clone())
            Fn.set_grad_sink(None)
    finally:
        Fn.set_grad_sink(None)
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])


def test_flat_sgd_is_torch_sgd(tmp_path):
    """parallel.FlatSGD (one kernel over the flat parameter / gradient / momentum buffers, mink_sgd_step) against
    torch.optim.SGD(fused) on the same model, batches and cosine schedule: five training steps, parameters equal to rounding
    (the same products and sums, one rounding each; torch's kernel may contract), momentum buffers too; optimizer checkpoints
    are interchangeable both ways (registration-order param_groups); the gradient buffer is cleared by the step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import batch_scenes

    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.parallel import BucketedGradAllReduce, FlatSGD

    dev = torch.device("cuda", 0)
    coords, feats = batch_scenes([81, 82, 83], grid=32, cin=28)
    batch = {"coordinates": coords.to(dev), "features": feats.to(dev)}
    labels = torch.tensor([1, 4, 2], device=dev)

    def run(flat, steps, resume=None):
        torch.manual_seed(21)
        m = get_model("ResNet14", 28, 5).to(dev)
        red = BucketedGradAllReduce(m)
        opt = FlatSGD(red, lr=0.05, momentum=0.9, weight_decay=1e-3) if flat else \
            torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, fused=True)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
        if resume is not None:
            m.load_state_dict(resume["model"]), opt.load_state_dict(copy.deepcopy(resume["opt"])), sched.load_state_dict(resume["sched"])
        for _ in range(steps):
            red.zero_grad()
            F.cross_entropy(m(m.process_input(batch)), labels).backward()
            red.finish()
            opt.step()
            sched.step()
        torch.cuda.synchronize()
        if flat:
            assert red.cleared and float(red.flat.abs().max()) == 0.0  # cleared behind the update: zero_grad() skips its memset
            assert all(p.data_ptr() >= opt.flat_w.data_ptr() for p in m.parameters())
        w = torch.cat([p.detach().flatten() for p in m.parameters()])
        mom = torch.cat([opt.state[p]["momentum_buffer"].flatten() for p in m.parameters()])
        ck = {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": copy.deepcopy(opt.state_dict()),
              "sched": sched.state_dict()}  # (deep copy: load_state_dict does not clone, a resumed torch SGD would step the checkpoint itself)
        return w, mom, ck

    # (trajectories of this network amplify rounding differences -- tests/test_gpu_parity_full.py -- so steps are compared ONE at
    #  a time from a common state: the first step from the initial weights, the second and third from a checkpoint)
    def close(a, b, rel):
        return float((a - b).abs().max()) <= rel * float(b.abs().max()) + 1e-7

    w_t, m_t, ck_t = run(False, 1)
    w_f, m_f, ck_f = run(True, 1)
    assert close(w_f, w_t, 1e-6) and close(m_f, m_t, 1e-6), (float((w_f - w_t).abs().max()), float((m_f - m_t).abs().max()))
    # the next step (momentum now non-zero, learning rate moved by the schedule) from the OTHER optimizer's checkpoint
    w_t2, m_t2, _ = run(False, 1, resume=ck_t)
    w_f2, m_f2, _ = run(True, 1, resume=ck_t)   # torch's checkpoint into FlatSGD
    w_t3, m_t3, _ = run(False, 1, resume=ck_f)  # FlatSGD's checkpoint into torch
    assert close(w_f2, w_t2, 1e-6) and close(m_f2, m_t2, 1e-5)
    # (the two checkpoints differ by the first step's rounding, which one forward/backward of this network amplifies ~100x)
    assert close(w_t3, w_t2, 5e-3) and close(m_t3, m_t2, 5e-2), (float((w_t3 - w_t2).abs().max()), float((m_t3 - m_t2).abs().max()))
    # and it trains: five steps stay within the band in which two correct trainers of this network drift apart
    w_t5, _, _ = run(False, 5)
    w_f5, _, _ = run(True, 5)
    assert float((w_f5 - w_t5).abs().max()) < 0.2 and float((w_f5 - w_t).abs().max()) > 1e-3
    # a parameter that leaves the flat buffer is refused, not silently skipped
    torch.manual_seed(21)
    m = get_model("ResNet14", 28, 5).to(dev)
    red = BucketedGradAllReduce(m)
    opt = FlatSGD(red, lr=0.05, momentum=0.9)
    m.final.bias.data = m.final.bias.data.clone()
    with pytest.raises(RuntimeError, match="no longer lives in the flat buffers"):
        opt.step()


@pytest.mark.long
@pytest.mark.timeout(80)
def test_segmentation_and_augmented_training_runs(tmp_path):
    """train() end to end on the GPU for the two widened rows: (a) SegmentationTraining + Res16UNet on per-voxel
    labels (two-phase prepare replays the plan with the transposed tables and parity-class orders of the decoder),
    loss falls and mIoU rises; (b) classification with the co3d_aug3 augmentation recipe drawn by DataLoader
    workers and applied on the device."""
    from nerf_downstream_amd.co3d_3d.train import train

    gin.clear_config()
    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/synthetic_seg.gin"],
        ["train.gpus=1", "train.max_steps=40", "train.val_every_n_steps=40", "train.log_every_n_steps=5", "train.batch_size=4",
         "train.val_batch_size=4", "SparseVoxelSegDataset.grid=32", "SparseVoxelSegDataset.num_samples=16", "train.lr=0.05",
         "train.scheduler_name='PolyLR'"],
    )
    res = train(save_path=str(tmp_path / "seg"), resume_training=False, run_name="s", run_name_postfix=None)
    logged = [x for x in res["history"] if "train/loss" in x]
    assert logged[-1]["train/loss"] < 0.6 * logged[0]["train/loss"], [x["train/loss"] for x in logged]
    assert logged[-1]["train/mIoU"] > logged[0]["train/mIoU"]
    val = [x for x in res["history"] if "val/mIoU" in x][-1]
    assert np.isfinite(val["val/loss"]) and 0.0 < val["val/mIoU"] <= 100.0, val  # (40 steps: running BN statistics are still young)

    gin.clear_config()
    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin", f"{CFG}/synthetic_cls.gin", f"{CFG}/co3d_aug3.gin"],
        ["train.gpus=1", "train.max_steps=8", "train.val_every_n_steps=8", "train.log_every_n_steps=1", "train.batch_size=4",
         "train.val_batch_size=4", "SparseVoxelDataset.grid=32", "SparseVoxelDataset.num_samples=16", "SparseVoxelDataset.num_classes=4",
         "get_model.out_channel=4", "train.train_num_workers=2", "train.val_num_workers=0",
         "SparseVoxelDataset.train_transformations=['RandomRotation', 'RandomAffine', 'CoordinateDropout', 'RandomHorizontalFlip', "
         "'CoordinateUniformTranslation', 'CoordinateJitter', 'RandomScale', 'RandomFeatureJitter']"],
    )
    res = train(save_path=str(tmp_path / "aug"), resume_training=False, run_name="a", run_name_postfix=None)
    logged = [x for x in res["history"] if "train/loss" in x]
    assert res["global_step"] == 8 and len(logged) == 8 and all(np.isfinite(x["train/loss"]) for x in logged)
    gin.clear_config()


@pytest.mark.long
@pytest.mark.timeout(60)
def test_bench_two_ranks_self_launched(tmp_path):
    """`python bench.py --gpus 2` as the driver would type it (no launcher): bench.py starts its own two ranks under
    torch.distributed.run and rank 0's JSON line comes back.  Rehearsal transport: gloo, both ranks on this one card
    (BENCH_DEVICE=0) -- the RCCL transport itself needs a node with two GPUs."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_DEVICE="0", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2",
           "--grid", "32", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["ranks_seen"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["scaling"] == "weak" and res["steps"] == 3 and res["warmup"] == 2 and res["value"] > 0
    assert res["config"]["global_batch"] == 4 and "gloo" in res["config"]["collective"]
    assert np.isfinite(res["config"]["final_loss"])


@pytest.mark.long
@pytest.mark.timeout(100)
def test_one_rank_rccl_group_costs_little(tmp_path):
    """The whole data-parallel machinery (RCCL process group of ONE rank, gradient sink, collectives issued from inside the backward
    call on the weight-gradient stream, branch on that stream too) against the plain single-GPU step, both at the bench's default
    shape: within 15 % -- at bench.py's own choice of hardware queues AND at sixteen.  Guards the cliff round 4 found (5.7-6.0 ms
    against 3.7 at eight or more hardware queues) and round 5 explained (a FIFTH busy queue: the stream the collectives were issued
    from): with four busy queues by construction the queue count must not matter.  Measured overhead: +3.5 %."""
    import json
    import subprocess

    def run(force, queues=None):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GPU_MAX_HW_QUEUES", "MINK_DP_LAUNCH")}
        env.update(MASTER_PORT=str(_free_port()), MASTER_ADDR="127.0.0.1")
        if queues:
            env["GPU_MAX_HW_QUEUES"] = str(queues)
        if force:
            env["BENCH_FORCE_REDUCER"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
               "--no-kernel-timing"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=33)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    plain, dp, dp16 = run(False), run(True), run(True, queues=16)
    assert "RCCL" in dp["config"]["collective"] and "none" in plain["config"]["collective"]
    for d in (dp, dp16):
        assert abs(d["config"]["final_loss"] - plain["config"]["final_loss"]) <= 1e-3 * max(1.0, abs(plain["config"]["final_loss"]))
        assert d["ms_per_step"] <= 1.15 * plain["ms_per_step"], (d["ms_per_step"], plain["ms_per_step"])


def test_scannet_plenoxel_segmentation_on_gpu(tmp_path):
    """train.py on a tiny PeRFception-ScanNet tree (reference scannet.py:450-660 format) with the HIP backend: metric
    float coordinates are floored and the features sharing a voxel averaged by TensorField.sparse(), predictions are
    carried back to every input row by out.slice(field); the first-step loss equals the CPU oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_train_cpu import _write_scannet_tree

    from nerf_downstream_amd.co3d_3d.train import train
    from oracle import me_cpu as OME

    data_root, _, _, _ = _write_scannet_tree(tmp_path)
    losses = {}
    for tag, ME_ in (("hip", None), ("cpu", OME)):
        gin.clear_config()
        gin.parse_config_files_and_bindings(
            [f"{CFG}/scannet_plenoxel.gin", f"{CFG}/res16unet.gin"],
            ["train.gpus=1", "train.max_steps=6", "train.val_every_n_steps=6", "train.log_every_n_steps=1", "train.batch_size=2",
             "train.val_batch_size=1", "train.train_num_workers=0", "train.val_num_workers=0", "train.lr=0.01",
             f"PlenoxelScannetDataset.data_root='{data_root}'", "get_model.name='Res16UNet14A'"])
        try:
            res = train(save_path=str(tmp_path / tag), resume_training=False, run_name="s", run_name_postfix=None, ME=ME_, seed=3)
        finally:
            gin.clear_config()
        losses[tag] = [h["train/loss"] for h in res["history"] if "train/loss" in h]
        assert any("val/mIoU" in h for h in res["history"])
    assert len(losses["hip"]) == 6 and abs(losses["hip"][0] - losses["cpu"][0]) < 1e-3, (losses["hip"][:2], losses["cpu"][:2])
    assert np.allclose(losses["hip"][:3], losses["cpu"][:3], atol=2e-2)


@pytest.mark.gpu
def test_reserved_segment_serves_later_allocations():
    """memory.reserve(): one segment handed to torch's caching allocator up front; allocations of a training step are then splits of it
    (no growth of the reserved total), and a second call is a no-op."""
    from nerf_downstream_amd import memory

    dev = torch.device("cuda", 0)
    memory._RESERVED.clear()
    torch.cuda.empty_cache()
    got = memory.reserve(dev, gigabytes=1.0)
    assert got == 1 << 30 and memory.reserve(dev, gigabytes=8.0) == got
    before = torch.cuda.memory_reserved(dev)
    assert before >= got
    blocks = [torch.empty(s << 20, dtype=torch.uint8, device=dev) for s in (3, 40, 200, 17, 256)]
    assert torch.cuda.memory_reserved(dev) == before  # carved out of the segment: no new hipMalloc
    del blocks
    assert memory.reserve(torch.device("cpu")) == 0
    # the allocator's free blocks belong to a stream: what is allocated under another stream grows the pool, unless that stream
    # has a segment of its own (memory.reserve_on; one per device)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        b0 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    grown = torch.cuda.memory_reserved(dev)
    assert grown > before  # (not served by the first segment)
    del b0
    assert memory.reserve_on(side, gigabytes=0.5) == 1 << 29 and memory.reserve_on(torch.cuda.Stream(device=dev), gigabytes=4.0) == 1 << 29
    at = torch.cuda.memory_reserved(dev)
    with torch.cuda.stream(side):
        blocks = [torch.empty(s << 20, dtype=torch.uint8, device=dev) for s in (100, 200, 90)]
    assert torch.cuda.memory_reserved(dev) == at
    del blocks
    memory._RESERVED.clear()
    torch.cuda.empty_cache()
