"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce used for data parallelism
(one process per GPU; RCCL on the MI355X node, gloo here) averages the per-rank gradients,
keeps BatchNorm statistics per rank (reference default, train.py:83) and leaves every rank with
identical parameters after the optimizer step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import batch_scenes

    coords, feats = batch_scenes([100 + 2 * rank, 101 + 2 * rank], grid=24, cin=8)
    return {"coordinates": coords, "features": feats, "labels": torch.tensor([rank, 1 - rank])}


def _model():
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    torch.manual_seed(5)
    return get_model("ResNet14", 8, 3, ME=OME)


def _local_grads(rank):
    m = _model()
    b = _batch(rank)
    F.cross_entropy(m(m.process_input(b)), b["labels"]).backward()
    return m, torch.cat([p.grad.flatten() for p in m.parameters()])


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    torch.set_num_threads(2)
    m = _model()
    red = BucketedGradAllReduce(m, bucket_bytes=4 << 20)  # several buckets for the 14 M parameters
    assert len(red.buckets) > 3
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    b = _batch(rank)
    red.zero_grad()
    F.cross_entropy(m(m.process_input(b)), b["labels"]).backward()
    assert all(red._launched)  # every bucket's all-reduce was started from the backward hooks
    red.finish()
    g = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    opt.step()
    w = torch.cat([p.detach().flatten() for p in m.parameters()])
    rm = m.bn1.bn.running_mean.clone()
    torch.save({"g": g, "w": w, "rm": rm}, f"{out}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_bucketed_allreduce_world2(tmp_path, oracle_maps):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["w"], r1["w"])  # ranks stay in lock-step
    assert not torch.equal(r0["rm"], r1["rm"])  # BatchNorm statistics are per rank
    torch.set_num_threads(2)  # same reduction order as the workers
    g0, g1 = _local_grads(0)[1], _local_grads(1)[1]
    ref = 0.5 * (g0 + g1)
    assert torch.allclose(r0["g"], ref, atol=1e-4 * float(ref.abs().max()), rtol=1e-3)


def _imbalance_worker(rank, world, port, out):
    """Rank 0: two scenes, gradients reported the way the module path reports them (autograd order, collectives issued
    from the hooks); rank 1: twelve scenes, gradients reported the way the native trunk reports them (per residual stage,
    in REGISTRATION order, after the stage's backward; deferred launches, a flush per stage)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import batch_scenes

    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    torch.set_num_threads(2)
    m = _model()
    red = BucketedGradAllReduce(m, bucket_bytes=64 << 10)  # a bucket per weight tensor or two: boundaries fall inside the residual stages
    assert len(red.buckets) > 8, len(red.buckets)
    seeds = [200, 201] if rank == 0 else list(range(300, 312))  # (two scenes: batch norm needs two rows at the coarsest level)
    coords, feats = batch_scenes(seeds, grid=24, cin=8)
    labels = torch.arange(len(seeds)) % 3
    for step in range(2):
        red.zero_grad()
        loss = F.cross_entropy(m(m.process_input({"coordinates": coords, "features": feats})), labels)
        if rank == 0:
            loss.backward()
        else:
            for h in red._hooks:
                h.remove()
            red._hooks, red.defer = [], True
            loss.backward()
            assert not red.launch_log or step > 0
            stages = [m.final] + [blk for li in (4, 3, 2, 1) for blk in list(getattr(m, f"layer{li}"))[::-1]] + [m.bn1, m.conv1]
            for st in stages:
                for p_ in st.parameters():  # registration order inside the stage, as minkowski/trunk.py reports
                    red.ready(p_)
                red.flush()
        red.finish()
    g = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    torch.save({"g": g, "log": list(red.launch_log), "n": len(red.buckets)}, f"{out}/i{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(150)
def test_collectives_are_issued_in_one_order_under_imbalance(tmp_path, oracle_maps):
    """The deadlock hazard of data parallelism with deferred launches is a rank-dependent ISSUE ORDER of the bucket
    all-reduces (RCCL matches collectives by order).  Two ranks with very different batches (2 scenes vs 12) that report
    their gradients in different orders -- the module path's autograd order on one, the native trunk's per-stage
    registration order with static flush points on the other, as happens when only one rank's batch is large enough
    for the trunk -- must issue the same sequence of (bucket, start, end), twice in a row, and end with the same mean
    gradients.  (With launches in completion order this test hangs or mismatches: bucket k+1 completes before bucket k
    on the trunk-style rank wherever a boundary falls inside a stage.)"""
    mp.spawn(_imbalance_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "i0.pt"), torch.load(tmp_path / "i1.pt")
    assert r0["log"] == r1["log"] and len(r0["log"]) == 2 * r0["n"]
    assert [b for b, _, _ in r0["log"]] == 2 * list(range(r0["n"]))  # ascending bucket order, every step
    assert torch.equal(r0["g"], r1["g"]) and bool(torch.isfinite(r0["g"]).all())


def test_flat_buffer_layout_single_process():
    sys.path.insert(0, ROOT)
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    m = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 2))
    red = BucketedGradAllReduce(m, bucket_bytes=64)
    params = list(m.parameters())
    # every slice starts on a 256-byte boundary (the fused optimizer kernel needs aligned pointers for its 16-byte path)
    assert red.flat.numel() == sum(-(-p.numel() // 64) * 64 for p in params) == 4 * 64
    assert all((p.grad.data_ptr() - red.flat.data_ptr()) % 256 == 0 for p in params)
    # reverse registration order: the last layer's gradients (ready first in backward) come first
    assert params[-1].grad.data_ptr() == red.flat.data_ptr()
    m(torch.randn(3, 4)).sum().backward()
    red.finish()  # world 1: no-op
    assert torch.equal(params[0].grad.flatten(), red.flat[3 * 64 : 3 * 64 + params[0].numel()])
    assert torch.equal(red.gradients(), torch.cat([p.grad.flatten() for p in params[::-1]]))
    assert float(red.flat.abs().sum()) == float(red.gradients().abs().sum())  # the padding stays zero
    red.zero_grad()
    assert float(red.flat.abs().sum()) == 0.0


@pytest.mark.timeout(150)
def test_bench_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher environment starts its own ranks under torch.distributed.run (as
    a child process -- the parent never touches the GPU) and relays rank 0's JSON line.  --dry-run: rendezvous and one
    all-reduce only (this container has no GPU); the real step runs the same way in tests/test_gpu_train.py."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2, "rank_sum": 3.0}
    # asking for a rank count that the launcher environment does not provide is an error, not a silent N=1 run
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_length_balanced_sampler_matches_distributed_sampler_sets(tmp_path, monkeypatch):
    """Length-aware placement across ranks: every global step draws the same sample set as DistributedSampler would
    (same averaged gradient), but the per-rank voxel totals are balanced to within one scene."""
    import numpy as np
    from torch.utils.data.distributed import DistributedSampler

    sys.path.insert(0, ROOT)
    from nerf_downstream_amd.co3d_3d.src.data.data_module import LengthBalancedDistributedSampler as LB

    rng = np.random.default_rng(0)
    n, W, B = 203, 4, 6
    lengths = rng.integers(8_000, 120_000, n)  # CO3D scenes vary by more than 10x
    samplers = [LB(lengths, B, W, r, seed=7) for r in range(W)]
    for s in samplers:
        s.set_epoch(3)
    per_rank = [list(s) for s in samplers]
    assert all(len(p) == (n // (W * B)) * B == len(samplers[0]) for p in per_rank)
    flat = sorted(i for p in per_rank for i in p)
    assert len(set(flat)) == len(flat)  # disjoint across ranks
    worst_lb, worst_plain = 0.0, 0.0
    plain = []
    for r in range(W):
        ds = DistributedSampler(range(n), W, r, shuffle=True, seed=7, drop_last=True)
        ds.set_epoch(3)
        plain.append(list(ds))
    for t in range(n // (W * B)):
        got = [p[t * B : (t + 1) * B] for p in per_rank]
        ref = [p[t * B : (t + 1) * B] for p in plain]
        assert sorted(i for g in got for i in g) == sorted(i for g in ref for i in g)  # same global batch
        tot = np.array([lengths[g].sum() for g in got], dtype=np.float64)
        tot_ref = np.array([lengths[g].sum() for g in ref], dtype=np.float64)
        assert tot.max() - tot.min() <= lengths[[i for g in got for i in g]].max()
        worst_lb, worst_plain = max(worst_lb, tot.max() / tot.mean()), max(worst_plain, tot_ref.max() / tot_ref.mean())
    assert worst_lb < 1.05 < worst_plain  # slowest rank: < 5 % over the mean instead of tens of percent

    # the CO3D dataset reads its lengths from the npz headers
    from nerf_downstream_amd.co3d_3d.src.data.co3d import Co3DDataset

    (tmp_path / "filelist").mkdir()
    names = []
    for j, k in enumerate([300, 57, 1234]):
        d = tmp_path / "data" / f"plenoxel_co3d_s{j}"
        d.mkdir(parents=True)
        np.savez(d / "data.npz", links=np.arange(k, dtype=np.int32), density=np.zeros((k, 1), np.float32),
                 sh=np.zeros((k, 27), np.uint8), sh_min=np.float32(0), sh_scale=np.float32(1))
        names.append(f"cup s{j}")
    (tmp_path / "filelist" / "train.txt").write_text("\n".join(names) + "\n")
    monkeypatch.chdir(tmp_path)
    assert Co3DDataset(phase="train", data_root=str(tmp_path / "data")).sample_lengths().tolist() == [300, 57, 1234]
