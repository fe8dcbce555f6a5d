#!/usr/bin/env python3
"""bench.py -- voxels/s, forward+backward(+all-reduce+SGD) of Mink-ResNet14 on synthetic
PeRFception-CO3D-shaped plenoxel grids (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one full training iteration over one batch of `--batch` scenes per GPU
(~51.6 k voxels x 28 features each, resident in HBM before the timed region): coordinate maps
+ kernel maps rebuilt from the raw float coordinates (as ME does every iteration), forward,
cross-entropy, backward, bucketed gradient all-reduce (N>1) and the SGD(momentum) update.
Weak scaling: per-GPU batch is fixed (reference semantics: train.batch_size is per GPU).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle = restatement of ME's CPU algorithm, timed on
the host cores on a bounded sample; N=1 only).
"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if os.environ.get("BENCH_DEVICE") is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # Rehearsal on a one-GPU box: several ranks share ONE card.  Two processes time-slicing a GPU switch context per
    # hardware queue, and every cross-stream dependency forces such a switch: with the full multi-stream schedule a step
    # takes seconds instead of milliseconds (measured: 13.8 s; compute + prepare streams only: 0.55 s).  That says nothing
    # about one process per GPU -- but it makes the rehearsal useless, so it runs the conservative schedule.
    os.environ.setdefault("MINK_DP_MULTISTREAM", "0")
elif int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("BENCH_FORCE_REDUCER") == "1":
    # Data parallelism: one rank drives four busy streams -- compute, map preparation, weight gradients (+ shortcut branch; the bucket
    # collectives are issued from this stream inside the backward call) and the RCCL process group's own.  HIP multiplexes streams
    # over GPU_MAX_HW_QUEUES (default 4) hardware queues and two busy streams on one queue serialise; the device serves four queues
    # side by side and time-slices a fifth.  The HIP runtime reads the variable when it is loaded -- before `import torch`;
    # nerf_downstream_amd/hwqueues.py has the measurements and the rule.
    from nerf_downstream_amd.hwqueues import configure as _configure_hw_queues

    _configure_hw_queues(data_parallel=True)
    # (this pool's host driver only supports dmabuf IPC: without it RCCL fails in hipIpcGetMemHandle; already exported on the
    #  boxes -- kept here for an environment that lost it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16 MFMA, dense (never the 2:1-sparsity figure)
MFMA_PEAK_TFLOPS = MFMA_F32_PEAK_TFLOPS  # of the arithmetic the dominant kernel runs in (main() sets it from --math)
HBM_PEAK_GBS = 8000.0


def make_batches(n_batches, batch, rank, num_classes, grid, cin):
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import SparseVoxelDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink

    feats = ["density", "sh"] if cin == 28 else ["sh"]
    ds = SparseVoxelDataset(phase="train", num_samples=1 << 20, num_classes=num_classes, grid=grid, features=feats)
    out = []
    for b in range(n_batches):
        base = (rank * n_batches + b) * batch
        out.append(collate_mink([ds[base + i] for i in range(batch)]))
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model_name, cin, num_classes, grid, state_dict, seconds_budget=24.0):
    """Times the CPU oracle (oracle/me_cpu.py + oracle/mink_maps.c: sequential hash insert, OpenMP kernel-map
    search, per offset gather -> SGEMM -> scatter-add, torch BatchNorm1d) on a bounded sample: full fwd+bwd steps
    of 2-scene batches (BASELINE config #0), once with every host core and once with OMP_NUM_THREADS=12 (the
    reference's own job script, sbatch.sh:34), ~seconds_budget/2 each.  `value` is the all-cores figure."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import maps as omaps
    from oracle import me_cpu as OME

    ref = get_model(model_name, cin, num_classes, ME=OME)
    ref.load_state_dict(state_dict)
    b = make_batches(1, 2, 977, num_classes, grid, cin)[0]
    nproc = os.cpu_count() or 1
    all_threads = torch.get_num_threads()

    def run(threads, budget):
        torch.set_num_threads(threads)
        omaps.set_threads(threads)
        vox, t_total, steps = 0, 0.0, 0
        while t_total < budget and steps < 8:
            t0 = time.perf_counter()
            out = ref(ref.process_input(b))
            loss = F.cross_entropy(out, b["labels"].long())
            loss.backward()
            t_total += time.perf_counter() - t0
            vox += b["coordinates"].shape[0]
            steps += 1
            ref.zero_grad(set_to_none=True)
        return vox / t_total, steps, t_total

    v_all, steps_all, t_all = run(all_threads, seconds_budget / 2)
    v_12, steps_12, t_12 = run(min(12, nproc), seconds_budget / 2)
    torch.set_num_threads(all_threads)
    omaps.set_threads(all_threads)
    t12 = min(12, nproc)
    best = (v_all, all_threads, steps_all, t_all) if v_all >= v_12 else (v_12, t12, steps_12, t_12)
    return {
        "value": best[0],  # the faster of the two thread counts (on a 128-thread host the small per-offset GEMMs of the
        "unit": "voxels/s",  # deep layers are slower with every core than with 12)
        "cores": best[1],
        "kind": "port",
        "sample": f"{best[2]} fwd+bwd steps of {model_name} on 2-scene batches ({b['coordinates'].shape[0]} voxels/step), "
        f"CPU restatement of the ME CPU algorithm (ME binary unavailable), {best[3]:.1f} s",
        "cpu_model": _cpu_model(),
        "nproc": nproc,
        "all_cores": {"value": v_all, "cores": all_threads, "steps": steps_all, "seconds": round(t_all, 1)},
        "omp12": {"value": v_12, "cores": t12, "steps": steps_12, "seconds": round(t_12, 1),
                  "note": "OMP_NUM_THREADS=12 as in the reference's job script (sbatch.sh:34)"},
    }


PMC_CONFIG = "fp32"  # which committed PMC profile `roofline.traffic` may quote: "fp32", "bf16s", None (no profile of this configuration)


def pmc_traffic(tag, meta):
    """(HBM bytes per launch of the dominant kernel, {"profile", "git_head_of_profile"}) from the newest committed PMC
    summary (profiles/*_pmc.json, produced by scripts/pmc_summary.py from separate rocprofv3 --pmc passes;
    2*FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md).  (None, ...) if no matching entry."""
    import glob

    import re

    def natural(path):  # r02_v10 after r02_v9
        return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]

    # a profile of the configuration that is running: profiles/<tag>_pmc.json for the default (fp32) bench,
    # profiles/<tag>_bf16s_pmc.json for --math bf16 --storage bf16 (scripts/profile_round.sh with BENCH_ARGS)
    # (profiles of other configurations -- profiles/<tag>_r34b4_pmc.json: Mink-ResNet34 at four scenes -- are not this run's kernels)
    want_name = re.compile(r"r\d+_v\d+_bf16s_pmc\.json$" if PMC_CONFIG == "bf16s" else r"r\d+_v\d+_pmc\.json$")
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")) if want_name.search(os.path.basename(f))), key=natural)
    if not files or PMC_CONFIG is None:
        return None, None
    source = {"profile": os.path.basename(files[-1]), "git_head_of_profile": "unknown"}
    meta_file = files[-1].replace(".json", ".meta.json")
    if os.path.exists(meta_file):
        source["git_head_of_profile"] = json.load(open(meta_file)).get("git_head", "unknown")
    stem_wgrad = tag.startswith("wgrad") and meta["cin"] <= 32 and meta["K"] == 27
    want = "wgrad_stream" if stem_wgrad else "wgrad_kernel" if tag.startswith("wgrad") else "gather_gemm2_kernel"
    best = None
    for e in json.load(open(files[-1])):
        if want in e["kernel"] and "hbm_traffic_bytes_per_launch" in e:
            if not tag.startswith("wgrad") and e["workgroups"] != -(-meta["n_out"] // 128):
                continue
            if best is None or e["hbm_traffic_bytes_per_launch"] > best:
                best = e["hbm_traffic_bytes_per_launch"]
    return best, source


def _conv_bytes(meta, pairs):
    """Algorithmic bytes of one convolution launch: 4 (N_in Cin + N_out Cout + K Cin Cout) + 8 P."""
    n_in = meta["n_in"] if meta["n_in"] is not None and meta["n_in"] >= 0 else meta["n_out"]
    return 4.0 * (n_in * meta["cin"] + meta["n_out"] * meta["cout"] + meta["K"] * meta["cin"] * meta["cout"]) + 8.0 * pairs


def _parse_kernel(tag):
    """'wgrad[825149x27:28->64]' -> ('wgrad', '27:28->64'): the kernel identity without the row count."""
    kind, rest = tag.split("[", 1)
    return kind, rest.split("x", 1)[1].rstrip("]")


def roofline_from_timings(timings, pair_table):
    """Dominant conv kernel of the timed region: achieved = algorithmic FLOPs per launch
    (2 * pairs * Cin * Cout, pairs = valid neighbour-table entries) / mean HIP-event duration."""
    groups = {}
    for tag, ent in timings.items():  # the two alternating batches give two tags (different row counts) of one kernel
        m = ent["meta"]
        key = (m["kind"], m["K"], m["cin"], m["cout"])
        g = groups.setdefault(key, {"ms": [], "flops": 0.0, "bytes": 0.0, "pairs": 0, "tags": [], "meta": m})
        pairs = m["pairs"] if m["pairs"] is not None else pair_table.get(tag)
        if pairs is None or not ent["ms"]:
            continue
        g["ms"] += ent["ms"]
        g["flops"] += 2.0 * pairs * m["cin"] * m["cout"] * len(ent["ms"])
        g["bytes"] += _conv_bytes(m, pairs) * len(ent["ms"])
        g["pairs"] = max(g["pairs"], pairs)
        g["tags"].append(tag)
    groups = {k: g for k, g in groups.items() if g["ms"]}
    if not groups:
        return None
    g = max(groups.values(), key=lambda g: sum(g["ms"]))
    n, tot = len(g["ms"]), sum(g["ms"])
    avg_ms = tot / n
    achieved = g["flops"] / (tot * 1e-3) / 1e12
    # (since round 5 the mid-layer weight gradients run on the bf16 matrix cores under --math bf16 too: wgrad16_kernel)
    peak = MFMA_PEAK_TFLOPS
    return {
        "bound": "mfma",
        "kernel": g["tags"][0],
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "traffic": pmc_traffic(g["tags"][0], g["meta"])[0],
        # `traffic` is NOT measured in this run: it is read from the newest committed PMC profile, named here with the
        # commit it was taken at -- compare with the head this line was produced from before trusting it
        "traffic_source": pmc_traffic(g["tags"][0], g["meta"])[1],
        "avg_ms": avg_ms,
        "launches": n,
        "flops_per_launch": g["flops"] / n,
        "algorithmic_bytes_per_launch": g["bytes"] / n,
        "hbm_frac_at_algorithmic_bytes": g["bytes"] / (tot * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "pairs": g["pairs"],
    }


def _reserved_total(dev):
    from nerf_downstream_amd import memory

    return memory.reserved_bytes(dev)


class Job:
    """One configuration's training step as the timed region queues it: model + flat gradient buffer (the backward kernels' sink,
    all-reduced in buckets when there is more than one rank) + SGD over the flat buffers + two alternating resident batches, and the
    software pipeline of a step (the coordinate pyramid of batch i+1 launched on the prepare stream before forward i, its tables
    after backward i is queued)."""

    def __init__(self, args, dev, rank, force_reducer=False, prepare_stream=None):
        from nerf_downstream_amd.co3d_3d.src.models import get_model
        from nerf_downstream_amd.co3d_3d.src.modules.classification_training import cross_entropy
        from nerf_downstream_amd.parallel import BucketedGradAllReduce

        self.args, self.dev, self.cross_entropy = args, dev, cross_entropy
        torch.manual_seed(777)  # same initial weights on every rank (reference: pl.seed_everything)
        self.model = model = get_model(args.model, args.in_channel, args.num_classes).to(dev)
        if prepare_stream is not None:  # (a second configuration in one process: the same four streams, not a fifth)
            model._side = prepare_stream
        self.state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        # one flat gradient buffer the backward kernels write into; with N > 1 ranks its buckets are all-reduced (overlapped
        # with backward), with one rank that is all it is -- the step is the same program at every N
        self.reducer = BucketedGradAllReduce(model, bucket_bytes=int(os.environ.get("BENCH_BUCKET_BYTES", 32 << 20)), force=force_reducer)
        # SGD with momentum and weight decay (configs/co3d_cls.gin) as one kernel over the flat buffers (parallel.FlatSGD;
        # BENCH_TORCH_SGD=1: torch's fused multi-tensor SGD, the same update)
        if dev.type == "cuda" and os.environ.get("BENCH_TORCH_SGD", "0") == "0":
            from nerf_downstream_amd.parallel import FlatSGD

            # (BENCH_SGD_IN_BACKWARD=1, one rank: the update of a bucket goes out from inside the backward call as soon as its gradients
            #  are complete -- parallel.FlatSGD(in_backward=True).  MEASURED (A/B on one box, two alternations): 3.50 / 3.52 ms against
            #  3.46 / 3.46 for the one kernel behind the backward pass, --math bf16 2.16 / 2.27 against 2.02 / 2.04, Mink-ResNet34 at
            #  four scenes 4.05 against 3.87: what the per-block ordering of the weight-gradient stream behind the compute stream costs
            #  is more than the 55 / 280 us of update it moves off the end of the step -- OFF by default.)
            self.opt = FlatSGD(self.reducer, lr=0.1, momentum=0.9, weight_decay=1e-4,
                               in_backward=os.environ.get("BENCH_SGD_IN_BACKWARD", "0") != "0")
        else:
            self.opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=dev.type == "cuda")
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, T_max=200000)
        batches = make_batches(2, args.batch, rank, args.num_classes, args.grid, args.in_channel)
        self.batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
        self.vox_per_step = [int(b["coordinates"].shape[0]) for b in self.batches]
        self.labels_dev = [b["labels"].long() for b in self.batches]
        if os.environ.get("BENCH_PREPARE_ON_COMPUTE") == "1" and dev.type == "cuda":  # (timing experiment: the map build in line with the step)
            model._side = torch.cuda.current_stream()
        self.state = {"tf": model.process_input(self.batches[0])}
        self.reuse_maps, self.tf_cache = os.environ.get("BENCH_ABLATE_MAPS", "0") == "1", {}
        # The next batch's maps from a HELPER THREAD (round 6): process_input (pyramid, row-count read-back, tables: 0.7 ms of the
        # host's 2.1 ms per step, most of it inside native calls and an event wait that release the GIL) runs beside the step's own
        # queuing instead of in two slices of it.  Same kernels on the same prepare stream.  MEASURED (A/B, one box, two alternations):
        # fp32 B=16 3.52 / 3.52 ms with the thread against 3.48 / 3.48 without (the build then lands beside the stem's forward),
        # --math bf16 2.04 / 2.05 against 2.07 / 2.07, Mink-ResNet34 at four scenes 3.91 / 3.88 against 3.92 / 3.89: nothing where the
        # GPU is the limit, 1 % where the host is -- OFF by default (BENCH_PREPARE_THREAD=1: on).
        self.prep_pool = None
        if dev.type == "cuda" and os.environ.get("BENCH_PREPARE_THREAD", "0") != "0" and not self.reuse_maps \
                and os.environ.get("BENCH_PREPARE_ON_COMPUTE") != "1":
            from concurrent.futures import ThreadPoolExecutor

            self.prep_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mink-prepare", initializer=torch.cuda.set_device, initargs=(dev,))

    def step_body(self, i):
        # software pipeline: the TensorField (+ coordinate/kernel maps, built on a side stream) of
        # batch i+1 is prepared while batch i computes; every step builds exactly one set.  The
        # coordinate pyramid is launched first, its row counts are read back (and the kernel maps
        # launched) after forward+backward have been queued, so the host never waits for them.
        from nerf_downstream_amd.minkowski import functional as Fn

        model, reducer, opt, state, batches = self.model, self.reducer, self.opt, self.state, self.batches
        tf = state["tf"]
        side = getattr(model, "_side", None)
        cur = Fn.current_stream() if Fn._PHASE_LOG is not None else None  # (the phase marks are a diagnostic: no Stream object per mark otherwise)
        Fn.log_phase("step_begin", cur)
        Fn.log_phase("pyramid_begin", side)
        nb = (i + 1) % len(batches)
        fut = None
        if self.prep_pool is not None:
            fut, nxt = self.prep_pool.submit(model.process_input, batches[nb]), None
        elif self.reuse_maps and nb in self.tf_cache:  # (timing-only ablation: what a step costs WITHOUT building the next batch's maps)
            nxt = None
        else:
            nxt = model.process_input(batches[nb], defer=True)
        Fn.log_phase("pyramid_end", side)
        reducer.zero_grad()
        Fn.log_phase("grads_cleared", cur)
        out = model(tf)
        Fn.log_phase("forward_queued", cur)
        loss = self.cross_entropy(out, self.labels_dev[i % len(batches)])  # the trainer's own loss call (classification_training.py)
        loss.backward()
        Fn.log_phase("backward_queued", cur)
        Fn.log_phase("maps_begin", side)
        if fut is not None:
            state["tf"] = fut.result()
        elif nxt is None:
            state["tf"] = self.tf_cache[nb]
        else:
            state["tf"] = model.finish_input(nxt)
            if self.reuse_maps:
                self.tf_cache[nb] = state["tf"]
        Fn.log_phase("maps_end", side)
        reducer.finish()
        opt.step()
        Fn.log_phase("step_end", cur)
        self.sched.step()
        return loss

    def forward_only_ms(self, steps=10):
        """The network forward on an already prepared batch, training-mode batch norm, no autograd graph (north_star: "fraction of
        HBM roofline on the sparse-conv forward")."""
        with torch.no_grad():
            self.model(self.state["tf"])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                self.model(self.state["tf"])
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps * 1e3


def layer_table_from(warm, pair_table=None):
    """(N_in, N_out, P, Cin, Cout) per layer from the event-timed launches, so the roofline can be recomputed."""
    table = []
    for t, v in warm.items():
        m = v["meta"]
        if m["pairs"] is None or not v["ms"]:
            continue
        if pair_table is not None:
            pair_table[t] = m["pairs"]
        table.append({
            "op": t, "n_in": m["n_in"], "n_out": m["n_out"], "pairs": m["pairs"], "cin": m["cin"],
            "cout": m["cout"], "avg_ms_hip_events_on_a_contended_stream": round(sum(v["ms"]) / len(v["ms"]), 4),
            "gflop": round(2e-9 * m["pairs"] * m["cin"] * m["cout"], 3),
            "algorithmic_mb": round(_conv_bytes(m, m["pairs"]) / 1e6, 2),
        })
    return table


def forward_bytes(layer_table, storage):
    """Algorithmic bytes of the forward convolutions of ONE batch at the storage type the configuration keeps them in.
    The table holds one row per (kernel, row count): the two alternating batches are two rows of every layer, so rows are
    averaged per kernel identity (until round 5 they were SUMMED, which counted every layer twice: the forward's
    `hbm_frac_at_algorithmic_bytes` of BENCH_r02..r05 is 2x too high -- 0.116 there is 0.058)."""
    per_kernel = {}
    for r in layer_table:
        if r["op"].startswith("fwd"):
            per_kernel.setdefault(_parse_kernel(r["op"]), []).append(r["algorithmic_mb"])
    fwd_bytes = sum(sum(v) / len(v) for v in per_kernel.values()) * 1e6
    stem_w = [r for r in layer_table if r["op"].startswith("wgrad") and r["cin"] <= 32 and r["op"].split("x")[1].startswith("27")]
    if storage == "bf16" and stem_w and not any(r["op"].startswith("fwd") and r["cin"] <= 32 for r in layer_table):
        # the bf16-storage stem forward (stem16.hip) is not one of the instrumented gather-GEMM calls: its algorithmic
        # bytes with 2-byte rows in and out (SURVEY 8d with the storage type: x, y, fp32 weights, table)
        r = stem_w[0]
        fwd_bytes += 2.0 * (r["n_in"] * r["cin"] + r["n_out"] * r["cout"]) + 4.0 * 27 * r["cin"] * r["cout"] + 8.0 * r["pairs"]
    return fwd_bytes


def run_other_config(base_args, dev, prepare_stream, model, batch, math, storage, steps, warmup):
    """BASELINE configs #3 (per-GPU shape) and #4 beside the headline, in the same process: a fresh model, optimizer, map plan and
    batches, the same step program, `warmup` untimed steps, then `steps` timed ones between two device synchronisations.  The
    kernel event timing is on for the first warm-up steps only (it yields the per-layer pair counts the forward's algorithmic
    bytes are computed from) and off in the timed region."""
    import copy

    from nerf_downstream_amd.minkowski import functional as Fn

    a = copy.copy(base_args)
    a.model, a.batch, a.math, a.storage = model, batch, math, storage
    old_math, old_storage = Fn.set_conv_math(math), Fn.set_conv_storage(storage)
    try:
        job = Job(a, dev, 0, prepare_stream=prepare_stream)
        Fn.enable_kernel_timing(True)
        n_timed = min(3, warmup)
        for i in range(n_timed):
            job.step_body(i)
        table = layer_table_from(Fn.kernel_timings())
        Fn.enable_kernel_timing(False)
        for i in range(n_timed, warmup):
            job.step_body(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = job.step_body(warmup + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        vox = sum(job.vox_per_step[(warmup + i) % 2] for i in range(steps))
        loss_val = float(loss.item())
        if not (loss_val == loss_val and abs(loss_val) < 1e30):
            raise SystemExit(f"non-finite loss {loss_val} in {model} B={batch} {math}: invalid run")
        fwd_ms = job.forward_only_ms()
        fb = forward_bytes(table, storage)
        return {
            "workload": f"Mink-{model}, batch={batch}/GPU, {a.grid}^3, {math} math, {storage} storage of the full-resolution stage, "
                        "fwd+bwd+SGD step incl. coordinate/kernel map build",
            "ms_per_step": dt / steps * 1e3, "voxels_per_s": vox / dt, "steps": steps, "warmup": warmup,
            "voxels_per_step": job.vox_per_step[0], "final_loss": loss_val,
            "forward_ms": fwd_ms, "forward_voxels_per_s": job.vox_per_step[0] / (fwd_ms * 1e-3),
            "forward_conv_algorithmic_bytes": fb,
            "hbm_frac_at_" + ("bf16" if storage == "bf16" else "fp32") + "_bytes": fb / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fb else None,
        }
    finally:
        Fn.enable_kernel_timing(False)
        Fn.set_conv_math(old_math), Fn.set_conv_storage(old_storage)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="scenes per GPU (co3d_cls.gin: train.batch_size = 16)")
    ap.add_argument("--model", default="ResNet14")
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--in-channel", type=int, default=28)
    ap.add_argument("--num-classes", type=int, default=51)
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="matrix-core arithmetic of conv forward/dgrad (fp32 = exact, the headline; bf16 = BASELINE config #4)")
    ap.add_argument("--storage", default="auto", choices=["auto", "fp32", "bf16"],
                    help="HBM storage of the full-resolution stage (input features, stem output): bf16 needs --math bf16 and is "
                    "what 'auto' picks there (mixed precision = bf16 matrix operands AND bf16 activations where they are large); "
                    "fp32 everywhere else")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (CPU test): rendezvous over gloo, one all-reduce, print ranks_seen; no compute, no number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` leg (BASELINE configs #4 and #3's per-GPU shape, K steps each after the headline)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timeline", action="store_true", help="after the timed region: print where the phases of a step "
                    "fall on the GPU clock of every stream and when the host queued them (diagnostic)")
    args = ap.parse_args()
    if os.environ.get("BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit if stuck
        import faulthandler

        faulthandler.enable()  # also on a GPU fault (SIGABRT): which launch was the host at
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD job and hand its exit code back
        # (nothing in this process has touched the GPU yet; a process that has must never exec another program).  The
        # ranks inherit stdout, so rank 0's JSON line is this command's JSON line.  Same launch as the reference's
        # Trainer(accelerator="gpu", devices=gpus, strategy=DDP) (co3d_3d/train.py:174-186) and as co3d_3d/train.py here.
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29537"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run")
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "ranks_seen": dist.get_world_size() if world > 1 else 1,
                              "rank_sum": t.item()}))
        if world > 1:
            dist.destroy_process_group()
        return
    # rehearsal hooks (single-GPU box): BENCH_DEVICE pins every rank to one card and
    # BENCH_DIST_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised without a node
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_reducer = world == 1 and os.environ.get("BENCH_FORCE_REDUCER") == "1"  # measure the DP machinery on one GPU
    if force_reducer:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
    _dummies = []
    if int(os.environ.get("BENCH_DUMMY_STREAMS", "0")) > 0:
        # measurement hook (scripts/cliff_profile.sh): N idle streams, each used once BEFORE the process group exists, so that the
        # hardware queues the HIP runtime hands out in order of first use are taken and RCCL's stream lands N queues further on
        for _ in range(int(os.environ["BENCH_DUMMY_STREAMS"])):
            st_ = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st_):
                torch.zeros(16, device=dev).add_(1.0)
            _dummies.append(st_)
        torch.cuda.synchronize()
    if world > 1 or force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    collective_desc = "none (one rank)"
    if dist.is_initialized():
        collective_desc = f"{dist.get_backend()} all-reduce of gradients, bucketed, overlapped with backward"
        if dist.get_backend() == "nccl":  # on ROCm the "nccl" backend IS RCCL
            collective_desc += " (RCCL %s)" % ".".join(str(v) for v in torch.cuda.nccl.version())

    from nerf_downstream_amd import _lib
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.minkowski import functional as Fn
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    _lib.lib()  # fail loudly if the HIP backend is missing
    Fn.set_conv_math(args.math)
    if args.storage == "auto":
        args.storage = "bf16" if args.math == "bf16" else "fp32"
    if args.storage == "bf16" and args.math != "bf16":
        raise SystemExit("--storage bf16 needs --math bf16")
    Fn.set_conv_storage(args.storage)
    global PMC_CONFIG, MFMA_PEAK_TFLOPS
    # the stem weight gradient (the dominant kernel) runs on the bf16 matrix cores under --math bf16; split-bf16 issues three
    # bf16 products per useful one
    MFMA_PEAK_TFLOPS = {"fp32": MFMA_F32_PEAK_TFLOPS, "bf16": MFMA_BF16_PEAK_TFLOPS, "bf16x3": MFMA_F32_PEAK_TFLOPS}[args.math]
    PMC_CONFIG = "bf16s" if args.storage == "bf16" else "fp32" if (args.math == "fp32" and args.model == "ResNet14" and args.batch == 16) else None
    if os.environ.get("BENCH_COMPUTE_STREAM", "0") != "0":
        # compute on a stream of its own instead of the legacy default stream: a CU-subset stream
        # (hipExtStreamCreateWithCUMask has no non-blocking flag) synchronises implicitly with the default stream
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    if dev.type == "cuda":
        # one large segment for the caching allocator to carve from (nerf_downstream_amd/memory.py; MINK_RESERVE_GB=0: off): the two
        # alternating batches differ in size, and the pool otherwise grows by hipMalloc (up to 13 ms a call) inside timed steps
        from nerf_downstream_amd.memory import reserve

        reserve(dev)
    job = Job(args, dev, rank, force_reducer=force_reducer)
    model, reducer, state, batches, vox_per_step, state0 = job.model, job.reducer, job.state, job.batches, job.vox_per_step, job.state0
    step_body = job.step_body
    if dev.type == "cuda":
        # the two up-front reservations (24 GB under the compute stream, 16 GB under the prepare stream, which the first
        # process_input has just created) were each ONE allocation, freed at once: without this reset they ARE `max_memory_allocated`
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
    if "BENCH_WGRAD_OVERLAP" in os.environ:
        Fn.set_wgrad_overlap(os.environ["BENCH_WGRAD_OVERLAP"] != "0")

    host_phase = collections.defaultdict(float) if os.environ.get("BENCH_HOST_PHASES") else None  # (diagnostic: host ms per phase)

    def step(i):
        if host_phase is None or i < args.warmup:
            return step_body(i)
        marks = []
        real = Fn.log_phase
        Fn.log_phase = lambda name, stream: (marks.append((name, time.perf_counter())), real(name, stream))
        try:
            t_ = time.perf_counter()
            loss = step_body(i)
            marks.append(("returned", time.perf_counter()))
        finally:
            Fn.log_phase = real
        for name, t in marks:
            host_phase["-> " + name] += t - t_
            t_ = t
        host_phase["steps"] += 1
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: every conv kernel is event-timed to find the dominant one; the timed region then
    # instruments only that kernel (one event pair per step) so the measurement is not perturbed.
    if not args.no_kernel_timing:
        Fn.enable_kernel_timing(True)
    # The host-side housekeeping between warm-up and the timed region (reading the warm-up's kernel timings and the pair counts
    # of their tables back, a full garbage collection: 50-100 ms with the GPU idle) is done BEFORE the last two warm-up steps,
    # so the timed region starts behind real work and a bare fence: started from a card that had idled for 0.1 s, the first
    # ~10 ms of a 76 ms region ran measurably slower (20 timed steps: 3.80 ms/step, 100: 3.72, 300: 3.70 -- whatever the
    # warm-up length; with the housekeeping moved: 3.72-3.74 at 20 steps).
    n_tail = min(2, max(args.warmup - 1, 0))
    for i in range(args.warmup - n_tail):
        step(i)
    dominant = None
    layer_table, pair_table = [], {}
    if not args.no_kernel_timing:
        warm = Fn.kernel_timings()  # (synchronises on the recorded events)
        layer_table = layer_table_from(warm, pair_table)
        if warm:
            # the kernel with the longest typical launch (median over the launches of one kernel identity -- kind, K, cin, cout;
            # the two alternating batches are two tags of it: the first launches of a process run long, and a tag may have
            # a single warm-up sample)
            by_kernel = {}
            for t, v in warm.items():
                by_kernel.setdefault(_parse_kernel(t), []).append((t, v["ms"]))
            med = {}
            for k, ents in by_kernel.items():
                ms = sorted(x for _, m_ in ents for x in m_)
                if ms:
                    med[max(ents, key=lambda e: len(e[1]))[0]] = ms[len(ms) // 2]
            dominant = max(med, key=med.get)
        # (every pass of the dominant kernel's LAYER stays timed -- the stem's forward and weight gradient are within 15 % of each
        #  other, and three warm-up samples do not always rank them: the dominant one is decided on the timed region's own launches)
        Fn.enable_kernel_timing(dominant is not None, only=dominant, any_kind=True)
    # The warmed-up model / optimizer / map plans are permanent: move them out of the cyclic
    # collector's reach so its periodic full collections stop re-traversing them (measured: 0.45 ms
    # per step on average over 200+ steps, pauses of tens of ms); young garbage is still collected.
    import gc

    gc.collect()
    if os.environ.get("BENCH_GC_FREEZE", "1") != "0":
        gc.freeze()
    for i in range(args.warmup - n_tail, args.warmup):
        step(i)
    if n_tail and not args.no_kernel_timing:
        Fn.kernel_timings()  # (drop the tail's launches: the record covers the K timed steps only; ~20 us, no table read-back)
    fence()
    step_marks = [] if (os.environ.get("BENCH_STEP_TIMES") and dev.type == "cuda") else None  # (diagnostic: GPU time of every timed step)
    probe_lib = None
    if step_marks is not None and os.environ.get("BENCH_CLOCK_PROBE"):  # (diagnostic: scripts/ubench/clock_probe.hip after every step)
        import ctypes

        probe_lib = ctypes.CDLL(os.environ["BENCH_CLOCK_PROBE"])
        probe_out = torch.zeros(args.steps, 2, dtype=torch.int64, device=dev)
        trace_ms = int(os.environ.get("BENCH_CLOCK_TRACE_MS", "0"))  # a resident one-wave kernel sampling the clock under load
        if trace_ms:
            trace_out = torch.zeros(trace_ms * 10, 2, dtype=torch.int32, device=dev)
            trace_stream = torch.cuda.Stream()
            torch.cuda.synchronize()
            probe_lib.clock_trace(ctypes.c_void_p(trace_out.data_ptr()), trace_ms * 10, 10000, ctypes.c_void_p(trace_stream.cuda_stream))
        else:
            torch.cuda.synchronize()
    if step_marks is not None and os.environ.get("BENCH_STEP_PHASES"):
        Fn._PHASE_LOG = []
    gc_log = None
    if step_marks is not None:  # (diagnostic: which host pauses are garbage collections, which are device allocations)
        import gc as gc_

        gc_log, gc_t = [], [0.0]

        def gc_cb(phase, info):
            if phase == "start":
                gc_t[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], time.perf_counter() - gc_t[0], gc_t[0]))

        gc_.callbacks.append(gc_cb)
        dev_alloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    prof_ = None
    if os.environ.get("BENCH_CPROFILE"):  # (diagnostic: where the host's time per step goes; backward on the calling thread so it is seen)
        import cProfile

        if os.environ["BENCH_CPROFILE"] != "mt":
            torch.autograd.set_multithreading_enabled(False)
        prof_ = cProfile.Profile()
        prof_.enable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if step_marks is not None and i == int(os.environ.get("BENCH_PAUSE_AT", "-1")):  # (diagnostic: what a pause in mid-run does)
            torch.cuda.synchronize()
            if os.environ.get("BENCH_PAUSE_BUSY") == "1":
                a_ = torch.randn(4096, 4096, device=dev)
                t_ = time.perf_counter()
                while time.perf_counter() - t_ < 0.2:
                    a_ = (a_ @ a_) * 1e-4
                torch.cuda.synchronize()
            else:
                time.sleep(0.2)
        loss = step(args.warmup + i)
        if step_marks is not None:
            if probe_lib is not None:
                probe_lib.clock_probe(ctypes.c_void_p(probe_out[i].data_ptr()), 1000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            ev_ = torch.cuda.Event(enable_timing=True)
            ev_.record()
            step_marks.append((ev_, time.perf_counter()))
            lead_ = int(os.environ.get("BENCH_MAX_LEAD", "0"))  # (diagnostic: hold the host at most this many steps ahead)
            if lead_ and len(step_marks) > lead_:
                step_marks[-1 - lead_][0].synchronize()
    if prof_ is not None:
        prof_.disable()
    fence()
    dt = time.perf_counter() - t0
    if prof_ is not None and rank == 0:
        import io
        import pstats

        for key in ("tottime", "cumulative"):
            buf = io.StringIO()
            pstats.Stats(prof_, stream=buf).sort_stats(key).print_stats(45)
            print(f"[bench] host profile of {args.steps} timed steps by {key}:\n" + buf.getvalue(), file=sys.stderr)
    if step_marks and rank == 0:
        if gc_log is not None:
            import gc as gc_

            gc_.callbacks.remove(gc_cb)
            ms_ = torch.cuda.memory_stats(dev)
            print("[bench] garbage collections inside the timed region (generation, ms, at host ms): " +
                  " ".join(f"g{g}:{d * 1e3:.2f}@{(t - t0) * 1e3:.0f}" for g, d, t in gc_log) +
                  f" | device allocations {ms_.get('num_device_alloc', 0) - dev_alloc0}, allocator retries {ms_.get('num_alloc_retries', 0)}", file=sys.stderr)
        print("[bench] GPU ms between the ends of consecutive timed steps: " +
              " ".join(f"{a[0].elapsed_time(b[0]):.2f}" for a, b in zip(step_marks, step_marks[1:])), file=sys.stderr)
        print("[bench] host ms between queuing the ends of consecutive timed steps: " +
              " ".join(f"{(b[1] - a[1]) * 1e3:.2f}" for a, b in zip(step_marks, step_marks[1:])), file=sys.stderr)
        print("[bench] how far the GPU's end of a step lies behind the host's queuing of it, relative to the first timed step (ms): " +
              " ".join(f"{step_marks[0][0].elapsed_time(m[0]) - (m[1] - step_marks[0][1]) * 1e3:.1f}" for m in step_marks[1:]), file=sys.stderr)
    if probe_lib is not None and rank == 0:
        khz = probe_lib.clock_probe_ref_khz()
        t_ = probe_out.cpu().double()
        print(f"[bench] shader clock seen by a probe kernel after every timed step (MHz; reference counter {khz} kHz): " +
              " ".join(f"{x:.0f}" for x in (t_[:, 0] / t_[:, 1] * khz / 1e3).tolist()), file=sys.stderr)
    if host_phase and rank == 0:
        n_ = host_phase.pop("steps")
        print("[bench] host ms per step spent before each mark (timed steps): " +
              "  ".join(f"{k} {v / n_ * 1e3:.3f}" for k, v in host_phase.items()), file=sys.stderr)
    if probe_lib is not None and trace_ms and rank == 0:
        torch.cuda.synchronize()
        t_ = trace_out.cpu().double().view(-1, 20, 2).sum(1)  # 2 ms per printed value
        print("[bench] shader clock under load, every 2 ms from the start of the timed region (MHz): " +
              " ".join(f"{x:.0f}" for x in (t_[:, 0] / t_[:, 1] * 100).tolist()), file=sys.stderr)
    if step_marks is not None and Fn._PHASE_LOG and rank == 0:
        log, Fn._PHASE_LOG = Fn._PHASE_LOG, None
        begins = [k for k, (n, _, _) in enumerate(log) if n == "step_begin"]
        per = []
        for b0, b1 in zip(begins[:-1], begins[1:]):
            e0 = log[b0][1]
            d_ = {n: e0.elapsed_time(e) for n, e, _ in log[b0:b1]}
            d_["next_step_begin"] = e0.elapsed_time(log[b1][1])
            per.append(d_)
        names = sorted(per[-1], key=lambda n: per[-1][n])
        print("[bench] phases of every timed step (GPU ms after its step_begin): " + " ".join(names), file=sys.stderr)
        for k, d_ in enumerate(per):
            print(f"[bench]   step {k:3d}: " + " ".join(f"{d_.get(n, float('nan')):7.3f}" for n in names), file=sys.stderr)
    Fn._PHASE_LOG = None
    timings = Fn.kernel_timings() if not args.no_kernel_timing else {}
    Fn.enable_kernel_timing(False)

    if args.timeline and rank == 0:
        # diagnostic: where the phases of a step fall on the GPU clock (every stream) and when the host queued them;
        # events are per mark and per step, read after one synchronisation, so the host runs ahead as in the timed region
        Fn._PHASE_LOG = log = []
        nt = 12
        for i in range(nt):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        Fn._PHASE_LOG = None
        begins = [k for k, (n, _, _) in enumerate(log) if n == "step_begin"]
        acc = {}
        for b0, b1 in zip(begins[3:-1], begins[4:]):  # steady-state steps: marks of one step relative to its first
            e0, h0 = log[b0][1], log[b0][2]
            for n, e, h in log[b0:b1]:
                a_ = acc.setdefault(n, [0.0, 0.0, 0])
                a_[0] += e0.elapsed_time(e)
                a_[1] += (h - h0) * 1e3
                a_[2] += 1
            a_ = acc.setdefault("next_step_begin", [0.0, 0.0, 0])
            a_[0] += e0.elapsed_time(log[b1][1])
            a_[1] += (log[b1][2] - h0) * 1e3
            a_[2] += 1
        print("[bench] timeline (ms after step_begin; GPU clock / host clock when queued):", file=sys.stderr)
        for n, (g_, h_, c_) in sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2]):
            print(f"[bench]   {n:22s} gpu {g_ / c_:7.3f}   host {h_ / c_:7.3f}", file=sys.stderr)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    vox = torch.tensor([float(sum(vox_per_step[(args.warmup + i) % len(batches)] for i in range(args.steps)))],
                       dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vox, op=dist.ReduceOp.SUM)
    loss_val = float(loss.item())
    if not (loss_val == loss_val and abs(loss_val) < 1e30):
        raise SystemExit(f"non-finite loss {loss_val}: invalid run")

    if rank == 0:
        storage_note = " math, bf16 storage at full resolution" if args.storage == "bf16" else ""
        res = {
            "metric": f"voxels/sec fwd+bwd Mink-{args.model} on CO3D plenoxels",
            "value": vox.item() / tmax.item(),
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": tmax.item() / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32",
                      "bf16": "bf16 (MFMA operands of every convolution: forward, data gradient and weight gradient; fp32 accumulate; "
                      + ("bf16 storage of the input features and the stem output, fp32 storage below" if args.storage == "bf16" else "fp32 storage")
                      + ")",
                      "bf16x3": "f32 via split-bf16 MFMA (3 products)"}[args.math],
            "data": "synthetic",
            "config": {
                "workload": f"Mink-{args.model} full CO3D-category classification ({args.num_classes} classes), "
                f"batch={args.batch}/GPU, {args.grid}^3 synthetic plenoxel grids (~{vox_per_step[0] // args.batch} voxels x "
                f"{args.in_channel} SH+density features per scene), {args.math}{storage_note}, fwd+bwd+SGD step incl. coordinate/kernel map build",
                "global_batch": args.batch * world,
                "voxels_per_step_per_gpu": vox_per_step[0],
                "parallelism": f"dp{world}",
                "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
                "collective": collective_desc,
                "final_loss": loss_val,
                # allocated = the working set (peak of live tensors since the model was built: weights, flat gradient / momentum
                # buffers, activations, the maps of the batches prepared ahead); pool = what the caching allocator holds from the
                # driver, of which `reserved_up_front` are the two segments taken before the first step (memory.py)
                "peak_memory_gb": ({"allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                                    "pool": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
                                    "reserved_up_front": round(_reserved_total(dev) / 2 ** 30, 2)} if dev.type == "cuda" else None),
            },
        }
        if timings:
            res["roofline"] = roofline_from_timings(timings, pair_table)
        if world == 1:
            # forward only (north_star: "fraction of HBM roofline on the sparse-conv forward")
            fwd_ms = job.forward_only_ms()
            fwd_bytes = forward_bytes(layer_table, args.storage)
            res.setdefault("roofline", {})["forward_only"] = {
                "voxels_per_s": vox_per_step[0] / (fwd_ms * 1e-3), "ms": fwd_ms,
                "conv_algorithmic_bytes": fwd_bytes,
                "hbm_frac_at_algorithmic_bytes": fwd_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fwd_bytes else None,
            }
        if layer_table:
            # (avg_ms here: HIP events around launches on streams that other streams' kernels contend with -- a weight-gradient
            #  kernel reads long when it shares the CUs with the data-gradient chain; per-kernel durations come from the rocprofv3
            #  kernel trace under profiles/, not from this table)
            print("[bench] per-layer conv kernels (warm-up; durations are HIP events on contended streams, see profiles/*_trace_summary.txt "
                  "for per-kernel times): " + json.dumps(layer_table), file=sys.stderr)
        if world == 1 and not args.no_other_configs and (args.model, args.batch, args.math) == ("ResNet14", 16, "fp32"):
            # BASELINE configs #4 and #3 (per-GPU shape) on the driver's record, measured after the headline in the same process;
            # the headline fields above are final by now
            k_o, w_o = min(args.steps, 20), 10  # (a fresh model, plan and allocator state: ten untimed steps whatever the headline's warm-up)
            res["other_configs"] = {
                "bf16_b16": run_other_config(args, dev, model._side, "ResNet14", 16, "bf16", "bf16", k_o, w_o),
                "resnet34_b4": run_other_config(args, dev, model._side, "ResNet34", 4, "fp32", "fp32", k_o, w_o),
            }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.model, args.in_channel, args.num_classes, args.grid, state0)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
