"""End-to-end rate of the thing north_star names: `python -m nerf_downstream_amd.co3d_3d.train` (BASELINE config #2: Mink-ResNet14,
51 classes, 16 scenes of 128^3 per step, fp32) fed by its own DataLoader from scenes ON DISK in the reference's PeRFception-CO3D
format (data.npz: links / density / uint8 sh + scale, min -- scripts/preprocess.py:30-57 of the reference), next to bench.py's step.

    python scripts/train_e2e.py [--scenes 256] [--steps 300] [--workers 14] [--form compact|decoded] [--staging 1|0]

Writes the scenes once (synthetic plenoxel shells of the co3d_3d default shape, quantised as the reference's preprocessing does),
runs the trainer through its command line in a child process (its log lines carry train/iter_time), and prints the sustained
iteration time over the last two thirds of the run.  Stages are timed separately with --stages (loader alone, no GPU work)."""
import argparse
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, "nerf_downstream_amd", "co3d_3d", "configs")


def _write_scene(args):
    root, j = args
    from nerf_downstream_amd.co3d_3d.src.data.co3d import CLASSES
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import make_scene

    xyz, density, sh, label = make_scene(j, 0, 51, 128)
    links = (xyz[:, 0] * 128 * 128 + xyz[:, 1] * 128 + xyz[:, 2]).astype(np.int32)
    lo, hi = float(sh.min()), float(sh.max())
    scale = np.float32((hi - lo) / 255.0)
    q = np.clip(np.rint((sh - lo) / scale), 0, 255).astype(np.uint8)
    d = os.path.join(root, "data", f"plenoxel_co3d_s{j}")
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, "data.npz"), links=links, density=density.astype(np.float32), sh=q, sh_min=np.float32(lo), sh_scale=scale)
    return f"{CLASSES[label]} s{j}"


def write_scenes(root, n, procs, repeat=16):
    from multiprocessing import Pool

    os.makedirs(os.path.join(root, "filelist"), exist_ok=True)
    t0 = time.perf_counter()
    with Pool(procs) as pool:
        lines = pool.map(_write_scene, [(root, j) for j in range(n)])
    # the training list names every scene `repeat` times: an epoch of 16 batches (256 scenes) would measure the DataLoader's start of
    # an epoch (every worker idle, then all of them at once), not its steady state -- PeRFception-CO3D has ~18.6 k scenes
    for phase in ("train", "test"):
        with open(os.path.join(root, "filelist", f"{phase}.txt"), "w") as f:
            f.write("\n".join(lines * repeat if phase == "train" else lines[:16]) + "\n")
    print(f"[e2e] wrote {n} scenes to {root} in {time.perf_counter() - t0:.1f} s", flush=True)


def loader_alone(root, workers, form, batches=40):
    """The DataLoader by itself (workers: npz -> tensors -> collate -> shared memory -> this process), then the pinned pack."""
    os.chdir(root)
    import torch

    from nerf_downstream_amd import gin_lite as gin
    from nerf_downstream_amd.co3d_3d.src.data.data_module import DataModule

    gin.clear_config()
    gin.parse_config_files_and_bindings([f"{CFG}/co3d_cls.gin"], [f"Co3DDatasetBase.data_root='{root}/data'", "Co3DDatasetBase.features=['density','sh']",
                                                                  f"Co3DDatasetBase.compact={form == 'compact'}"])
    dm = DataModule("train", "val", "test", 16, 8, workers, 0, "collate_mink")
    dl = dm.train_dataloader()
    n, t0, nbytes = 0, None, 0
    while n < batches + 5:
        for b in dl:
            n += 1
            if n == 5:
                t0 = time.perf_counter()
            nbytes = sum(v.numel() * v.element_size() for v in b.values() if torch.is_tensor(v))
            if n >= batches + 5:
                break
    per = (time.perf_counter() - t0) / batches
    print(f"[e2e] DataLoader alone, {dl.num_workers} workers, {form}: {per * 1e3:.2f} ms per batch of 16 scenes ({nbytes / 1e6:.1f} MB)", flush=True)
    if torch.cuda.is_available():
        from nerf_downstream_amd.co3d_3d.src.data.staging import PinnedStager

        st = PinnedStager(torch.device("cuda", 0))
        st.upload(st.pack(b))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pk = st.pack(b)
        t_pack = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            st.upload(pk)
        torch.cuda.synchronize()
        t_up = (time.perf_counter() - t0) / 20
        print(f"[e2e] pack into pinned memory {t_pack * 1e3:.2f} ms per batch ({st.PACK_THREADS} copy threads), one H2D copy of the packed "
              f"batch {t_up * 1e3:.2f} ms ({nbytes / 1e9 / t_up:.1f} GB/s over the bus)", flush=True)
    gin.clear_config()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default="/tmp/mink_e2e")
    ap.add_argument("--scenes", type=int, default=256)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--form", default="compact", choices=["compact", "decoded"])
    ap.add_argument("--staging", default="1")
    ap.add_argument("--stages", action="store_true", help="time the loader and the pinned pack alone, then exit")
    args = ap.parse_args()
    marker = os.path.join(args.root, f".done_{args.scenes}_x16")
    if not os.path.exists(marker):
        write_scenes(args.root, args.scenes, min(14, os.cpu_count() or 1))
        open(marker, "w").close()
    if args.stages:
        loader_alone(args.root, args.workers, args.form)
        return
    log_every = 25
    cmd = [sys.executable, "-m", "nerf_downstream_amd.co3d_3d.train", "--ginc", f"{CFG}/co3d_cls.gin", "--ginc", f"{CFG}/resnet14.gin",
           "--save_path", os.path.join(args.root, f"run_{args.form}_{args.staging}_{args.workers}"), "--run_name", "e2e",
           "--ginb", f"Co3DDatasetBase.data_root='{args.root}/data'", "--ginb", "Co3DDatasetBase.features=['density','sh']",
           "--ginb", f"Co3DDatasetBase.compact={args.form == 'compact'}", "--ginb", "get_model.in_channel=28",
           "--ginb", f"train.max_steps={args.steps}", "--ginb", f"train.val_every_n_steps={10 ** 9}", "--ginb", f"train.log_every_n_steps={log_every}",
           "--ginb", "train.loggers=['csv']", "--ginb", f"train.train_num_workers={args.workers}", "--ginb", "train.val_num_workers=0",
           "--ginb", "train.val_batch_size=8", "--ginb", "train.lr=0.01"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MINK_STAGING=args.staging)
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=args.root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    wall = time.perf_counter() - t0
    its = [float(m.group(1)) for m in re.finditer(r"train/iter_time=([0-9.e+-]+)", p.stdout)]
    if p.returncode != 0 or not its:
        print(p.stdout[-4000:])
        raise SystemExit(f"trainer failed (rc {p.returncode})")
    tail = its[len(its) // 3:]
    direct = args.form == "compact" and args.staging != "0" and os.environ.get("MINK_DIRECT_LOADER", "1") != "0"
    feed = (f"direct reader ({max(2, min(8, args.workers))} threads: scene files -> pinned buffers)" if direct else
            f"DataLoader, {args.workers} worker processes" + (", pinned staging + copy stream" if args.staging != "0" else ", pageable copies on the compute stream"))
    print(f"[e2e] co3d_3d.train CLI, ResNet14 B=16 fp32, {args.form} samples, {feed}, "
          f"{args.steps} steps ({wall:.0f} s wall incl. start-up and the final validation): train/iter_time over windows of {log_every} steps (ms): "
          + " ".join(f"{t * 1e3:.2f}" for t in its) + f" | sustained (last two thirds): {np.mean(tail) * 1e3:.2f} ms per step", flush=True)


if __name__ == "__main__":
    main()
