#!/usr/bin/env python3
"""Trainer of the plenoxel voxel-grid classifier: same command line, gin names and training
recipe as the reference's co3d_3d/train.py (argparse :198-238, `train()` parameters :50-97),
driven by the same config files, e.g.

    python -m nerf_downstream_amd.co3d_3d.train --ginc nerf_downstream_amd/co3d_3d/configs/co3d_cls.gin \
        --ginc nerf_downstream_amd/co3d_3d/configs/resnet14.gin --gpus 8

The reference builds a PyTorch-Lightning Trainer with DDP (:174-187); this is a plain loop with
the same semantics -- per-rank batches of `train.batch_size`, per-rank BatchNorm statistics,
mean of the gradients over the ranks each step (bucketed RCCL all-reduce overlapped with
backward, nerf_downstream_amd/parallel.py), SGD / cosine schedule stepped per iteration,
validation top-1/top-5, `last.ckpt` / best checkpoint with `model.`-prefixed keys.
One process per GPU: with --gpus N > 1 and no torchrun environment it re-launches itself under
`python -m torch.distributed.run` (as a child process, before touching the GPU).
"""
import argparse
import csv
import json
import logging
import os
import subprocess
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # one hardware queue for every stream of a rank (compute, prepare, weight-gradient, bucket launches + RCCL's): the HIP
    # default of 4 makes two of them share one and serialise; read when the HIP runtime is loaded, i.e. before `import torch`.
    # Seven: at eight the step falls off a cliff (bench.py: 3.81 ms at 7, 5.7-6.0 at 8 with a one-rank RCCL group)
    # (round 5: the cliff is a fifth busy hardware queue -- nerf_downstream_amd/hwqueues.py; the collectives no longer have a stream
    #  of their own, and an inherited value on the wrong side is refused there)
    from nerf_downstream_amd.hwqueues import configure as _configure_hw_queues

    _configure_hw_queues(data_parallel=True)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.co3d_3d.src.data.data_module import DataModule
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.co3d_3d.src.modules.classification_training import ClassificationTraining
from nerf_downstream_amd.co3d_3d.src.modules.segmentation_training import SegmentationTraining

TRAINING_MODULES = {"ClassificationTraining": ClassificationTraining, "SegmentationTraining": SegmentationTraining}
from nerf_downstream_amd.co3d_3d.src.modules.optim import get_optimizer, get_scheduler
from nerf_downstream_amd.parallel import BucketedGradAllReduce
from nerf_downstream_amd.safe_load import load_checkpoint_file

logger = logging.getLogger(__name__)


class CSVLogger:
    """metrics.csv of the run: one line appended per log call (the file is rewritten only when a new metric name
    widens the header); an existing file -- a run being resumed -- is continued, not overwritten."""

    def __init__(self, save_path, run_name, resume=False):
        self.dir = os.path.join(save_path, run_name)
        os.makedirs(self.dir, exist_ok=True)
        self.path = os.path.join(self.dir, "metrics.csv")
        self.rows, self.keys = [], []
        if resume and os.path.exists(self.path):
            with open(self.path, newline="") as f:
                rd = csv.DictReader(f)
                self.keys = list(rd.fieldnames or [])
                self.rows = [{k: v for k, v in r.items() if v != ""} for r in rd]
        elif os.path.exists(self.path):
            os.remove(self.path)

    def log_dict(self, metrics, step):
        row = {"global_step": step, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in metrics.items()}}
        self.rows.append(row)
        if self.keys and all(k in self.keys for k in row):
            with open(self.path, "a", newline="") as f:
                csv.DictWriter(f, fieldnames=self.keys).writerow(row)
            return
        self.keys = sorted(set(self.keys) | set(row))
        with open(self.path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=self.keys)
            w.writeheader()
            w.writerows(self.rows)


def _to_device(batch, device):
    # the augmentation parameter rows stay on the host too: `augment_batch` uploads them itself and can tell
    # from them, without a read-back, whether any voxel will be dropped
    out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) and k != "aug_params" else v) for k, v in batch.items()}
    if device.type == "cuda":  # process_input prepares the batch on another stream: it waits for the copies through this
        out["h2d_event"] = torch.cuda.current_stream(device).record_event()
    return out


def _dist_env():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def save_checkpoint(path, model, optimizer, scheduler, step, epoch, best, batch_in_epoch=0):
    torch.save(
        {
            "state_dict": {"model." + k: v for k, v in model.state_dict().items()},  # Lightning key layout
            "optimizer_states": [optimizer.state_dict()],
            "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else [],
            "global_step": step,
            "epoch": epoch,
            "batch_in_epoch": batch_in_epoch,  # batches of `epoch` already consumed: a resumed run skips them
            "best": best,
        },
        path,
    )


def load_checkpoint(path, model, optimizer=None, scheduler=None, weights_only=False):
    ckpt = load_checkpoint_file(path)  # this trainer's checkpoints and the reference's (Lightning layout): tensors and plain containers only
    sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in ckpt["state_dict"].items()}
    model.load_state_dict(sd)
    if not weights_only:
        if optimizer is not None and ckpt.get("optimizer_states"):
            optimizer.load_state_dict(ckpt["optimizer_states"][0])
        if scheduler is not None and ckpt.get("lr_schedulers"):
            scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return ckpt


@torch.no_grad()
def validate(module, loader, device, world):
    model = module.model
    was_training = model.training
    model.eval()
    tot = None  # the module's accumulable vector (classification: loss*n, correct@1, correct@5, n; segmentation: + confusion matrix)
    for batch in loader:
        batch = _to_device(batch, device)
        v = module.val_accumulate(batch)
        tot = v if tot is None else tot + v
    if world > 1:
        dist.all_reduce(tot)
    model.train(was_training)
    return module.val_metrics(tot)


@gin.configurable
def train(
    save_path: str,
    gpus: int,
    run_name: str,
    run_name_postfix: str,
    project_name: str,
    max_steps: int,
    max_epochs: int,
    warmup_steps: int = -1,
    model=None,
    training_module: str = "SegmentationTraining",
    optimizer_name: str = "SGD",
    scheduler_name: str = "PolyLR",
    scheduler_interval: str = "step",
    lr: float = 1e-3,
    weight_decay: float = 1e-4,
    batch_size: int = 8,
    val_batch_size: int = 6,
    prune_batch_size: int = 8,
    train_num_workers: int = 4,
    val_num_workers: int = 2,
    collate_func_name: str = "collate_mink",
    val_every_n_steps: int = 1000,
    log_every_n_steps: int = 10,
    reset_profiler_every_n_steps: int = 1000,
    progressbar_refresh_rate: int = 1,
    loggers: list = ["csv"],
    resume_training: bool = False,
    checkpoint_path: str = None,
    load_weights: bool = False,
    load_optimizers: bool = False,
    transfer_self_supervised: bool = False,
    use_sync_batchnorm: bool = False,
    use_sync_grad: bool = False,
    ignore_label: int = -100,
    train_phase="train",
    val_phase="val",
    test_phase="test",
    monitor_metric: str = "val/mIoU",
    evaluate: bool = False,
    void_weight=None,
    debug: bool = False,
    ME=None,
    device=None,
    seed: int = 777,
):
    """Same parameters (and gin bindings `train.*`) as the reference.  `ME` / `device` are test
    hooks: the CPU tests inject the oracle namespace to run BASELINE config #1 on the host."""
    if scheduler_interval != "step":
        raise NotImplementedError("the classification configs step the scheduler per iteration")
    world, rank, local_rank = _dist_env()
    if device is None:
        if ME is None and not torch.cuda.is_available():
            raise RuntimeError("co3d_3d.train needs a GPU: the HIP backend has no CPU fallback (reference: accelerator='gpu')")
        device = torch.device("cuda", local_rank) if torch.cuda.is_available() and ME is None else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on hosts without the legacy path)
        dist.init_process_group("nccl" if device.type == "cuda" else "gloo")
    os.makedirs(save_path, exist_ok=True)
    if device.type == "cuda":
        from nerf_downstream_amd.memory import reserve

        reserve(device)  # one segment for the caching allocator to split, instead of hipMalloc calls in the middle of steps

    torch.manual_seed(seed)  # identical initial weights on every rank (reference: pl.seed_everything)
    if model is None:
        model = get_model(ME=ME) if ME is not None else get_model()
    if use_sync_batchnorm and world > 1:
        from nerf_downstream_amd import minkowski

        model = minkowski.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model)
    model = model.to(device)
    if run_name is None or "default" in run_name.lower() or run_name == "":
        run_name = f"b{batch_size}x{gpus}-{model.__class__.__name__}"
    if run_name_postfix is not None:
        run_name += "-" + run_name_postfix

    data = DataModule(train_phase, val_phase, test_phase, batch_size, val_batch_size, train_num_workers, val_num_workers,
                      collate_func_name, world_size=world, rank=rank, seed=seed)
    if training_module not in TRAINING_MODULES:
        raise ValueError(f"train.training_module = {training_module!r}; available: {sorted(TRAINING_MODULES)}")
    module = TRAINING_MODULES[training_module](model)
    optimizer = get_optimizer(optimizer_name, model.parameters(), lr=lr, weight_decay=weight_decay)
    # flat gradient buffer written by the backward kernels (HIP backend); all-reduced in buckets when world > 1
    reducer = BucketedGradAllReduce(model) if (world > 1 or (device.type == "cuda" and ME is None)) else None
    if reducer is not None and reducer.flat.is_cuda and optimizer_name == "SGD" and os.environ.get("MINK_TORCH_SGD", "0") == "0":
        # the same update as torch.optim.SGD, as one kernel over the flat parameter / gradient / momentum buffers
        from nerf_downstream_amd.parallel import FlatSGD

        # (MINK_SGD_IN_BACKWARD=1, one rank: a bucket's update is launched from inside the backward call; measured slower, bench.py)
        optimizer = FlatSGD.like(optimizer, reducer, in_backward=world == 1 and os.environ.get("MINK_SGD_IN_BACKWARD", "0") != "0")
    scheduler = get_scheduler(scheduler_name, optimizer, warmup_steps)
    csv_logger = CSVLogger(save_path, run_name, resume=resume_training) if rank == 0 and "csv" in loggers else None
    for name in loggers:
        if name != "csv" and rank == 0:
            logger.warning(f"logger {name!r} is not available here; only 'csv' is written")

    step, epoch, best, skip_batches = 0, 0, -float("inf"), 0
    last_ckpt = os.path.join(save_path, run_name, "last.ckpt")
    if resume_training or load_weights:
        path = checkpoint_path or last_ckpt
        ck = load_checkpoint(path, model, optimizer if (resume_training or load_optimizers) else None,
                             scheduler if resume_training else None, weights_only=not resume_training and not load_optimizers)
        if resume_training:
            step, epoch, best = ck["global_step"], ck["epoch"], ck.get("best", best)
            skip_batches = int(ck.get("batch_in_epoch", 0))
        logger.info(f"loaded {path} (step {step})")

    train_loader, val_loader = data.train_dataloader(), data.val_dataloader()
    # training batches reach the device through pinned staging buffers and a copy stream of their own (data/staging.py):
    # a helper thread packs every collated batch into pinned memory, the loop below uploads it with ONE asynchronous copy
    # whose event the model's prepare stream waits for.  MINK_STAGING=0: a pageable copy per tensor on the compute stream
    # (what the reference's Lightning loop does with pin_memory=False).
    stager = None
    if device.type == "cuda" and ME is None and os.environ.get("MINK_STAGING", "1") != "0":
        from nerf_downstream_amd.co3d_3d.src.data.staging import DirectCompactLoader, PinnedStager, StagedLoader

        stager = PinnedStager(device)
    total_steps = max_steps + (warmup_steps if warmup_steps > 0 else 0)
    history = []
    model.train()
    t_iter = time.perf_counter()
    done = step >= total_steps
    while not done and (max_epochs <= 0 or epoch < max_epochs):
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        batch_in_epoch = 0
        if skip_batches and hasattr(train_loader.sampler, "skip"):  # resumed inside this epoch: drop what was trained on, by index
            train_loader.sampler.skip(skip_batches * train_loader.batch_size)
            batch_in_epoch, skip_batches = skip_batches, 0
        direct = (stager is not None and os.environ.get("MINK_DIRECT_LOADER", "1") != "0" and train_loader.sampler is not None
                  and DirectCompactLoader.usable(train_loader.dataset))
        if direct:
            # compact scenes on disk: read straight into the pinned staging buffers by a few threads of this process (data/staging.py)
            idx = iter(train_loader.sampler)
            for _ in range(skip_batches * train_loader.batch_size):  # (samplers without skip(): the indices are drawn and dropped)
                if next(idx, None) is None:
                    break
            batch_in_epoch += skip_batches
            skip_batches = 0
            next_batch = DirectCompactLoader(train_loader.dataset, idx, train_loader.batch_size, stager,
                                             threads=max(2, min(8, train_num_workers or 2))).next
            it = None
        else:
            it = iter(train_loader)
        for _ in range(skip_batches):  # (samplers without skip(): the batches are drawn and dropped)
            if next(it, None) is None:
                break
            batch_in_epoch += 1
        skip_batches = 0
        if direct:
            pass
        elif stager is not None:
            staged = StagedLoader(it, stager)
            next_batch = staged.next
        else:
            def next_batch(it=it):
                b = next(it, None)
                return _to_device(b, device) if b is not None else None
        batch = next_batch()
        if batch is not None:
            field = model.process_input(batch)
        while batch is not None:
            # prepare the next batch on the side stream under this one's compute: its coordinate
            # pyramid is launched now, the row counts are read back and the kernel maps launched
            # once forward+backward are queued, so the host never waits for the device here
            nxt_batch = next_batch()
            if nxt_batch is not None:
                nxt_field = model.process_input(nxt_batch, defer=True)
            if reducer is not None:
                reducer.zero_grad()
            else:
                optimizer.zero_grad(set_to_none=True)
            if stager is not None:  # (the labels and features of a staged batch were written by the copy stream)
                torch.cuda.current_stream(device).wait_event(batch["h2d_event"])
            loss, out = module.training_step(batch, field)
            loss.backward()
            cur_batch = batch
            batch = nxt_batch
            if batch is not None:
                field = model.finish_input(nxt_field)
            if reducer is not None:
                reducer.finish()
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            step += 1
            batch_in_epoch += 1
            if step == 3:
                # the warmed-up model / optimizer / coordinate plans are permanent: keep the cyclic collector's
                # periodic full collections from re-traversing them (0.4 ms per step on average at ~5 ms steps)
                import gc

                gc.collect()
                gc.freeze()
            if step % log_every_n_steps == 0:
                loss_float = float(loss.detach().cpu())
                module.check_finite(loss_float)
                now = time.perf_counter()
                m = {"train/loss": loss_float, **module.train_metrics(out.detach(), cur_batch),
                     "train/iter_time": (now - t_iter) / log_every_n_steps, "lr": optimizer.param_groups[0]["lr"]}
                t_iter = now
                history.append({"global_step": step, **m})
                if csv_logger:
                    csv_logger.log_dict(m, step)
                if rank == 0:
                    logger.info(f"step {step}: " + " ".join(f"{k}={v:.4g}" for k, v in m.items()))
            if stager is not None:
                stager.release(cur_batch)  # its device buffer may be overwritten behind everything queued for it so far
            if step % val_every_n_steps == 0 or step >= total_steps:
                vm = validate(module, val_loader, device, world)
                history.append({"global_step": step, **vm})
                if csv_logger:
                    csv_logger.log_dict(vm, step)
                if rank == 0:
                    logger.info(f"step {step}: " + " ".join(f"{k}={v:.4g}" for k, v in vm.items()))
                    os.makedirs(os.path.dirname(last_ckpt), exist_ok=True)
                    save_checkpoint(last_ckpt, model, optimizer, scheduler, step, epoch, best, batch_in_epoch)
                    score = vm.get(monitor_metric, vm[module.monitor])
                    if score > best:
                        best = score
                        save_checkpoint(os.path.join(save_path, run_name, "best.ckpt"), model, optimizer, scheduler, step, epoch, best,
                                        batch_in_epoch)
            if step >= total_steps:
                done = True
                break
        epoch += 1
    results = {"global_step": step, "best": best, "history": history}
    if evaluate and rank == 0:
        with open(os.path.join(save_path, "eval_results.json"), "w") as f:
            json.dump({k: v for k, v in results.items() if k != "history"}, f)
    return results


def setup_logger(exp_name, debug):
    tag = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "0"))
    logging.basicConfig(level=logging.DEBUG if debug else logging.INFO, format=f"{tag}:[{exp_name}] %(asctime)s %(message)s",
                        datefmt="[%X]", handlers=[logging.StreamHandler(sys.stdout)], force=True)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--ginc", action="append", help="gin config file")
    p.add_argument("--ginb", action="append", help="gin bindings")
    p.add_argument("--save_path", type=str, default="experiments", help="path for logging")
    p.add_argument("--resume", action="store_true", help="resume training")
    p.add_argument("--run_name", type=str, default=None)
    p.add_argument("--run_name_postfix", type=str, default=None)
    p.add_argument("--gpus", type=int, default=1, help="num_gpus")
    p.add_argument("--seed", type=int, default=777)
    p.add_argument("--debug", action="store_true")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU; start the workers as a child job and return its exit code
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               "-m", "nerf_downstream_amd.co3d_3d.train"] + (argv if argv is not None else sys.argv[1:])
        return subprocess.call(cmd)
    run_name = args.run_name if args.run_name is not None else "default"
    if args.run_name_postfix is not None:
        run_name = f"{run_name}-{args.run_name_postfix}"
    setup_logger(f"{run_name}_{args.seed}", args.debug)
    ginbs = [f"train.gpus={args.gpus}"] + (args.ginb or [])
    logging.info(f"Gin configuration files: {args.ginc}")
    logging.info(f"Gin bindings: {ginbs}")
    np.random.seed(args.seed)
    gin.parse_config_files_and_bindings(args.ginc, ginbs)
    train(save_path=args.save_path, resume_training=args.resume, run_name=args.run_name,
          run_name_postfix=args.run_name_postfix, seed=args.seed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
