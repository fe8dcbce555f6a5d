#!/usr/bin/env python3
"""Trainer of the 2-D comparison network: same command line (--ginc / --ginb / --ckpt_path / --resume / --seed) and
gin names (`run.*`, `LitModel.*`, `DataModule.*`, `ResNetBased.*`) as the reference's co3d_2d/train.py:39-122,164-171,
as a plain loop instead of a Lightning Trainer.  `run.precision = 16` runs the convolutions on the bf16 matrix cores
(fp32 accumulate and storage); every convolution, batch norm and pooling is a hand-written gfx950 kernel
(src/model/dense.py).

    python -m nerf_downstream_amd.co3d_2d.train --ginc nerf_downstream_amd/co3d_2d/configs/resnet18.gin"""
import argparse
import logging
import os
import sys

import torch

from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.safe_load import load_checkpoint_file
from nerf_downstream_amd.co3d_2d.src.data.loader import DataModule
from nerf_downstream_amd.co3d_2d.src.modules.classification import LitModel, lr_at

logger = logging.getLogger("co3d_2d")


@gin.configurable
def run(ckpt_path, resume_training, seed, run_name="resnet18", num_gpus=1, log_every_n_steps=100, max_epochs=1000,
        check_val_every_n_epoch=10, precision=16, progressbar_refresh_rate=20, run_train=True, run_eval=True,
        log_dir="co3d_2d/logs", max_steps=-1):
    """`max_steps` (extension): stop after this many optimizer steps (tests, benchmarks)."""
    from nerf_downstream_amd.minkowski import functional as Fn

    if not torch.cuda.is_available():
        raise RuntimeError("co3d_2d.train needs a GPU: the HIP backend has no CPU fallback")
    if num_gpus != 1:
        raise NotImplementedError("the dense baseline is a single-GPU comparison point (BASELINE config #5)")
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    run_name = f"{run_name}_{seed}"
    out_dir = os.path.join(log_dir, run_name)
    os.makedirs(out_dir, exist_ok=True)
    data = DataModule()
    model = LitModel().to(dev)
    opt = model.configure_optimizers()
    step, epoch0 = 0, 0
    if resume_training and ckpt_path:
        ck = load_checkpoint_file(ckpt_path)
        model.load_state_dict(ck["state_dict"])
        opt.load_state_dict(ck["optimizer"])
        step, epoch0 = ck["global_step"], ck["epoch"]
    old_math = Fn.set_conv_math("bf16" if int(precision) == 16 else "fp32")
    history, best = [], -1.0
    try:
        train_loader = data.train_dataloader()
        total = len(train_loader) * max_epochs if max_steps < 0 else max_steps
        model.train()
        done = not run_train
        for epoch in range(epoch0, max_epochs):
            if done:
                break
            for batch in train_loader:
                batch = {k: v.to(dev, non_blocking=True) for k, v in batch.items()}
                for pg in opt.param_groups:
                    pg["lr"] = lr_at(step, model.lr, total)
                opt.zero_grad(set_to_none=True)
                loss, logs = model.training_step(batch)
                loss.backward()
                opt.step()
                step += 1
                if step % log_every_n_steps == 0 or step == total:
                    row = {"global_step": step, "lr": opt.param_groups[0]["lr"], **{k: float(v) for k, v in logs.items()}}
                    history.append(row)
                    logger.info(" ".join(f"{k}={v:.4g}" for k, v in row.items()))
                if step >= total:
                    done = True
                    break
            if (epoch + 1) % check_val_every_n_epoch == 0 or done:
                vm = model.evaluation(data.val_dataloader(), dev, "val")
                history.append({"global_step": step, **vm})
                logger.info(f"val Acc: {vm['val/acc']}, val Loss: {vm['val/loss']}")
                ck = {"state_dict": model.state_dict(), "optimizer": opt.state_dict(), "global_step": step, "epoch": epoch + 1}
                torch.save(ck, os.path.join(out_dir, "last.ckpt"))
                if vm["val/acc"] > best:
                    best = vm["val/acc"]
                    torch.save(ck, os.path.join(out_dir, "best.ckpt"))
        if run_eval:
            tm = model.evaluation(data.test_dataloader(), dev, "test")
            history.append({"global_step": step, **tm})
    finally:
        Fn.set_conv_math(old_math)
    return {"global_step": step, "history": history, "best": best}


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--ginc", action="append", help="gin config file")
    parser.add_argument("--ginb", action="append", help="gin bindings")
    parser.add_argument("--ckpt_path", type=str, default=None, help="path to load the ckpt")
    parser.add_argument("--resume", action="store_true", default=False, help="resume training")
    parser.add_argument("--seed", type=int, default=333)
    args = parser.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %H:%M:%S",
                        handlers=[logging.StreamHandler(sys.stdout)], force=True)
    logging.info(f"Gin configuration files: {args.ginc}")
    logging.info(f"Gin bindings: {args.ginb or []}")
    gin.parse_config_files_and_bindings(args.ginc, args.ginb or [])
    run(ckpt_path=args.ckpt_path, resume_training=args.resume, seed=args.seed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
