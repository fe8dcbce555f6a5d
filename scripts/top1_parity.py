"""north_star: "top-1 on a fixed synthetic split matching reference +-0.1%".  Exploration tool behind
tests/test_gpu_parity_full.py::test_fixed_split_top1_matches_oracle: trains Mink-ResNet14 on SURVEY 8d's fixed split
(512 train / 128 val, 51 classes) with the HIP backend and with the CPU oracle -- identical init, data order, recipe --
and prints validation top-1 of both, how many validation predictions differ and the smallest top-2 margins.

    python scripts/top1_parity.py [steps] [class_sep] [scene_sigma] [lr] [grid]
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.co3d_3d.src.data.datasets import get_dataset
from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.co3d_3d.train import load_checkpoint, train
from oracle import me_cpu as OME

CFG = os.path.join(os.getcwd(), "nerf_downstream_amd", "co3d_3d", "configs")
arg = lambda i, d, t=float: t(sys.argv[i]) if len(sys.argv) > i else d  # noqa: E731
steps, sep, sigma, lr, grid = arg(1, 300, int), arg(2, 0.25), arg(3, 0.35), arg(4, 0.02), arg(5, 32, int)


def _heartbeat():
    while True:
        time.sleep(60)
        print(f"[heartbeat] {time.strftime('%X')}", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
torch.set_num_threads(min(16, os.cpu_count() or 1))


def bindings():
    return ["train.gpus=1", f"train.max_steps={steps}", f"train.val_every_n_steps={steps}", "train.log_every_n_steps=10",
            f"SparseVoxelDataset.grid={grid}", "SparseVoxelDataset.num_samples=512", "SparseVoxelDataset.num_classes=51",
            f"SparseVoxelDataset.class_sep={sep}", f"SparseVoxelDataset.scene_sigma={sigma}",
            "get_model.out_channel=51", "train.batch_size=8", "train.val_batch_size=16", f"train.lr={lr}",
            "train.train_num_workers=0", "train.val_num_workers=0"]


def run(tag, ME):
    gin.clear_config()
    gin.parse_config_files_and_bindings([f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin", f"{CFG}/synthetic_cls.gin"], bindings())
    t = time.time()
    res = train(save_path=f"/tmp/top1_{tag}", resume_training=False, run_name="r", run_name_postfix=None, ME=ME, seed=11)
    val = [h for h in res["history"] if "val/acc1" in h][-1]
    losses = [h["train/loss"] for h in res["history"] if "train/loss" in h]
    print(f"{tag}: {steps} steps in {time.time() - t:.0f}s  val top-1 {val['val/acc1']:.3f}  val loss {val['val/loss']:.4f}  "
          f"train losses first/last {np.round(losses[:2], 3).tolist()} {np.round(losses[-3:], 3).tolist()}", flush=True)
    # logits of the whole validation split with the final weights, on the backend that trained them
    model = get_model(ME=ME) if ME is not None else get_model().cuda()
    load_checkpoint(f"/tmp/top1_{tag}/r/last.ckpt", model, weights_only=True)
    model.eval()
    ds = get_dataset()(phase="val")
    outs, labels = [], []
    with torch.no_grad():
        for s in range(0, len(ds), 16):
            b = collate_mink([ds[i] for i in range(s, min(s + 16, len(ds)))])
            if ME is None:
                b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
            outs.append(model(model.process_input(b)).float().cpu())
            labels.append(b["labels"].cpu())
    gin.clear_config()
    return val, torch.cat(outs), torch.cat(labels)


va, la, y = run("hip", None)
vb, lb, _ = run("oracle", OME)
pa, pb = la.argmax(1), lb.argmax(1)
top2 = la.topk(2, 1).values
margin = (top2[:, 0] - top2[:, 1]).sort().values
print(f"val top-1: hip {va['val/acc1']:.3f}  oracle {vb['val/acc1']:.3f}  difference {abs(va['val/acc1'] - vb['val/acc1']):.3f} points; "
      f"val-loss difference {abs(va['val/loss'] - vb['val/loss']):.4f}")
print(f"predictions that differ: {(pa != pb).sum().item()} / {len(pa)};  max |logit difference| {(la - lb).abs().max():.3e};  "
      f"five smallest top-2 margins (hip) {np.round(margin[:5].numpy(), 4).tolist()}")
