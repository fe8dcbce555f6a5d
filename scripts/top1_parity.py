"""north_star: "top-1 on a fixed synthetic split matching reference +-0.1%".  Trains Mink-ResNet14 for
`steps` steps on the fixed synthetic split with the HIP backend and with the CPU oracle (identical init,
data order, recipe) and prints final validation top-1 / loss of both.  Run on the GPU box (a few minutes)."""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.co3d_3d.train import train
from oracle import me_cpu as OME

CFG = os.path.join(os.getcwd(), "nerf_downstream_amd", "co3d_3d", "configs")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300

def _heartbeat():
    while True:
        time.sleep(60)
        print(f"[heartbeat] {time.strftime('%X')}", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()


def run(tag, ME):
    gin.clear_config()
    gin.parse_config_files_and_bindings(
        [f"{CFG}/co3d_cls.gin", f"{CFG}/resnet14.gin", f"{CFG}/synthetic_cls.gin"],
        ["train.gpus=1", f"train.max_steps={steps}", f"train.val_every_n_steps={steps}", "train.log_every_n_steps=10",
         "SparseVoxelDataset.grid=32", "SparseVoxelDataset.num_samples=640", "SparseVoxelDataset.num_classes=8",
         "get_model.out_channel=8", "train.batch_size=8", "train.val_batch_size=16", "train.lr=0.003",
         "train.train_num_workers=0", "train.val_num_workers=0"])
    t = time.time()
    res = train(save_path=f"/tmp/top1_{tag}", resume_training=False, run_name="r", run_name_postfix=None, ME=ME, seed=11)
    gin.clear_config()
    val = [h for h in res["history"] if "val/acc1" in h][-1]
    losses = [h["train/loss"] for h in res["history"] if "train/loss" in h]
    print(f"{tag}: {steps} steps in {time.time() - t:.0f}s  val top-1 {val['val/acc1']:.3f}  val loss {val['val/loss']:.4f}  "
          f"last train losses {np.round(losses[-3:], 4).tolist()}", flush=True)
    return val

a = run("hip", None)
b = run("oracle", OME)
print(f"top-1 difference {abs(a['val/acc1'] - b['val/acc1']):.3f} points, val-loss difference {abs(a['val/loss'] - b['val/loss']):.4f}")
