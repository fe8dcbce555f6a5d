"""TEST INFRASTRUCTURE (see oracle/__init__.py).  Trains the CPU oracle on SURVEY 8d's fixed split and writes
tests/golden/top1_oracle_v1.npz -- the oracle half of tests/test_gpu_parity_full.py::test_fixed_split_top1_statistics.

The oracle's 300-step runs cost ~1 s per step on eight cores: they belong in the build container, once, not on the GPU
box inside the driver's time limit.  The GPU test trains the HIP seeds only and compares with this file.

    python oracle/make_top1_fixture.py [n_seeds=6] [threads=6]     (an existing file of the same recipe is continued)

Per seed s (initial weights 11 + 1000 s, data order 1234 + s; tests/top1_recipe.py): the per-step training loss, the
oracle's own argmax over the 1,024-scene validation split, its top-1; plus the labels and the recipe hash the GPU test
recomputes before it trusts the file.  What the reference step is: co3d_3d/src/modules/classification_training.py:52-97.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import top1_recipe as R  # noqa: E402

from oracle import maps, me_cpu as OME  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    out = os.path.join(ROOT, "tests", "golden", "top1_oracle_v1.npz")
    torch.set_num_threads(threads)
    maps.build()
    maps.set_threads(threads)
    cpu = torch.device("cpu")
    t0 = time.time()
    val = R.stat_val_batches()
    labels = torch.cat([y for _, y in val]).numpy()
    print(f"recipe {R.recipe_hash()}: {len(labels)} validation scenes generated in {time.time() - t0:.0f}s", flush=True)
    losses, preds, top1 = [], [], []
    if os.path.exists(out):  # continue a file of the same recipe (seeds are independent runs)
        old = np.load(out)
        if str(old["recipe"]) == R.recipe_hash() and np.array_equal(old["labels"], labels.astype(np.uint8)):
            losses, preds, top1 = list(old["losses"]), list(old["preds"]), [float(v) for v in old["top1"]]
            print(f"continuing {out}: seeds 0..{len(top1) - 1} present", flush=True)
    for s in range(len(top1), n_seeds):
        t = time.time()
        model, ls = R.fit(OME, cpu, seed=s, progress=lambda k, l: print(f"  seed {s} step {k}: loss {l:.4f} ({time.time() - t:.0f}s)", flush=True))
        p = R.stat_predictions(model, val, cpu)
        losses.append(ls.astype(np.float32)), preds.append(p.astype(np.uint8))
        top1.append(100.0 * float((p == labels).mean()))
        print(f"seed {s}: oracle top-1 {top1[-1]:.3f} % ({time.time() - t:.0f}s)", flush=True)
        np.savez_compressed(out, recipe=np.array(R.recipe_hash()), seeds=np.arange(s + 1), labels=labels.astype(np.uint8),
                            top1=np.array(top1), preds=np.stack(preds), losses=np.stack(losses),
                            threads=np.array(threads), torch_version=np.array(torch.__version__))
    print(f"wrote {out}: top-1 {np.round(top1, 2).tolist()}, mean {np.mean(top1):.2f}, sd {np.std(top1, ddof=1):.2f}")


if __name__ == "__main__":
    main()
