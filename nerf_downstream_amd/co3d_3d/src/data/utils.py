"""Batch collation (counterpart of the reference's co3d_3d/src/data/utils.py:25-50)."""
import numpy as np
import torch

from nerf_downstream_amd.minkowski import utils as me_utils


def _cat(tensors):
    """torch.cat; in a DataLoader worker the result is allocated in SHARED memory (as torch's default_collate does), so handing the
    batch to the trainer passes a file descriptor instead of copying its 29-105 MB a second time."""
    if torch.utils.data.get_worker_info() is None:
        return torch.cat(tensors)
    first = tensors[0]
    shape = (sum(int(t.shape[0]) for t in tensors),) + tuple(first.shape[1:])
    numel = int(np.prod(shape))
    storage = first._typed_storage()._new_shared(numel, device=first.device)
    out = first.new(storage).resize_(*shape)
    return torch.cat(tensors, out=out)


def collate_mink(list_data):
    """list of sample dicts -> {"coordinates": f32 [sumN,4] (batch,x,y,z), "features": f32 [sumN,C],
    "labels": int64 [B]}.  Runs in DataLoader workers: CPU only."""
    if "links" in list_data[0]:  # compact on-disk form (Co3DDatasetBase(compact=True)): decoded on the GPU later
        n = [int(d["links"].shape[0]) for d in list_data]
        package = {
            "links": _cat([d["links"] for d in list_data]),
            "density": _cat([d["density"] for d in list_data]),
            "sh_q": _cat([d["sh_q"] for d in list_data]),
            "scene_offsets": torch.tensor(np.concatenate([[0], np.cumsum(n)]), dtype=torch.int32),
            "sh_scale": torch.stack([d["sh_scale"] for d in list_data]),
            "sh_min": torch.stack([d["sh_min"] for d in list_data]),
            "labels": torch.from_numpy(np.concatenate([np.asarray(d["labels"]) for d in list_data])),
            "feature_names": list_data[0]["feature_names"],
        }
        resos = {tuple(d.get("reso", (128, 128, 128))) for d in list_data}
        if len(resos) != 1:
            raise ValueError(f"one batch mixes grid resolutions {sorted(resos)} (data.npz scenes are 128^3, last.ckpt scenes 256^3)")
        package["reso"] = resos.pop()
        return _with_programs(package, list_data, n)
    coords, feats = me_utils.sparse_collate(
        [d["coordinates"] for d in list_data], [d["features"] for d in list_data], dtype=torch.float32
    )
    package = {"coordinates": coords, "features": feats}
    for key in list_data[0].keys():
        if "label" in key or "instance" in key or "dists" in key:
            package[key] = torch.from_numpy(np.concatenate([np.asarray(d[key]) for d in list_data]))
    for key in ("metadata", "dataset", "colors"):
        if key in list_data[0]:
            package[key] = [d[key] for d in list_data]
    return _with_programs(package, list_data, [int(d["coordinates"].shape[0]) for d in list_data])


def _with_programs(package, list_data, n):
    """Augmentation programs drawn by the dataset (transforms.Compose.sample): one parameter row and one
    Philox stream id per scene, plus a seed for the batch; applied on the GPU in process_input."""
    if "aug_params" in list_data[0]:
        package["aug_params"] = torch.stack([d["aug_params"] for d in list_data])
        streams = np.array([d["aug_stream"] for d in list_data], dtype=np.uint32)
        package["aug_streams"] = torch.from_numpy(streams.view(np.int32).copy())
        package["aug_seed"] = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
        package.setdefault("scene_offsets", torch.tensor(np.concatenate([[0], np.cumsum(n)]), dtype=torch.int32))
        package.setdefault("feature_names", list_data[0].get("feature_names"))
    return package
