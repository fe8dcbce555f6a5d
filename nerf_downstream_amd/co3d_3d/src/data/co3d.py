"""PeRFception-CO3D plenoxel dataset (counterpart of the reference's
co3d_3d/src/data/co3d.py:70-268; on-disk format documented by scripts/preprocess.py:30-57).

Scene directory `<data_root>/plenoxel_co3d_<scene>/data.npz` holds
    links   int32 [N]     flat index into the 128^3 grid (x*128*128 + y*128 + z)
    density f32   [N,1]
    sh      uint8 [N,27]  de-quantised as sh * sh_scale + sh_min
`filelist/<phase>.txt` (relative to the CWD, like the reference :100) lists "<class> <scene>".
Sample dict / feature selection follow co3d.py:185-242.  Augmentations (`train_transformations`, names of
classes in transforms.py as in configs/co3d_aug3.gin) are only DRAWN here: the sample carries its
parameter row (`aug_params`, `aug_stream`) and the batch is transformed on the GPU (transforms.py)."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from nerf_downstream_amd import gin_lite as gin
from nerf_downstream_amd.safe_load import load_checkpoint_file

from . import transforms

CLASSES = (
    "apple backpack ball banana baseballbat baseballglove bench bicycle book bottle bowl broccoli cake car carrot "
    "cellphone chair couch cup donut frisbee hairdryer handbag hotdog hydrant keyboard kite laptop microwave "
    "motorcycle mouse orange parkingmeter pizza plant remote sandwich skateboard stopsign suitcase teddybear toaster "
    "toilet toybus toyplane toytrain toytruck tv umbrella vase wineglass"
).split()
assert len(CLASSES) == 51


_NPY_HEADER = None


def npz_members(path):
    """({name: (byte offset of the array data in `path`, dtype, shape)}, [names that are not plain stored .npy members]).
    The reference writes its scenes with `np.savez` (scripts/preprocess.py:49): STORED zip members, each a .npy file, so an array
    can be read -- or mapped -- where it lies in the file; the header is parsed with one regular expression."""
    import re
    import struct
    import zipfile

    global _NPY_HEADER
    if _NPY_HEADER is None:
        _NPY_HEADER = re.compile(rb"\{'descr': '([<>|=][a-zA-Z]\d+)', 'fortran_order': (False|True), 'shape': \(([\d, ]*)\), \}")
    out, slow = {}, []
    with zipfile.ZipFile(path) as zf, open(path, "rb") as f:
        for info in zf.infolist():
            name = info.filename[:-4] if info.filename.endswith(".npy") else info.filename
            if info.compress_type != zipfile.ZIP_STORED:
                slow.append(name)
                continue
            f.seek(info.header_offset)
            local = f.read(30)
            if local[:4] != b"PK\x03\x04":
                slow.append(name)
                continue
            n_name, n_extra = struct.unpack("<HH", local[26:30])
            start = info.header_offset + 30 + n_name + n_extra
            f.seek(start)
            head = f.read(12)
            if head[:6] != b"\x93NUMPY":
                slow.append(name)
                continue
            hlen, skip = (struct.unpack("<H", head[8:10])[0], 10) if head[6] == 1 else (struct.unpack("<I", head[8:12])[0], 12)
            f.seek(start + skip)
            m = _NPY_HEADER.match(f.read(hlen).strip())
            if m is None or m.group(2) == b"True":
                slow.append(name)
                continue
            shape = tuple(int(v) for v in m.group(3).replace(b" ", b"").split(b",") if v)
            out[name] = (start + skip + hlen, np.dtype(m.group(1).decode()), shape)
    return out, slow


def _read_npz(path):
    """{name: array} of a `data.npz`.  `np.load` spends ~1 ms per member parsing the header through `ast.literal_eval` and copies
    (and CRC-checks) every byte -- 4 ms per scene, 64 ms per batch of 16 in a DataLoader worker, which is the trainer's bound at a
    3.4 ms step.  Stored members are instead MAPPED where they lie in the file (`npz_members`; no copy: the bytes are read by whoever
    concatenates the batch); anything else -- a compressed member, an unusual header -- goes through `np.load`."""
    members, slow = npz_members(path)
    out = {}
    for name, (off, dtype, shape) in members.items():
        count = int(np.prod(shape)) if shape else 1
        if count == 0:
            out[name] = np.zeros(shape, dtype)
        elif count * dtype.itemsize < 4096:  # (scalars, the two SH constants: read, not mapped)
            with open(path, "rb") as f:
                f.seek(off)
                out[name] = np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype).reshape(shape).copy()
        else:
            out[name] = np.memmap(path, dtype=dtype, mode="c", offset=off, shape=shape)  # (copy-on-write: private, never written back)
    if slow:
        z = np.load(path)
        for name in slow:
            out[name] = z[name]
    return out


def links_to_coordinates(links, reso):
    """flat grid index -> float (x,y,z) voxel coordinates (co3d.py:196-203)."""
    links = torch.as_tensor(links).long()
    yz = reso[1] * reso[2]
    return torch.stack([links // yz, (links % yz) // reso[2], links % reso[2]], 1).float()


def select_features(coordinates, density, sh, names):
    """raw columns [xyzs | density | sh] selected by name (co3d.py:205-229)."""
    xyzs = coordinates - coordinates.mean(dim=1, keepdim=True)  # per-point mean, as the reference (:211)
    xyzs = xyzs / torch.linalg.norm(xyzs, dim=1).max()
    cols = {"xyzs": xyzs, "density": density, "sh": sh, "ones": torch.ones_like(density)}
    return torch.cat([cols[n] for n in names], dim=1).float(), xyzs


@gin.configurable()
class Co3DDatasetBase(Dataset):
    def __init__(self, phase, data_root="co3d_3d/datasets/co3d", train_transformations=(), eval_transformations=(),
                 downsample_mode=1, downsample_stride=2, num_points=-1, features=("sh",), filelist_dir="filelist",
                 compact=False):
        """`compact` (extension): samples stay in the on-disk form (links / density / uint8 sh + scale, min:
        35 bytes per voxel instead of 128) and are decoded on the GPU by `mink_decode_plenoxel` when the
        batch reaches the model (`MinkowskiBaseModel.process_input`), "xyzs" (a per-scene reduction) included."""
        phase = "test" if phase in ("val", "test") else "train"  # reference :84 (val == test list)
        names = list(train_transformations if phase == "train" else eval_transformations)
        unknown = [t for t in names if not hasattr(getattr(transforms, t, None), "draw")]
        if unknown:
            raise NotImplementedError(f"augmentations {unknown} have no GPU counterpart (supported: the classes of "
                                      "data/transforms.py)")
        self.transformations = transforms.Compose([getattr(transforms, t)() for t in names]) if names else None
        self.phase, self.data_root, self.features = phase, data_root, list(features)
        self.compact = bool(compact)
        with open(os.path.join(filelist_dir, f"{phase}.txt")) as f:
            self.files = [line.split()[:2] for line in f if line.strip()]
        self.CLASS_LABELS, self.NUM_CLASSES = CLASSES, len(CLASSES)

    def _load_raw(self, inst_id):
        """The scene in its on-disk form: (links int32 [N], density f32 [N], sh uint8 [N,27], sh_scale, sh_min, reso).
        Two formats, as in the reference (co3d.py:133-176): the pre-processed `data.npz` (128^3 grid,
        scripts/preprocess.py:30-57) and, failing that, the Plenoxel training checkpoint `last.ckpt` (256^3 grid;
        `state_dict["model.links_idx" / "model.density_data" / "model.sh_data"]`, top-level `model.sh_data_min` /
        `model.sh_data_scale`)."""
        scene = os.path.join(self.data_root, f"plenoxel_co3d_{inst_id}")
        numpy_file, torch_file = os.path.join(scene, "data.npz"), os.path.join(scene, "last.ckpt")
        if os.path.exists(numpy_file):
            z = _read_npz(numpy_file)
            return (z["links"].astype(np.int32, copy=False), z["density"].astype(np.float32, copy=False).reshape(-1),
                    np.ascontiguousarray(z["sh"]), z["sh_scale"], z["sh_min"], [128, 128, 128])
        if os.path.exists(torch_file):
            ck = load_checkpoint_file(torch_file)  # tensors / numpy values only: nothing from the file is executed
            sd = ck["state_dict"]
            as_np = lambda t: t.numpy() if torch.is_tensor(t) else np.asarray(t)  # noqa: E731
            return (as_np(sd["model.links_idx"]).astype(np.int32), as_np(sd["model.density_data"]).astype(np.float32).reshape(-1),
                    np.ascontiguousarray(as_np(sd["model.sh_data"])), as_np(ck["model.sh_data_scale"]), as_np(ck["model.sh_data_min"]),
                    [256, 256, 256])
        raise ValueError(f"{inst_id} not exist in {self.data_root}")

    def load_data(self, inst_id):
        links, density, sh_q, scale, mn, reso = self._load_raw(inst_id)
        sh = sh_q.reshape(len(links), -1).astype(np.float32) * np.asarray(scale, np.float32) + np.asarray(mn, np.float32)
        return torch.from_numpy(links), torch.from_numpy(density), torch.from_numpy(sh.astype(np.float32)), reso

    def load_compact(self, inst_id):
        links, density, sh_q, scale, mn, reso = self._load_raw(inst_id)
        bc = lambda a: np.broadcast_to(np.asarray(a, np.float32).reshape(-1), (27,)).copy()  # noqa: E731
        return {"links": torch.from_numpy(links), "density": torch.from_numpy(density),
                "sh_q": torch.from_numpy(sh_q.reshape(len(links), sh_q.size // len(links) if len(links) else 27)), "sh_scale": torch.from_numpy(bc(scale)),
                "sh_min": torch.from_numpy(bc(mn)), "reso": tuple(reso)}

    def __getitem__(self, index):
        label, inst_id = self.files[index]
        if self.compact:
            sample = self.load_compact(inst_id)
            keep = self.transformations.row_mask(sample["density"].numpy()) if self.transformations is not None else None
            if keep is not None:  # per-scene row filters of the recipe (DensityBasedSample), on the on-disk form
                k = torch.from_numpy(keep)
                sample = dict(sample, links=sample["links"][k], density=sample["density"][k], sh_q=sample["sh_q"][k])
            sample["labels"] = np.array([self.CLASS_LABELS.index(label)])
            sample["feature_names"] = tuple(self.features)
            return self._with_program(sample)
        links, density, sh, reso = self.load_data(inst_id)
        keep = self.transformations.row_mask(density.numpy()) if self.transformations is not None else None
        if keep is not None:
            # (the reference filters after it has normalised xyzs over the whole scene, co3d.py:209-219; here the filter
            #  comes first, so "xyzs" is normalised over the kept voxels -- the recipes that bind it do not select xyzs)
            k = torch.from_numpy(keep)
            links, density, sh = links[k], density[k], sh[k]
        coordinates = links_to_coordinates(links, reso)
        feats, xyzs = select_features(coordinates, density.reshape(-1, 1), sh.reshape(len(links), -1), self.features)
        return self._with_program({"coordinates": coordinates, "features": feats, "xyzs": xyzs,
                                   "labels": np.array([self.CLASS_LABELS.index(label)]),
                                   "feature_names": tuple(self.features)})

    def sample_lengths(self):
        """Voxel count of every scene, read from the `links` array header inside each data.npz (no array data is
        loaded): what the length-aware data-parallel sampler balances (data_module.py).  None if a scene is missing."""
        import zipfile

        if getattr(self, "_lengths", None) is None:
            out = np.zeros(len(self.files), dtype=np.int64)
            for i, (_, inst_id) in enumerate(self.files):
                path = os.path.join(self.data_root, f"plenoxel_co3d_{inst_id}", "data.npz")
                try:
                    with zipfile.ZipFile(path) as z, z.open("links.npy") as f:
                        version = np.lib.format.read_magic(f)
                        shape = (np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0)(f)[0]
                    out[i] = int(shape[0])
                except (OSError, KeyError, ValueError):
                    return None
            self._lengths = out
        return self._lengths

    def _with_program(self, sample):
        if self.transformations is not None:  # drawn here (DataLoader worker), applied on the GPU
            params, stream = self.transformations.sample()
            sample["aug_params"], sample["aug_stream"] = torch.from_numpy(params), stream
        return sample

    def __len__(self):
        return len(self.files)


class Co3DDataset(Co3DDatasetBase):
    pass


class Co3D10pDataset(Co3DDatasetBase):
    pass
