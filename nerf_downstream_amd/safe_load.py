"""Loaders for files this project did not write -- PeRFception / Plenoxel checkpoints (`last.ckpt`), reference training
checkpoints, the ScanNet `scene_scales.data` pickle -- that execute nothing from the file.

The reference loads them with plain `torch.load` / `pickle.load` (co3d_3d/src/data/co3d.py:168-176, train.py:95-99,
src/data/scannet.py:520-521), i.e. arbitrary code on load.  Here:

* `load_checkpoint_file(path)`: `torch.load(..., weights_only=True)` -- tensors, containers and plain scalars only -- with
  exactly the numpy array / scalar / dtype reconstructors allow-listed (the Plenoxel checkpoints keep `sh_data_scale` /
  `sh_data_min` as numpy values); anything else in the file is an error, never a fallback to the unsafe loader.
* `load_plain_pickle(path)`: a `pickle.Unpickler` whose `find_class` refuses every global but the numpy scalar / dtype
  reconstructors: a pickle of dicts / lists / strings / numbers loads, one that names any callable does not.
"""
import io
import pickle

import numpy as np
import torch


def _numpy_safe_globals():
    try:
        from numpy._core import multiarray as ma  # numpy >= 2
    except ImportError:  # pragma: no cover
        from numpy.core import multiarray as ma
    allowed = [ma._reconstruct, ma.scalar, np.ndarray, np.dtype]
    for t in ("float16", "float32", "float64", "uint8", "int8", "int16", "int32", "int64", "uint16", "uint32", "uint64", "bool"):
        allowed.append(type(np.dtype(t)))  # numpy >= 1.25 pickles a dtype through its per-type class
    return allowed


def load_checkpoint_file(path, map_location="cpu"):
    """A checkpoint as a tree of tensors / numpy arrays / containers / scalars; raises `pickle.UnpicklingError` (from
    torch) for a file that would need anything else."""
    with torch.serialization.safe_globals(_numpy_safe_globals()):
        return torch.load(path, map_location=map_location, weights_only=True)


class _PlainUnpickler(pickle.Unpickler):
    _ALLOWED = None

    def find_class(self, module, name):
        if _PlainUnpickler._ALLOWED is None:
            _PlainUnpickler._ALLOWED = {(f.__module__, getattr(f, "__qualname__", getattr(f, "__name__", ""))): f for f in _numpy_safe_globals()}
            # the public aliases older numpy versions wrote
            for f in list(_PlainUnpickler._ALLOWED.values()):
                nm = getattr(f, "__qualname__", getattr(f, "__name__", ""))
                for mod in ("numpy.core.multiarray", "numpy._core.multiarray", "numpy"):
                    _PlainUnpickler._ALLOWED.setdefault((mod, nm), f)
        f = _PlainUnpickler._ALLOWED.get((module, name))
        if f is None:
            raise pickle.UnpicklingError(f"refusing to load global {module}.{name}: only plain data (dict / list / str / numbers / "
                                         "numpy values) is accepted from this file")
        return f


def load_plain_pickle(path_or_bytes):
    """dict / list / tuple / str / numbers (and numpy scalars / arrays) from a pickle; anything that names another global
    raises `pickle.UnpicklingError` before it is constructed."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return _PlainUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as f:
        return _PlainUnpickler(f).load()
