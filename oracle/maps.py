"""ctypes/numpy front-end of oracle/mink_maps.c (TEST INFRASTRUCTURE, see __init__).

Also holds pure-numpy brute-force versions (``*_bruteforce``) used by the tests to
pin the C restatement on small cases.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_maps.so")
_SRC = os.path.join(_HERE, "mink_maps.c")


def build(force=False):
    """gcc-compile the C restatement next to its source (oracle/liboracle_maps.so)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c99", "-Wall", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        i64, i32, p = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        L.orc_quantize.argtypes = [p, i64, p]
        L.orc_quantize.restype = None
        L.orc_unique.argtypes = [p, i64, p, p]
        L.orc_unique.restype = i64
        L.orc_stride_map.argtypes = [p, i64, i32, p, p]
        L.orc_stride_map.restype = i64
        L.orc_kernel_map.argtypes = [p, i64, p, i64, p, i32, p]
        L.orc_kernel_map.restype = ctypes.c_int
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP thread count of the C restatement's kernel-map search (bench.py times it at two thread counts)."""
    lib()  # liboracle_maps.so brings libgomp in
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def kernel_offsets(kernel_size, in_ts, dilation=1):
    """Kernel region offsets (A4): x fastest, z slowest; odd k centred, even k {0..k-1}.

    Witness for the index order: z-axis line = kernel indices [4,13,22]
    (reference co3d_3d/src/models/mink/modules/sparse_conv.py:375-379).
    Returns int32 [k^3, 3] in (dx,dy,dz) column order, scaled by dilation*in_ts.
    """
    k = int(kernel_size)
    r = np.arange(k) - (k - 1) // 2 if k % 2 == 1 else np.arange(k)
    dz, dy, dx = np.meshgrid(r, r, r, indexing="ij")  # x fastest after ravel
    off = np.stack([dx.ravel(), dy.ravel(), dz.ravel()], 1) * int(dilation) * int(in_ts)
    return np.ascontiguousarray(off, dtype=np.int32)


def quantize(fcoords):
    fcoords = np.ascontiguousarray(fcoords, dtype=np.float32)
    out = np.empty(fcoords.shape, dtype=np.int32)
    lib().orc_quantize(_ptr(fcoords), fcoords.shape[0], _ptr(out))
    return out


def unique(coords):
    """A2. -> (unique_index int32[nu], inverse int32[n])."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    n = coords.shape[0]
    ui = np.empty(n, dtype=np.int32)
    inv = np.empty(n, dtype=np.int32)
    nu = lib().orc_unique(_ptr(coords), n, _ptr(ui), _ptr(inv))
    if nu < 0:
        raise RuntimeError("oracle: coordinate outside the packable range")
    return ui[:nu].copy(), inv


def stride_map(coords, out_ts):
    """A3. -> (out_coords int32[no,4], in2out int32[n])."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    n = coords.shape[0]
    oc = np.empty((n, 4), dtype=np.int32)
    i2o = np.empty(n, dtype=np.int32)
    no = lib().orc_stride_map(_ptr(coords), n, int(out_ts), _ptr(oc), _ptr(i2o))
    if no < 0:
        raise RuntimeError("oracle: coordinate outside the packable range")
    return oc[:no].copy(), i2o


def kernel_map_table(in_coords, out_coords, offsets):
    """A4/A5 as a dense table nbr[n_out, K] (input row or -1)."""
    in_coords = np.ascontiguousarray(in_coords, dtype=np.int32)
    out_coords = np.ascontiguousarray(out_coords, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    K = offsets.shape[0]
    nbr = np.empty((out_coords.shape[0], K), dtype=np.int32)
    rc = lib().orc_kernel_map(
        _ptr(in_coords), in_coords.shape[0], _ptr(out_coords), out_coords.shape[0], _ptr(offsets), K, _ptr(nbr)
    )
    if rc:
        raise RuntimeError("oracle: kernel map failed")
    return nbr


def table_to_lists(nbr):
    """Dense table -> ME-format kernel map {k: int32[2,n]} (row 0 = in, row 1 = out),

    pairs ordered by output row (canonical order; ME's own in-list order is
    OpenMP-nondeterministic).  Empty lists are omitted like ME does.
    Format witness: reference sparse_conv.py:124-143.
    """
    out = {}
    for k in range(nbr.shape[1]):
        o = np.nonzero(nbr[:, k] >= 0)[0].astype(np.int32)
        if o.size:
            out[k] = np.stack([nbr[o, k], o], 0)
    return out


def transpose_table(nbr, n_in):
    """nbr_t[i, k] = the output row o with nbr[o, k] == i (or -1): the kernel map read from the input side, which the
    data gradient of a strided convolution gathers through.  Each (input row, offset) pair has at most one output."""
    nbr = np.asarray(nbr)
    nbr_t = np.full((int(n_in), nbr.shape[1]), -1, np.int32)
    for k in range(nbr.shape[1]):
        o = np.nonzero(nbr[:, k] >= 0)[0]
        assert np.unique(nbr[o, k]).size == o.size, "an (input row, offset) pair reaches two outputs"
        nbr_t[nbr[o, k], k] = o.astype(np.int32)
    return nbr_t


def class_partition(coords, ts, pad):
    """Row order used by the data gradient of stride-2 convolutions: rows grouped by the parity of coordinate / ts per
    axis (class = px | py << 1 | pz << 2), input order kept inside a class, every class segment padded with -1 to a
    multiple of `pad` rows; total length n + 8 (pad - 1), the tail -1.  Derived data of this design (no ME counterpart):
    all rows of one class reach their outputs through the same kernel offsets."""
    c = np.asarray(coords, dtype=np.int64)
    cls = ((c[:, 1] // ts) & 1) | (((c[:, 2] // ts) & 1) << 1) | (((c[:, 3] // ts) & 1) << 2)
    out = np.full(c.shape[0] + 8 * (pad - 1), -1, np.int32)
    pos = 0
    for k in range(8):
        rows = np.nonzero(cls == k)[0].astype(np.int32)
        out[pos : pos + rows.size] = rows
        pos += -(-rows.size // pad) * pad
    return out


# ----------------------------------------------------------------------------- brute force
def unique_bruteforce(coords):
    seen, ui, inv = {}, [], []
    for i, c in enumerate(map(tuple, np.asarray(coords).tolist())):
        if c not in seen:
            seen[c] = len(ui)
            ui.append(i)
        inv.append(seen[c])
    return np.array(ui, np.int32), np.array(inv, np.int32)


def stride_map_bruteforce(coords, out_ts):
    c = np.asarray(coords).astype(np.int64).copy()
    c[:, 1:] = np.floor(c[:, 1:] / float(out_ts)).astype(np.int64) * out_ts
    ui, inv = unique_bruteforce(c)
    return c[ui].astype(np.int32), inv


def kernel_map_bruteforce(in_coords, out_coords, offsets):
    table = {tuple(c): i for i, c in enumerate(np.asarray(in_coords).tolist())}
    nbr = np.full((len(out_coords), len(offsets)), -1, np.int32)
    for o, c in enumerate(np.asarray(out_coords).tolist()):
        for k, d in enumerate(np.asarray(offsets).tolist()):
            nbr[o, k] = table.get((c[0], c[1] + d[0], c[2] + d[1], c[3] + d[2]), -1)
    return nbr


def canonical_rows(coords):
    """Permutation that sorts rows lexicographically by (b,x,y,z)."""
    c = np.asarray(coords)
    return np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))
