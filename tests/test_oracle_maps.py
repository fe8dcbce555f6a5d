"""Pins oracle/mink_maps.c (C restatement) against brute-force numpy/python sets."""
import os

import numpy as np
import pytest

from helpers import shell_scene


def _coords(seed, batch=2, negative=False):
    out = []
    for b in range(batch):
        xyz, _ = shell_scene(seed + b, grid=16, negative=negative)
        out.append(np.concatenate([np.full((len(xyz), 1), b), xyz], 1))
    return np.concatenate(out).astype(np.int32)


def test_kernel_offsets_order(oracle_maps):
    off = oracle_maps.kernel_offsets(3, 2)
    assert off.shape == (27, 3)
    # x fastest, z slowest; z-axis line = kernel indices 4, 13, 22 (sparse_conv.py:375-379)
    assert off[4].tolist() == [0, 0, -2] and off[13].tolist() == [0, 0, 0] and off[22].tolist() == [0, 0, 2]
    assert off[0].tolist() == [-2, -2, -2] and off[1].tolist() == [0, -2, -2] and off[3].tolist() == [-2, 0, -2]
    off2 = oracle_maps.kernel_offsets(2, 4)
    assert off2.tolist() == [[0, 0, 0], [4, 0, 0], [0, 4, 0], [4, 4, 0], [0, 0, 4], [4, 0, 4], [0, 4, 4], [4, 4, 4]]
    assert oracle_maps.kernel_offsets(1, 8).tolist() == [[0, 0, 0]]


def test_quantize_is_floor_not_trunc(oracle_maps):
    f = np.array([[0, -0.5, 1.5, -1.0], [1, 2.999, -2.001, 0.0]], np.float32)
    assert oracle_maps.quantize(f).tolist() == [[0, -1, 1, -1], [1, 2, -3, 0]]


@pytest.mark.parametrize("negative", [False, True])
def test_unique_first_occurrence(oracle_maps, negative):
    c = _coords(0, negative=negative)
    rng = np.random.default_rng(1)
    dup = np.concatenate([c, c[rng.integers(0, len(c), 200)]])
    dup = dup[rng.permutation(len(dup))]
    ui, inv = oracle_maps.unique(dup)
    ui_b, inv_b = oracle_maps.unique_bruteforce(dup)
    assert np.array_equal(ui, ui_b) and np.array_equal(inv, inv_b)
    assert np.all(np.diff(ui) > 0)  # first-occurrence order
    assert np.array_equal(dup[ui][inv], dup)


def test_unique_empty_and_range(oracle_maps):
    ui, inv = oracle_maps.unique(np.zeros((0, 4), np.int32))
    assert ui.size == 0 and inv.size == 0
    with pytest.raises(RuntimeError):
        oracle_maps.unique(np.array([[0, 40000, 0, 0]], np.int32))


@pytest.mark.parametrize("negative", [False, True])
@pytest.mark.parametrize("ts", [2, 4, 8])
def test_stride_map(oracle_maps, negative, ts):
    c = _coords(3, negative=negative)
    if ts > 2:
        c, _ = oracle_maps.stride_map(c, ts // 2)
    oc, i2o = oracle_maps.stride_map(c, ts)
    oc_b, i2o_b = oracle_maps.stride_map_bruteforce(c, ts)
    assert np.array_equal(oc, oc_b) and np.array_equal(i2o, i2o_b)
    assert np.all(oc[:, 1:] % ts == 0)
    assert len(np.unique(oc, axis=0)) == len(oc)
    # floor semantics for negatives
    assert np.all(oc[i2o][:, 1:] <= c[:, 1:]) and np.all(c[:, 1:] - oc[i2o][:, 1:] < ts)


@pytest.mark.parametrize("ksize,stride", [(3, 1), (3, 2), (1, 2), (2, 2)])
def test_kernel_map(oracle_maps, ksize, stride):
    cin = _coords(5, negative=True)
    cout = cin if stride == 1 else oracle_maps.stride_map(cin, stride)[0]
    off = oracle_maps.kernel_offsets(ksize, 1)
    nbr = oracle_maps.kernel_map_table(cin, cout, off)
    assert np.array_equal(nbr, oracle_maps.kernel_map_bruteforce(cin, cout, off))
    lists = oracle_maps.table_to_lists(nbr)
    for k, io in lists.items():
        assert np.array_equal(cin[io[0]][:, 1:], cout[io[1]][:, 1:] + off[k])
        assert np.all(np.diff(io[1]) > 0)
    if ksize == 2:  # pooling region == stride map (each input row has exactly one parent)
        i2o = oracle_maps.stride_map(cin, 2)[1]
        got = np.full(len(cin), -1)
        for k, io in lists.items():
            got[io[0]] = io[1]
        assert np.array_equal(got, i2o)
    if ksize == 3 and stride == 1:  # centre offset is the identity, map is symmetric
        assert np.array_equal(nbr[:, 13], np.arange(len(cin)))
        for k in range(27):
            v = nbr[:, k] >= 0
            assert np.array_equal(nbr[nbr[v, k], 26 - k], np.nonzero(v)[0])


def test_c_restatement_is_clean_under_asan_ubsan(tmp_path):
    """oracle/mink_maps.c built with -fsanitize=address,undefined (CPU build only: GPU sanitizers are not available
    on this pool) and driven over duplicates, negatives, range corners, out-of-range rows and empty inputs."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "orc_san"
    cc = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp", "-std=c99", "-Wall",
          "-Werror", "-o", str(exe), os.path.join(root, "oracle", "mink_maps.c"), os.path.join(root, "oracle", "sanitize_driver.c"), "-lm"]
    subprocess.check_call(cc)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="4"))
    assert r.returncode == 0 and "sanitize_driver ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("negative", [False, True])
def test_derived_maps_transpose_and_class_partition(oracle_maps, negative):
    """The two derived structures the data gradient of a strided convolution reads (no ME counterpart): pinned by their
    definitions, row by row, on a scene with negative coordinates as well (floor, not truncation, decides the parity)."""
    c1 = _coords(5, negative=negative)
    c2, _ = oracle_maps.stride_map(c1, 2)
    nbr = oracle_maps.kernel_map_table(c1, c2, oracle_maps.kernel_offsets(3, 1))
    nbr_t = oracle_maps.transpose_table(nbr, len(c1))
    assert nbr_t.shape == (len(c1), 27)
    pairs = {(int(nbr[o, k]), k): o for o in range(len(c2)) for k in range(27) if nbr[o, k] >= 0}
    assert (nbr_t >= 0).sum() == len(pairs)
    for (i, k), o in pairs.items():
        assert nbr_t[i, k] == o
    for ts, c in ((1, c1), (2, c2)):
        pad = 16
        perm = oracle_maps.class_partition(c, ts, pad)
        assert perm.shape == (len(c) + 8 * (pad - 1),)
        rows = perm[perm >= 0]
        assert sorted(rows.tolist()) == list(range(len(c)))  # a permutation of the rows
        cls = [(int(np.floor(r[1] / ts)) & 1) | ((int(np.floor(r[2] / ts)) & 1) << 1) | ((int(np.floor(r[3] / ts)) & 1) << 2) for r in c]
        pos = 0
        for k in range(8):
            want = [i for i in range(len(c)) if cls[i] == k]  # input order kept inside a class
            seg = perm[pos : pos + -(-len(want) // pad) * pad]
            assert seg[: len(want)].tolist() == want and np.all(seg[len(want):] == -1)
            assert pos % pad == 0
            pos += len(seg)
        assert np.all(perm[pos:] == -1)
