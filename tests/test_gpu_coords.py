"""-m gpu: HIP coordinate maps vs the C oracle, BIT-EXACT (integer work)."""
import numpy as np
import pytest
import torch

from helpers import batch_scenes

pytestmark = pytest.mark.gpu


def _mgr(coords_f):
    from nerf_downstream_amd import minkowski as ME

    tf = ME.TensorField(coordinates=coords_f.cuda(), features=torch.zeros(coords_f.shape[0], 4).cuda())
    return ME, tf


def _noisy_field(seed, grid, negative, dup):
    coords, _ = batch_scenes([seed, seed + 1, seed + 2], grid=grid, cin=1, negative=negative)
    rng = np.random.default_rng(seed)
    if dup:  # float jitter inside the voxel + repeated rows -> duplicates after flooring
        extra = coords[rng.integers(0, len(coords), len(coords) // 3)]
        coords = torch.cat([coords, extra])
        coords = coords[torch.from_numpy(np.sort(rng.permutation(len(coords))))]
        order = torch.argsort(coords[:, 0], stable=True)
        coords = coords[order]
        coords[:, 1:] += torch.from_numpy(rng.uniform(0, 0.999, (len(coords), 3)).astype(np.float32))
    return coords


@pytest.mark.parametrize("grid,negative,dup", [(16, False, False), (24, True, True), (48, True, False)])
def test_unique_stride_kernel_maps(oracle_maps, grid, negative, dup):
    coords = _noisy_field(7, grid, negative, dup)
    ME, tf = _mgr(coords)
    m = tf.coordinate_manager
    q = oracle_maps.quantize(coords.numpy())
    ui, inv = oracle_maps.unique(q)
    assert np.array_equal(m.field_unique_index.cpu().numpy(), ui)
    assert np.array_equal(m.field_inverse.cpu().numpy(), inv)
    c_ref = {1: q[ui]}
    assert np.array_equal(m.levels[1].coords.cpu().numpy(), c_ref[1])
    key = ME.CoordinateMapKey(1)
    keys = {1: key}
    for ts in (2, 4, 8):
        keys[ts] = m.stride(keys[ts // 2], 2)
        oc, i2o = oracle_maps.stride_map(c_ref[ts // 2], ts)
        c_ref[ts] = oc
        assert np.array_equal(m.levels[ts].coords.cpu().numpy(), oc)
        assert np.array_equal(m.stride_map(keys[ts // 2], keys[ts]).cpu().numpy(), i2o)
    for ts_in, ts_out, ks in [(1, 1, 3), (1, 2, 2), (2, 4, 3), (2, 4, 1), (4, 4, 3), (4, 8, 3), (8, 8, 3)]:
        off = oracle_maps.kernel_offsets(ks, ts_in)
        ref = oracle_maps.kernel_map_table(c_ref[ts_in], c_ref[ts_out], off)
        nbr, nbr_t = m.kernel_table(keys[ts_in], keys[ts_out], ks, 1, transposed=True)
        assert np.array_equal(nbr.cpu().numpy(), ref)
        ref_t = np.full((len(c_ref[ts_in]), off.shape[0]), -1, np.int32)
        o, k = np.nonzero(ref >= 0)
        ref_t[ref[o, k], k] = o
        assert np.array_equal(nbr_t.cpu().numpy(), ref_t)
        # ME-format rulebook (ballot / prefix-sum compaction), canonical order
        km = m.kernel_map(keys[ts_in], keys[ts_out], kernel_size=ks)
        ref_lists = oracle_maps.table_to_lists(ref)
        assert sorted(km.keys()) == sorted(ref_lists.keys())
        for kk, v in km.items():
            assert v.dtype == torch.int32 and np.array_equal(v.cpu().numpy(), ref_lists[kk])
    boff = m.batch_offsets(keys[4]).cpu().numpy()
    assert np.array_equal(boff, np.searchsorted(c_ref[4][:, 0], np.arange(4)))


@pytest.mark.parametrize("case", ["ascending", "one_duplicate", "one_swap", "last_pair_swapped", "single_row", "float_jitter"])
def test_ascending_rows_skip_the_level_0_insert_and_nothing_else_does(oracle_maps, case):
    """mink_coords_build_levels compares every row's key with the row before it: strictly ascending rows (a voxel grid's
    `links` order) are their own unique rows and level 0 runs without its hash insert (the level's map stays empty until
    `hash_map()` asks for it); ONE pair out of order -- a duplicate, a swap, anywhere -- and level 0 goes through the hash map.
    Either way: unique rows, inverse map, the strided levels and a table equal the C oracle's, bit for bit."""
    from nerf_downstream_amd import minkowski as ME

    coords, _ = batch_scenes([11, 12], grid=40, cin=1, negative=True)
    n = len(coords)
    expect_fast = case in ("ascending", "single_row", "float_jitter")
    if case == "one_duplicate":
        coords = torch.cat([coords[: n // 2 + 1], coords[n // 2 :]])  # row n//2 twice, adjacent: ascending but not strictly
    elif case == "one_swap":
        coords[[n // 3, n // 3 + 1]] = coords[[n // 3 + 1, n // 3]]
    elif case == "last_pair_swapped":
        coords[[n - 2, n - 1]] = coords[[n - 1, n - 2]]
    elif case == "single_row":
        coords = coords[:1]
    elif case == "float_jitter":  # (jitter inside the voxel: the floored keys are still ascending)
        coords[:, 1:] += torch.from_numpy(np.random.default_rng(0).uniform(0, 0.999, (n, 3)).astype(np.float32))
    tf = ME.TensorField(coordinates=coords.cuda(), features=torch.zeros(len(coords), 4).cuda())
    m = tf.coordinate_manager
    assert m.levels[1].hash_empty == expect_fast, case
    q = oracle_maps.quantize(coords.numpy())
    ui, inv = oracle_maps.unique(q)
    assert np.array_equal(m.field_unique_index.cpu().numpy(), ui)
    assert np.array_equal(m.field_inverse.cpu().numpy(), inv)
    c1 = q[ui]
    assert np.array_equal(m.levels[1].coords.cpu().numpy(), c1)
    k1 = ME.CoordinateMapKey(1)
    k2 = m.stride(k1, 2)
    c2, i2o = oracle_maps.stride_map(c1, 2)
    assert np.array_equal(m.levels[2].coords.cpu().numpy(), c2)
    assert np.array_equal(m.stride_map(k1, k2).cpu().numpy(), i2o)
    nbr, _ = m.kernel_table(k1, k2, 3, 1)
    assert np.array_equal(nbr.cpu().numpy(), oracle_maps.kernel_map_table(c1, c2, oracle_maps.kernel_offsets(3, 1)))
    tkeys, tvals, cap = m.hash_map(1)  # filled on demand after the shortcut: every row is found under its own id
    assert not m.levels[1].hash_empty
    from nerf_downstream_amd._lib import check, lib

    off = oracle_maps.kernel_offsets(1, 1)
    own = torch.empty(len(c1), 1, dtype=torch.int32, device="cuda")
    check(lib().mink_kernel_map(tkeys.data_ptr(), tvals.data_ptr(), cap, m.levels[1].coords.data_ptr(), len(c1),
                                np.ascontiguousarray(off, np.int32).ctypes.data, 1, own.data_ptr(), None, None))
    assert torch.equal(own[:, 0].cpu(), torch.arange(len(c1), dtype=torch.int32))


def test_block_index_capacity_is_the_block_count_and_a_short_table_is_reported(oracle_maps, monkeypatch):
    """The block index of the map at tensor stride ts is sized for the rows of the map at 4 ts (its blocks, exactly) when the
    pyramid holds that level; tables through it equal the oracle's.  A table with too few slots (a caller's error at the C ABI)
    does not hang: probing is bounded, the build reports it through `blk_counter`."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import lib

    coords, _ = batch_scenes([21, 22], grid=48, cin=1)
    q = oracle_maps.quantize(coords.numpy())
    c1 = q[oracle_maps.unique(q)[0]]
    off = oracle_maps.kernel_offsets(3, 1)
    ref = oracle_maps.kernel_map_table(c1, c1, off)

    def field():
        tf = ME.TensorField(coordinates=coords.cuda(), features=torch.zeros(len(coords), 4).cuda())
        return tf.coordinate_manager

    m = field()
    k1 = ME.CoordinateMapKey(1)
    m.stride(m.stride(k1, 2), 2)  # the level at 4 ts: the block count
    nbr, _ = m.kernel_table(k1, k1, 3)
    assert np.array_equal(nbr.cpu().numpy(), ref) and m.block_index_ok()
    assert m._blk_pool.numel() < 6 * len(c1)  # (sized for the blocks: the capacity for the rows alone took 10 words per row)
    m = field()  # no coarser level: the capacity for the rows
    nbr, _ = m.kernel_table(k1, k1, 3)
    assert np.array_equal(nbr.cpu().numpy(), ref) and m.block_index_ok()
    m = field()
    L = lib()
    monkeypatch.setattr(L, "mink_table_capacity", lambda n: 64)  # far fewer slots than blocks
    nbr, _ = m.kernel_table(k1, k1, 3)
    torch.cuda.synchronize()
    monkeypatch.undo()
    # the violated promise is LOUD: the next batch's table build of this process raises (mink_set_overflow_sink: a pinned host
    # word the insert kernel sets, read without a synchronisation), once
    from nerf_downstream_amd.minkowski.coords import check_block_index_overflow

    m2 = field()
    with pytest.raises(RuntimeError, match="fewer slots"):
        m2.kernel_table(k1, k1, 3)
    check_block_index_overflow()  # (reported once: the word is cleared)
    nbr2, _ = m2.kernel_table(k1, k1, 3)
    assert np.array_equal(nbr2.cpu().numpy(), ref) and m2.block_index_ok()
    assert not m.block_index_ok()
    got = nbr.cpu().numpy()
    assert got.min() >= -1 and got.max() < len(c1)  # rows are missing, nothing points outside the map
    check_block_index_overflow()


def test_full_size_properties():
    """BASELINE-size grid (128^3 shell, ~50k voxels x 4 samples): size-independent properties."""
    from nerf_downstream_amd import minkowski as ME

    coords, _ = batch_scenes([1, 2, 3, 4], grid=128, cin=1)
    tf = ME.TensorField(coordinates=coords.cuda(), features=torch.zeros(len(coords), 4).cuda())
    m = tf.coordinate_manager
    k1 = ME.CoordinateMapKey(1)
    assert m.levels[1].n == len(coords)  # integer coords: no duplicates
    nbr, _ = m.kernel_table(k1, k1, 3)
    n = nbr.shape[0]
    ar = torch.arange(n, device="cuda", dtype=torch.int32)
    assert torch.equal(nbr[:, 13], ar)  # centre offset = identity
    for k in range(13):  # symmetry: nbr[nbr[o,k], 26-k] == o
        v = nbr[:, k] >= 0
        assert torch.equal(nbr[nbr[v, k].long(), 26 - k], ar[v])
    k2 = m.stride(k1, 2)
    i2o = m.stride_map(k1, k2).long()
    c1, c2 = m.levels[1].coords, m.levels[2].coords
    assert torch.equal(c2[i2o][:, 0], c1[:, 0])
    assert torch.equal(c2[i2o][:, 1:], torch.div(c1[:, 1:], 2, rounding_mode="floor") * 2)
    assert torch.unique(c2, dim=0).shape[0] == c2.shape[0]
    ch, _ = m.kernel_table(k1, k2, 2)  # children table: every input row appears exactly once
    flat = ch[ch >= 0]
    assert flat.numel() == n and torch.equal(torch.sort(flat).values, ar)


def test_errors():
    from nerf_downstream_amd import minkowski as ME

    bad = torch.tensor([[0.0, 40000.0, 0.0, 0.0]])
    with pytest.raises(ValueError):
        ME.TensorField(coordinates=bad.cuda(), features=torch.zeros(1, 4).cuda())
    unsorted = torch.tensor([[1.0, 0, 0, 0], [0.0, 1, 1, 1]])
    tf = ME.TensorField(coordinates=unsorted.cuda(), features=torch.zeros(2, 4).cuda())
    with pytest.raises(ValueError):
        tf.coordinate_manager.batch_size()
    with pytest.raises(RuntimeError):
        ME.TensorField(coordinates=unsorted, features=torch.zeros(2, 4))


@pytest.mark.parametrize("features", [("density", "sh"), ("sh",), ("sh", "ones", "density"), ("xyzs", "density", "sh"), ("xyzs",)])
def test_decode_plenoxel_batch_matches_oracle(features):
    """GPU-side decode of a compact PeRFception batch == the CPU restatement of the
    reference loader, bit for bit (integer coordinates; float32 multiply-then-add de-quantisation),
    and a model fed the compact batch sees exactly the field it sees from the decoded tensors."""
    from nerf_downstream_amd import minkowski as ME
    from oracle.decode import decode_batch

    rng = np.random.default_rng(7)
    scenes = []
    for j in range(3):
        n = 4000 + 333 * j
        scenes.append({
            "links": np.sort(rng.choice(128 ** 3, n, replace=False)).astype(np.int32),
            "density": rng.random(n).astype(np.float32) * 10,
            "sh_q": rng.integers(0, 256, (n, 27)).astype(np.uint8),
            "sh_scale": (rng.random(27) * 0.02 + 1e-3).astype(np.float32),
            "sh_min": rng.standard_normal(27).astype(np.float32),
        })
    ns = [len(s["links"]) for s in scenes]
    batch = {
        "links": torch.from_numpy(np.concatenate([s["links"] for s in scenes])).cuda(),
        "density": torch.from_numpy(np.concatenate([s["density"] for s in scenes])).cuda(),
        "sh_q": torch.from_numpy(np.concatenate([s["sh_q"] for s in scenes])).cuda(),
        "scene_offsets": torch.tensor(np.concatenate([[0], np.cumsum(ns)]), dtype=torch.int32).cuda(),
        "sh_scale": torch.from_numpy(np.stack([s["sh_scale"] for s in scenes])).cuda(),
        "sh_min": torch.from_numpy(np.stack([s["sh_min"] for s in scenes])).cuda(),
        "feature_names": features,
    }
    coords, feats = ME.utils.decode_plenoxel_batch(batch)
    ocoords, ofeats = decode_batch(scenes, features=features)
    assert np.array_equal(coords.cpu().numpy(), ocoords)
    # de-quantised SH / density / ones: bit for bit.  "xyzs" (a chain of float32 divisions and a square root): within one
    # unit in the last place of the oracle's numpy float32 evaluation
    xcols = np.zeros(feats.shape[1], bool)
    if "xyzs" in features:
        x0 = sum({"xyzs": 3, "density": 1, "sh": 27, "ones": 1}[f] for f in features[: features.index("xyzs")])
        xcols[x0 : x0 + 3] = True
    got = feats.cpu().numpy()
    assert np.array_equal(got[:, ~xcols], ofeats[:, ~xcols])
    if xcols.any():
        assert np.abs(got[:, xcols] - ofeats[:, xcols]).max() <= 1.2e-7 and np.abs(got[:, xcols]).max() <= 1.0 + 1e-6
    # `last.ckpt` scenes live on a 256^3 grid (reference co3d.py:152): the batch carries its resolution
    links256 = [np.sort(rng.choice(256 ** 3, len(s["links"]), replace=False)).astype(np.int32) for s in scenes]
    b256 = dict(batch, links=torch.from_numpy(np.concatenate(links256)).cuda(), reso=(256, 256, 256))
    c256, f256 = ME.utils.decode_plenoxel_batch(b256)
    oc256, _ = decode_batch([dict(s, links=l) for s, l in zip(scenes, links256)], features=features, reso=(256, 256, 256))
    assert np.array_equal(c256.cpu().numpy(), oc256) and int(c256[:, 1:].max()) > 127
    assert torch.equal(f256[:, torch.from_numpy(~xcols)], feats[:, torch.from_numpy(~xcols)])  # (xyzs follows the coordinates)
    if features == ("density", "sh"):
        from nerf_downstream_amd.co3d_3d.src.models import get_model

        torch.manual_seed(0)
        net = get_model("ResNet14", 28, 7).cuda().eval()
        with torch.no_grad():
            a = net(net.process_input(batch))
            b = net(net.process_input({"coordinates": coords, "features": feats}))
            c = net(net.process_input({"coordinates": coords.float(), "features": feats}))  # the reference's float field
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("grid,negative,dup", [(16, False, False), (24, True, True), (48, True, False), (128, False, False)])
def test_block_index_tables_match_oracle(oracle_maps, grid, negative, dup):
    """The batched table builder looks neighbours up through the 4^3-cell block index (MinkKernelMapDesc.blk_*) instead
    of the per-voxel hash map: every table of a ResNet pass, transposed ones included, bit-exact against the C oracle
    -- negative coordinates (blocks must floor, not truncate), duplicates and the full 128^3 shape included."""
    from nerf_downstream_amd import minkowski as ME

    coords = _noisy_field(9, grid, negative, dup)
    ME_, tf = _mgr(coords)
    m = tf.coordinate_manager
    q = oracle_maps.quantize(coords.numpy())
    ui, _ = oracle_maps.unique(q)
    c_ref = {1: q[ui]}
    keys = {1: ME.CoordinateMapKey(1)}
    for ts in (2, 4, 8, 16):
        keys[ts] = m.stride(keys[ts // 2], 2)
        c_ref[ts], _ = oracle_maps.stride_map(c_ref[ts // 2], ts)
    ops = [("ktable", 1, 1, 3, 1, False), ("ktable", 1, 2, 2, 1, False), ("ktable", 2, 4, 3, 1, True), ("ktable", 2, 4, 1, 1, True),
           ("ktable", 4, 4, 3, 1, False), ("ktable", 4, 8, 3, 1, True), ("ktable", 8, 8, 3, 1, False), ("ktable", 8, 16, 1, 1, True),
           ("ktable", 16, 16, 3, 1, False), ("ktable", 2, 2, 2, 1, True)]
    m._build_tables_batched(ops)
    torch.cuda.synchronize()
    assert len(m.tables) == len(ops)
    for _, ts_in, ts_out, ks, dil, transposed in ops:
        off = oracle_maps.kernel_offsets(ks, ts_in)
        ref = oracle_maps.kernel_map_table(c_ref[ts_in], c_ref[ts_out], off)
        nbr, nbr_t = m.tables[(ts_in, ts_out, ks, dil)]
        assert np.array_equal(nbr.cpu().numpy(), ref), (ts_in, ts_out, ks)
        if transposed:
            ref_t = np.full((len(c_ref[ts_in]), off.shape[0]), -1, np.int32)
            o, k = np.nonzero(ref >= 0)
            ref_t[ref[o, k], k] = o
            assert np.array_equal(nbr_t.cpu().numpy(), ref_t), (ts_in, ts_out, ks)


@pytest.mark.gpu
def test_block_index_tables_equal_the_per_voxel_hash():
    """Neighbour tables come from the 4^3-block index (mink_kernel_map_batch); the per-voxel hash look-up it replaced
    (mink_kernel_map, still in the C ABI for callers that keep ME's own map) must give the same table bit for bit, for
    every kernel shape the networks use: 3^3 stride 1 / 2 (with the transposed table), 2^3 stride 2, 1^3 stride 2."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import check, lib
    from nerf_downstream_amd.minkowski.coords import kernel_offsets

    coords, feats = batch_scenes([31, 32, 33], grid=48, cin=4)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    m, k1 = x.coordinate_manager, x.coordinate_map_key
    k2 = m.stride(k1, 2)
    for kin, kout, ks, tr in ((k1, k1, 3, False), (k1, k2, 3, True), (k1, k2, 2, True), (k1, k2, 1, True), (k2, k2, 3, False)):
        nbr, nbr_t = m.kernel_table(kin, kout, ks, 1, transposed=tr)
        lin, lout = m.levels[kin.ts], m.levels[kout.ts]
        off = kernel_offsets(ks, kin.ts, 1)
        K = off.shape[0]
        ref = torch.empty(lout.n, K, dtype=torch.int32, device="cuda")
        ref_t = torch.full((lin.n, K), -1, dtype=torch.int32, device="cuda") if tr else None
        tkeys, tvals, cap = m.hash_map(kin.ts)  # (level 0 of an ascending field skipped its insert: filled on demand)
        check(lib().mink_kernel_map(tkeys.data_ptr(), tvals.data_ptr(), cap, lout.coords.data_ptr(), lout.n,
                                    off.ctypes.data, K, ref.data_ptr(), None if ref_t is None else ref_t.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(nbr, ref), (kin.ts, kout.ts, ks)
        assert (nbr_t is None) == (ref_t is None) and (ref_t is None or torch.equal(nbr_t, ref_t)), (kin.ts, kout.ts, ks)


def test_class_partition_batch_equals_the_per_map_calls():
    """mink_class_partition_batch (four launches over all maps of a plan) against mink_class_partition map by map: the same
    permutations, bit for bit -- maps of very different sizes in one batch, an empty one included."""
    import ctypes

    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import ClassPartitionDesc, check, lib

    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    coords, feats = batch_scenes([3, 4, 5], grid=48, cin=4)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    m = x.coordinate_manager
    levels = [(1, m.levels[1])]
    for ts in (1, 2, 4):
        m.stride(ME.CoordinateMapKey(ts), 2)
        levels.append((2 * ts, m.levels[2 * ts]))
    pad = 128
    want, keep = [], []
    descs = (ClassPartitionDesc * (len(levels) + 1))()
    got = []
    for d, (ts, lev) in zip(descs, levels):
        rows = int(L.mink_class_partition_rows(lev.n, pad))
        wsb = int(L.mink_class_partition_workspace_bytes(lev.n))
        ref = torch.empty(rows, dtype=torch.int32, device="cuda")
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        check(L.mink_class_partition(lev.coords.data_ptr(), lev.n, ts, pad, ref.data_ptr(), ws.data_ptr(), wsb, st))
        want.append(ref)
        out, ws2 = torch.full((rows,), 7, dtype=torch.int32, device="cuda"), torch.empty(wsb, dtype=torch.uint8, device="cuda")
        d.coords, d.n, d.ts, d.pad, d.perm, d.workspace, d.workspace_bytes = lev.coords.data_ptr(), lev.n, ts, pad, out.data_ptr(), ws2.data_ptr(), wsb
        got.append(out)
        keep.append(ws2)
    e = descs[len(levels)]  # an empty map: only its (empty) permutation is filled
    empty = torch.full((int(L.mink_class_partition_rows(0, pad)) + 4,), 7, dtype=torch.int32, device="cuda")
    e.coords, e.n, e.ts, e.pad, e.perm, e.workspace, e.workspace_bytes = None, 0, 1, pad, empty.data_ptr(), None, 0
    check(L.mink_class_partition_batch(len(levels) + 1, ctypes.byref(descs), st))
    torch.cuda.synchronize()
    for (ts, lev), a, b in zip(levels, want, got):
        assert torch.equal(a, b), (ts, lev.n)
        assert sorted(b[b >= 0].tolist()) == list(range(lev.n))
    assert bool((empty[-4:] == 7).all())


def test_batched_map_construction_beyond_one_batch_equals_table_by_table():
    """mink_kernel_map_batch issues one launch per kind over up to eight maps; a plan with more (ten block indices, ten 3x3x3
    tables, nine strided ones with their transposed tables, nine pooling tables) takes the flush-and-continue path.  Every
    table must equal the one a fresh manager builds alone (single-descriptor calls), bit for bit."""
    from nerf_downstream_amd import minkowski as ME

    coords, feats = batch_scenes([8, 9], grid=40, cin=4)

    def manager():
        x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
        m = x.coordinate_manager
        ts = 1
        for _ in range(9):
            m.stride(ME.CoordinateMapKey(ts), 2)
            ts *= 2
        return m

    tss = [2 ** i for i in range(10)]
    ops = [("ktable", t, t, 3, 1, False) for t in tss] + [("ktable", t, 2 * t, 3, 1, True) for t in tss[:-1]] + \
          [("ktable", t, 2 * t, 2, 1, False) for t in tss[:-1]]
    batched = manager()
    batched._build_tables_batched(ops)
    single = manager()
    for _, ti, to, ks, dil, tr in ops:
        a = batched.tables[(ti, to, ks, dil)]
        b = single.kernel_table(ME.CoordinateMapKey(ti), ME.CoordinateMapKey(to), ks, dil, transposed=tr)
        assert torch.equal(a[0], b[0]), (ti, to, ks)
        assert int((a[0] >= 0).sum()) > 0
        if tr:
            assert torch.equal(a[1], b[1]), (ti, to, ks, "transposed")
