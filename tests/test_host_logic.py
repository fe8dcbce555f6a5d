"""Host-side pieces of the map preparation that need no GPU: the arena the maps of one batch are carved from, the cached
descriptor offsets, the prepare-stream gate's parsing, the memory reservation's no-ops."""
import ctypes

import numpy as np
import torch


def test_arena_takes_are_aligned_typed_views_of_a_few_chunks():
    from nerf_downstream_amd.minkowski.coords import _Arena

    old = _Arena.last_used
    try:
        _Arena.last_used = 1 << 12
        a = _Arena(torch.device("cpu"))
        t1 = a.take((3, 5), torch.int32)
        t2 = a.take(7, torch.int64)
        t3 = a.take((2, 3, 4), torch.float32)
        t4 = a.take(0, torch.int32)  # (an empty map: a valid, empty view)
        t5 = a.take(100, torch.uint8)
        assert t1.shape == (3, 5) and t1.dtype == torch.int32 and t2.shape == (7,) and t2.dtype == torch.int64
        assert t3.shape == (2, 3, 4) and t4.numel() == 0 and t5.shape == (100,)
        base = a.chunks[0].data_ptr()
        ptrs = [t.data_ptr() - base for t in (t1, t2, t3, t5)]  # (an empty view has no address)
        assert all(p % 256 == 0 for p in ptrs) and ptrs == sorted(set(ptrs))  # 256-byte steps from the chunk's start, no two takes share one
        t1.fill_(1), t2.fill_(2), t3.fill_(3.0), t5.fill_(5)
        assert int(t1.sum()) == 15 and int(t2.sum()) == 14 and float(t3.sum()) == 72.0 and int(t5.sum()) == 500  # no overlap
        big = a.take(1 << 16, torch.int32)  # larger than what is left: a new chunk, typed views start over
        big.fill_(7)
        assert len(a.chunks) == 2 and a.chunks[-1].numel() % 256 == 0 and int(t1.sum()) == 15
        after = a.take(4, torch.int64)  # (the chunk `big` asked for is full: another one)
        after.fill_(9)
        assert (after.data_ptr() - a.chunks[-1].data_ptr()) % 256 == 0 and int(big.sum()) == 7 << 16 and int(after.sum()) == 36
        assert a.used == _Arena.last_used and a.used % 256 == 0
    finally:
        _Arena.last_used = old


def test_descriptor_offsets_are_the_kernel_offsets():
    from nerf_downstream_amd.minkowski.coords import _kernel_offsets_ct, kernel_offsets

    for ks, ts, dil in ((3, 1, 1), (3, 4, 1), (2, 2, 1), (1, 8, 1), (3, 2, 2)):
        ct = _kernel_offsets_ct(ks, ts, dil)
        assert isinstance(ct, ctypes.c_int32 * 81) and _kernel_offsets_ct(ks, ts, dil) is ct  # (built once per shape)
        ref = kernel_offsets(ks, ts, dil).ravel()
        got = np.ctypeslib.as_array(ct)
        assert np.array_equal(got[: ref.size], ref) and not got[ref.size :].any()
    o = kernel_offsets(3, 2, 1)
    assert o.shape == (27, 3) and tuple(o[0]) == (-2, -2, -2) and tuple(o[1]) == (0, -2, -2) and tuple(o[13]) == (0, 0, 0)  # x fastest


def test_prepare_gate_parsing_and_no_ops():
    from nerf_downstream_amd import memory
    from nerf_downstream_amd.minkowski import functional as Fn

    assert Fn._parse_gate("0") is None and Fn._parse_gate("") is None and Fn._parse_gate(None) is None
    assert Fn._parse_gate("f0") == (0, 0) and Fn._parse_gate("f3") == (0, 3) and Fn._parse_gate("b1") == (1, 1) and Fn._parse_gate("b-1") == (1, -1)
    Fn.wait_prepare_gate(None)  # nothing recorded, no stream: nothing to wait for
    assert memory.reserve_on(None) == 0 and memory.reserve(torch.device("cpu")) == 0
