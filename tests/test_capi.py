"""CPU: the C-ABI library builds, loads and exports every symbol include/mink_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mink_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mink_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from nerf_downstream_amd import _lib

    _lib.build()
    names = _declared()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in mink_hip.h but not exported"
    handle = _lib.lib()
    assert handle.mink_abi_version() == 4  # 2: every scratch buffer travels with its size; 3: MinkStem.xb (bf16 storage); 4: mink_net_*
    # pure host helpers can run without a GPU
    assert handle.mink_table_capacity(1000) == 2048
    assert handle.mink_conv_plan_ksplit(1_000_000, 27, 64, 0) == 1
    assert handle.mink_conv_plan_ksplit(512, 27, 512, 0) > 1
    assert handle.mink_unique_workspace_bytes(1000) > 5000


def test_argument_validation_without_gpu():
    from nerf_downstream_amd import _lib

    L = _lib.lib()
    assert L.mink_coords_make_keys(None, 0, 10, 1, None, None, None) == -1
    assert b"NULL" in L.mink_last_error()
    assert L.mink_kernel_map(None, None, 64, None, 5, None, 28, None, None, None) == -1
    assert L.mink_bn_apply(None, 4, 6, None, None, None, None, None, 0, None, None) == -1
    assert b"multiple of 4" in L.mink_last_error()


def test_undersized_workspace_is_refused_before_any_launch():
    """ABI v2: the size of a scratch buffer is an argument, checked against the plan the call is about to launch.  The
    split factors / row splits of a convolution depend on planner state (tuning knobs), so a size cached by the caller
    for another plan used to be written past; now it is MINK_EINVAL.  Dummy non-NULL pointers: the check comes before
    the first launch, so this runs without a GPU."""
    from nerf_downstream_amd import _lib

    L = _lib.lib()
    P = 0x10000  # aligned, never dereferenced
    # weight gradient of a mid layer: the plan wants row-split slabs
    n_out, K, cin, cout = 40000, 27, 64, 64
    need = L.mink_conv_wgrad_workspace_bytes(n_out, K, cin, cout)
    assert need > 0
    assert L.mink_conv_wgrad(P, n_out, cin, cin, P, cout, cout, P, n_out, K, P, P, need - 1, None) == -1
    assert b"workspace" in L.mink_last_error() and str(need).encode() in L.mink_last_error()
    # ... and a size that was right for the default plan is too small once a knob asks for more row splits
    old = L.mink_conv_set_stagger((2 << 12) | (200 << 16))  # force G = 3 and 200 row splits (scripts/kbench.py wsweep)
    try:
        assert L.mink_conv_wgrad_workspace_bytes(n_out, K, cin, cout) > need
        assert L.mink_conv_wgrad(P, n_out, cin, cin, P, cout, cout, P, n_out, K, P, P, need, None) == -1
        assert b"row splits of this plan" in L.mink_last_error()
    finally:
        L.mink_conv_set_stagger(old)
    # split-K forward: slabs
    assert L.mink_conv_gather_gemm(P, 512, 64, 64, P, 0, 0, P, 512, 27, None, 0, P, 64, 64, None, 4, P, 4 * 4 * 512 * 64 - 1, None) == -1
    assert b"slabs" in L.mink_last_error()
    # batch norm backward / statistics, coordinate pyramid, class partition
    assert L.mink_bn_bwd(P, P, P, 1000, 64, P, P, P, 1, P, None, P, P, P, L.mink_bn_workspace_bytes(1000, 64) - 1, None) == -1
    assert b"workspace" in L.mink_last_error()
    assert L.mink_bn_stats(P, 1000, 64, 1e-5, 0.1, P, P, None, None, P, 16, None) == -1
    assert L.mink_class_partition(P, 1000, 2, 128, P, P, 8, None) == -1
    assert b"workspace" in L.mink_last_error()


def test_trunk_branch_policy():
    """Where the native trunk runs its shortcut branch (minkowski/functional.py: trunk_branch_mode): a stream of its own by
    default; under a data-parallel reducer (fork off, branch-on-side on) the weight-gradient stream with fp32 / split-bf16
    math and no branch at all with bf16 math.  Host logic only: the math mode is read through the C ABI, no kernel runs."""
    from nerf_downstream_amd.minkowski import functional as Fn

    old_fork, old_side, old_math = Fn.set_branch_fork(True), Fn.set_trunk_branch_on_side(False), Fn.set_conv_math("fp32")
    try:
        assert Fn.conv_math() == "fp32" and Fn.trunk_branch_mode() == "own"
        Fn.set_branch_fork(False)
        assert Fn.trunk_branch_mode() is None
        Fn.set_trunk_branch_on_side(True)
        assert Fn.trunk_branch_mode() == "side"
        Fn.set_conv_math("bf16x3")
        assert Fn.trunk_branch_mode() == "side"
        Fn.set_conv_math("bf16")
        assert Fn.conv_math() == "bf16" and Fn.trunk_branch_mode() is None
        Fn.set_branch_fork(True)
        assert Fn.trunk_branch_mode() == "own"
    finally:
        Fn.set_branch_fork(old_fork), Fn.set_trunk_branch_on_side(old_side), Fn.set_conv_math(old_math)
    assert Fn.set_conv_storage("bf16") == "fp32" and Fn.set_conv_storage("fp32") == "bf16"
    with pytest.raises(ValueError):
        Fn.set_conv_storage("fp16")


def test_net_sizes_follow_the_documented_arena_layout():
    """mink_net_sizes (host arithmetic only: no GPU): the activation arena of mink_net_forward is, per block and 256-byte
    aligned, [y1 | h1 | y2 | out | (yd | sd)] of n_out x C floats + six C-vectors of statistics; the gradient arena the block's
    mink_block_grad_scratch_floats + its input gradient -- the layout include/mink_hip.h documents and
    nerf_downstream_amd/minkowski/trunk.py::_Saved walks."""
    import ctypes

    from nerf_downstream_amd._lib import BasicBlock, LevelMaps, Net, lib

    L = lib()
    blocks = (BasicBlock * 3)()
    spec = [(64, 64, 2, True), (64, 64, 1, False), (64, 128, 2, True)]  # cin, C, stride, shortcut convolution
    for b, (cin, C, stride, down) in zip(blocks, spec):
        b.conv1.K, b.conv1.cin, b.conv1.cout, b.conv1.stride = 27, cin, C, stride
        b.conv2.K, b.conv2.cin, b.conv2.cout, b.conv2.stride = 27, C, C, 1
        if down:
            b.down.w, b.down.K, b.down.cin, b.down.cout, b.down.stride = 1, 1, cin, C, 2  # (any non-NULL pointer: never dereferenced here)
    net = Net()
    net.blocks, net.n_blocks, net.with_stem = ctypes.cast(blocks, ctypes.POINTER(BasicBlock)), 3, 1
    levels = (LevelMaps * 3)()
    rows = [10840, 2316, 527]
    for lv, n in zip(levels, rows):
        lv.n = n
    a, g, w = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert L.mink_net_sizes(ctypes.byref(net), levels, 3, ctypes.byref(a), ctypes.byref(g), ctypes.byref(w)) == 0
    up = lambda v: -(-v // 64) * 64  # noqa: E731
    shapes = [(rows[0], rows[1], 64, 64, True), (rows[1], rows[1], 64, 64, False), (rows[1], rows[2], 64, 128, True)]
    want_a = sum(up((6 if d else 4) * no * C) + up(6 * C) for _, no, _, C, d in shapes)
    want_g = sum(up(L.mink_block_grad_scratch_floats(ni, no, cin, C, int(d))) + up(ni * cin) for ni, no, cin, C, d in shapes)
    assert a.value == want_a and g.value == want_g
    assert w.value == max(L.mink_block_workspace_bytes(ni, no, cin, C) for ni, no, cin, C, _ in shapes)
    # a block that strides past the last level given is refused
    assert L.mink_net_sizes(ctypes.byref(net), levels, 2, ctypes.byref(a), ctypes.byref(g), ctypes.byref(w)) != 0
    assert b"level" in L.mink_last_error()
