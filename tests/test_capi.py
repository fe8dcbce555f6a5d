"""CPU: the C-ABI library builds, loads and exports every symbol include/mink_hip.h declares."""
import ctypes
import os
import re

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
    assert handle.mink_abi_version() == 1
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
