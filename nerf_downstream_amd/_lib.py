"""ctypes binding of libmink_hip.so (the C ABI declared in include/mink_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing the
compute path raises.  ``build()`` compiles it in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MINK_HIP_LIB") or os.path.join(_HERE, "libmink_hip.so")  # (override: A/B builds of the same ABI)
CSRC = os.path.join(_HERE, "csrc")

_i32, _i64, _f32, _p = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class KernelMapDesc(ctypes.Structure):
    """MinkKernelMapDesc of include/mink_hip.h"""

    _fields_ = [("in_table_keys", _p), ("in_table_vals", _p), ("in_cap", _i64), ("out_coords", _p), ("n_out", _i64),
                ("n_in", _i64), ("nbr", _p), ("nbr_t", _p), ("K", _i32), ("offsets", _i32 * 81),
                ("in_coords", _p), ("in_ts", _i32), ("blk_build", _i32), ("blk_table", _p), ("blk_base", _p), ("blk_slot", _p),
                ("blk_rowids", _p), ("blk_counter", _p), ("blk_cap", _i64)]


class ConvLayer(ctypes.Structure):
    """MinkConvLayer of include/mink_hip.h"""

    _fields_ = [("w", _p), ("dw", _p), ("nbr", _p), ("nbr_t", _p), ("perm", _p), ("n_perm", _i64), ("K", _i32), ("cin", _i32),
                ("cout", _i32), ("stride", _i32)]


class NormLayer(ctypes.Structure):
    """MinkNormLayer"""

    _fields_ = [("gamma", _p), ("beta", _p), ("running_mean", _p), ("running_var", _p), ("dgamma", _p), ("dbeta", _p),
                ("mean", _p), ("invstd", _p), ("momentum", _f32), ("eps", _f32)]


class Exec(ctypes.Structure):
    """MinkExec"""

    _fields_ = [("compute", _p), ("branch", _p), ("wgrad", _p), ("ws_compute", _p), ("ws_branch", _p), ("ws_wgrad", _p),
                ("ws_bytes", _i64)]


class Stem(ctypes.Structure):
    """MinkStem"""

    _fields_ = [("conv", ConvLayer), ("norm", NormLayer), ("nbr_pool", _p), ("in2out", _p), ("n", _i64), ("n_pool", _i64),
                ("x", _p), ("y", _p), ("out", _p), ("g_out", _p), ("xb", _p), ("xb_ready", _i32)]


class BasicBlock(ctypes.Structure):
    """MinkBasicBlock"""

    _fields_ = [("conv1", ConvLayer), ("conv2", ConvLayer), ("down", ConvLayer), ("norm1", NormLayer), ("norm2", NormLayer),
                ("normd", NormLayer), ("n_in", _i64), ("n_out", _i64), ("x", _p), ("y1", _p), ("h1", _p), ("y2", _p),
                ("yd", _p), ("sd", _p), ("out", _p), ("g_out", _p), ("g_x", _p), ("g_tmp", _p)]


class LevelMaps(ctypes.Structure):
    """MinkLevelMaps"""

    _fields_ = [("n", _i64), ("nbr3", _p), ("down3", _p), ("down1", _p), ("down3_t", _p), ("perm", _p), ("n_perm", _i64)]


class Net(ctypes.Structure):
    """MinkNet"""

    _fields_ = [("stem", Stem), ("blocks", ctypes.POINTER(BasicBlock)), ("n_blocks", _i32), ("with_stem", _i32), ("out", _p),
                ("out_rows", _i64), ("g_stem_out", _p)]


STAGE_HOOK = ctypes.CFUNCTYPE(None, _i32, _i32)  # MinkStageHook
BLOCK_DONE_HOOK = ctypes.CFUNCTYPE(None, _i32)  # MinkBlockDoneHook


class ClassPartitionDesc(ctypes.Structure):
    """MinkClassPartitionDesc"""

    _fields_ = [("coords", _p), ("n", _i64), ("ts", _i32), ("pad", _i32), ("perm", _p), ("workspace", _p), ("workspace_bytes", _i64)]


class TimingEntry(ctypes.Structure):
    """MinkTimingEntry"""

    _fields_ = [("kind", _i32), ("K", _i32), ("cin", _i32), ("cout", _i32), ("n_in", _i64), ("n_out", _i64), ("nbr", _p),
                ("ms", _f32)]


# name -> (restype, argtypes); mirrors include/mink_hip.h one to one
SIGNATURES = {
    "mink_last_error": (ctypes.c_char_p, []),
    "mink_abi_version": (ctypes.c_int, []),
    "mink_table_capacity": (_i64, [_i64]),
    "mink_unique_workspace_bytes": (_i64, [_i64]),
    "mink_coords_make_keys": (ctypes.c_int, [_p, ctypes.c_int, _i64, _i32, _p, _p, _p]),
    "mink_coords_unique": (ctypes.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_levels_workspace_bytes": (_i64, [_i64]),
    "mink_coords_build_levels": (ctypes.c_int, [_p, ctypes.c_int, _i64, _i32, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_kernel_map": (ctypes.c_int, [_p, _p, _i64, _p, _i64, _p, _i32, _p, _p, _p]),
    "mink_kernel_map_batch": (ctypes.c_int, [_i32, _p, _p]),
    "mink_set_overflow_sink": (ctypes.c_int, [_p]),
    "mink_rulebook_workspace_bytes": (_i64, [_i64, _i32]),
    "mink_rulebook": (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "mink_class_partition_rows": (_i64, [_i64, _i32]),
    "mink_class_partition_workspace_bytes": (_i64, [_i64]),
    "mink_class_partition": (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p, _i64, _p]),
    "mink_batch_offsets": (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p]),
    "mink_decode_plenoxel": (ctypes.c_int, [_p, _p, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p, _p, _i32, _p]),
    "mink_augment_workspace_bytes": (_i64, [_i64, _i32]),
    "mink_augment_scenes": (
        ctypes.c_int,
        [_p, _i32, _p, _i64, _i32, _i64, _p, _i32, _p, _p, ctypes.c_uint64, _p, _p, _p, _i64, _p, _p, _i64, _p],
    ),
    "mink_conv_set_stagger": (ctypes.c_int, [ctypes.c_int]),
    "mink_conv_set_pipeline": (ctypes.c_int, [ctypes.c_int]),
    "mink_conv_trace": (ctypes.c_int, [_p, _i64]),
    "mink_conv_get_math": (ctypes.c_int, []),
    "mink_conv_set_math": (ctypes.c_int, [ctypes.c_int]),
    "mink_conv_plan_ksplit": (ctypes.c_int, [_i64, _i32, _i32, _i32]),
    "mink_conv_plan": (ctypes.c_int, [_i64, _i32, _i32, _i32, _i32]),
    "mink_conv_gather_gemm": (
        ctypes.c_int,
        [_p, _i64, _i32, _i32, _p, _i32, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _p, _i32, _p, _i64, _p],
    ),
    "mink_conv_stats_workspace_bytes": (_i64, [_i64, _i32]),
    "mink_conv_gather_gemm_stats": (
        ctypes.c_int,
        [_p, _i64, _i32, _i32, _p, _p, _i64, _i32, _p, _i32, _i32, _p, _i32, _p, _i64, _p, _p, _p, _i64, _p],
    ),
    "mink_bn_stats_from_partials": (ctypes.c_int, [_p, _i32, _i64, _i32, _f32, _f32, _p, _p, _p, _p, _p]),
    "mink_conv_wgrad_workspace_bytes": (_i64, [_i64, _i32, _i32, _i32]),
    "mink_conv_wgrad": (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _i32, _i32, _p, _i64, _i32, _p, _p, _i64, _p]),
    "mink_conv_wgrad_bn_relu_pool_supported": (ctypes.c_int, [_i64, _i32, _i32, _i64, _i32, _i32]),
    "mink_conv_wgrad_bn_relu_pool": (
        ctypes.c_int,
        [_p, _i64, _i32, _i32, _p, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _p],
    ),
    "mink_dense_xwt": (ctypes.c_int, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "mink_rows_scatter_add": (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "mink_pool_sum_fwd": (ctypes.c_int, [_p, _i32, _i32, _p, _i64, _i32, _p, _p]),
    "mink_pool_sum_bwd": (ctypes.c_int, [_p, _i32, _p, _i64, _p, _p]),
    "mink_pool_max_fwd": (ctypes.c_int, [_p, _i32, _p, _i64, _i32, _p, _p, _p]),
    "mink_pool_max_bwd": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _i32, _p, _p]),
    "mink_global_avg_fwd": (ctypes.c_int, [_p, _i32, _p, _i32, _p, _p]),
    "mink_global_avg_bwd": (ctypes.c_int, [_p, _i32, _p, _i32, _i64, _p, _p]),
    "mink_head_forward": (ctypes.c_int, [_p, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p]),
    "mink_head_backward": (ctypes.c_int, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p]),
    "mink_softmax_ce_forward": (ctypes.c_int, [_p, _p, _i32, _i32, _p, _p, _p]),
    "mink_softmax_ce_backward": (ctypes.c_int, [_p, _p, _p, _i32, _i32, _p, _p]),
    "mink_segment_mean": (ctypes.c_int, [_p, _i32, _i32, _p, _p, _i64, _p, _p]),
    "mink_bn_workspace_bytes": (_i64, [_i64, _i32]),
    "mink_bn_stats": (ctypes.c_int, [_p, _i64, _i32, _f32, _f32, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_bn_apply": (ctypes.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p]),
    "mink_bn_fwd": (ctypes.c_int, [_p, _i64, _i32, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_bn_bwd": (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_bn_reduce": (ctypes.c_int, [_i32, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "mink_bn_stats_from_sums": (ctypes.c_int, [_p, _p, _i32, _f32, _f32, _p, _p, _p, _p, _p]),
    "mink_bn_bwd_from_sums": (ctypes.c_int, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "mink_bn_relu_pool_fwd": (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "mink_bn_relu_pool_bwd": (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_eltwise": (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "mink_sgd_step": (ctypes.c_int, [_p, _p, _p, _i64, _f32, _f32, _f32, _i32, _p]),
    "mink_activation": (ctypes.c_int, [_p, _p, _p, _i32, _i64, _i32, _f32, _p, _p]),
    "mink_block_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "mink_block_grad_scratch_floats": (_i64, [_i64, _i64, _i32, _i32, _i32]),
    "mink_class_partition_batch": (ctypes.c_int, [_i32, _p, _p]),
    "mink_rows_to_bf16": (ctypes.c_int, [_p, _i64, _i32, _i32, _p, _p]),
    "mink_stem_conv_bf16s_supported": (ctypes.c_int, [_i64, _i64, _i32, _i32, _i32]),
    "mink_stem_conv_bf16s_stats_rows": (_i32, []),
    "mink_stem_conv_bf16s": (ctypes.c_int, [_p, _i64, _p, _i32, _p, _i64, _i32, _p, _i32, _p, _i32, _p]),
    "mink_bn_relu_pool_fwd_b16": (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "mink_bn_relu_pool_bwd_b16": (ctypes.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_conv_wgrad_bn_relu_pool_b16": (
        ctypes.c_int,
        [_p, _i64, _i32, _p, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _i64, _p],
    ),
    "mink_stem_supported": (ctypes.c_int, [_i64, _i32, _i32, _i32]),
    "mink_stem_forward": (ctypes.c_int, [_p, _p]),
    "mink_stem_backward": (ctypes.c_int, [_p, _p]),
    "mink_block_forward": (ctypes.c_int, [_p, _p]),
    "mink_block_backward": (ctypes.c_int, [_p, _p]),
    "mink_bn_set_fold": (ctypes.c_int, [_i32]),
    "mink_bn_bwd_slabs": (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _i64, _p]),
    "mink_bn_apply_from_partials": (ctypes.c_int, [_p, _i64, _i32, _p, _i32, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p]),
    "mink_bn_small_rows": (_i32, []),
    "mink_bn_set_small": (ctypes.c_int, [_i32]),
    "mink_conv_gather_gemm_slabs": (
        ctypes.c_int,
        [_p, _i64, _i32, _i32, _p, _i32, _i32, _p, _i64, _i32, _p, _i64, _p, _i32, _i32, _i32, _p, _i64, _p, _p],
    ),
    "mink_bn_small_fwd": (ctypes.c_int, [_p, _i32, _i64, _i32, _p, _f32, _f32, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p]),
    "mink_bn_small_bwd": (ctypes.c_int, [_p, _i32, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "mink_set_stage_hook": (ctypes.c_int, [_p]),
    "mink_set_block_done_hook": (ctypes.c_int, [_p]),
    "mink_event_create": (ctypes.c_int, [_p]),
    "mink_event_destroy": (ctypes.c_int, [_p]),
    "mink_stream_wait_event": (ctypes.c_int, [_p, _p]),
    "mink_net_sizes": (ctypes.c_int, [_p, _p, _i32, _p, _p, _p]),
    "mink_net_forward": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _p]),
    "mink_net_backward": (ctypes.c_int, [_p, _p, _i32, _p, _i64, _p, _p, _i64, _p, _p]),
    "mink_stream_create_cu_subset": (ctypes.c_int, [_i32, _i32, _i32, _p]),
    "mink_stream_destroy": (ctypes.c_int, [_p]),
    "mink_conv_timing": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "mink_conv_timing_fetch": (_i64, [_p, _i64]),
}


def build(force=False):
    """Compile libmink_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_PATH


_lib = None


class MinkHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the native library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MinkHipError(
                f"{LIB_PATH} not found: the HIP backend is mandatory (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C nerf_downstream_amd/csrc`."
            )
        # One HIP runtime per process: torch ships its own libamdhip64; load it first so that
        # libmink_hip.so (linked against the same soname) binds to that instance instead of
        # bringing /opt/rocm's copy in beside it (two runtimes = "no ROCm-capable device").
        import torch  # noqa: F401

        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        # A/B knobs of the launch fusions (scripts/ab_bn.py): MINK_BN_FOLD = fold limit in partial rows, MINK_BN_SMALL = 0 / 1 / 8 / 16
        if os.environ.get("MINK_BN_FOLD"):
            L.mink_bn_set_fold(int(os.environ["MINK_BN_FOLD"]))
        if os.environ.get("MINK_BN_SMALL"):
            L.mink_bn_set_small(int(os.environ["MINK_BN_SMALL"]))
        if os.environ.get("MINK_CONV_PIPELINE"):  # (A/B runs: 0 = two-stage mid-layer kernel, 1 / 2 / 3 = three-stage, see mink_hip.h)
            L.mink_conv_set_pipeline(int(os.environ["MINK_CONV_PIPELINE"]))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MinkHipError(f"libmink_hip: {lib().mink_last_error().decode()} (code {rc})")
