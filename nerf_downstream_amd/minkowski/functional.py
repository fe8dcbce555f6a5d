"""Autograd functions of the HIP backend: every forward/backward is a call into libmink_hip.so
(C ABI, include/mink_hip.h) on torch-owned device buffers and the current HIP stream.

Semantics follow SURVEY.md Appendix A (A6 convolution, A8 batch norm, A9 sum pooling,
A10 global average pooling); the reference call sites are cited on the modules.
"""
import ctypes
import os

import torch

from .._lib import check, lib


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _f32c(t):
    assert t.is_cuda, "the HIP backend has no CPU path"
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _bytes(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Scratch buffers of the main (compute) stream are recycled: kernels on one stream run in
# order, so the next user may overwrite a workspace as soon as it is enqueued behind the last.
_SCRATCH = {}


def _scratch(nbytes, device, slot, on=None):
    """`on`: the torch stream the buffer will be used on when that is not the current one."""
    stream = _stream() if on is None else on.cuda_stream
    key = (device.index, stream, slot)
    buf = _SCRATCH.get(key)
    nbytes = max(int(nbytes), 256)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=device)
        if on is not None:  # allocated under the current stream, used on `on`: a later free must wait for that stream
            buf.record_stream(on)
        _SCRATCH[key] = buf
    return buf


# ------------------------------------------------------------------- matrix-core math
_MATH = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 3}


def set_conv_math(mode="fp32"):
    """Arithmetic of the convolution forward / input-gradient GEMMs: "fp32" (exact, default),
    "bf16" (bf16 MFMA operands, fp32 accumulate -- BASELINE config "bf16 mixed precision") or
    "bf16x3" (split-bf16, three products per step).  Tensors in HBM stay fp32; the weight
    gradient always runs in exact fp32.  Returns the previous mode name."""
    if mode not in _MATH:
        raise ValueError(f"conv math {mode!r}: choose from {sorted(set(_MATH))}")
    old = lib().mink_conv_set_math(_MATH[mode])
    _PLAN_CACHE.clear()  # the split plan depends on which kernel the mode selects
    return {0: "fp32", 1: "bf16", 3: "bf16x3"}[old]


_STORAGE_B16 = False


def set_conv_storage(mode="fp32"):
    """HBM storage of the FULL-RESOLUTION stage of a Mink-ResNet trunk (input features and the stem convolution's
    output, three quarters of the activation bytes of a step): "fp32" (default) or "bf16".  "bf16" takes effect under
    `set_conv_math("bf16")` on the native trunk (minkowski/trunk.py); every tensor from the pooled level down, the
    parameters, their gradients and the batch-norm statistics stay fp32.  Returns the previous mode name."""
    global _STORAGE_B16
    if mode not in ("fp32", "bf16"):
        raise ValueError(f"conv storage {mode!r}: choose 'fp32' or 'bf16'")
    old, _STORAGE_B16 = _STORAGE_B16, mode == "bf16"
    return "bf16" if old else "fp32"


def rows_to_bf16(x):
    """bf16 copy [n, 32] (zero-padded) of fp32 rows with at most 32 channels (mink_rows_to_bf16) on the current stream."""
    xb = torch.empty(x.shape[0], 32, dtype=torch.bfloat16, device=x.device)
    check(lib().mink_rows_to_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), xb.data_ptr(), _stream()))
    return xb


def conv_math():
    """The current mode name of set_conv_math (read without changing it)."""
    return {0: "fp32", 1: "bf16", 3: "bf16x3"}[lib().mink_conv_get_math()]


# ---------------------------------------------------------------------- kernel timing
# bench.py measures the dominant kernel live with HIP events recorded on the launch stream.  The events are recorded
# inside the native library around each convolution call (mink_conv_timing), so the module-by-module path and the
# native trunk are instrumented alike.  HIP event records are not free on this stack (each one is a barrier packet),
# so the timed region instruments ONE kernel tag only; the warm-up instruments all.
_TIMING_MODE = 0     # 0 off, 1 every convolution launch, 2 only `_TIMING_ONLY`
_TIMING_ONLY = None
_TIMING_TABLES = {}  # warm-up (mode 1): data_ptr -> neighbour table, kept until the entries are fetched (pair counts)
_TIMING_PAIRS = {}   # tag -> {n_out: valid entries of its table}
_KINDS = ("fwd", "dgrad", "wgrad")


def _parse_tag(tag):
    import re

    kind, n_out, K, cin, cout = re.fullmatch(r"(\w+)\[(\d+)x(\d+):(\d+)->(\d+)\]", tag).groups()
    return _KINDS.index(kind), int(n_out), int(K), int(cin), int(cout)


def enable_kernel_timing(on=True, only=None, any_kind=False):
    """`only`: a tag "kind[n_out x K:cin->cout]" -- every launch of that kind / K / cin / cout is timed (`any_kind`: of that
    K / cin / cout, whatever the kind -- forward, data gradient and weight gradient of one layer)."""
    global _TIMING_MODE, _TIMING_ONLY
    L = lib()
    n = L.mink_conv_timing_fetch(None, 0)
    if n:  # drop entries nobody asked for
        from .._lib import TimingEntry

        L.mink_conv_timing_fetch((TimingEntry * n)(), n)
    _TIMING_TABLES.clear()
    if not on:
        _TIMING_MODE, _TIMING_ONLY = 0, None
        L.mink_conv_timing(0, 0, 0, 0, 0)
    elif only is None:
        _TIMING_MODE, _TIMING_ONLY = 1, None
        L.mink_conv_timing(1, 0, 0, 0, 0)
    else:
        kind, _, K, cin, cout = _parse_tag(only)
        _TIMING_MODE, _TIMING_ONLY = 2, only
        L.mink_conv_timing(2, -1 if any_kind else kind, K, cin, cout)


def note_table(*tables):
    """Warm-up instrumentation: keep the neighbour tables of timed launches alive so their pair counts can be read."""
    if _TIMING_MODE == 1:
        for t in tables:
            if t is not None:
                _TIMING_TABLES[t.data_ptr()] = t


def kernel_timings():
    """{tag: {"ms": [per-launch milliseconds], "meta": {kind, n_in, n_out, K, cin, cout, pairs (per launch)}}} of the
    launches timed since the last call (synchronises on their events)."""
    from .._lib import TimingEntry

    L = lib()
    n = L.mink_conv_timing_fetch(None, 0)
    out = {}
    if n == 0:
        return out
    buf = (TimingEntry * n)()
    n = L.mink_conv_timing_fetch(buf, n)
    for e in buf[:n]:
        tag = f"{_KINDS[e.kind]}[{e.n_out}x{e.K}:{e.cin}->{e.cout}]"
        pairs = _TIMING_PAIRS.setdefault(tag, {})
        t = _TIMING_TABLES.get(e.nbr)
        if t is not None and e.n_out not in pairs:
            # (the row count of the gathered operand is not an argument of a forward / data-gradient call: every row of
            # it is referenced by the table of the layers timed here, so it is the largest index + 1)
            pairs[e.n_out] = (int((t >= 0).sum().item()), e.n_in if e.n_in >= 0 else int(t.max().item()) + 1)
        pr, n_in = pairs.get(e.n_out, (None, e.n_in))
        ent = out.setdefault(tag, {"ms": [], "meta": {"kind": _KINDS[e.kind], "n_in": n_in, "n_out": e.n_out, "K": e.K,
                                                      "cin": e.cin, "cout": e.cout, "pairs": pr}})
        if e.ms >= 0:
            ent["ms"].append(e.ms)
    _TIMING_TABLES.clear()
    return out


# ------------------------------------------------------------------------- convolution
_FORCE_KSPLIT = 0  # benchmarking hook (scripts/kbench.py ksweep)

def gather_gemm(x, w, nbr, cout, w_transposed=False, flip_k=False, bias=None, row_perm=None, stats=False):
    """y[o] = sum_k x[nbr[o,k]] @ W[k] (+bias) on the fp32 matrix cores.  `stats=True` (forward
    only) also returns the column (sum, sum of squares) partials [rows,2,cout] (float64) of y for
    the batch norm that follows, or None when the launch shape cannot produce them."""
    L = lib()
    n_out, K = nbr.shape
    cin = x.shape[1]
    y = torch.empty(n_out, cout, dtype=torch.float32, device=x.device)
    n_rows = n_out if row_perm is None else row_perm.numel()
    ksplit = _FORCE_KSPLIT or _plan_ksplit(L, n_rows, K, cin, cout, int(row_perm is not None))
    ws = _scratch(4 * ksplit * n_out * cout, x.device, "splitk") if ksplit > 1 else None
    partial = None
    note_table(nbr)
    if stats and not w_transposed and row_perm is None and n_out > 0:
        partial = torch.empty(512, 2, cout, dtype=torch.float64, device=x.device)
        sws = _scratch(L.mink_conv_stats_workspace_bytes(n_out, cout), x.device, "convstats")
        rows = ctypes.c_int32(0)
        check(
            L.mink_conv_gather_gemm_stats(
                x.data_ptr(), x.shape[0], x.stride(0), cin, w.data_ptr(), nbr.data_ptr(), n_out, K, y.data_ptr(), cout, cout,
                _ptr(bias), ksplit, _ptr(ws), 0 if ws is None else ws.numel(), partial.data_ptr(), ctypes.addressof(rows), sws.data_ptr(),
                sws.numel(), _stream(),
            )
        )
        partial = partial[: rows.value] if rows.value > 0 else None
    else:
        check(
            L.mink_conv_gather_gemm(
                x.data_ptr(), x.shape[0], x.stride(0), cin, w.data_ptr(), int(w_transposed), int(flip_k), nbr.data_ptr(), n_out, K,
                _ptr(row_perm), 0 if row_perm is None else row_perm.numel(),
                y.data_ptr(), cout, cout, _ptr(bias), ksplit, _ptr(ws), 0 if ws is None else ws.numel(), _stream(),
            )
        )
    return (y, partial) if stats else y


def dense_xwt(x, w):
    """x [n, Kd] @ w[N, Kd]^T on the fp32 matrix cores (mink_dense_xwt)."""
    x, w = _f32c(x), _f32c(w)
    y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
    check(lib().mink_dense_xwt(x.data_ptr(), w.data_ptr(), x.shape[0], x.shape[1], w.shape[0], y.data_ptr(), _stream()))
    return y


def conv_wgrad(x, dy, nbr, kernel_shape, out=None, on=None):
    """dW[k] = X[nbr[:, k]]^T dY.  `out`: write into this contiguous fp32 tensor (a slice of a
    data-parallel reducer's flat gradient buffer) instead of a fresh one.  `on`: launch on this torch
    stream instead of the current one (the caller orders it and records the buffers it reads; a fresh
    result tensor is recorded here) -- cheaper than switching the current stream around the call."""
    L = lib()
    n_out, K = nbr.shape
    cin, cout = x.shape[1], dy.shape[1]
    if out is not None:
        dw = out
    else:
        dw = torch.empty(kernel_shape, dtype=torch.float32, device=x.device)
        if on is not None:
            dw.record_stream(on)
    ws = _scratch(_wgrad_ws_bytes(L, n_out, K, cin, cout), x.device, "wgrad", on)
    raw = _stream() if on is None else on.cuda_stream
    note_table(nbr)
    check(
        L.mink_conv_wgrad(
            x.data_ptr(), x.shape[0], x.stride(0), cin, dy.data_ptr(), dy.stride(0), cout, nbr.data_ptr(), n_out, K,
            dw.data_ptr(), ws.data_ptr(), ws.numel(), raw,
        )
    )
    return dw


_PLAN_CACHE = {}


def _wgrad_ws_bytes(L, n_out, K, cin, cout):
    key = ("wws", n_out, K, cin, cout)
    v = _PLAN_CACHE.get(key)
    if v is None:
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        v = _PLAN_CACHE[key] = int(L.mink_conv_wgrad_workspace_bytes(n_out, K, cin, cout))
    return v


def _plan_ksplit(L, n_rows, K, cin, cout, classes):
    key = ("ks", n_rows, K, cin, cout, classes)
    v = _PLAN_CACHE.get(key)
    if v is None:
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        v = _PLAN_CACHE[key] = int(L.mink_conv_plan(n_rows, K, cin, cout, classes))
    return v


_OVERLAP_WGRAD = True  # weight gradients on a side stream, joined at the end of backward (B=16 ResNet14: 5.11 -> 4.84 ms/step)
_SIDE_STREAMS = {}
_CU_STREAMS = []  # (raw handle) of the CU-subset streams created here: they live as long as the process


def new_stream(device, role):
    """A HIP stream for one of the auxiliary roles ("prepare": the next batch's coordinate maps, "wgrad": the weight
    gradients, "branch": the shortcut branch).  MINK_CUS_<ROLE>="first:count" (or "count") confines it to that range
    of compute units (mink_stream_create_cu_subset): what runs on it then cannot take execution slots from the
    kernels of the compute stream on the other CUs."""
    spec = os.environ.get("MINK_CUS_" + role.upper(), _CU_DEFAULTS.get(role, ""))
    if not spec or spec == "0":
        return torch.cuda.Stream(device=device)
    first, _, count = spec.rpartition(":")
    total = torch.cuda.get_device_properties(device).multi_processor_count
    first, count = int(first or 0), int(count)
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(lib().mink_stream_create_cu_subset(first, count, total, ctypes.byref(h)))
    _CU_STREAMS.append(h.value)
    return torch.cuda.ExternalStream(h.value, device=device)


_CU_DEFAULTS = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device.index)
    if s is None:
        s = _SIDE_STREAMS[device.index] = new_stream(device, "wgrad")
    return s


def set_bn_small(on=True):
    """Few-row layers of the native trunk: the one-launch batch norm + split-K sum (mink_bn_small_fwd / _bwd; csrc/
    elementwise.hip).  Off: the trunk sequences exactly the kernels of the module-by-module path (the bitwise tests)."""
    return bool(lib().mink_bn_set_small(1 if on else 0))


def set_bn_fold(max_rows=0):
    """Batch-norm finalize inside the apply pass (mink_bn_apply_from_partials, mink_bn_bwd) when the producer left at most
    `max_rows` partial rows (0: never; at most 128).  Returns the previous limit."""
    return int(lib().mink_bn_set_fold(int(max_rows)))


def set_wgrad_overlap(on=True):
    global _OVERLAP_WGRAD
    old, _OVERLAP_WGRAD = _OVERLAP_WGRAD, bool(on)
    return old


_GRAD_SINK = None  # a data-parallel reducer that owns the gradient memory (see set_grad_sink)
# Where a deferred bucket collective may be launched: behind the backward of a convolution with at most this many
# weights per offset -- the wide-and-shallow layers, whose kernels keep the GPU busy for >100 us.  A static property
# of the layer on purpose: every rank then issues its collectives at the same points of the backward pass (a
# row-count rule would depend on each rank's batch and could interleave them differently with SyncBatchNorm's).
_FLUSH_MAX_WEIGHTS = 128 * 128


def set_grad_sink(sink):
    """`sink.view_for(param)` -> the tensor the weight gradient of `param` is to be WRITTEN into (or
    None: hand the gradient to autograd as usual); `sink.ready(param)` is called once that write
    has been queued, `sink.release(param)` if it will not be after all, `sink.flush()` where the host has
    time to spare (a large layer's kernels have just been queued).  This lets a bucketed all-reduce keep the weight-gradient side stream: without
    it every gradient would have to be accumulated into the reducer's buffer on the compute stream,
    i.e. joined layer by layer."""
    global _GRAD_SINK
    old, _GRAD_SINK = _GRAD_SINK, sink
    return old


_DEFERRED = {"pending": False, "task": -1}  # task: the autograd graph task the end-of-backward join is queued for
_HOME_STREAMS = {}  # device index -> the stream the network itself runs on (noted where a branch forks off)
_BRANCH_STREAMS = {}


def _parse_gate(v):
    """"f<i>" / "b<i>": before block i of the native trunk's forward / backward pass; "b-1": before the stem's backward; "0": off."""
    if not v or v == "0":
        return None
    import re

    if not re.fullmatch(r"[fb]-?\d+", v):
        raise ValueError(f"MINK_PREPARE_GATE={v!r}: expected 'f<i>' or 'b<i>' (block index; 'b-1' = before the stem's backward) or '0'")
    return (1 if v[0] == "b" else 0, int(v[1:]))


# MINK_PREPARE_GATE (off by default): the prepare stream may start a batch's pyramid / plan only once the compute stream has
# reached this point of its latest native-trunk pass -- one map build per step, beside the middle of the step, instead of a
# prepare stream that follows the host however far ahead it is.  Steady step times and fewer prepared batches in memory (peak
# reserved 38.8 -> 26.7 GB), no throughput gain (3,000 steps: 3.43-3.44 ms against 3.41-3.43); a short timed window
# loses its map-free tail and host-bound shapes lose ~1 % (profiles/r05_transient.txt).
_PREPARE_GATE = _parse_gate(os.environ.get("MINK_PREPARE_GATE", "0"))  # (backward?, stage): see trunk._stage_hook
_PREPARE_GATE_EVENT = {}  # device index -> event of the latest gate point


def note_prepare_gate(stream):
    _PREPARE_GATE_EVENT[stream.device.index] = stream.record_event()


def wait_prepare_gate(stream):
    ev = _PREPARE_GATE_EVENT.get(stream.device.index) if (_PREPARE_GATE_EVENT and stream is not None) else None
    if ev is not None:
        stream.wait_event(ev)
_PHASE_LOG = None  # diagnostic (bench.py --timeline): [(name, timing event, host clock)] of every mark
_STREAM_OBJS = {}


def current_stream(device=None):
    """torch.cuda.current_stream(device) without building a Stream object per call (8 us each, ~15 per step on the hot path: 0.12 ms
    of a host-bound step): the raw handle of the current stream is one C call, the object comes out of a cache."""
    idx = device.index if (device is not None and getattr(device, "index", None) is not None) else (
        device if isinstance(device, int) else torch.cuda.current_device())
    raw = torch._C._cuda_getCurrentRawStream(idx)
    s = _STREAM_OBJS.get((idx, raw))
    if s is None:
        s = _STREAM_OBJS[(idx, raw)] = torch.cuda.current_stream(idx)
    return s


def log_phase(name, stream):
    if _PHASE_LOG is not None:
        import time

        t = torch.cuda.Event(enable_timing=True)
        t.record(stream)
        _PHASE_LOG.append((name, t, time.perf_counter()))


_SKEW = int(os.environ.get("MINK_STREAM_SKEW", "0"))  # race hunting: delay every auxiliary stream by this many GPU cycles


def skew(stream):
    """Test hook: stall `stream` for MINK_STREAM_SKEW cycles before its next piece of work, so a
    missing cross-stream dependency shows up as a wrong result instead of passing by luck."""
    if _SKEW:
        with torch.cuda.stream(stream):
            torch.cuda._sleep(_SKEW)


_WAIT_EVENTS = {}


def stream_wait(waiter, awaited):
    """`waiter.wait_stream(awaited)` without creating a HIP event per call (9 us each on this stack, ~20 per
    step): one cached event per awaited stream.  Re-recording it later does not disturb a wait already
    queued -- a stream wait captures the record that precedes it."""
    ev = _WAIT_EVENTS.get(awaited)
    if ev is None:
        ev = _WAIT_EVENTS[awaited] = torch.cuda.Event()
    ev.record(awaited)
    waiter.wait_event(ev)


def side_stream_if_any(device):
    return _SIDE_STREAMS.get(device.index)


def compute_streams(device):
    """The auxiliary compute streams this module has created on `device` (weight-gradient stream,
    shortcut-branch stream): whoever consumes gradients from another stream joins these."""
    return [d[device.index] for d in (_SIDE_STREAMS, _BRANCH_STREAMS) if device.index in d]


_BRANCH_FORK = True
_TRUNK_BRANCH_ON_SIDE = False  # data parallelism: the native trunk runs its shortcut branch on the weight-gradient stream


def set_branch_fork(on=True):
    """Allow / forbid residual blocks to run their shortcut branch on the second compute stream."""
    global _BRANCH_FORK
    old, _BRANCH_FORK = _BRANCH_FORK, bool(on)
    return old


def branch_fork_enabled():
    return _BRANCH_FORK


def trunk_branch_mode():
    """Where the native trunk runs the shortcut branch of its residual blocks: "own" (a stream of its own), "side" (the
    weight-gradient stream: data parallelism with fp32 matrix math, where that stream has room -- under bf16 math its
    fp32 weight gradients are the long pole and the branch behind them stalls the chain: 2.64 -> 2.69 ms) or None."""
    if _BRANCH_FORK:
        return "own"
    if _TRUNK_BRANCH_ON_SIDE and conv_math() != "bf16":
        return "side"
    return None


def set_trunk_branch_on_side(on=True):
    """The native trunk's shortcut branch on the weight-gradient stream instead of a stream of its own (data parallelism:
    the process group's streams already take hardware queues; the module-by-module path then does not fork at all)."""
    global _TRUNK_BRANCH_ON_SIDE
    old, _TRUNK_BRANCH_ON_SIDE = _TRUNK_BRANCH_ON_SIDE, bool(on)
    return old


def branch_stream(device, home=None):
    """A second compute stream for an independent branch of the network (the shortcut
    convolution of a residual block runs beside the main branch: both are small kernels that do
    not fill the chip on their own).  `home`: the stream the caller forks from."""
    s = _BRANCH_STREAMS.get(device.index)
    if s is None:
        s = _BRANCH_STREAMS[device.index] = new_stream(device, "branch")
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:  # parameters of the branch get their gradient from this stream on purpose
            quiet(False)
    if home is not None:
        _HOME_STREAMS[device.index] = home
    return s


def _join_side_streams():
    """End-of-backward callback: the compute stream waits for the weight gradients still running
    on the side stream, so everything after backward() (optimizer, clipping, ...) is ordered."""
    _DEFERRED["task"] = -1
    join_side_streams()


_AFTER_JOIN = []  # callables run once the compute stream is ordered after the side stream (release of kept-alive buffers)


def join_side_streams():
    """Make the current stream wait for weight gradients still in flight on the side stream (no-op when none are)."""
    if _DEFERRED["pending"]:
        _DEFERRED["pending"] = False
        for index, side in _SIDE_STREAMS.items():
            stream_wait(current_stream(index), side)
    for fn in _AFTER_JOIN:
        fn()


def _defer_join():
    """Queue the end-of-backward join once per backward pass.  Keyed by the autograd graph task: a backward that
    raised drops its queued callbacks, and a flag that merely said "already queued" would then stay set for the
    rest of the process -- every later optimizer step would read weight gradients the side stream is still writing."""
    _DEFERRED["pending"] = True
    task = torch._C._current_graph_task_id()
    if _DEFERRED["task"] != task or task < 0:
        _DEFERRED["task"] = task
        torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)


def _sink_views(*params):
    """Data parallelism: the slices of the reducer's flat buffer to write these parameters' gradients into --
    all of them or none (then autograd accumulates as usual).  Saves one accumulate launch per parameter."""
    sink = _GRAD_SINK
    if sink is None:
        return None
    bulk = getattr(sink, "views_for", None)
    if bulk is not None:
        return bulk(params)
    views = []
    for p in params:
        v = sink.view_for(p)
        if v is None:
            for q in params[: len(views)]:  # hand back what was claimed
                sink.release(q)
            return None
        views.append(v)
    return views


class ConvolutionFunction(torch.autograd.Function):
    """MinkowskiConvolution forward/backward (reference modules/common.py:116-125; A6).

    `table_fn(transposed)` returns (nbr, nbr_t) lazily so the transposed table is only built
    when an input gradient is really needed (the stem conv needs wgrad only).
    """

    @staticmethod
    def forward(ctx, x, kernel, table_fn, same_map, stats_holder=None):
        """`stats_holder`: a list that receives the column-statistics partials of the output (or
        None) -- the batch norm that follows then skips its own reduction pass."""
        x = _f32c(x)
        w = _f32c(kernel)
        cin = x.shape[1]
        ctx.cin = cin
        if cin % 4 and not ctx.needs_input_grad[0]:
            # e.g. the reference default features=["sh"] (27 channels): one zero column puts the
            # rows on 16-byte boundaries so the vectorised / flattened-K kernels apply
            pad = 4 - cin % 4
            x = torch.nn.functional.pad(x, (0, pad))
            w = torch.nn.functional.pad(w, (0, 0, 0, pad))
        tables = table_fn(False)
        nbr = tables[0]
        ctx.save_for_backward(x, w)
        ctx.table_fn, ctx.same_map, ctx.nbr = table_fn, same_map, nbr
        if stats_holder is None:  # a transposed convolution brings the parity-class row order of its output
            return gather_gemm(x, w, nbr, w.shape[-1], row_perm=tables[2] if len(tables) > 2 else None)
        y, partial = gather_gemm(x, w, nbr, w.shape[-1], stats=True)
        stats_holder.append(partial)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _f32c(gy)
        gx = gw = None
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # dgrad and wgrad only share their inputs: run wgrad on a second HIP stream so the two
        # (latency-bound for the deep, small layers) overlap; the compute stream re-joins before
        # anything can consume the weight gradient.
        side = _side_stream(gy.device) if (want_gx and want_gw and _OVERLAP_WGRAD) else None
        if side is not None:
            main = current_stream()
            stream_wait(side, main)  # gy / x are ready on the side stream
            skew(side)
        if want_gx:
            # dgrad = the same gather-GEMM over the transposed map with W[k]^T (read in place)
            if ctx.same_map:  # stride 1: nbr_t[i][k] == nbr[i][K-1-k]
                gx = gather_gemm(gy, w, ctx.nbr, w.shape[-2], w_transposed=True, flip_k=True)
            elif w.shape[0] == 1 and w.shape[-1] % 4 == 0 and w.shape[-2] % 4 == 0:
                # kernel volume 1, strided (the shortcut of a residual block): the product is a plain GEMM over the
                # OUTPUT rows, scattered to the input rows it reaches through the forward table -- the same two
                # kernels the native trunk sequences (no transposed table, no class permutation)
                gx = torch.zeros(x.shape[0], w.shape[-2], dtype=torch.float32, device=gy.device)
                gsc = dense_xwt(gy, w[0])
                check(lib().mink_rows_scatter_add(gsc.data_ptr(), ctx.nbr.data_ptr(), gsc.shape[0], gsc.shape[1], gx.data_ptr(), _stream()))
            else:
                _, nbr_t, perm = ctx.table_fn(True)
                gx = gather_gemm(gy, w, nbr_t, w.shape[-2], w_transposed=True, row_perm=perm)
        if want_gw:
            sink = _GRAD_SINK
            out = sink.view_for(w) if (sink is not None and w.shape[1] == ctx.cin) else None
            if side is not None:
                gw = conv_wgrad(x, gy, ctx.nbr, w.shape, out=out, on=side)  # (a fresh gw is recorded on the side stream there)
                for t in (x, gy, ctx.nbr):  # every buffer the side stream reads
                    t.record_stream(side)
                if out is not None:  # written in place into the reducer's buffer: nothing for autograd to do
                    _defer_join()
                    sink.ready(w)
                    if w.shape[-1] * w.shape[-2] <= _FLUSH_MAX_WEIGHTS:  # long kernels are queued: the host has time to launch collectives
                        sink.flush()
                    return gx, None, None, None, None
                # Nothing reads gw before backward ends when autograd merely installs it as
                # w.grad: then the join is deferred to the end-of-backward callback and the
                # wgrad kernels overlap the rest of the backward chain.  An existing .grad
                # (accumulation, DP flat buffers), a gradient hook or the channel-padding slice
                # below consume it right away: join now.
                # (No reference to gw may be kept here: AccumulateGrad only installs the tensor
                # itself when nobody else holds it -- otherwise it CLONES it, at once, on the
                # compute stream.  Double backward clones as well.)
                home = _HOME_STREAMS.get(gw.device.index)
                if home is not None and home != main:
                    gw.record_stream(home)  # a branch's gradient is consumed by the network's own stream later
                if w.is_leaf and w.grad is None and not getattr(w, "_post_accumulate_grad_hooks", None) and \
                        not w._backward_hooks and gw.shape[1] == ctx.cin and not torch.is_grad_enabled():
                    _defer_join()
                else:
                    stream_wait(main, side)
            else:
                gw = conv_wgrad(x, gy, ctx.nbr, w.shape, out=out)
                if out is not None:
                    sink.ready(w)
                    return gx, None, None, None, None
            if gw.shape[1] != ctx.cin:  # drop the gradient of the zero-padded input channels
                gw = gw[:, : ctx.cin].contiguous()
        return gx, gw, None, None, None


class PointwiseConvolutionFunction(torch.autograd.Function):
    """1x1x1 stride-1 convolution = F @ W (`use_mm`, witness sparse_conv.py:323-335).  Forward and input
    gradient are plain library GEMMs with a long M; the weight gradient X^T dY is [cin, cout] small with the
    row count as its reduction dimension -- the library runs that on a few dozen workgroups (14 ms for
    825 k x 128 -> 96), so it goes through the streaming weight-gradient kernel with an identity table."""

    @staticmethod
    def forward(ctx, x, w, ident_fn):
        ctx.save_for_backward(x, w)
        ctx.ident_fn = ident_fn
        return x.mm(w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        # (W^T materialised: for the transposed-operand form the library picks a 12-wide macro tile -- 10 ms
        # instead of 0.7 ms for 825 k x 96 -> 128)
        gx = gy.mm(w.t().contiguous()) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = conv_wgrad(_f32c(x), _f32c(gy), ctx.ident_fn(), (1,) + tuple(w.shape)).view_as(w)
        return gx, gw, None


# -------------------------------------------------------------------------- batch norm
def _bn_statistics(L, x, n, C, eps, momentum, running_mean, running_var, partial):
    """mean / invstd of the batch (+ running-stat update): from the producer's column partials
    when it supplied them, else by a reduction pass over x."""
    dev = x.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    mom = momentum if running_mean is not None else 0.0
    if partial is not None:
        check(
            L.mink_bn_stats_from_partials(
                partial.data_ptr(), partial.shape[0], n, C, eps, mom, mean.data_ptr(), invstd.data_ptr(),
                _ptr(running_mean), _ptr(running_var), _stream(),
            )
        )
    else:
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
        check(
            L.mink_bn_stats(
                x.data_ptr(), n, C, eps, mom, mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                ws.data_ptr(), ws.numel(), _stream(),
            )
        )
    return mean, invstd


class BatchNormFunction(torch.autograd.Function):
    """BatchNorm1d over the rows of F, optionally fused with the residual add and ReLU that
    follow it in the reference block (modules/resnet_block.py:53-69; A8)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, residual, relu, partial=None):
        L = lib()
        x = _f32c(x)
        n, C = x.shape
        dev = x.device
        if residual is not None:
            residual = _f32c(residual)
        y = torch.empty_like(x)
        if training and partial is None:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
            check(
                L.mink_bn_fwd(
                    x.data_ptr(), n, C, eps, momentum if running_mean is not None else 0.0, gamma.data_ptr(), beta.data_ptr(),
                    _ptr(residual), int(relu), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean),
                    _ptr(running_var), ws.data_ptr(), ws.numel(), _stream(),
                )
            )
        elif training and partial is not None:
            # statistics partials from the producing convolution: finalize + apply, one launch when the partial rows are few
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            check(
                L.mink_bn_apply_from_partials(
                    x.data_ptr(), n, C, partial.data_ptr(), partial.shape[0], eps, momentum if running_mean is not None else 0.0,
                    gamma.data_ptr(), beta.data_ptr(), _ptr(residual), int(relu), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                    _ptr(running_mean), _ptr(running_var), _stream(),
                )
            )
        else:
            if training:
                mean, invstd = _bn_statistics(L, x, n, C, eps, momentum, running_mean, running_var, partial)
            else:
                mean = running_mean.float()
                invstd = torch.rsqrt(running_var.float() + eps)
            check(
                L.mink_bn_apply(
                    x.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                    _ptr(residual), int(relu), y.data_ptr(), _stream(),
                )
            )
        ctx.save_for_backward(x, y if relu else None, mean, invstd, gamma)
        ctx.training, ctx.relu, ctx.has_res = training, relu, residual is not None
        ctx.beta = beta  # only as the key of its gradient slot under data parallelism
        return y

    @staticmethod
    def backward(ctx, gy):
        L = lib()
        x, y, mean, invstd, gamma = ctx.saved_tensors
        gy = _f32c(gy)
        n, C = x.shape
        dev = x.device
        if not ctx.training:  # eval-mode BN is an affine map; rarely differentiated
            g = gy * (y > 0) if ctx.relu else gy
            xhat = (x - mean) * invstd
            return (g * (gamma * invstd), (g * xhat).sum(0), g.sum(0), None, None, None, None, None,
                    g if ctx.has_res else None, None, None)
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        views = _sink_views(gamma, ctx.beta) if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] else None
        if views is not None:
            dgamma, dbeta = views
        else:
            dgamma = torch.empty(C, dtype=torch.float32, device=dev)
            dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
        check(
            L.mink_bn_bwd(
                gy.data_ptr(), x.data_ptr(), _ptr(y), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                int(ctx.relu), gx.data_ptr(), _ptr(gres), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
            )
        )
        if views is not None:  # written in place into the reducer's buffer: nothing for autograd to accumulate
            _GRAD_SINK.ready(gamma), _GRAD_SINK.ready(ctx.beta)
            return gx, None, None, None, None, None, None, None, gres, None, None
        return gx, dgamma, dbeta, None, None, None, None, None, gres, None, None


class SyncBatchNormFunction(torch.autograd.Function):
    """BatchNorm with statistics over all ranks (ME.MinkowskiSyncBatchNorm, reference
    train.py:106-107): per-channel (sum x, sum x^2, rows) are all-reduced between the reduction and
    the apply pass; backward all-reduces (sum g, sum g*xhat).  Parameter gradients stay local
    sums -- the data-parallel gradient all-reduce averages them like every other gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, group):
        import torch.distributed as dist

        L = lib()
        x = _f32c(x)
        n, C = x.shape
        dev = x.device
        if residual is not None:
            residual = _f32c(residual)
        buf = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        buf[2 * C] = float(n)
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
        check(L.mink_bn_reduce(0, x.data_ptr(), None, None, n, C, None, None, buf.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        dist.all_reduce(buf, group=group)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        check(
            L.mink_bn_stats_from_sums(
                buf.data_ptr(), buf[2 * C :].data_ptr(), C, eps, momentum if running_mean is not None else 0.0,
                mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var), _stream(),
            )
        )
        y = torch.empty_like(x)
        check(
            L.mink_bn_apply(
                x.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                _ptr(residual), int(relu), y.data_ptr(), _stream(),
            )
        )
        ctx.save_for_backward(x, y if relu else None, mean, invstd, gamma, buf[2 * C :].clone())
        ctx.relu, ctx.has_res, ctx.group = relu, residual is not None, group
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist

        L = lib()
        x, y, mean, invstd, gamma, n_total = ctx.saved_tensors
        gy = _f32c(gy)
        n, C = x.shape
        dev = x.device
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
        check(
            L.mink_bn_reduce(1, gy.data_ptr(), x.data_ptr(), _ptr(y), n, C, mean.data_ptr(), invstd.data_ptr(),
                             sums.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        )
        dbeta, dgamma = sums[:C].float(), sums[C:].float()  # local sums: averaged later with the other grads
        dist.all_reduce(sums, group=ctx.group)
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if ctx.has_res else None
        tmp = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(
            L.mink_bn_bwd_from_sums(
                gy.data_ptr(), x.data_ptr(), _ptr(y), n, C, sums.data_ptr(), n_total.data_ptr(), mean.data_ptr(),
                invstd.data_ptr(), gamma.data_ptr(), int(ctx.relu), gx.data_ptr(), _ptr(gres), tmp.data_ptr(), _stream(),
            )
        )
        return gx, dgamma, dbeta, None, None, None, None, gres, None, None


class BNReLUSumPoolFunction(torch.autograd.Function):
    """relu(BN(x)) summed over the 2^3 children of every coarse voxel, in one pass over x
    (reference resnet.py:58-64: bn1 -> relu -> pool).  The normalised fine-level tensor -- the
    largest activation of the network -- is never written; backward recomputes the ReLU mask."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, nbr, in2out, partial=None):
        L = lib()
        x = _f32c(x)
        n, C = x.shape
        dev = x.device
        if training:
            mean, invstd = _bn_statistics(L, x, n, C, eps, momentum, running_mean, running_var, partial)
        else:
            mean = running_mean.float()
            invstd = torch.rsqrt(running_var.float() + eps)
        n_out, K = nbr.shape
        y = torch.empty(n_out, C, dtype=torch.float32, device=dev)
        check(
            L.mink_bn_relu_pool_fwd(
                x.data_ptr(), C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), nbr.data_ptr(),
                n_out, K, y.data_ptr(), _stream(),
            )
        )
        ctx.save_for_backward(x, mean, invstd, gamma, beta)
        ctx.in2out, ctx.training = in2out, training
        return y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.training:
            raise NotImplementedError("fused bn+relu+pool backward needs batch statistics (training mode)")
        L = lib()
        x, mean, invstd, gamma, beta = ctx.saved_tensors
        gy = _f32c(gy)
        n, C = x.shape
        gx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), x.device, "bn")
        check(
            L.mink_bn_relu_pool_bwd(
                gy.data_ptr(), x.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                ctx.in2out.data_ptr(), gx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
            )
        )
        return gx, dgamma, dbeta, None, None, None, None, None, None, None, None


class ConvBNReLUSumPoolFunction(torch.autograd.Function):
    """pool(relu(bn(conv(x)))) of the network stem (reference resnet.py:58-64) as ONE autograd node,
    for an input that needs no gradient: forward = convolution (+ column statistics in its epilogue)
    and the fused bn+relu+sum-pool pass; backward = (dgamma, dbeta) reduction and the streaming
    weight-gradient kernel recomputing the gradient w.r.t. the conv output inside its operand load
    -- neither bn(conv(x)) nor its gradient (825 k x 64 floats each at B=16) is ever written."""

    @staticmethod
    def forward(ctx, x, kernel, gamma, beta, running_mean, running_var, momentum, eps, nbr, nbr_pool, in2out):
        L = lib()
        x, w = _f32c(x), _f32c(kernel)
        cin = x.shape[1]
        ctx.cin = cin
        if cin % 4:
            pad = 4 - cin % 4
            x = torch.nn.functional.pad(x, (0, pad))
            w = torch.nn.functional.pad(w, (0, 0, 0, pad))
        C = w.shape[-1]
        y, partial = gather_gemm(x, w, nbr, C, stats=True)
        n = y.shape[0]
        mean, invstd = _bn_statistics(L, y, n, C, eps, momentum, running_mean, running_var, partial)
        n_pool, K = nbr_pool.shape
        out = torch.empty(n_pool, C, dtype=torch.float32, device=x.device)
        check(
            L.mink_bn_relu_pool_fwd(
                y.data_ptr(), C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), nbr_pool.data_ptr(),
                n_pool, K, out.data_ptr(), _stream(),
            )
        )
        ctx.save_for_backward(x, w, y, mean, invstd, gamma, beta)
        ctx.nbr, ctx.in2out = nbr, in2out
        ctx.kernel = kernel  # only as the key of its gradient slot under data parallelism
        return out

    @staticmethod
    def backward(ctx, gy):
        L = lib()
        x, w, y, mean, invstd, gamma, beta = ctx.saved_tensors
        gy = _f32c(gy)
        n, C = y.shape
        dev = y.device
        need = ctx.needs_input_grad
        pv = _sink_views(gamma, beta) if need[2] and need[3] else None  # data parallelism: write straight into the reducer's buffer
        kv = _sink_views(ctx.kernel) if need[1] and w.shape[1] == ctx.cin else None
        if pv is not None:
            dgamma, dbeta = pv
        else:
            dgamma = torch.empty(C, dtype=torch.float32, device=dev)
            dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        gw = kv[0] if kv is not None else None
        ws = _scratch(L.mink_bn_workspace_bytes(n, C), dev, "bn")
        check(
            L.mink_bn_relu_pool_bwd(
                gy.data_ptr(), y.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                ctx.in2out.data_ptr(), None, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
            )
        )
        nbr = ctx.nbr
        K = nbr.shape[1]
        if gw is None:
            gw = torch.empty(w.shape, dtype=torch.float32, device=dev)
        wws = _scratch(L.mink_conv_wgrad_workspace_bytes(n, K, x.shape[1], C), dev, "wgrad")
        note_table(nbr)
        check(
            L.mink_conv_wgrad_bn_relu_pool(
                x.data_ptr(), x.shape[0], x.stride(0), x.shape[1], y.data_ptr(), C, gy.data_ptr(), gy.shape[0],
                ctx.in2out.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                dgamma.data_ptr(), dbeta.data_ptr(), nbr.data_ptr(), n, K, gw.data_ptr(), wws.data_ptr(), wws.numel(), _stream(),
            )
        )
        if _GRAD_SINK is not None and hasattr(_GRAD_SINK, "flush"):
            _GRAD_SINK.flush()  # the stem's weight gradient (0.9 ms) is queued: everything completed so far goes out beside it
        if pv is not None:
            _GRAD_SINK.ready(gamma), _GRAD_SINK.ready(beta)
            dgamma = dbeta = None
        if kv is not None:
            _GRAD_SINK.ready(ctx.kernel)
            gw = None
        elif gw.shape[1] != ctx.cin:
            gw = gw[:, : ctx.cin].contiguous()
        return None, gw, dgamma, dbeta, None, None, None, None, None, None, None

    @staticmethod
    def supported(x, kernel, nbr):
        cin = x.shape[1] + (-x.shape[1]) % 4
        return bool(lib().mink_conv_wgrad_bn_relu_pool_supported(x.shape[0], cin, cin, nbr.shape[0], nbr.shape[1], kernel.shape[-1]))


# ----------------------------------------------------------------------------- eltwise
def _eltwise(a, b, mode):
    y = torch.empty_like(a)
    check(lib().mink_eltwise(a.data_ptr(), _ptr(b), a.numel(), mode, y.data_ptr(), _stream()))
    return y


class ReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = _eltwise(_f32c(x), None, 0)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return _eltwise(_f32c(gy), y, 1)


ACT_KINDS = {"leaky_relu": 1, "elu": 2, "celu": 3, "selu": 4, "gelu": 5, "prelu": 6}


class ActivationFunction(torch.autograd.Function):
    """Pointwise activations beyond ReLU (ME.MinkowskiLeakyReLU / ELU / CELU / SELU / GELU / PReLU; the reference's layer
    factory lists them at import, modules/common.py:36-43): out = f(x), backward gy * f'(x), one HIP pass each.  The
    PReLU weight gradient (a column sum of gy * min(x, 0)) is a torch reduction."""

    @staticmethod
    def forward(ctx, x, kind, alpha, slope):
        x = _f32c(x)
        y = torch.empty_like(x)
        C = 1 if slope is None or slope.numel() == 1 else x.shape[1]
        check(lib().mink_activation(x.data_ptr(), None, _ptr(slope), C, x.numel(), ACT_KINDS[kind], float(alpha), y.data_ptr(), _stream()))
        ctx.save_for_backward(x, slope)
        ctx.kind, ctx.alpha, ctx.C = kind, float(alpha), C
        return y

    @staticmethod
    def backward(ctx, gy):
        x, slope = ctx.saved_tensors
        gy = _f32c(gy)
        gx = torch.empty_like(x)
        check(lib().mink_activation(x.data_ptr(), gy.data_ptr(), _ptr(slope), ctx.C, x.numel(), ACT_KINDS[ctx.kind], ctx.alpha,
                                    gx.data_ptr(), _stream()))
        gslope = None
        if slope is not None and ctx.needs_input_grad[3]:
            t = gy * x.clamp(max=0)
            gslope = t.sum().reshape(1) if slope.numel() == 1 else t.sum(0)
        return gx, None, None, gslope


class AddFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return _eltwise(_f32c(a), _f32c(b), 2)

    @staticmethod
    def backward(ctx, gy):
        return gy, gy


# ----------------------------------------------------------------------------- pooling
class SumPoolFunction(torch.autograd.Function):
    """MinkowskiSumPooling(kernel==stride) (reference resnet.py:62-64; A9)."""

    @staticmethod
    def forward(ctx, x, nbr, in2out):
        x = _f32c(x)
        n_out, K = nbr.shape
        C = x.shape[1]
        y = torch.empty(n_out, C, dtype=torch.float32, device=x.device)
        check(lib().mink_pool_sum_fwd(x.data_ptr(), x.stride(0), C, nbr.data_ptr(), n_out, K, y.data_ptr(), _stream()))
        ctx.in2out, ctx.n_in = in2out, x.shape[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _f32c(gy)
        C = gy.shape[1]
        gx = torch.empty(ctx.n_in, C, dtype=torch.float32, device=gy.device)
        check(lib().mink_pool_sum_bwd(gy.data_ptr(), C, ctx.in2out.data_ptr(), ctx.n_in, gx.data_ptr(), _stream()))
        return gx, None, None


class MaxPoolFunction(torch.autograd.Function):
    """Max pooling over a neighbour table (windows may overlap); `nbr_t` = the transposed table."""

    @staticmethod
    def forward(ctx, x, nbr, nbr_t):
        x = _f32c(x)
        n_out, K = nbr.shape
        C = x.shape[1]
        y = torch.empty(n_out, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(n_out, C, dtype=torch.int32, device=x.device)
        check(lib().mink_pool_max_fwd(x.data_ptr(), C, nbr.data_ptr(), n_out, K, y.data_ptr(), arg.data_ptr(), _stream()))
        ctx.save_for_backward(arg, nbr_t)
        ctx.n_in = x.shape[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        arg, nbr_t = ctx.saved_tensors
        gy = _f32c(gy)
        C = gy.shape[1]
        gx = torch.empty(ctx.n_in, C, dtype=torch.float32, device=gy.device)
        check(lib().mink_pool_max_bwd(gy.data_ptr(), arg.data_ptr(), C, nbr_t.data_ptr(), ctx.n_in, nbr_t.shape[1], gx.data_ptr(), _stream()))
        return gx, None, None


class GlobalAvgPoolFunction(torch.autograd.Function):
    """MinkowskiGlobalAvgPooling (reference resnet.py:15-22,175; A10)."""

    @staticmethod
    def forward(ctx, x, boff):
        x = _f32c(x)
        B, C = boff.numel() - 1, x.shape[1]
        y = torch.empty(B, C, dtype=torch.float32, device=x.device)
        check(lib().mink_global_avg_fwd(x.data_ptr(), C, boff.data_ptr(), B, y.data_ptr(), _stream()))
        ctx.boff, ctx.n = boff, x.shape[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _f32c(gy)
        B, C = gy.shape
        gx = torch.empty(ctx.n, C, dtype=torch.float32, device=gy.device)
        check(lib().mink_global_avg_bwd(gy.data_ptr(), C, ctx.boff.data_ptr(), B, ctx.n, gx.data_ptr(), _stream()))
        return gx, None


def segment_mean(x, members, seg, n_out):
    x = _f32c(x)
    y = torch.empty(n_out, x.shape[1], dtype=torch.float32, device=x.device)
    check(
        lib().mink_segment_mean(x.data_ptr(), x.stride(0), x.shape[1], members.data_ptr(), seg.data_ptr(), n_out, y.data_ptr(), _stream())
    )
    return y


# ----------------------------------------------------------------------- classifier head
class GlobalAvgLinearFunction(torch.autograd.Function):
    """logits = MinkowskiGlobalAvgPooling(x) @ kernel + bias in one launch each way (mink_head_forward/backward):
    the tail of the reference network, `self.final(self.glob_avg(out))` (models/mink/resnet.py:175-177)."""

    @staticmethod
    def forward(ctx, x, boff, kernel, bias):
        x = _f32c(x)
        B, C, ncls = boff.numel() - 1, x.shape[1], kernel.shape[1]
        pooled = torch.empty(B, C, dtype=torch.float32, device=x.device)
        logits = torch.empty(B, ncls, dtype=torch.float32, device=x.device)
        w = kernel.contiguous()
        check(lib().mink_head_forward(x.data_ptr(), boff.data_ptr(), B, C, w.data_ptr(), _ptr(bias), ncls, pooled.data_ptr(),
                                      logits.data_ptr(), _stream()))
        ctx.save_for_backward(pooled, w, boff)
        ctx.n, ctx.has_bias, ctx.bias_shape = x.shape[0], bias is not None, None if bias is None else bias.shape
        ctx.params = (kernel, bias) if (w is kernel and (bias is None or bias.is_contiguous())) else None
        return logits

    @staticmethod
    def backward(ctx, gl):
        pooled, w, boff = ctx.saved_tensors
        gl = _f32c(gl)
        B, C = pooled.shape
        ncls = w.shape[1]
        gx = torch.empty(ctx.n, C, dtype=torch.float32, device=gl.device) if ctx.needs_input_grad[0] else None
        # a data-parallel reducer's flat buffer as the gradient sink: the kernel writes dW / db in place (no accumulate launches)
        views = None
        if _GRAD_SINK is not None and ctx.params is not None:
            views = _sink_views(*[p for p in ctx.params if p is not None])
        if views is not None:
            gw, gb = views[0], (views[1] if ctx.has_bias else None)
        else:
            gw = torch.empty_like(w)
            gb = torch.empty(ctx.bias_shape, dtype=torch.float32, device=gl.device) if ctx.has_bias else None
        check(lib().mink_head_backward(gl.data_ptr(), pooled.data_ptr(), w.data_ptr(), boff.data_ptr(), B, C, ncls, gw.data_ptr(),
                                       _ptr(gb), _ptr(gx), _stream()))
        if views is not None:
            for p in ctx.params:
                if p is not None:
                    _GRAD_SINK.ready(p)
            return gx, None, None, None
        return gx, None, gw, gb


def global_avg_linear(x, batch_offsets, kernel, bias=None):
    return GlobalAvgLinearFunction.apply(x, batch_offsets, kernel, bias)


class SoftmaxCrossEntropyFunction(torch.autograd.Function):
    """F.cross_entropy(logits, labels) with its defaults (mean over the batch) as one launch each way."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = _f32c(logits)
        labels = labels.contiguous()
        assert labels.dtype == torch.int64 and labels.shape == (logits.shape[0],), "labels: int64 [B]"
        B, ncls = logits.shape
        prob = torch.empty_like(logits)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        check(lib().mink_softmax_ce_forward(logits.data_ptr(), labels.data_ptr(), B, ncls, prob.data_ptr(), loss.data_ptr(), _stream()))
        ctx.save_for_backward(prob, labels)
        return loss

    @staticmethod
    def backward(ctx, g):
        prob, labels = ctx.saved_tensors
        g = _f32c(g)
        dl = torch.empty_like(prob)
        check(lib().mink_softmax_ce_backward(prob.data_ptr(), labels.data_ptr(), g.data_ptr(), prob.shape[0], prob.shape[1],
                                             dl.data_ptr(), _stream()))
        return dl, None


def cross_entropy(logits, labels):
    """Mean softmax cross-entropy; HIP kernels for device logits, torch otherwise (CPU oracle runs of the same trainer)."""
    if logits.is_cuda and logits.dim() == 2 and labels.dim() == 1:
        return SoftmaxCrossEntropyFunction.apply(logits, labels)
    return torch.nn.functional.cross_entropy(logits, labels)
