"""The Mink-ResNet trunk (stem + BasicBlocks) as ONE autograd node whose forward and backward are ONE native call each
(`mink_net_forward` / `mink_net_backward`, include/mink_hip.h) instead of one Python autograd node and one FFI call per
operator -- or, as until round 3, one call per residual block.  Same kernels in the same order as the module-by-module
path (which stays, and is what the tests compare this against bit for bit -- with the one-launch batch norm of the few-row
layers, `mink_bn_set_small`, on BOTH sides or off on both: its sums run in another order than the three-launch norm's, and the
module path only takes it where functional.py does; the bitwise tests pin the setting); what changes is the host cost: ~300 launches
per training step at ~20 us of Python each were as slow as the GPU's own 4 ms; one call per block still cost Mink-ResNet34
at four scenes per GPU (BASELINE config #3's per-GPU shape) 3.6 ms of host per 4.7 ms step.  Here the host's per-step work
is a handful of per-LEVEL map records, two allocations and two calls, whatever the depth.

Reference composition this mirrors: models/mink/resnet.py:58-64,107-161,163-177 (stem, _make_layer, layer1..4) and
modules/resnet_block.py:53-69 (BasicBlock).

The descriptors are persistent ctypes objects: what never changes between steps (weight / batch-norm pointers, shapes,
gradient-sink addresses) is written once; per step only the level records and the stem's few fields are stored."""
import ctypes
import os

import torch

from .._lib import BLOCK_DONE_HOOK, STAGE_HOOK, Exec, LevelMaps, Net
from .._lib import BasicBlock as _BlockDesc
from .._lib import check, lib
from . import functional as Fn
from .coords import CoordinateMapKey

# Data parallelism: a block's collective is issued from inside mink_net_backward, on the weight-gradient stream, at the point behind
# which its gradients are complete (mink_set_block_done_hook).  MINK_DP_LAUNCH=stream: the round-4 scheme -- one event per block and
# a stream of its own for the launches, a FIFTH busy hardware queue (the cliff of DESIGN section 6 at GPU_MAX_HW_QUEUES >= 8).
_DP_LAUNCH_IN_CALL = os.environ.get("MINK_DP_LAUNCH", "call") != "stream"
_KEEPALIVE = []  # gradient scratch of the running backward pass: the weight-gradient stream reads it until the final join
Fn._AFTER_JOIN.append(_KEEPALIVE.clear)
_ALIGN = 64  # floats: kNetAlign of csrc/trunk.hip (every block's region starts on a 256-byte boundary)


class _Stage:
    """One residual block: its modules, where its parameters sit in the flat parameter list, its descriptor."""

    __slots__ = ("conv1", "norm1", "conv2", "norm2", "down", "normd", "pidx", "np", "stride", "cin", "C", "desc", "level")


class _Plan:
    __slots__ = ("stages", "params", "norms", "stem", "out_ts", "ptrs", "ex", "sources", "net", "blocks", "levels", "n_levels",
                 "sizes", "grad_key", "events", "need_nbr3")


class _Events:
    """The per-block completion events of a plan (mink_net_backward's done_events); destroyed with the plan -- a model whose plan
    is rebuilt (stale parameters) or dropped no longer leaks them."""

    def __init__(self, L, n):
        self.L, self.n = L, n
        self.arr = (ctypes.c_void_p * n)()
        for i in range(n):
            check(L.mink_event_create(ctypes.byref(self.arr, i * ctypes.sizeof(ctypes.c_void_p))))

    def __del__(self):
        try:
            for i in range(self.n):
                if self.arr[i]:
                    self.L.mink_event_destroy(self.arr[i])
        except Exception:  # (interpreter shutdown: the library may already be gone)
            pass


def plan_for(model):
    """The trunk of a ResNetBase as a list of stages, or None if a layer is not what the native path sequences
    (Bottleneck blocks, biases, dilation, non-affine / synchronised norms ...)."""
    from . import modules as M

    def conv_ok(c, k, stride=None):
        return (type(c) is M.MinkowskiConvolution and c.kernel_size == k and c.bias is None and c.dilation == 1
                and (stride is None or c.stride == stride) and not c.use_mm)

    def norm_ok(n):
        return type(n) is M.MinkowskiBatchNorm and n.bn.affine and n.bn.track_running_stats and n.bn.momentum is not None

    if not (conv_ok(model.conv1, 3, 1) and norm_ok(model.bn1) and type(model.pool) is M.MinkowskiSumPooling
            and model.pool.kernel_size == 2 and model.pool.stride == 2):
        return None
    plan = _Plan()
    params, stages, ts, level = [model.conv1.kernel, model.bn1.bn.weight, model.bn1.bn.bias], [], 2, 0
    for li in range(1, 5):
        for blk in getattr(model, f"layer{li}"):
            if type(blk).__name__ != "BasicBlock" or not (conv_ok(blk.conv1, 3) and conv_ok(blk.conv2, 3, 1)
                                                         and norm_ok(blk.norm1) and norm_ok(blk.norm2)):
                return None
            st = _Stage()
            st.conv1, st.norm1, st.conv2, st.norm2, st.stride = blk.conv1, blk.norm1, blk.conv2, blk.norm2, blk.conv1.stride
            st.down = st.normd = None
            st.cin, st.C = blk.conv1.in_channels, blk.conv1.out_channels
            if st.stride not in (1, 2) or blk.conv2.in_channels != st.C or blk.conv2.out_channels != st.C:
                return None
            if blk.downsample is not None:
                d, dn = blk.downsample[0], blk.downsample[1]
                if not (conv_ok(d, 1, 2) and st.stride == 2 and norm_ok(dn) and d.in_channels == st.cin and d.out_channels == st.C):
                    return None
                st.down, st.normd = d, dn
            elif st.stride != 1 or st.cin != st.C:
                return None
            st.pidx = len(params)
            params += [st.conv1.kernel, st.norm1.bn.weight, st.norm1.bn.bias, st.conv2.kernel, st.norm2.bn.weight, st.norm2.bn.bias]
            if st.down is not None:
                params += [st.down.kernel, st.normd.bn.weight, st.normd.bn.bias]
            st.np = len(params) - st.pidx
            ts *= st.stride
            level += st.stride == 2
            st.level = level
            stages.append(st)
    plan.stages, plan.params, plan.out_ts = stages, params, ts
    # where every parameter lives (module, attribute), in the order of `params`: `stale()` compares identities
    src = [(model.conv1, "kernel"), (model.bn1.bn, "weight"), (model.bn1.bn, "bias")]
    for st in stages:
        src += [(st.conv1, "kernel"), (st.norm1.bn, "weight"), (st.norm1.bn, "bias"), (st.conv2, "kernel"), (st.norm2.bn, "weight"),
                (st.norm2.bn, "bias")]
        if st.down is not None:
            src += [(st.down, "kernel"), (st.normd.bn, "weight"), (st.normd.bn, "bias")]
    plan.sources = src
    plan.norms = [model.bn1] + [n for s in stages for n in (s.norm1, s.norm2, s.normd) if n is not None]
    # the native descriptors: one MinkNet over a contiguous array of blocks, one level record per tensor stride
    plan.blocks = (_BlockDesc * len(stages))()
    for i, st in enumerate(stages):
        st.desc = plan.blocks[i]  # (a view of the array element)
    plan.net = Net()
    plan.net.blocks = ctypes.cast(plan.blocks, ctypes.POINTER(_BlockDesc))
    plan.net.n_blocks, plan.net.with_stem = len(stages), 1
    plan.stem = plan.net.stem  # (a view of the embedded MinkStem)
    plan.n_levels = level + 1
    plan.levels = (LevelMaps * plan.n_levels)()
    plan.need_nbr3 = [any(s.level == lv for s in stages) for lv in range(plan.n_levels)]
    plan.need_nbr3[0] = any(s.level == 0 for s in stages)  # (only a stride-1 block on the stem's own level needs it)
    plan.ptrs, plan.ex, plan.sizes, plan.grad_key, plan.events = None, Exec(), {}, None, None
    return plan


def _fill_norm(nd, norm):
    bn = norm.bn
    nd.gamma, nd.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
    nd.running_mean, nd.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    nd.momentum, nd.eps = float(bn.momentum), float(bn.eps)


def _fill_static(plan):
    """Everything of the descriptors that does not change from step to step."""
    sd = plan.stem
    w0 = plan.params[0]
    sd.conv.K, sd.conv.cout, sd.conv.stride = w0.shape[0], w0.shape[2], 1
    _fill_norm(sd.norm, plan.norms[0])
    for st in plan.stages:
        d = st.desc
        p = plan.params[st.pidx : st.pidx + st.np]
        d.conv1.w, d.conv1.K, d.conv1.cin, d.conv1.cout, d.conv1.stride = p[0].data_ptr(), 27, st.cin, st.C, st.stride
        d.conv2.w, d.conv2.K, d.conv2.cin, d.conv2.cout, d.conv2.stride = p[3].data_ptr(), 27, st.C, st.C, 1
        _fill_norm(d.norm1, st.norm1)
        _fill_norm(d.norm2, st.norm2)
        if st.down is not None:
            d.down.w, d.down.K, d.down.cin, d.down.cout, d.down.stride = p[6].data_ptr(), 1, st.cin, st.C, 2
            _fill_norm(d.normd, st.normd)


def stale(plan):
    """True when a module no longer holds the Parameter object the plan captured (load_state_dict(assign=True),
    torch.utils.swap_tensors, a re-assigned attribute): the plan must be rebuilt -- its kernels would read, and its
    gradients go to, tensors the model no longer owns."""
    for (mod, name), p in zip(plan.sources, plan.params):
        if getattr(mod, name) is not p:
            return True
    return False


def _refresh(plan):
    """Re-write the static part when a parameter or buffer has moved in place (.to(), .data swaps) or a norm's
    momentum / eps was changed."""
    ptrs = tuple(p.data_ptr() for p in plan.params) + tuple(n.bn.running_mean.data_ptr() for n in plan.norms) + \
        tuple((n.bn.momentum, n.bn.eps) for n in plan.norms)
    if ptrs != plan.ptrs:
        _fill_static(plan)
        plan.ptrs = ptrs


def usable(model, plan, x):
    """Training-mode batch norm everywhere, an input that needs no gradient, a stem the fused kernels accept."""
    if plan is None or not x.F.is_cuda or x.F.requires_grad or x.F.dtype != torch.float32:
        return False
    for n in plan.norms:
        if not n.bn.training:
            return False
    w0 = plan.params[0]
    cin = x.F.shape[1] + (-x.F.shape[1]) % 4
    return bool(lib().mink_stem_supported(x.F.shape[0], cin, w0.shape[-1], w0.shape[0]))


def out_key_of(plan):
    return CoordinateMapKey(plan.out_ts)


def _table(m, ts_in, ts_out, ks, transposed=False):
    """Neighbour table of the manager: straight from its cache once the batch's maps were prepared ahead, through the
    recording accessor otherwise (the request trace is what the next batch's plan is compiled from)."""
    if m.prepared:
        ent = m.tables.get((ts_in, ts_out, ks, 1))
        if ent is not None and (not transposed or ent[1] is not None):
            return ent
    return m.kernel_table(CoordinateMapKey(ts_in), CoordinateMapKey(ts_out), ks, 1, transposed=transposed)


def _branch(dev, cur):
    """The stream of the shortcut branch: its own, or (data parallelism) the weight-gradient stream."""
    return Fn._side_stream(dev) if Fn.trunk_branch_mode() == "side" else Fn.branch_stream(dev, home=cur)


def _exec(plan, cur, br, side, nbytes, device):
    ws_c = Fn._scratch(nbytes, device, "trunk")
    ws_b = Fn._scratch(nbytes, device, "trunk", br) if br != cur else ws_c
    ws_s = Fn._scratch(nbytes, device, "trunk", side) if side != cur else ws_c
    ex = plan.ex
    ex.compute, ex.branch, ex.wgrad = cur.cuda_stream, br.cuda_stream, side.cuda_stream
    ex.ws_compute, ex.ws_branch, ex.ws_wgrad = ws_c.data_ptr(), ws_b.data_ptr(), ws_s.data_ptr()
    ex.ws_bytes = min(ws_c.numel(), ws_b.numel(), ws_s.numel())
    return ctypes.byref(ex)


_STEM_WS = {}


def _stem_ws(L, n0, cin, C0):
    key = (n0, cin, C0)
    v = _STEM_WS.get(key)
    if v is None:
        if len(_STEM_WS) > 4096:
            _STEM_WS.clear()
        v = _STEM_WS[key] = int(L.mink_block_workspace_bytes(n0, n0, cin, C0))
    return v


def _sizes(L, plan, rows):
    """(activation floats, gradient floats, scratch bytes per stream) of the blocks for these level row counts."""
    v = plan.sizes.get(rows)
    if v is None:
        if len(plan.sizes) > 4096:
            plan.sizes.clear()
        a, g, w = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(L.mink_net_sizes(ctypes.byref(plan.net), plan.levels, plan.n_levels, ctypes.byref(a), ctypes.byref(g), ctypes.byref(w)))
        v = plan.sizes[rows] = (a.value, g.value, w.value)
    return v


# ---- instrumentation between the stages of a native call (race tests: skewed auxiliary streams; bench.py --timeline)
_HOOK_STATE = {"installed": False, "cur": None, "fwd_skew": (), "bwd_skew": ()}


def _stage_hook(stage, backward):
    st = _HOOK_STATE
    if Fn._PREPARE_GATE and Fn._PREPARE_GATE == (int(bool(backward)), stage):
        # (the next batches' map builds start beside THIS point of a step and not before: functional._PREPARE_GATE)
        Fn.note_prepare_gate(st["cur"])
    if backward:
        if stage < 0:
            Fn.log_phase("stem_backward_begin", st["cur"])
        else:
            for s_ in st["bwd_skew"]:
                Fn.skew(s_)
    elif stage < 0:
        Fn.log_phase("stem_forward_begin", st["cur"])
    else:
        if stage == 0:
            Fn.log_phase("stem_forward", st["cur"])
        for s_ in st["fwd_skew"]:
            Fn.skew(s_)


_HOOK_C = STAGE_HOOK(_stage_hook)


def _arm_hook(L, cur, fwd_skew=(), bwd_skew=(), n_stages=None):
    if Fn._PREPARE_GATE and n_stages is not None and not _HOOK_STATE.get("gate_checked"):
        _HOOK_STATE["gate_checked"] = True
        if not -1 <= Fn._PREPARE_GATE[1] < n_stages:  # (a gate point no pass ever reaches: the map builds would simply never be held back)
            import warnings

            warnings.warn(f"MINK_PREPARE_GATE names block {Fn._PREPARE_GATE[1]} of a trunk with {n_stages} blocks: the gate never fires "
                          "and the map builds follow the host as without it")
    want = bool(Fn._SKEW) or Fn._PHASE_LOG is not None or bool(Fn._PREPARE_GATE)
    st = _HOOK_STATE
    if want:
        st["cur"], st["fwd_skew"], st["bwd_skew"] = cur, fwd_skew, bwd_skew
    if want != st["installed"]:
        check(L.mink_set_stage_hook(ctypes.cast(_HOOK_C, ctypes.c_void_p) if want else None))
        st["installed"] = want


def _align(v):
    return -(-v // _ALIGN) * _ALIGN


class _Saved:
    """What the forward pass keeps for backward, indexable the way the tests read it: [0] = the stem's tuple
    (x, w0, arena0, nbr0, nbr_pool, i2o, pad, bf16 storage?, bf16 input copy), [1 + i] = block i's
    (activations [y1 | h1 | y2 | out | ...] as a view of the one arena, nbr1, nbr2, nbrd, ts_in, ts_out, n_in, n_out)."""

    def __init__(self, stem, arena, rows, tables, plan):
        self.stem, self.arena, self.rows, self.tables, self.plan = stem, arena, rows, tables, plan

    def __len__(self):
        return 1 + len(self.plan.stages)

    def _stage(self, i):
        off = 0
        for j, st in enumerate(self.plan.stages):
            n_out = self.rows[st.level]
            width = (6 if st.down is not None else 4) * n_out * st.C
            if j == i:
                lv_in = st.level - (st.stride == 2)
                t = self.tables[st.level]
                return (self.arena[off : off + width], t[1] if st.stride == 2 else t[0], t[0], t[2] if st.stride == 2 else None,
                        2 << lv_in, 2 << st.level, self.rows[lv_in], n_out)
            off += _align(width) + _align(6 * st.C)
        raise IndexError(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return self.stem if i == 0 else self._stage(i - 1)


class TrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, manager, fork, *params):
        L = lib()
        dev = x.device
        m = manager
        _refresh(plan)
        # ---- stem: its activations are laid out here (their size depends on the storage type)
        w0 = params[0]
        cin = x.shape[1]
        pad = (-cin) % 4
        if pad:  # e.g. features=["sh"] (27 channels): one zero column puts the rows on 16-byte boundaries
            x = torch.nn.functional.pad(x, (0, pad))
            w0 = torch.nn.functional.pad(w0, (0, 0, 0, pad))
        x = x.contiguous()
        cin = x.shape[1]
        C0 = w0.shape[-1]
        nbr0 = _table(m, 1, 1, 3)[0]
        m.stride(CoordinateMapKey(1), 2)
        nbr_pool = _table(m, 1, 2, 2)[0]
        i2o = m.in2out[(1, 2)] if m.prepared else m.stride_map(CoordinateMapKey(1), CoordinateMapKey(2))
        n0, n1 = x.shape[0], nbr_pool.shape[0]
        cur = Fn.current_stream(dev)
        br = _branch(dev, cur) if (fork and m.prepared) else cur
        # ---- the level records of this batch (level l = tensor stride 2 << l)
        lv, tables, rows = plan.levels, [], []
        for li in range(plan.n_levels):
            ts = 2 << li
            if li:
                m.stride(CoordinateMapKey(ts >> 1), 2)
            rec = lv[li]
            n = m.levels[ts].n
            rec.n = n
            rows.append(n)
            t3 = _table(m, ts, ts, 3)[0] if plan.need_nbr3[li] else None
            rec.nbr3 = t3.data_ptr() if t3 is not None else None
            if li:
                d3, d1 = _table(m, ts >> 1, ts, 3)[0], _table(m, ts >> 1, ts, 1)[0]
                rec.down3, rec.down1 = d3.data_ptr(), d1.data_ptr()
                tables.append((t3, d3, d1))
            else:
                rec.down3 = rec.down1 = None
                tables.append((t3, None, None))
            rec.down3_t = rec.perm = None
            rec.n_perm = 0
        rows = tuple(rows)
        act_floats, _, ws_net = _sizes(L, plan, rows)
        exp = _exec(plan, cur, br, cur, max(ws_net, _stem_ws(L, n0, cin, C0)), dev)
        # bf16 storage of the full-resolution stage (Fn.set_conv_storage): y in bf16 and a bf16 copy of x behind it
        b16 = bool(Fn._STORAGE_B16 and Fn.conv_math() == "bf16" and L.mink_stem_conv_bf16s_supported(n0, n0, w0.shape[0], cin, C0))
        ny = (n0 * C0 // 2 + 16 * n0 + 3) // 4 * 4 if b16 else n0 * C0  # floats of the y (+ xb) region: 16-byte multiples
        arena0 = torch.empty(ny + n1 * C0 + 2 * C0, dtype=torch.float32, device=dev)
        a0 = arena0.data_ptr()
        sd = plan.stem
        sd.conv.w, sd.conv.dw, sd.conv.nbr, sd.conv.cin = w0.data_ptr(), None, nbr0.data_ptr(), cin
        sd.norm.mean, sd.norm.invstd = a0 + 4 * (ny + n1 * C0), a0 + 4 * (ny + n1 * C0 + C0)
        sd.nbr_pool, sd.in2out, sd.n, sd.n_pool = nbr_pool.data_ptr(), i2o.data_ptr(), n0, n1
        sd.x, sd.y, sd.out, sd.g_out = x.data_ptr(), a0, a0 + 4 * ny, None
        sd.xb, sd.xb_ready = (a0 + 2 * n0 * C0 if b16 else None), 0
        xb_pre = getattr(m, "xb", None)
        if b16 and xb_pre is not None and xb_pre[0] == x.data_ptr() and xb_pre[1].shape[0] == n0 and cin <= 32:
            sd.xb, sd.xb_ready = xb_pre[1].data_ptr(), 1  # made beside the previous step (TensorField.finish)
        else:
            xb_pre = None
        arena = torch.empty(act_floats, dtype=torch.float32, device=dev)
        _arm_hook(L, cur, fwd_skew=(br,) if (Fn._SKEW and br != cur) else (), n_stages=len(plan.stages))
        net = plan.net
        check(L.mink_net_forward(ctypes.byref(net), lv, plan.n_levels, arena.data_ptr(), act_floats, exp))
        if Fn._TIMING_MODE == 1:
            Fn.note_table(nbr0, *[t for ts_ in tables for t in ts_])
        last = plan.stages[-1]
        n_out = net.out_rows
        off = (net.out - arena.data_ptr()) // 4
        out = arena[off : off + n_out * last.C].view(n_out, last.C)
        ctx.plan, ctx.manager, ctx.fork = plan, m, fork
        ctx.saved = _Saved((x, w0, arena0, nbr0, nbr_pool, i2o, pad, b16, xb_pre), arena, rows, tables, plan)
        ctx.params = params  # keys of the data-parallel gradient sink; deciding where the weight gradients may run
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = lib()
        plan, m, saved, params = ctx.plan, ctx.manager, ctx.saved, ctx.params
        dev = g_out.device
        g_out = Fn._f32c(g_out)
        sink = Fn._GRAD_SINK
        x, w0p, arena0, nbr0, nbr_pool, i2o, pad, b16, xb_pre = saved.stem
        arena, rows, tables = saved.arena, saved.rows, saved.tables
        views = Fn._sink_views(*params) if (sink is not None and pad == 0) else None
        # the weight gradients may run on the side stream (joined once, at the end of backward) when nothing consumes a
        # gradient earlier: the gradient buffer's owner (the sink) has them written in place, or autograd merely
        # installs them as .grad
        overlap = Fn._OVERLAP_WGRAD
        if overlap and views is None:
            for p in params:
                if p.grad is not None or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                    overlap = False
                    break
            overlap = overlap and not torch.is_grad_enabled()
        cur = Fn.current_stream(dev)
        br = _branch(dev, cur) if (ctx.fork and m.prepared) else cur
        side = Fn._side_stream(dev) if overlap else cur
        C0 = w0p.shape[-1]
        n0 = x.shape[0]
        # ---- the level records again (another forward pass of this model may have run in between), with the transposed
        # tables and parity-class orders the strided data gradients use
        lv = plan.levels
        bwd_tables = []
        for li in range(plan.n_levels):
            rec = lv[li]
            t3, d3, d1 = tables[li]
            rec.n = rows[li]
            rec.nbr3 = t3.data_ptr() if t3 is not None else None
            if li:
                ts = 2 << li
                rec.down3, rec.down1 = d3.data_ptr(), d1.data_ptr()
                nbr_t = _table(m, ts >> 1, ts, 3, transposed=True)[1]
                perm = m.tables.get(("perm", ts >> 1, 128)) if m.prepared else None
                if perm is None:
                    perm = m.class_perm(CoordinateMapKey(ts >> 1))
                rec.down3_t, rec.perm, rec.n_perm = nbr_t.data_ptr(), perm.data_ptr(), perm.numel()
                bwd_tables += [nbr_t, perm]
            else:
                rec.down3 = rec.down1 = rec.down3_t = rec.perm = None
                rec.n_perm = 0
        act_floats, grad_floats, ws_net = _sizes(L, plan, rows)
        exp = _exec(plan, cur, br, side, max(ws_net, _stem_ws(L, n0, x.shape[1], C0)), dev)
        if overlap:
            Fn._defer_join()
        # ---- where the parameter gradients go
        grads = [None] * len(params)
        flat = None
        if views is not None:
            gkey = (id(sink), views[0].data_ptr(), views[-1].data_ptr())
            ptrs = None if plan.grad_key == gkey else [v.data_ptr() for v in views]
            plan.grad_key = gkey
        else:  # one fresh buffer for all parameter gradients of the trunk, handed to autograd as views of it
            sizes = [_align(p.numel()) for p in params]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            if side != cur:
                flat.record_stream(side)
            base, ptrs, off = flat.data_ptr(), [], 0
            for i, p in enumerate(params):
                ptrs.append(base + 4 * off)
                if not (i == 0 and pad):
                    grads[i] = flat[off : off + p.numel()].view(p.shape)
                off += sizes[i]
            plan.grad_key = None
        gw_pad = None
        if ptrs is not None:
            for st in plan.stages:
                d, q = st.desc, ptrs[st.pidx : st.pidx + st.np]
                d.conv1.dw, d.norm1.dgamma, d.norm1.dbeta, d.conv2.dw, d.norm2.dgamma, d.norm2.dbeta = q[:6]
                if st.down is not None:
                    d.down.dw, d.normd.dgamma, d.normd.dbeta = q[6:9]
        sd = plan.stem
        if sd.x != x.data_ptr() or sd.y != arena0.data_ptr():
            TrunkFunction._restore_stem(plan, saved.stem)
        if views is not None:
            sd.conv.dw, sd.norm.dgamma, sd.norm.dbeta = views[0].data_ptr(), views[1].data_ptr(), views[2].data_ptr()
        else:
            if pad:  # the padded stem kernel's gradient has its own buffer; the caller gets the un-padded columns
                gw_pad = torch.empty(w0p.shape, dtype=torch.float32, device=dev)
            sd.conv.dw = gw_pad.data_ptr() if pad else ptrs[0]
            sd.norm.dgamma, sd.norm.dbeta = ptrs[1], ptrs[2]
        g_buf = torch.empty(grad_floats, dtype=torch.float32, device=dev)
        collect = views is not None and (getattr(sink, "_collect", False) or getattr(sink, "_bucket_cb", None) is not None)
        evp, hook_err = None, []
        in_call = collect and _DP_LAUNCH_IN_CALL and hasattr(sink, "stage_stream")
        if in_call:
            # the library calls back after every block, at the point of its weight-gradient stream behind which that block's
            # gradients are complete: the block is reported, and a bucket it completes is all-reduced, right there
            many = getattr(sink, "ready_many", None)
            stages = plan.stages

            def _block_done(si):
                try:
                    st = stages[si]
                    sink.stage_stream = side
                    if many is not None:
                        many(params[st.pidx : st.pidx + st.np])
                    else:
                        for i in range(st.pidx, st.pidx + st.np):
                            sink.ready(params[i])
                except BaseException as exc:  # (an exception must not unwind through the native frame)
                    hook_err.append(exc)
                finally:
                    sink.stage_stream = None

            cb = BLOCK_DONE_HOOK(_block_done)
            check(L.mink_set_block_done_hook(ctypes.cast(cb, ctypes.c_void_p)))
        elif collect:
            if plan.events is None:
                plan.events = _Events(L, len(plan.stages))
            evp = plan.events.arr
        _arm_hook(L, cur, bwd_skew=tuple({br, side} - {cur}) if Fn._SKEW else ())
        try:
            check(L.mink_net_backward(ctypes.byref(plan.net), lv, plan.n_levels, arena.data_ptr(), act_floats, g_out.data_ptr(),
                                      g_buf.data_ptr(), grad_floats, exp, evp))
        finally:
            if in_call:
                L.mink_set_block_done_hook(None)
        if hook_err:
            raise hook_err[0]
        Fn.log_phase("stem_backward_end", cur)
        if Fn._TIMING_MODE == 1:
            Fn.note_table(nbr0, *[t for ts_ in tables for t in ts_], *bwd_tables)
        _KEEPALIVE.append((g_buf, g_out, flat, gw_pad, bwd_tables))  # (never the gradients handed to autograd: a second reference makes it clone them at once)
        if in_call:
            sink.flush()
            for i in range(3):
                sink.ready(params[i])
        elif collect:
            # every block's gradients are queued: report them in backward order, each with the event behind which they are
            # complete, so that a bucket's all-reduce waits for exactly its own blocks and overlaps the rest of backward
            many = getattr(sink, "ready_many", None)
            for si in range(len(plan.stages) - 1, -1, -1):
                st = plan.stages[si]
                sink.stage_event = evp[si]
                if many is not None:
                    many(params[st.pidx : st.pidx + st.np])
                else:
                    for i in range(st.pidx, st.pidx + st.np):
                        sink.ready(params[i])
            sink.stage_event = None
            sink.flush()  # everything complete so far goes out beside the stem's weight gradient (0.9 ms)
            for i in range(3):
                sink.ready(params[i])
        if views is None and pad:
            grads[0] = gw_pad[:, : gw_pad.shape[1] - pad].contiguous()
        if not overlap:
            _KEEPALIVE.clear()
        return (None, None, None, None, *grads)

    @staticmethod
    def _restore_stem(plan, sv):
        x, w0, arena0, nbr0, nbr_pool, i2o, pad, b16, xb_pre = sv
        sd = plan.stem
        n0, n1, C0 = x.shape[0], nbr_pool.shape[0], w0.shape[-1]
        a0 = arena0.data_ptr()
        ny = arena0.numel() - n1 * C0 - 2 * C0
        sd.conv.w, sd.conv.nbr, sd.conv.cin = w0.data_ptr(), nbr0.data_ptr(), x.shape[1]
        sd.norm.mean, sd.norm.invstd = a0 + 4 * (ny + n1 * C0), a0 + 4 * (ny + n1 * C0 + C0)
        sd.nbr_pool, sd.in2out, sd.n, sd.n_pool = nbr_pool.data_ptr(), i2o.data_ptr(), n0, n1
        sd.xb = (xb_pre[1].data_ptr() if xb_pre is not None else a0 + 2 * n0 * C0) if b16 else None
        sd.x, sd.y, sd.out = x.data_ptr(), a0, a0 + 4 * ny
