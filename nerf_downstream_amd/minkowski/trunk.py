"""The Mink-ResNet trunk (stem + BasicBlocks) as ONE autograd node whose forward / backward issue one native call
per stage (`mink_stem_*` / `mink_block_*`, include/mink_hip.h) instead of one Python autograd node and one FFI call
per operator.  Same kernels in the same order as the module-by-module path (which stays, and is what the tests
compare this against bit for bit); what changes is the host cost: ~300 launches per training step at ~20 us of
Python each were as slow as the GPU's own 4 ms, here a stage costs one descriptor + ~12 x hipLaunchKernel.

Reference composition this mirrors: models/mink/resnet.py:58-64,163-177 (stem, layer1..4) and
modules/resnet_block.py:53-69 (BasicBlock).

Host cost is the whole point, so the descriptors are persistent ctypes objects: what never changes between steps
(weight / batch-norm pointers, shapes) is written once, per step only the map pointers, row counts and activation
addresses are stored -- plain integer arithmetic on one arena per stage, no tensor views."""
import ctypes

import torch

from .._lib import BasicBlock as _BlockDesc
from .._lib import Exec
from .._lib import Stem as _StemDesc
from .._lib import check, lib
from . import functional as Fn
from .coords import CoordinateMapKey

_KEEPALIVE = []  # gradient scratch of the running backward pass: the weight-gradient stream reads it until the final join
Fn._AFTER_JOIN.append(_KEEPALIVE.clear)


class _Stage:
    """One residual block: its modules, where its parameters sit in the flat parameter list, its descriptor."""

    __slots__ = ("conv1", "norm1", "conv2", "norm2", "down", "normd", "pidx", "np", "stride", "cin", "C", "desc")


class _Plan:
    __slots__ = ("stages", "params", "norms", "stem", "out_ts", "ptrs", "ex", "sources")


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
    params, stages, ts = [model.conv1.kernel, model.bn1.bn.weight, model.bn1.bn.bias], [], 2
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
            st.desc = _BlockDesc()
            ts *= st.stride
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
    plan.stem, plan.ptrs, plan.ex = _StemDesc(), None, Exec()
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


_WS_CACHE = {}


def _ws_need(L, n_in, n_out, cin, cout):
    key = (n_in, n_out, cin, cout)
    v = _WS_CACHE.get(key)
    if v is None:
        if len(_WS_CACHE) > 4096:
            _WS_CACHE.clear()
        v = _WS_CACHE[key] = (int(L.mink_block_workspace_bytes(n_in, n_out, cin, cout)),
                              int(L.mink_block_grad_scratch_floats(n_in, n_out, cin, cout, 0)),
                              int(L.mink_block_grad_scratch_floats(n_in, n_out, cin, cout, 1)))
    return v


class TrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, manager, fork, *params):
        L = lib()
        dev = x.device
        m = manager
        _refresh(plan)
        # ---- stem
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
        cur = torch.cuda.current_stream(dev)
        br = _branch(dev, cur) if (fork and m.prepared) else cur
        # rows per stage and the scratch the largest stage needs
        need, n_in, ts = _ws_need(L, n0, n0, cin, C0)[0], n1, 2
        shapes = []
        for st in plan.stages:
            ts_out = ts * st.stride
            if st.stride != 1:
                m.stride(CoordinateMapKey(ts), st.stride)
            n_out = m.levels[ts_out].n
            shapes.append((ts, ts_out, n_in, n_out))
            need = max(need, _ws_need(L, n_in, n_out, st.cin, st.C)[0])
            ts, n_in = ts_out, n_out
        exp = _exec(plan, cur, br, cur, need, dev)
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
        Fn.log_phase("stem_forward_begin", cur)
        check(L.mink_stem_forward(ctypes.byref(sd), exp))
        Fn.log_phase("stem_forward", cur)
        Fn.note_table(nbr0)
        saved = [(x, w0, arena0, nbr0, nbr_pool, i2o, pad, b16, xb_pre)]
        # ---- residual blocks
        hp = a0 + 4 * ny
        skew = Fn._SKEW and br != cur
        arena = None
        for st, (ts_in, ts_out, n_in, n_out) in zip(plan.stages, shapes):
            nbr1 = _table(m, ts_in, ts_out, 3)[0]
            nbr2 = _table(m, ts_out, ts_out, 3)[0]
            has_down = st.down is not None
            C = st.C
            cnt = n_out * C
            arena = torch.empty((6 if has_down else 4) * cnt + 6 * C, dtype=torch.float32, device=dev)
            a = arena.data_ptr()
            s0 = a + 4 * (6 if has_down else 4) * cnt
            d = st.desc
            d.conv1.nbr, d.conv2.nbr = nbr1.data_ptr(), nbr2.data_ptr()
            d.norm1.mean, d.norm1.invstd, d.norm2.mean, d.norm2.invstd = s0, s0 + 4 * C, s0 + 8 * C, s0 + 12 * C
            d.n_in, d.n_out, d.x = n_in, n_out, hp
            d.y1, d.h1, d.y2, d.out = a, a + 4 * cnt, a + 8 * cnt, a + 12 * cnt
            nbrd = None
            if has_down:
                nbrd = _table(m, ts_in, ts_out, 1)[0]
                d.down.nbr, d.yd, d.sd = nbrd.data_ptr(), a + 16 * cnt, a + 20 * cnt
                d.normd.mean, d.normd.invstd = s0 + 16 * C, s0 + 20 * C
            d.g_out = d.g_x = d.g_tmp = None
            if skew:
                Fn.skew(br)
            check(L.mink_block_forward(ctypes.byref(d), exp))
            Fn.note_table(nbr1, nbr2, nbrd)
            saved.append((arena, nbr1, nbr2, nbrd, ts_in, ts_out, n_in, n_out, hp))
            hp = a + 12 * cnt
        out = arena[3 * cnt : 4 * cnt].view(n_out, C)
        ctx.plan, ctx.manager, ctx.saved, ctx.fork = plan, m, saved, fork
        ctx.params = params  # keys of the data-parallel gradient sink; deciding where the weight gradients may run
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = lib()
        plan, m, saved, params = ctx.plan, ctx.manager, ctx.saved, ctx.params
        dev = g_out.device
        g_out = Fn._f32c(g_out)
        sink = Fn._GRAD_SINK
        x, w0p, arena0, nbr0, nbr_pool, i2o, pad, b16, xb_pre = saved[0]
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
        cur = torch.cuda.current_stream(dev)
        br = _branch(dev, cur) if (ctx.fork and m.prepared) else cur
        side = Fn._side_stream(dev) if overlap else cur
        C0 = w0p.shape[-1]
        n0 = x.shape[0]
        need = _ws_need(L, n0, n0, x.shape[1], C0)[0]
        for st, sv in zip(plan.stages, saved[1:]):
            need = max(need, _ws_need(L, sv[6], sv[7], st.cin, st.C)[0])
        exp = _exec(plan, cur, br, side, need, dev)
        if overlap:
            Fn._defer_join()
        skew = [s_ for s_ in {br, side} - {cur}] if Fn._SKEW else ()
        grads = [None] * len(params)
        g, gp = g_out, g_out.data_ptr()
        for si in range(len(plan.stages) - 1, -1, -1):
            st = plan.stages[si]
            arena, nbr1, nbr2, nbrd, ts_in, ts_out, n_in, n_out, hp = saved[si + 1]
            has_down = st.down is not None
            C, cin = st.C, st.cin
            d = st.desc
            if views is not None:
                gs = views[st.pidx : st.pidx + st.np]
            else:  # one fresh buffer for the stage's parameter gradients, handed to autograd as views of it
                like = params[st.pidx : st.pidx + st.np]
                flat = torch.empty(sum(t.numel() for t in like), dtype=torch.float32, device=dev)
                if side != cur:
                    flat.record_stream(side)
                gs, off = [], 0
                for t in like:
                    gs.append(flat[off : off + t.numel()].view(t.shape))
                    off += t.numel()
            d.conv1.dw, d.norm1.dgamma, d.norm1.dbeta = gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr()
            d.conv2.dw, d.norm2.dgamma, d.norm2.dbeta = gs[3].data_ptr(), gs[4].data_ptr(), gs[5].data_ptr()
            nbr1_t = None
            if st.stride == 2:
                nbr1_t = _table(m, ts_in, ts_out, 3, transposed=True)[1]
                perm1 = m.tables.get(("perm", ts_in, 128)) if m.prepared else None
                if perm1 is None:
                    perm1 = m.class_perm(CoordinateMapKey(ts_in))
                d.conv1.nbr_t, d.conv1.perm, d.conv1.n_perm = nbr1_t.data_ptr(), perm1.data_ptr(), perm1.numel()
            if has_down:  # (its data gradient goes through the forward table: no transposed table)
                d.down.dw = gs[6].data_ptr()
                d.normd.dgamma, d.normd.dbeta = gs[7].data_ptr(), gs[8].data_ptr()
            # (the descriptor still holds this batch's forward fields: the activations it points at are kept in `saved`)
            if d.y1 != arena.data_ptr():  # another forward pass ran on this model in between: restore this batch's fields
                TrunkFunction._restore(st, saved[si + 1])
            ntmp = _ws_need(L, n_in, n_out, cin, C)[2 if has_down else 1]
            g_buf = torch.empty(ntmp + n_in * cin, dtype=torch.float32, device=dev)
            gb = g_buf.data_ptr()
            d.g_out, d.g_tmp, d.g_x = gp, gb, gb + 4 * ntmp
            for s_ in skew:
                Fn.skew(s_)
            check(L.mink_block_backward(ctypes.byref(d), exp))
            Fn.note_table(nbr1, nbr2, nbrd, nbr1_t)
            _KEEPALIVE.append((g_buf, g))  # (never the gradients handed to autograd: a second reference makes it clone them at once)
            if views is not None:
                for i in range(st.pidx, st.pidx + st.np):
                    sink.ready(params[i])
                if C <= 128:  # a wide-and-shallow stage is queued (static rule, the same on every rank): the host has time
                    sink.flush()
            else:
                grads[st.pidx : st.pidx + st.np] = gs
            g, gp = g_buf, gb + 4 * ntmp
        # ---- stem
        if views is not None:
            gs = views[:3]
        else:
            gw = torch.empty(w0p.shape, dtype=torch.float32, device=dev)
            gbn = torch.empty(2 * C0, dtype=torch.float32, device=dev)
            gs = [gw, gbn[:C0], gbn[C0:]]
        sd = plan.stem
        if sd.x != x.data_ptr() or sd.y != arena0.data_ptr():
            TrunkFunction._restore_stem(plan, saved[0])
        sd.conv.dw, sd.norm.dgamma, sd.norm.dbeta, sd.g_out = gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), gp
        Fn.log_phase("stem_backward_begin", torch.cuda.current_stream(dev))
        check(L.mink_stem_backward(ctypes.byref(sd), exp))
        Fn.log_phase("stem_backward_end", torch.cuda.current_stream(dev))
        Fn.note_table(nbr0)
        _KEEPALIVE.append(g)
        if views is not None:
            sink.flush()  # everything complete so far goes out beside the stem's weight gradient (0.9 ms)
            for i in range(3):
                sink.ready(params[i])
        else:
            grads[0] = gs[0][:, : gs[0].shape[1] - pad].contiguous() if pad else gs[0]
            grads[1], grads[2] = gs[1], gs[2]
        if not overlap:
            _KEEPALIVE.clear()
        return (None, None, None, None, *grads)

    @staticmethod
    def _restore(st, sv):
        """Point a stage descriptor back at the batch whose backward is running (a second forward pass of the same
        model overwrote the per-step fields)."""
        arena, nbr1, nbr2, nbrd, ts_in, ts_out, n_in, n_out, hp = sv
        d, C = st.desc, st.C
        cnt = n_out * C
        a = arena.data_ptr()
        s0 = a + 4 * (6 if st.down is not None else 4) * cnt
        d.conv1.nbr, d.conv2.nbr = nbr1.data_ptr(), nbr2.data_ptr()
        d.norm1.mean, d.norm1.invstd, d.norm2.mean, d.norm2.invstd = s0, s0 + 4 * C, s0 + 8 * C, s0 + 12 * C
        d.n_in, d.n_out, d.x = n_in, n_out, hp
        d.y1, d.h1, d.y2, d.out = a, a + 4 * cnt, a + 8 * cnt, a + 12 * cnt
        if st.down is not None:
            d.down.nbr, d.yd, d.sd = nbrd.data_ptr(), a + 16 * cnt, a + 20 * cnt
            d.normd.mean, d.normd.invstd = s0 + 16 * C, s0 + 20 * C

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
