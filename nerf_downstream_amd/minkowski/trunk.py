"""The Mink-ResNet trunk (stem + BasicBlocks) as ONE autograd node whose forward / backward issue one native call
per stage (`mink_stem_*` / `mink_block_*`, include/mink_hip.h) instead of one Python autograd node and one FFI call
per operator.  Same kernels in the same order as the module-by-module path (which stays, and is what the tests
compare this against bit for bit); what changes is the host cost: ~300 launches per training step at ~20 us of
Python each were as slow as the GPU's own 4 ms, here a stage costs one descriptor + ~12 x hipLaunchKernel.

Reference composition this mirrors: models/mink/resnet.py:58-64,163-177 (stem, layer1..4) and
modules/resnet_block.py:53-69 (BasicBlock)."""
import ctypes

import torch

from .._lib import BasicBlock as _BlockDesc
from .._lib import ConvLayer, Exec, NormLayer
from .._lib import Stem as _StemDesc
from .._lib import check, lib
from . import functional as Fn
from .coords import CoordinateMapKey

_KEEPALIVE = []  # gradient scratch of the running backward pass: the weight-gradient stream reads it until the final join


Fn._AFTER_JOIN.append(_KEEPALIVE.clear)


class _Stage:
    """Static description of one stage (built once per model): modules and parameter positions."""

    __slots__ = ("conv1", "norm1", "conv2", "norm2", "down", "normd", "pidx", "stride")


def plan_for(model):
    """The trunk of a ResNetBase as a list of stages, or None if a layer is not what the native path sequences
    (Bottleneck blocks, biases, dilation, non-affine / synchronised norms ...)."""
    from . import modules as M

    def conv_ok(c, k, stride=None):
        return (type(c) is M.MinkowskiConvolution and c.kernel_size == k and c.bias is None and c.dilation == 1
                and (stride is None or c.stride == stride) and not c.use_mm)

    def norm_ok(n):
        return type(n) is M.MinkowskiBatchNorm and n.bn.affine and n.bn.track_running_stats

    if not (conv_ok(model.conv1, 3, 1) and norm_ok(model.bn1) and type(model.pool) is M.MinkowskiSumPooling
            and model.pool.kernel_size == 2 and model.pool.stride == 2):
        return None
    params, stages = [model.conv1.kernel, model.bn1.bn.weight, model.bn1.bn.bias], []
    for li in range(1, 5):
        for blk in getattr(model, f"layer{li}"):
            if type(blk).__name__ != "BasicBlock" or not (conv_ok(blk.conv1, 3) and conv_ok(blk.conv2, 3, 1)
                                                         and norm_ok(blk.norm1) and norm_ok(blk.norm2)):
                return None
            st = _Stage()
            st.conv1, st.norm1, st.conv2, st.norm2, st.stride = blk.conv1, blk.norm1, blk.conv2, blk.norm2, blk.conv1.stride
            st.down = st.normd = None
            if blk.downsample is not None:
                d, dn = blk.downsample[0], blk.downsample[1]
                if not (conv_ok(d, 1, blk.conv1.stride) and norm_ok(dn) and d.stride in (1, 2)) or d.stride == 1:
                    return None
                st.down, st.normd = d, dn
            elif blk.conv1.stride != 1 or blk.conv1.in_channels != blk.conv1.out_channels:
                return None
            if st.stride not in (1, 2):
                return None
            st.pidx = len(params)
            params += [st.conv1.kernel, st.norm1.bn.weight, st.norm1.bn.bias, st.conv2.kernel, st.norm2.bn.weight, st.norm2.bn.bias]
            if st.down is not None:
                params += [st.down.kernel, st.normd.bn.weight, st.normd.bn.bias]
            stages.append(st)
    return {"stages": stages, "params": params, "norms": [model.bn1] + [n for s in stages for n in (s.norm1, s.norm2, s.normd) if n is not None]}


def usable(model, plan, x):
    """Training-mode batch norm everywhere, an input that needs no gradient, 16-byte rows."""
    if plan is None or not x.F.is_cuda or x.F.requires_grad or x.F.dtype != torch.float32:
        return False
    if not all(n.bn.training for n in plan["norms"]):
        return False
    w0 = plan["params"][0]
    cin = x.F.shape[1] + (-x.F.shape[1]) % 4
    return bool(lib().mink_stem_supported(x.F.shape[0], cin, w0.shape[-1], w0.shape[0]))


def out_key_of(plan):
    ts = 2
    for st in plan["stages"]:
        ts *= st.stride
    return CoordinateMapKey(ts)


def _p(t):
    return None if t is None else t.data_ptr()


def _norm_desc(norm, stat, C, grads=None):
    bn = norm.bn
    from .modules import _bn_momentum

    g = grads or (None, None)
    return NormLayer(bn.weight.data_ptr(), bn.bias.data_ptr(), _p(bn.running_mean), _p(bn.running_var), _p(g[0]), _p(g[1]),
                     stat.data_ptr(), stat.data_ptr() + 4 * C, float(_bn_momentum(bn)), float(bn.eps))


def _exec(device, fork, overlap):
    """Streams and per-stream scratch of this call."""
    cur = torch.cuda.current_stream(device)
    br = Fn.branch_stream(device, home=cur) if fork else cur
    side = Fn._side_stream(device) if overlap else cur
    return cur, br, side


def _exec_desc(cur, br, side, nbytes, device):
    ws_c = Fn._scratch(nbytes, device, "trunk")
    ws_b = Fn._scratch(nbytes, device, "trunk", br) if br != cur else ws_c
    ws_s = Fn._scratch(nbytes, device, "trunk", side) if side != cur else ws_c
    return Exec(cur.cuda_stream, br.cuda_stream, side.cuda_stream, ws_c.data_ptr(), ws_b.data_ptr(), ws_s.data_ptr(),
                min(ws_c.numel(), ws_b.numel(), ws_s.numel())), (ws_c, ws_b, ws_s)


class TrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, manager, fork, *params):
        L = lib()
        dev = x.device
        stages = plan["stages"]
        m = manager
        k1 = CoordinateMapKey(1)
        # ---- stem
        w0 = params[0]
        cin = x.shape[1]
        pad = (-cin) % 4
        if pad:  # e.g. features=["sh"] (27 channels): one zero column puts the rows on 16-byte boundaries
            x = torch.nn.functional.pad(x, (0, pad))
            w0 = torch.nn.functional.pad(w0, (0, 0, 0, pad))
        x = x.contiguous()
        C0 = w0.shape[-1]
        nbr0, _ = m.kernel_table(k1, k1, 3, 1)
        k2 = m.stride(k1, 2)
        nbr_pool, _ = m.kernel_table(k1, k2, 2, 1)
        i2o = m.stride_map(k1, k2)
        n0, n1 = x.shape[0], nbr_pool.shape[0]
        cur, br, _ = _exec(dev, fork and getattr(m, "prepared", False), False)
        # scratch requirement: the largest stage
        need, n_in, c_in, key = L.mink_block_workspace_bytes(n0, n0, x.shape[1], C0), n1, C0, k2
        shapes = []
        for st in stages:
            out_key = m.stride(key, st.stride)
            n_out, C = m.size(out_key), st.conv1.out_channels
            shapes.append((key, out_key, n_in, n_out, c_in, C))
            need = max(need, L.mink_block_workspace_bytes(n_in, n_out, c_in, C))
            key, n_in, c_in = out_key, n_out, C
        ex, ws_refs = _exec_desc(cur, br, cur, need, dev)
        exp = ctypes.byref(ex)
        arena0 = torch.empty(n0 * C0 + n1 * C0 + 2 * C0, dtype=torch.float32, device=dev)
        y0, out0, stat0 = arena0[: n0 * C0], arena0[n0 * C0 : (n0 + n1) * C0].view(n1, C0), arena0[(n0 + n1) * C0 :]
        sd = _StemDesc(ConvLayer(w0.data_ptr(), None, nbr0.data_ptr(), None, None, 0, 27, x.shape[1], C0, 1),
                       _norm_desc(plan["norms"][0], stat0, C0), nbr_pool.data_ptr(), i2o.data_ptr(), n0, n1, x.data_ptr(),
                       y0.data_ptr(), out0.data_ptr(), None)
        check(L.mink_stem_forward(ctypes.byref(sd), exp))
        Fn.note_table(nbr0)
        saved = [(x, w0, arena0, nbr0, nbr_pool, i2o, pad)]
        # ---- residual blocks
        h = out0
        for st, (in_key, out_key, n_in, n_out, c_in, C) in zip(stages, shapes):
            nbr1, _ = m.kernel_table(in_key, out_key, 3, 1)
            nbr2, _ = m.kernel_table(out_key, out_key, 3, 1)
            has_down = st.down is not None
            cnt = n_out * C
            arena = torch.empty((6 if has_down else 4) * cnt + 6 * C, dtype=torch.float32, device=dev)
            y1, h1, y2, out = (arena[i * cnt : (i + 1) * cnt] for i in range(4))
            yd = arena[4 * cnt : 5 * cnt] if has_down else None
            sdn = arena[5 * cnt : 6 * cnt] if has_down else None
            stat = arena[(6 if has_down else 4) * cnt :]
            p = params[st.pidx : st.pidx + (9 if has_down else 6)]
            nbrd = m.kernel_table(in_key, out_key, 1, 1)[0] if has_down else None
            zero = ConvLayer()
            bd = _BlockDesc(
                ConvLayer(p[0].data_ptr(), None, nbr1.data_ptr(), None, None, 0, 27, c_in, C, st.stride),
                ConvLayer(p[3].data_ptr(), None, nbr2.data_ptr(), None, None, 0, 27, C, C, 1),
                ConvLayer(p[6].data_ptr(), None, nbrd.data_ptr(), None, None, 0, 1, c_in, C, st.stride) if has_down else zero,
                _norm_desc(st.norm1, stat, C), _norm_desc(st.norm2, stat[2 * C :], C),
                _norm_desc(st.normd, stat[4 * C :], C) if has_down else NormLayer(),
                n_in, n_out, h.data_ptr(), y1.data_ptr(), h1.data_ptr(), y2.data_ptr(), _p(yd), _p(sdn), out.data_ptr(),
                None, None, None)
            if Fn._SKEW and br != cur:
                Fn.skew(br)
            check(L.mink_block_forward(ctypes.byref(bd), exp))
            Fn.note_table(nbr1, nbr2, nbrd)
            saved.append((h, arena, nbr1, nbr2, nbrd, in_key, out_key, n_in, n_out, c_in, C))
            h = out.view(n_out, C)
        ctx.plan, ctx.manager, ctx.saved, ctx.fork = plan, m, saved, fork
        ctx.params = params  # needed as keys of the data-parallel gradient sink and for the side-stream decision
        ctx.set_materialize_grads(False)
        return h

    @staticmethod
    def backward(ctx, g_out):
        L = lib()
        plan, m, saved, params = ctx.plan, ctx.manager, ctx.saved, ctx.params
        stages = plan["stages"]
        dev = g_out.device
        g_out = Fn._f32c(g_out)
        sink = Fn._GRAD_SINK
        # the weight gradients may run on the side stream (joined once, at the end of backward) when nothing consumes a
        # gradient earlier: autograd merely installs it as .grad, or the data-parallel reducer owns the memory
        installs_only = all(p.grad is None and not getattr(p, "_post_accumulate_grad_hooks", None) and not p._backward_hooks
                            for p in params) and not torch.is_grad_enabled()
        views = None
        if sink is not None and saved[0][6] == 0:
            views = Fn._sink_views(*params)
        overlap = Fn._OVERLAP_WGRAD and (views is not None or installs_only)
        cur, br, side = _exec(dev, ctx.fork and getattr(m, "prepared", False), overlap)
        need = max(L.mink_block_workspace_bytes(s[7], s[8], s[9], s[10]) for s in saved[1:])
        x0, w0 = saved[0][0], saved[0][1]
        need = max(need, L.mink_block_workspace_bytes(x0.shape[0], x0.shape[0], x0.shape[1], w0.shape[-1]))
        ex, ws_refs = _exec_desc(cur, br, side, need, dev)
        exp = ctypes.byref(ex)
        if overlap:
            Fn._defer_join()
            if Fn._SKEW:
                Fn.skew(side)
        grads = [None] * len(params)

        def grad_slots(idx, like):
            """Destination tensors for the gradients of params[idx...]: the reducer's slices, or one fresh buffer."""
            if views is not None:
                return [views[i] for i in idx]
            flat = torch.empty(sum(t.numel() for t in like), dtype=torch.float32, device=dev)
            out, off = [], 0
            for t in like:
                out.append(flat[off : off + t.numel()].view(t.shape))
                off += t.numel()
            if side != cur:
                flat.record_stream(side)
            return out

        g = g_out
        for st, sv in zip(reversed(stages), reversed(saved[1:])):
            h, arena, nbr1, nbr2, nbrd, in_key, out_key, n_in, n_out, c_in, C = sv
            has_down = st.down is not None
            cnt = n_out * C
            y1, h1, y2, out = (arena[i * cnt : (i + 1) * cnt] for i in range(4))
            yd = arena[4 * cnt : 5 * cnt] if has_down else None
            stat = arena[(6 if has_down else 4) * cnt :]
            idx = list(range(st.pidx, st.pidx + (9 if has_down else 6)))
            gs = grad_slots(idx, [params[i] for i in idx])
            nbr1_t = perm1 = nbrd_t = None
            n_perm = 0
            if st.stride == 2:
                _, nbr1_t = m.kernel_table(in_key, out_key, 3, 1, transposed=True)
                perm1 = m.class_perm(in_key)
                n_perm = perm1.numel()
                if has_down:
                    _, nbrd_t = m.kernel_table(in_key, out_key, 1, 1, transposed=True)
            g_tmp = torch.empty(L.mink_block_grad_scratch_floats(n_in, n_out, c_in, C, int(has_down)), dtype=torch.float32, device=dev)
            g_x = torch.empty(n_in, c_in, dtype=torch.float32, device=dev)
            p = [params[i] for i in idx]
            bd = _BlockDesc(
                ConvLayer(p[0].data_ptr(), gs[0].data_ptr(), nbr1.data_ptr(), _p(nbr1_t), _p(perm1), n_perm, 27, c_in, C, st.stride),
                ConvLayer(p[3].data_ptr(), gs[3].data_ptr(), nbr2.data_ptr(), None, None, 0, 27, C, C, 1),
                ConvLayer(p[6].data_ptr(), gs[6].data_ptr(), nbrd.data_ptr(), _p(nbrd_t), _p(perm1), n_perm, 1, c_in, C, st.stride)
                if has_down else ConvLayer(),
                _norm_desc(st.norm1, stat, C, (gs[1], gs[2])), _norm_desc(st.norm2, stat[2 * C :], C, (gs[4], gs[5])),
                _norm_desc(st.normd, stat[4 * C :], C, (gs[7], gs[8])) if has_down else NormLayer(),
                n_in, n_out, h.data_ptr(), y1.data_ptr(), h1.data_ptr(), y2.data_ptr(), _p(yd), None, out.data_ptr(),
                g.data_ptr(), g_x.data_ptr(), g_tmp.data_ptr())
            if Fn._SKEW:
                for s_ in {br, side} - {cur}:
                    Fn.skew(s_)
            check(L.mink_block_backward(ctypes.byref(bd), exp))
            Fn.note_table(nbr1, nbr2, nbrd, nbr1_t, nbrd_t)
            _KEEPALIVE.append((g_tmp, g))  # (never the gradients handed to autograd: a second reference makes it clone them at once)
            if views is not None:
                for i in idx:
                    sink.ready(params[i])
                if C <= 128:  # wide-and-shallow stage queued (static rule: the same on every rank): the host has time
                    sink.flush()
            else:
                for i, t in zip(idx, gs):
                    grads[i] = t
            g = g_x
        # ---- stem
        x, w0p, arena0, nbr0, nbr_pool, i2o, pad = saved[0]
        C0 = w0p.shape[-1]
        n0, n1 = x.shape[0], nbr_pool.shape[0]
        y0, stat0 = arena0[: n0 * C0], arena0[(n0 + n1) * C0 :]
        if views is not None:
            gs = [views[0], views[1], views[2]]
        else:
            gw = torch.empty(w0p.shape, dtype=torch.float32, device=dev)
            gb = torch.empty(2 * C0, dtype=torch.float32, device=dev)
            gs = [gw, gb[:C0], gb[C0:]]
        sd = _StemDesc(ConvLayer(w0p.data_ptr(), gs[0].data_ptr(), nbr0.data_ptr(), None, None, 0, 27, x.shape[1], C0, 1),
                       _norm_desc(plan["norms"][0], stat0, C0, (gs[1], gs[2])), nbr_pool.data_ptr(), i2o.data_ptr(), n0, n1,
                       x.data_ptr(), y0.data_ptr(), None, g.data_ptr())
        check(L.mink_stem_backward(ctypes.byref(sd), exp))
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
