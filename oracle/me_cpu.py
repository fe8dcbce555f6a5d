"""oracle/me_cpu.py -- CPU restatement of the MinkowskiEngine subset used by the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py): a "mini-ME" with the same Python
surface the reference imports as ``import MinkowskiEngine as ME``,
so model code written against that surface can be run on this CPU restatement and
compared with the HIP product path.  Coordinate structures come from the plain-C
restatement (oracle/mink_maps.c); float arithmetic is plain torch-CPU fp32 written
exactly as ME's CPU algorithm: per kernel offset gather -> GEMM -> scatter-add
(the loop re-stated by the reference at
co3d_3d/src/models/mink/modules/sparse_conv.py:122-144), BatchNorm as
``torch.nn.BatchNorm1d`` on the feature matrix (witness:
co3d_3d/src/models/mink/resnet.py:101-105).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import maps

ORIGIN_TS = 0  # tensor-stride sentinel of the global-pool "origin" map


class CoordinateMapKey:
    def __init__(self, tensor_stride, name=""):
        self.ts = int(tensor_stride)
        self.name = name

    def get_tensor_stride(self):
        return [self.ts] * 3

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and (self.ts, self.name) == (other.ts, other.name)

    def __hash__(self):
        return hash((self.ts, self.name))

    def __repr__(self):
        return f"CoordinateMapKey(ts={self.ts})"


class CoordinateManager:
    """Per-batch cache of coordinate maps / stride maps / kernel maps (A3, A5).

    Created fresh for every TensorField, exactly like ME (rebuilt every iteration,
    shared between forward and backward).
    """

    def __init__(self, D=3):
        self.D = D
        self.coords = {}  # ts -> int32 [n,4]
        self.in2out = {}  # (ts_in, ts_out) -> int32 [n_in]
        self.tables = {}  # (ts_in, ts_out, ksize, dilation) -> int32 [n_out,K]
        self.field_inverse = None
        self.field_unique_index = None

    # --- construction -------------------------------------------------------------
    def insert_field(self, fcoords):
        """A1 + A2: floor-quantise the float field and insert sequentially."""
        q = maps.quantize(fcoords.detach().cpu().numpy())
        ui, inv = maps.unique(q)
        self.coords[1] = np.ascontiguousarray(q[ui])
        self.field_unique_index, self.field_inverse = ui, inv
        return CoordinateMapKey(1)

    def stride(self, key, stride):
        """Output key of a stride-`stride` op on `key` (witness: sparse_conv.py:403-405)."""
        stride = int(stride[0] if isinstance(stride, (list, tuple)) else stride)
        if stride == 1:
            return key
        ts_out = key.ts * stride
        if ts_out not in self.coords:
            oc, i2o = maps.stride_map(self.coords[key.ts], ts_out)
            self.coords[ts_out] = oc
            self.in2out[(key.ts, ts_out)] = i2o
        return CoordinateMapKey(ts_out)

    def stride_map(self, in_key, out_key):
        k = (in_key.ts, out_key.ts)
        if k not in self.in2out:
            oc, i2o = maps.stride_map(self.coords[in_key.ts], out_key.ts)
            assert np.array_equal(oc, self.coords[out_key.ts])
            self.in2out[k] = i2o
        return self.in2out[k]

    def kernel_table(self, in_key, out_key, kernel_size, dilation=1):
        kk = (in_key.ts, out_key.ts, int(kernel_size), int(dilation))
        if kk not in self.tables:
            off = maps.kernel_offsets(kernel_size, in_key.ts, dilation)
            self.tables[kk] = maps.kernel_map_table(self.coords[in_key.ts], self.coords[out_key.ts], off)
        return self.tables[kk]

    def kernel_table_transposed(self, fine_key, coarse_key, kernel_size, dilation=1):
        """Table of a transposed convolution coarse -> fine [ME-recall]: the kernel map of the ordinary
        convolution fine -> coarse with in and out swapped, i.e. nbr_t[i][k] = o for every nbr[o][k] = i."""
        nbr = self.kernel_table(fine_key, coarse_key, kernel_size, dilation)
        nbr_t = np.full((self.coords[fine_key.ts].shape[0], nbr.shape[1]), -1, np.int32)
        o, k = np.nonzero(nbr >= 0)
        nbr_t[nbr[o, k], k] = o
        return nbr_t

    def kernel_map(self, in_key, out_key, stride=1, kernel_size=3, dilation=1, is_transpose=False, is_pool=False):
        """ME-format kernel map {k: IntTensor[2,n]} (witness: sparse_conv.py:90-96,124-143)."""
        assert not is_transpose
        lists = maps.table_to_lists(self.kernel_table(in_key, out_key, kernel_size, dilation))
        return {k: torch.from_numpy(v) for k, v in lists.items()}

    def size(self, key):
        return self.coords[key.ts].shape[0] if key.ts != ORIGIN_TS else self.batch_size()

    def batch_size(self):
        return int(self.coords[1][:, 0].max()) + 1

    def get_coordinates(self, key):
        if key.ts == ORIGIN_TS:
            c = np.zeros((self.batch_size(), 4), np.int32)
            c[:, 0] = np.arange(self.batch_size())
            return torch.from_numpy(c)
        return torch.from_numpy(self.coords[key.ts])


class SparseTensor:
    def __init__(self, features, coordinate_map_key=None, coordinate_manager=None):
        self._F = features
        self.coordinate_map_key = coordinate_map_key
        self._manager = coordinate_manager

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self._manager.get_coordinates(self.coordinate_map_key)

    @property
    def coordinate_manager(self):
        return self._manager

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def D(self):
        return self._manager.D

    @property
    def shape(self):
        return self._F.shape

    def _check(self, other):
        assert self.coordinate_map_key == other.coordinate_map_key and self._manager is other._manager

    def slice(self, field):
        """`out.slice(x)` (reference res16unet.py:435) [ME-recall]: the features of this tensor-stride-1
        sparse tensor read back at the rows of the field it was quantised from (F[inverse_mapping])."""
        assert self.coordinate_map_key.ts == 1 and field._manager is self._manager
        inv = torch.from_numpy(self._manager.field_inverse.astype(np.int64))
        return TensorField(features=self.F[inv], coordinates=field.C, _manager=self._manager)

    def __iadd__(self, other):  # reference resnet_block.py:66
        self._check(other)
        self._F = self._F + other._F
        return self

    def __add__(self, other):
        self._check(other)
        return SparseTensor(self._F + other._F, self.coordinate_map_key, self._manager)


def cat(*tensors):
    """ME.cat [ME-recall]: feature-wise concatenation of sparse tensors that share one coordinate map
    (reference res16unet.py:410,415,420,425)."""
    for t in tensors[1:]:
        tensors[0]._check(t)
    return SparseTensor(torch.cat([t.F for t in tensors], dim=1), tensors[0].coordinate_map_key, tensors[0]._manager)


class TensorField:
    """ME.TensorField(coordinates=[N,1+D] float, features=[N,C]) (base_model.py:10-13)."""

    def __init__(self, features=None, coordinates=None, **kw):
        assert coordinates is not None and features is not None
        self._F = features
        self._C = coordinates
        if kw.get("_manager") is not None:  # a slice(): shares the manager of the field it came from
            self._manager = kw["_manager"]
            return
        self._manager = CoordinateManager(D=coordinates.shape[1] - 1)
        self.coordinate_field_map_key = self._manager.insert_field(coordinates)

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self._C

    def sparse(self):
        """A2: duplicates after flooring are averaged (UNWEIGHTED_AVERAGE)."""
        m = self._manager
        inv = torch.from_numpy(m.field_inverse.astype(np.int64))
        nu = m.coords[1].shape[0]
        F = self._F if self._F.dtype == torch.float64 else self._F.float()  # (float64: the tests' high-precision reference run)
        if nu == F.shape[0]:
            Fs = F[torch.from_numpy(m.field_unique_index.astype(np.int64))]
        else:
            Fs = torch.zeros(nu, F.shape[1], dtype=F.dtype).index_add_(0, inv, F)
            cnt = torch.zeros(nu, dtype=F.dtype).index_add_(0, inv, torch.ones(F.shape[0], dtype=F.dtype))
            Fs = Fs / cnt[:, None]
        return SparseTensor(Fs, CoordinateMapKey(1), m)


# --------------------------------------------------------------------------------- functions
# Test hook: when set, _ConvFn passes every GEMM operand through OPERAND_HOOK(tensor, role, shape) before it is multiplied, with
# role in {"fwd_x", "fwd_w", "fwd_y", "dgrad_g", "dgrad_w", "wgrad_x", "wgrad_g"} ("fwd_y": the finished output, identity gradient)
# and shape = (K, cin, cout, n_in, n_out).  tests/test_gpu_parity_full.py uses it to run the oracle in float64 on operands ROUNDED TO
# bf16 exactly where the implementation under test rounds them (BASELINE config #4): what a bf16-operand / fp32-accumulate kernel is
# asked to compute, tensor by tensor, instead of a cosine against the fp32 network.
OPERAND_HOOK = None


def _op(t, role, shape):
    return t if OPERAND_HOOK is None else OPERAND_HOOK(t, role, shape)


class _ConvFn(torch.autograd.Function):
    """A6: out[o_list] += in[i_list] @ kernel[k], k ascending; and its backward."""

    @staticmethod
    def forward(ctx, x, kernel, nbr):
        nbr_t = torch.from_numpy(nbr.astype(np.int64))
        shape = (kernel.shape[0], kernel.shape[1], kernel.shape[2], x.shape[0], nbr.shape[0])
        xf, kf = _op(x, "fwd_x", shape), _op(kernel, "fwd_w", shape)
        out = x.new_zeros(nbr.shape[0], kernel.shape[2])
        lists = []
        for k in range(nbr.shape[1]):
            o = torch.nonzero(nbr_t[:, k] >= 0).squeeze(1)
            if o.numel() == 0:
                lists.append(None)
                continue
            i = nbr_t[o, k]
            lists.append((i, o))
            out[o] += xf[i] @ kf[k]
        ctx.save_for_backward(x, kernel)
        ctx.lists = lists
        ctx.shape = shape
        return _op(out, "fwd_y", shape)

    @staticmethod
    def backward(ctx, gout):
        x, kernel = ctx.saved_tensors
        gout = gout.contiguous()
        shape = ctx.shape
        gx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        gk = torch.zeros_like(kernel)
        if OPERAND_HOOK is not None:
            gd, kd = _op(gout, "dgrad_g", shape), _op(kernel, "dgrad_w", shape)
            x, gw = _op(x, "wgrad_x", shape), _op(gout, "wgrad_g", shape)
        else:
            gd, kd, gw = gout, kernel, gout
        for k, io in enumerate(ctx.lists):
            if io is None:
                continue
            i, o = io
            if gx is not None:
                gx[i] += gd[o] @ kd[k].t()
            gk[k] = x[i].t() @ gw[o]
        return gx, gk, None


class _SumPoolFn(torch.autograd.Function):
    """A9: out[parent(i)] += in[i]; backward dIn[i] = dOut[parent(i)]."""

    @staticmethod
    def forward(ctx, x, in2out, n_out):
        idx = torch.from_numpy(in2out.astype(np.int64))
        ctx.idx = idx
        return x.new_zeros(n_out, x.shape[1]).index_add_(0, idx, x)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.idx], None, None


class _GlobalAvgFn(torch.autograd.Function):
    """A10: out[b] = mean over rows of batch b; backward dIn[i] = dOut[b_i] / N_b."""

    @staticmethod
    def forward(ctx, x, batch_idx, B):
        idx = torch.from_numpy(batch_idx.astype(np.int64))
        cnt = torch.bincount(idx, minlength=B).to(x.dtype)
        ctx.idx, ctx.cnt = idx, cnt
        return x.new_zeros(B, x.shape[1]).index_add_(0, idx, x) / cnt[:, None]

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.cnt[:, None])[ctx.idx], None, None


# ----------------------------------------------------------------------------------- modules
class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class MinkowskiConvolution(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None, **kw):
        super().__init__()
        assert dimension == 3
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = int(kernel_size[0] if isinstance(kernel_size, (list, tuple)) else kernel_size)
        self.stride = int(stride[0] if isinstance(stride, (list, tuple)) else stride)
        self.dilation = int(dilation[0] if isinstance(dilation, (list, tuple)) else dilation)
        self.kernel_volume = self.kernel_size**3
        # use_mm: kernel volume 1 AND all strides 1 (sparse_conv.py:323-328)
        self.use_mm = self.kernel_volume == 1 and self.stride == 1
        shape = (in_channels, out_channels) if self.use_mm else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):  # A7, sparse_conv.py:427-435
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, input):
        m = input._manager
        if self.use_mm:
            out_key = input.coordinate_map_key
            out = input.F.mm(self.kernel)
        else:
            out_key = m.stride(input.coordinate_map_key, self.stride)
            nbr = m.kernel_table(input.coordinate_map_key, out_key, self.kernel_size, self.dilation)
            out = _ConvFn.apply(input.F, self.kernel, nbr)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, out_key, m)


class MinkowskiConvolutionTranspose(MinkowskiConvolution):
    """ME.MinkowskiConvolutionTranspose (reference modules/common.py:171-179) [ME-recall]: up-samples
    from tensor stride ts to ts / stride onto the coordinate map that already exists there (the
    encoder's), out[i] += in[o] @ kernel[k] over the pairs (i, o, k) of the ordinary convolution
    fine -> coarse.  Weights are initialised from the OUT channel count (ME reset_parameters(True))."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, dimension)
        assert not self.use_mm
        stdv = 1.0 / math.sqrt(self.out_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, input):
        m = input._manager
        ts = input.coordinate_map_key.ts
        assert ts % self.stride == 0 and ts // self.stride in m.coords, "no coordinate map to up-sample onto"
        out_key = CoordinateMapKey(ts // self.stride)
        nbr_t = m.kernel_table_transposed(out_key, input.coordinate_map_key, self.kernel_size, self.dilation)
        out = _ConvFn.apply(input.F, self.kernel, nbr_t)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, out_key, m)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)

    def forward(self, input):
        return SparseTensor(self.bn(input.F), input.coordinate_map_key, input._manager)


# Test hook: when set, MinkowskiReLU calls RELU_HOOK(z) instead of torch.relu(z).  The full-size parity test uses it to
# evaluate the network with the branch decisions (z > 0) of the implementation under test imposed -- the gradient of a
# ReLU network is only defined up to the branch taken where a pre-activation is zero to rounding, so two correct fp32
# implementations may legitimately differ there (tests/test_gpu_parity_full.py).
RELU_HOOK = None


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, input):
        f = torch.relu(input.F) if RELU_HOOK is None else RELU_HOOK(input.F)
        return SparseTensor(f, input.coordinate_map_key, input._manager)


def _pointwise(name, fn, arg=None, default=None):
    """ME's activation modules are torch's activations applied to F (ME MinkowskiNonlinearity.py wraps the torch modules)."""

    class _Act(nn.Module):
        def __init__(self, *args, **kw):
            super().__init__()
            self.value = (args[0] if args else kw.get(arg, default)) if arg else None

        def forward(self, input):
            out = fn(input.F) if arg is None else fn(input.F, self.value)
            return SparseTensor(out, input.coordinate_map_key, input._manager)

    _Act.__name__ = _Act.__qualname__ = name
    return _Act


MinkowskiLeakyReLU = _pointwise("MinkowskiLeakyReLU", torch.nn.functional.leaky_relu, "negative_slope", 0.01)
MinkowskiELU = _pointwise("MinkowskiELU", torch.nn.functional.elu, "alpha", 1.0)
MinkowskiCELU = _pointwise("MinkowskiCELU", torch.nn.functional.celu, "alpha", 1.0)
MinkowskiSELU = _pointwise("MinkowskiSELU", torch.nn.functional.selu)
MinkowskiGELU = _pointwise("MinkowskiGELU", torch.nn.functional.gelu)


class MinkowskiPReLU(nn.Module):
    def __init__(self, num_parameters=1, init=0.25):
        super().__init__()
        self.weight = nn.Parameter(torch.full((num_parameters,), float(init)))

    def forward(self, input):
        return SparseTensor(torch.nn.functional.prelu(input.F, self.weight), input.coordinate_map_key, input._manager)


class MinkowskiInstanceNorm(nn.Module):
    """ME.MinkowskiInstanceNorm: per batch sample and channel, (x - mean) / sqrt(biased var + 1e-8) * weight + bias [ME-recall; parity unpinned]."""

    def __init__(self, num_features):
        super().__init__()
        self.eps = 1e-8
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, input):
        m = input._manager
        b = torch.from_numpy(m.coords[input.coordinate_map_key.ts][:, 0].astype(np.int64))
        out = torch.empty_like(input.F)
        for j in range(m.batch_size()):
            rows = torch.nonzero(b == j).squeeze(1)
            x = input.F[rows]
            mu, var = x.mean(0, keepdim=True), x.var(0, unbiased=False, keepdim=True)
            out[rows] = (x - mu) / torch.sqrt(var + self.eps) * self.weight + self.bias
        return SparseTensor(out, input.coordinate_map_key, m)


class _Functional:
    """MinkowskiEngine.MinkowskiFunctional (reference modules/common.py:56-71)."""

    @staticmethod
    def _w(input, out):
        return SparseTensor(out, input.coordinate_map_key, input._manager)

    relu = staticmethod(lambda input, *a, **k: _Functional._w(input, torch.relu(input.F)))
    leaky_relu = staticmethod(lambda input, negative_slope=0.01, **k: _Functional._w(input, torch.nn.functional.leaky_relu(input.F, negative_slope)))
    elu = staticmethod(lambda input, alpha=1.0, **k: _Functional._w(input, torch.nn.functional.elu(input.F, alpha)))
    celu = staticmethod(lambda input, alpha=1.0, **k: _Functional._w(input, torch.nn.functional.celu(input.F, alpha)))
    selu = staticmethod(lambda input, **k: _Functional._w(input, torch.nn.functional.selu(input.F)))
    gelu = staticmethod(lambda input, **k: _Functional._w(input, torch.nn.functional.gelu(input.F)))
    prelu = staticmethod(lambda input, weight: _Functional._w(input, torch.nn.functional.prelu(input.F, weight)))


MinkowskiFunctional = _Functional()


class MinkowskiSumPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, dimension=None, **kw):
        super().__init__()
        ks = int(kernel_size[0] if isinstance(kernel_size, (list, tuple)) else kernel_size)
        st = int(stride[0] if isinstance(stride, (list, tuple)) else stride)
        assert ks == st, "oracle restates only the kernel==stride pooling used by the reference (resnet.py:62-64)"
        self.stride = st

    def forward(self, input):
        m = input._manager
        out_key = m.stride(input.coordinate_map_key, self.stride)
        i2o = m.stride_map(input.coordinate_map_key, out_key)
        return SparseTensor(_SumPoolFn.apply(input.F, i2o, m.size(out_key)), out_key, m)


class MinkowskiGlobalAvgPooling(nn.Module):
    def forward(self, input):
        m = input._manager
        b = m.coords[input.coordinate_map_key.ts][:, 0]
        return SparseTensor(_GlobalAvgFn.apply(input.F, b, m.batch_size()), CoordinateMapKey(ORIGIN_TS), m)


class _Utils:
    @staticmethod
    def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
        """A11 (reference data/utils.py:25-30)."""
        n = [int(c.shape[0]) for c in coords]
        bcoords = torch.zeros(sum(n), coords[0].shape[1] + 1, dtype=dtype)
        s = 0
        for j, c in enumerate(coords):
            c = torch.from_numpy(c) if isinstance(c, np.ndarray) else c
            bcoords[s : s + n[j], 1:] = c.to(dtype)
            bcoords[s : s + n[j], 0] = j
            s += n[j]
        bfeats = torch.cat([torch.from_numpy(f) if isinstance(f, np.ndarray) else f for f in feats], 0)
        return bcoords, bfeats


utils = _Utils()
