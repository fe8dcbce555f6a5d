"""nn.Module layer of the HIP backend: the ME modules the reference hot path instantiates
(SURVEY.md 8b).  Constructor signatures, parameter names (`kernel`, `bias`, `bn.*`) and
initialisation follow ME so reference checkpoints / configs map one to one."""
import math

import torch
import torch.nn as nn

from . import functional as Fn
from .coords import ORIGIN_TS, CoordinateMapKey, _as_int
from .tensor import SparseTensor


class MinkowskiNetwork(nn.Module):
    """Base class of reference models (models/mink/base_model.py:6-8)."""

    def __init__(self, D):
        super().__init__()
        self.D = D


class MinkowskiConvolution(nn.Module):
    """ME.MinkowskiConvolution as called by `conv()` (modules/common.py:116-125).

    kernel: (K, Cin, Cout), or (Cin, Cout) when the kernel volume is 1 and every stride is 1
    (`use_mm`, witness sparse_conv.py:323-335); bias: (1, Cout).  Init U(+-1/sqrt(Cin*K))
    (sparse_conv.py:427-435)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        assert dimension == 3, "the HIP backend implements dimension=3"
        assert kernel_generator is None and not expand_coordinates, "custom kernel generators are out of scope"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = _as_int(kernel_size), _as_int(stride), _as_int(dilation)
        assert self.kernel_size in (1, 2, 3), "kernel sizes 1, 2 (offsets {0,1}) and 3 (centred) are implemented"
        self.kernel_volume = self.kernel_size ** 3
        self.dimension = dimension
        self.use_mm = self.kernel_volume == 1 and self.stride == 1
        shape = (in_channels, out_channels) if self.use_mm else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, input, coordinates=None, bn_stats=False):
        """`bn_stats` (extension): also produce the per-channel (sum, sum of squares) partials of the
        output while it is being written (attribute `_bn_partial` of the result), so that a
        MinkowskiBatchNorm in training mode applied to it skips its own reduction pass."""
        assert coordinates is None, "explicit output coordinates are out of scope"
        m = input.coordinate_manager
        in_key = input.coordinate_map_key
        holder = [] if (bn_stats and self.bias is None and not self.use_mm) else None
        if self.use_mm:  # plain matmul on the feature matrix, same coordinates
            out_key = in_key
            if input.F.shape[0] >= 4096 and self.kernel.requires_grad and torch.is_grad_enabled():
                out = Fn.PointwiseConvolutionFunction.apply(input.F, self.kernel, lambda m=m, k=in_key: m.identity_table(k))
            else:
                out = input.F.mm(self.kernel)
        else:
            out_key = m.stride(in_key, self.stride)
            ks, dil = self.kernel_size, self.dilation

            def table_fn(transposed, m=m, in_key=in_key, out_key=out_key):
                nbr, nbr_t = m.kernel_table(in_key, out_key, ks, dil, transposed=transposed)
                perm = m.class_perm(in_key) if transposed and self.stride == 2 else None
                return nbr, nbr_t, perm

            # stride 1 and an odd (centred) kernel: the transposed table is the table itself with the offsets
            # flipped; an even kernel (offsets {0,1}) has no such symmetry and takes the explicit transposed table
            same_map = self.stride == 1 and self.kernel_size % 2 == 1
            out = Fn.ConvolutionFunction.apply(input.F, self.kernel, table_fn, same_map, holder)
        if self.bias is not None:
            out = out + self.bias
        res = SparseTensor(out, out_key, m)
        if holder:
            res._bn_partial = holder[0]
        return res

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, dilation={self.dilation}")


class MinkowskiConvolutionTranspose(MinkowskiConvolution):
    """ME.MinkowskiConvolutionTranspose as called by `conv_tr()` (reference modules/common.py:171-179;
    res16unet.py:196-206): up-samples from tensor stride ts to ts / stride onto the coordinate map that
    already exists there (the encoder's), out[i] += in[o] @ kernel[k] over the pairs (i, o, k) of the
    ordinary convolution fine -> coarse.  kernel: (K, Cin, Cout), init U(+-1/sqrt(Cout*K)) (ME initialises
    transposed kernels from the OUT channel count).

    On the device this is the data-gradient kernel of the down-sampling convolution run forward: a gather
    over the transposed neighbour table with rows grouped by parity class, so each 128-row tile multiplies
    only the one kernel offset its rows can have (kernel_size == stride: one parent per voxel)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         expand_coordinates, convolution_mode, dimension)
        assert self.stride == 2 and not self.use_mm, "the HIP backend up-samples by 2 (upsample_stride=2)"
        stdv = 1.0 / math.sqrt(self.out_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, input, coordinates=None, bn_stats=False):
        assert coordinates is None, "explicit output coordinates are out of scope"
        m = input.coordinate_manager
        in_key = input.coordinate_map_key
        ts = in_key.get_tensor_stride()[0]
        if ts % self.stride or not m.has_level(ts // self.stride):
            raise RuntimeError(f"MinkowskiConvolutionTranspose: no coordinate map at tensor stride {ts // self.stride} to "
                               "up-sample onto (generative up-sampling is out of scope)")
        out_key = CoordinateMapKey(ts // self.stride)
        ks, dil = self.kernel_size, self.dilation

        def table_fn(transposed, m=m, in_key=in_key, out_key=out_key):
            # tables of the ordinary convolution fine (out_key) -> coarse (in_key), used the other way round
            nbr, nbr_t = m.kernel_table(out_key, in_key, ks, dil, transposed=True)
            if transposed:  # towards the coarse input: its rows gather their children
                return nbr_t, nbr, None
            return nbr_t, nbr, m.class_perm(out_key)

        out = Fn.ConvolutionFunction.apply(input.F, self.kernel, table_fn, False, None)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, out_key, m)


def cat(*tensors):
    """ME.cat (reference res16unet.py:410-425): feature-wise concatenation of sparse tensors that share
    one coordinate map."""
    for t in tensors[1:]:
        tensors[0]._check(t)
    return SparseTensor(torch.cat([t.F for t in tensors], dim=1), tensors[0].coordinate_map_key,
                        tensors[0].coordinate_manager)


def _bn_momentum(bn):
    """The running-statistics update factor of `nn.BatchNorm1d`: `momentum`, or -- for momentum=None, the cumulative
    moving average -- 1 / num_batches_tracked (read after this step's increment; one host read-back, rare config)."""
    if bn.momentum is not None:
        return bn.momentum
    if bn.num_batches_tracked is None:
        return 0.0
    return 1.0 / max(float(bn.num_batches_tracked), 1.0)


class MinkowskiBatchNorm(nn.Module):
    """ME.MinkowskiBatchNorm: an `nn.BatchNorm1d` (attribute `bn`, so state-dict keys are
    `*.bn.weight` ... and the reference init loop resnet.py:101-105 finds it) applied to F.

    Extension used by this repo's fused blocks: `forward(x, relu=True, residual=r)` computes
    relu(bn(x) + r) in one pass (one HIP kernel forward, one reduction + one pass backward)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)
        self.counted_by_parent = False  # a parent network may bump all BN step counters in one launch

    def forward(self, input, relu=False, residual=None):
        bn = self.bn
        training = bn.training or not bn.track_running_stats
        if training and bn.track_running_stats and not self.counted_by_parent:
            bn.num_batches_tracked += 1
        if residual is not None:
            input._check(residual)
        gamma = bn.weight if bn.affine else torch.ones(bn.num_features, device=input.F.device)
        beta = bn.bias if bn.affine else torch.zeros(bn.num_features, device=input.F.device)
        out = Fn.BatchNormFunction.apply(
            input.F, gamma, beta, bn.running_mean, bn.running_var, training,
            _bn_momentum(bn), bn.eps,
            residual.F if residual is not None else None, bool(relu),
            getattr(input, "_bn_partial", None) if training else None)
        return SparseTensor(out, input.coordinate_map_key, input.coordinate_manager)


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """ME.MinkowskiSyncBatchNorm (reference train.py:106-107; off by default, train.py:83): batch
    statistics over the voxels of ALL ranks.  Falls back to local statistics when no process
    group is initialised or in eval mode."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)
        self.process_group = process_group

    def forward(self, input, relu=False, residual=None):
        import torch.distributed as dist

        bn = self.bn
        training = bn.training or not bn.track_running_stats
        if not (training and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1):
            return super().forward(input, relu=relu, residual=residual)
        if bn.track_running_stats and not self.counted_by_parent:
            bn.num_batches_tracked += 1
        if residual is not None:
            input._check(residual)
        out = Fn.SyncBatchNormFunction.apply(
            input.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, _bn_momentum(bn),
            bn.eps, residual.F if residual is not None else None, bool(relu), self.process_group)
        return SparseTensor(out, input.coordinate_map_key, input.coordinate_manager)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """Recursively replace every MinkowskiBatchNorm by a MinkowskiSyncBatchNorm sharing its
        parameters and buffers (same contract as ME / torch.nn.SyncBatchNorm)."""
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            bn = module.bn
            new = cls(bn.num_features, bn.eps, bn.momentum, bn.affine, bn.track_running_stats, process_group)
            new.bn = bn
            new.counted_by_parent = module.counted_by_parent
            new.train(module.training)
            return new
        for name, child in list(module.named_children()):
            setattr(module, name, cls.convert_sync_batchnorm(child, process_group))
        if hasattr(module, "_norms"):  # networks that keep a list of their norm layers
            module._norms = [m for m in module.modules() if isinstance(m, MinkowskiBatchNorm)]
        return module


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, input):
        return SparseTensor(Fn.ReLUFunction.apply(input.F), input.coordinate_map_key, input.coordinate_manager)


class _Activation(nn.Module):
    """Pointwise activation on the feature matrix; subclasses name the ME class (modules/common.py:36-51 looks the
    classes up by `__name__`) and take torch's constructor arguments (`ARG` = the name of the shape parameter)."""

    KIND, ALPHA, ARG = None, 0.0, None

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.alpha = self.ALPHA
        if self.ARG is not None:
            self.alpha = float(args[0] if args else kwargs.get(self.ARG, self.ALPHA))

    def forward(self, input):
        out = Fn.ActivationFunction.apply(input.F, self.KIND, self.alpha, None)
        return SparseTensor(out, input.coordinate_map_key, input.coordinate_manager)


class MinkowskiLeakyReLU(_Activation):
    KIND, ALPHA, ARG = "leaky_relu", 0.01, "negative_slope"


class MinkowskiELU(_Activation):
    KIND, ALPHA, ARG = "elu", 1.0, "alpha"


class MinkowskiCELU(_Activation):
    KIND, ALPHA, ARG = "celu", 1.0, "alpha"


class MinkowskiSELU(_Activation):
    KIND = "selu"


class MinkowskiGELU(_Activation):
    KIND = "gelu"


class MinkowskiPReLU(nn.Module):
    """ME.MinkowskiPReLU = torch.nn.PReLU on F: `weight` [num_parameters] initialised to `init`."""

    def __init__(self, num_parameters=1, init=0.25):
        super().__init__()
        self.weight = nn.Parameter(torch.full((num_parameters,), float(init)))

    def forward(self, input):
        out = Fn.ActivationFunction.apply(input.F, "prelu", 0.0, self.weight)
        return SparseTensor(out, input.coordinate_map_key, input.coordinate_manager)


class MinkowskiInstanceNorm(nn.Module):
    """ME.MinkowskiInstanceNorm(num_features) (reference modules/common.py:25-26): every batch sample's voxels are
    normalised per channel by that sample's own mean / biased variance, then scaled and shifted by `weight` (ones) /
    `bias` (zeros).  The rows of one sample are contiguous, so each sample is one batch-norm pass (training-mode
    statistics, no running buffers) over its row range.
    eps: ME's instance norm divides by sqrt(var + 1e-8) [ME-recall of MinkowskiInstanceNormFunction; parity unpinned: ME
    is absent and the reference holds no fixture for this layer -- it is only named by the layer factory, not used by a
    shipped model]; the value is an attribute so a caller that knows better can set it."""

    def __init__(self, num_features):
        super().__init__()
        self.num_features, self.eps = num_features, 1e-8
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, input):
        m = input.coordinate_manager
        boff = m.batch_offsets(input.coordinate_map_key).tolist()
        F, g, b = input.F, self.weight.reshape(-1), self.bias.reshape(-1)
        parts = [Fn.BatchNormFunction.apply(F[s:e], g, b, None, None, True, 0.0, self.eps, None, False, None)
                 for s, e in zip(boff[:-1], boff[1:]) if e > s]
        return SparseTensor(torch.cat(parts, 0), input.coordinate_map_key, m)


class MinkowskiLinear(nn.Module):
    """ME.MinkowskiLinear: torch.nn.Linear on the feature matrix (same parameter names through `.linear`)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, input):
        return SparseTensor(self.linear(input.F), input.coordinate_map_key, input.coordinate_manager)


class MinkowskiDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p

    def forward(self, input):
        return SparseTensor(nn.functional.dropout(input.F, self.p, self.training), input.coordinate_map_key,
                            input.coordinate_manager)


class MinkowskiSumPooling(nn.Module):
    """ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3) (resnet.py:62-64)."""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__()
        self.kernel_size, self.stride = _as_int(kernel_size), _as_int(stride)
        assert dimension == 3 and _as_int(dilation) == 1
        assert self.kernel_size == self.stride and self.kernel_size ** 3 <= 27, \
            "only non-overlapping pooling (kernel_size == stride) is implemented"

    def forward(self, input, norm=None, conv=None):
        """`norm` (extension): a MinkowskiBatchNorm to apply, followed by ReLU, to `input` on the
        fly -- pool(relu(norm(input))) in one pass without materialising the normalised tensor.
        `conv` (extension, with `norm`): a MinkowskiConvolution to apply first --
        pool(relu(norm(conv(input)))), the stem of the reference ResNets, as one autograd node
        whose backward never materialises the gradient of the convolution output either."""
        if conv is not None:
            fused = self._conv_norm_pool(input, norm, conv)
            if fused is not None:
                return fused
            input = conv(input, bn_stats=norm is not None and norm.bn.training)
        m, in_key = input.coordinate_manager, input.coordinate_map_key
        out_key = m.stride(in_key, self.stride)
        nbr, _ = m.kernel_table(in_key, out_key, self.kernel_size, 1)
        i2o = m.stride_map(in_key, out_key)
        fusable = norm is not None and norm.bn.affine and type(norm) is MinkowskiBatchNorm
        if fusable and (norm.bn.training or not torch.is_grad_enabled()):
            bn = norm.bn
            training = bn.training or not bn.track_running_stats
            if training and bn.track_running_stats and not norm.counted_by_parent:
                bn.num_batches_tracked += 1
            out = Fn.BNReLUSumPoolFunction.apply(input.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                                                 _bn_momentum(bn), bn.eps, nbr, i2o,
                                                 getattr(input, "_bn_partial", None) if training else None)
        else:
            if norm is not None:
                input = norm(input, relu=True)
            out = Fn.SumPoolFunction.apply(input.F, nbr, i2o)
        return SparseTensor(out, out_key, m)


    def _conv_norm_pool(self, input, norm, conv):
        """The fully fused stem, or None when it does not apply (eval mode, input gradient wanted,
        conv with bias / stride, SyncBN, or a shape the streaming weight-gradient kernel leaves
        to the general one)."""
        if (norm is None or type(norm) is not MinkowskiBatchNorm or not norm.bn.affine or not norm.bn.training
                or not norm.bn.track_running_stats or conv.bias is not None or conv.use_mm or conv.stride != 1
                or conv.kernel_volume != 27 or conv.dilation != 1 or input.F.requires_grad or not torch.is_grad_enabled()):
            return None
        m, in_key = input.coordinate_manager, input.coordinate_map_key
        nbr, _ = m.kernel_table(in_key, in_key, conv.kernel_size, conv.dilation)
        if not Fn.ConvBNReLUSumPoolFunction.supported(input.F, conv.kernel, nbr):
            return None
        out_key = m.stride(in_key, self.stride)
        nbr_pool, _ = m.kernel_table(in_key, out_key, self.kernel_size, 1)
        i2o = m.stride_map(in_key, out_key)
        bn = norm.bn
        if not norm.counted_by_parent:
            bn.num_batches_tracked += 1
        out = Fn.ConvBNReLUSumPoolFunction.apply(
            input.F, conv.kernel, bn.weight, bn.bias, bn.running_mean, bn.running_var,
            _bn_momentum(bn), bn.eps, nbr, nbr_pool, i2o)
        return SparseTensor(out, out_key, m)


class MinkowskiGlobalAvgPooling(nn.Module):
    """ME.MinkowskiGlobalAvgPooling() (resnet.py:15-22): row b of the output is batch index b."""

    def forward(self, input):
        m = input.coordinate_manager
        out = Fn.GlobalAvgPoolFunction.apply(input.F, m.batch_offsets(input.coordinate_map_key))
        return SparseTensor(out, CoordinateMapKey(ORIGIN_TS), m)
