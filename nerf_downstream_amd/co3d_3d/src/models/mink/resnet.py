"""Mink-ResNet14/18/34 for plenoxel voxel-grid classification (counterpart of the reference's
co3d_3d/src/models/mink/resnet.py:25-192).  Topology, tensor strides and parameter names:

    TensorField.sparse()                                   ts 1
    conv1 3^3 (Cin->64) - bn1 - relu - SumPool(2,2)        ts 1 -> 2
    layer1..4: BasicBlock x LAYERS[i], first block stride 2, planes 64/128/256/512
    glob_avg (global average over each batch sample) - final 1x1 conv with bias -> logits [B, classes]

State-dict keys: conv1.kernel, bn1.bn.*, layer{i}.{j}.{conv1,conv2}.kernel,
layer{i}.{j}.{norm1,norm2}.bn.*, layer{i}.0.downsample.{0.kernel,1.bn.*}, final.{kernel,bias}.
ResNet50/101 use the Bottleneck block (conv1/conv3 1x1x1, conv2 3x3x3 carrying the stride; expansion 4)."""
import os

import torch
import torch.nn as nn

from .base_model import MinkowskiBaseModel
from .modules.common import conv, get_norm
from .modules.resnet_block import BasicBlock, Bottleneck


class _Presparsed:
    """A field whose `.sparse()` has already been taken (forward() looked at it to choose its path)."""

    def __init__(self, st):
        self._st = st

    def sparse(self):
        return self._st


class GlobalAvgPool(nn.Module):
    def __init__(self, ME):
        super().__init__()
        self.global_avg_pool = ME.MinkowskiGlobalAvgPooling()

    def forward(self, tensor):
        return self.global_avg_pool(tensor)


class ResNetBase(MinkowskiBaseModel):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)
    NORM_TYPE = "BN"

    def __init__(self, in_channel, out_channel, D=3, ME=None):
        super().__init__(D, ME=ME)
        self.D = D
        ME = self._ME
        self._fused = bool(getattr(ME, "SUPPORTS_FUSED_NORM", False))
        self.inplanes = self.INIT_DIM
        self.conv1 = conv(in_channel, self.inplanes, kernel_size=3, stride=1, D=D, ME=ME)
        self.bn1 = get_norm(self.NORM_TYPE, self.inplanes, D=D, bn_momentum=0.1, ME=ME)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=D)
        for i, (planes, count) in enumerate(zip(self.PLANES, self.LAYERS), start=1):
            setattr(self, f"layer{i}", self._make_layer(planes, count, stride=2))
        self.glob_avg = GlobalAvgPool(ME)
        self.final = conv(self.PLANES[3] * self.BLOCK.expansion, out_channel, kernel_size=1, bias=True, D=D, ME=ME)
        self.weight_initialization()
        self._norms = []
        self._trunk_plan = None  # native trunk description (minkowski/trunk.py), built on first use; False = not applicable
        self._native_trunk = os.environ.get("MINK_NATIVE_TRUNK", "1") != "0"
        if self._fused:  # HIP backend: one foreach launch bumps every BN step counter
            self._norms = [m for m in self.modules() if isinstance(m, ME.MinkowskiBatchNorm)]
            for m in self._norms:
                m.counted_by_parent = True

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        ME, out_planes = self._ME, planes * self.BLOCK.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_planes:
            shortcut = nn.Sequential(
                conv(self.inplanes, out_planes, kernel_size=1, stride=stride, bias=False, D=self.D, ME=ME),
                get_norm("BN", out_planes, D=self.D, bn_momentum=0.1, ME=ME),
            )
        seq = [self.BLOCK(self.inplanes, planes, stride=stride, downsample=shortcut, D=self.D, ME=ME)]
        self.inplanes = out_planes
        seq += [self.BLOCK(self.inplanes, planes, stride=1, D=self.D, ME=ME) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        if self.training and self._norms:  # (a norm the user has put in eval() keeps its count, as torch's does)
            counters = [m.bn.num_batches_tracked for m in self._norms if m.bn.training and m.bn.track_running_stats]
            if counters:
                torch._foreach_add_(counters, 1)
        if self._fused and self._native_trunk:
            # HIP backend: stem + residual stages as ONE autograd node whose forward / backward make one native call per
            # stage (minkowski/trunk.py) -- the same kernels as the module-by-module path below, without ~20 us of
            # Python per launch.  Anything the native sequencer does not cover (eval-mode norms, Bottleneck blocks,
            # an input that needs a gradient) takes the module path.
            from nerf_downstream_amd.minkowski import trunk

            if self._trunk_plan is None or (self._trunk_plan and trunk.stale(self._trunk_plan)):
                self._trunk_plan = trunk.plan_for(self) or False
            xs = x.sparse()
            if self._trunk_plan and trunk.usable(self, self._trunk_plan, xs):
                fork = getattr(self.layer1[0], "_fork", False) and trunk.Fn.trunk_branch_mode() is not None
                feats = trunk.TrunkFunction.apply(xs.F, self._trunk_plan, xs.coordinate_manager, fork, *self._trunk_plan.params)
                out = self._ME.SparseTensor(feats, trunk.out_key_of(self._trunk_plan), xs.coordinate_manager)
                return self._head(out)
            x = _Presparsed(xs)
        if self._fused:  # bn1 -> relu -> pool in one pass over the finest-level activation; its
            # statistics come out of the stem convolution's epilogue (no extra pass over 825 k x 64)
            # -- and, in training, conv1 + bn1 + relu + pool are one autograd node (the input needs no
            # gradient) whose backward keeps the gradient of the convolution output in registers
            out = self.pool(x.sparse(), norm=self.bn1, conv=self.conv1)
        else:
            out = self.pool(self.relu(self.bn1(self.conv1(x.sparse()))))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        return self._head(out)

    def _head(self, out):
        """glob_avg -> final -> .F (reference resnet.py:175-177).  HIP backend: one launch each way instead of the
        pooling kernel + library GEMM + bias add (and five launches backward) on the latency-bound end of the chain."""
        fin = self.final
        if self._fused and out.F.is_cuda and getattr(fin, "use_mm", False) and out.F.dtype == torch.float32:
            m = out.coordinate_manager
            return self._ME.functional.global_avg_linear(out.F, m.batch_offsets(out.coordinate_map_key), fin.kernel, fin.bias)
        return self.final(self.glob_avg(out)).F


class ResNet14(ResNetBase):
    BLOCK = BasicBlock
    LAYERS = (1, 1, 1, 1)


class ResNet18(ResNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 2, 2, 2)


class ResNet34(ResNetBase):
    BLOCK = BasicBlock
    LAYERS = (3, 4, 6, 3)


class ResNet50(ResNetBase):
    BLOCK = Bottleneck
    LAYERS = (3, 4, 6, 3)


class ResNet101(ResNetBase):
    BLOCK = Bottleneck
    LAYERS = (3, 4, 23, 3)
