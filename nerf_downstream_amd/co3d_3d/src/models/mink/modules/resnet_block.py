"""Residual blocks of the Mink-ResNet family (counterpart of the reference's
co3d_3d/src/models/mink/modules/resnet_block.py: BasicBlock :11-73, Bottleneck :76-132; parameter
names conv1/norm1/conv2/norm2[/conv3/norm3]/downsample are kept for state-dict compatibility).

    BasicBlock:  y = relu( norm2(conv2( relu(norm1(conv1(x))) )) + (downsample(x) or x) )
    Bottleneck:  y = relu( norm3(conv3( relu(norm2(conv2( relu(norm1(conv1(x))) ))) )) + (downsample(x) or x) )
                 conv1, conv3 are 1x1x1 (plain feature-matrix products), conv2 is 3x3x3 and carries the stride

With a backend that advertises SUPPORTS_FUSED_NORM the three elementwise tails
(norm1+relu, downsample norm, norm2+add+relu) each run as ONE fused HIP pass."""
import os

import torch.nn as nn

from .common import conv, default_me, get_nonlinearity, get_norm


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, norm_type="BN",
                 nonlinearity_type="MinkowskiReLU", bn_momentum=0.1, D=3, conv_mode=0, ME=None):
        super().__init__()
        ME = ME or default_me()
        mk = dict(D=D, conv_mode=conv_mode, ME=ME)
        self.conv1 = conv(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, **mk)
        self.norm1 = get_norm(norm_type, planes, D, bn_momentum=bn_momentum, ME=ME)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=1, dilation=dilation, bias=False, **mk)
        self.norm2 = get_norm(norm_type, planes, D, bn_momentum=bn_momentum, ME=ME)
        self.downsample = downsample
        self.nonlinearity = get_nonlinearity(nonlinearity_type, ME)()
        self._fused = bool(getattr(ME, "SUPPORTS_FUSED_NORM", False))
        self._fork = bool(getattr(ME, "SUPPORTS_FUSED_NORM", False)) and os.environ.get("MINK_FORK_SHORTCUT", "1") != "0"
        self._fork_unprepared = False  # tests: fork even when maps are still built on demand (exercises the manager's stream fence)

    def _forked_shortcut(self, x):
        """HIP backend, maps prepared ahead: run the shortcut branch (1x1 strided conv + norm) on a
        second stream beside conv1/norm1/conv2 -- none of these kernels fills the chip alone.
        Returns (shortcut, join) where join() makes the current stream wait for the branch."""
        import torch

        from nerf_downstream_amd.minkowski import functional as Fn

        cur = torch.cuda.current_stream(x.F.device)
        br = Fn.branch_stream(x.F.device, home=cur)
        Fn.stream_wait(br, cur)
        Fn.skew(br)
        with torch.cuda.stream(br):
            shortcut = self._shortcut(x)
        x.F.record_stream(br)

        def join():
            Fn.stream_wait(cur, br)
            shortcut.F.record_stream(cur)

        return shortcut, join

    def _shortcut(self, x):
        """downsample(x) = norm(conv(x)) (reference resnet.py:120-128); on the fused backend the convolution hands the
        column statistics of its output to the norm, as conv1 / conv2 do (and as the native trunk sequences it)."""
        ds = self.downsample
        if self._fused and self.training and len(ds) == 2:
            return ds[1](ds[0](x, bn_stats=True))
        return ds(x)

    def _may_fork(self, x):
        if not (self._fused and self._fork and self.downsample is not None and x.F.is_cuda):
            return False
        from nerf_downstream_amd.minkowski import functional as Fn

        return Fn.branch_fork_enabled() and (getattr(x.coordinate_manager, "prepared", False) or self._fork_unprepared)

    def forward(self, x):
        if self._may_fork(x):
            shortcut, join = self._forked_shortcut(x)
            st = self.training  # the convolutions hand their output's column statistics to the norm that follows
            h = self.conv2(self.norm1(self.conv1(x, bn_stats=st), relu=True), bn_stats=st)
            join()
            return self.norm2(h, relu=True, residual=shortcut)
        shortcut = x if self.downsample is None else self._shortcut(x)
        if self._fused:
            st = self.training
            h = self.norm1(self.conv1(x, bn_stats=st), relu=True)
            return self.norm2(self.conv2(h, bn_stats=st), relu=True, residual=shortcut)
        h = self.nonlinearity(self.norm1(self.conv1(x)))
        h = self.norm2(self.conv2(h))
        h += shortcut
        return self.nonlinearity(h)


class Bottleneck(BasicBlock):
    """ResNet50/101 block (reference resnet_block.py:76-132): 1x1x1 reduce -> 3x3x3 (stride) -> 1x1x1
    expand to 4 * planes, residual add, ReLU.  Shares the forked-shortcut machinery of BasicBlock."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, norm_type="BN",
                 nonlinearity_type="MinkowskiReLU", bn_momentum=0.1, D=3, conv_mode=0, ME=None):
        nn.Module.__init__(self)
        ME = ME or default_me()
        mk = dict(D=D, conv_mode=conv_mode, ME=ME)
        self.conv1 = conv(inplanes, planes, kernel_size=1, **mk)
        self.norm1 = get_norm(norm_type, planes, D, bn_momentum=bn_momentum, ME=ME)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=stride, dilation=dilation, **mk)
        self.norm2 = get_norm(norm_type, planes, D, bn_momentum=bn_momentum, ME=ME)
        self.conv3 = conv(planes, planes * self.expansion, kernel_size=1, **mk)
        self.norm3 = get_norm(norm_type, planes * self.expansion, D, bn_momentum=bn_momentum, ME=ME)
        self.downsample = downsample
        self.nonlinearity = get_nonlinearity(nonlinearity_type, ME)()
        self._fused = bool(getattr(ME, "SUPPORTS_FUSED_NORM", False))
        self._fork = self._fused and os.environ.get("MINK_FORK_SHORTCUT", "1") != "0"
        self._fork_unprepared = False

    def forward(self, x):
        join = None
        if self._may_fork(x):
            shortcut, join = self._forked_shortcut(x)
        else:
            shortcut = x if self.downsample is None else self._shortcut(x)
        if self._fused:
            st = self.training
            h = self.norm1(self.conv1(x), relu=True)
            h = self.norm2(self.conv2(h, bn_stats=st), relu=True)
            h = self.conv3(h)
            if join is not None:
                join()
            return self.norm3(h, relu=True, residual=shortcut)
        h = self.nonlinearity(self.norm1(self.conv1(x)))
        h = self.nonlinearity(self.norm2(self.conv2(h)))
        h = self.norm3(self.conv3(h))
        h += shortcut
        return self.nonlinearity(h)
