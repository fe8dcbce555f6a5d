"""Residual block of Mink-ResNet14/18/34 (counterpart of the reference's
co3d_3d/src/models/mink/modules/resnet_block.py:11-73; parameter names conv1/norm1/conv2/
norm2/downsample are kept for state-dict compatibility).

    y = relu( norm2(conv2( relu(norm1(conv1(x))) )) + (downsample(x) or x) )

With a backend that advertises SUPPORTS_FUSED_NORM the three elementwise tails
(norm1+relu, downsample norm, norm2+add+relu) each run as ONE fused HIP pass."""
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

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        if self._fused:
            h = self.norm1(self.conv1(x), relu=True)
            return self.norm2(self.conv2(h), relu=True, residual=shortcut)
        h = self.nonlinearity(self.norm1(self.conv1(x)))
        h = self.norm2(self.conv2(h))
        h += shortcut
        return self.nonlinearity(h)
