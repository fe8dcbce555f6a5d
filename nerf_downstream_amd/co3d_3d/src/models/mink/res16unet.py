"""Res16UNet segmentation family (counterpart of the reference's co3d_3d/src/models/mink/res16unet.py:25-575;
SURVEY 8f-3).  U-shaped sparse network: a two-convolution stem at tensor stride 1, four down-sampling
stages (convolution k=2 s=2 + residual blocks), four up-sampling stages (transposed convolution k=2 s=2
onto the encoder's coordinate map, concatenation with the encoder activation of that stride, residual
blocks), a 1x1x1 classifier, and the per-point read-back `out.slice(x).F`.

Module names follow the reference so state dicts map one to one: conv0p1s1, conv1p1s2, block1, conv2p2s2,
block2, conv3p4s2, block3, conv4p8s2, block4, convtr4p16s2, block5, convtr5p8s2, block6, convtr6p4s2,
block7, convtr7p2s2, block8, final -- each conv*/convtr* an nn.Sequential whose entries 0 / 1 (/ 3 / 4 in
the stem) are the convolution and its norm.

Reference quirk kept visible: the size-named subclasses read `self.PLANES`, which only the *A / *B / *C / *D
variants define (res16unet.py:438-453 vs :528-575); here the base class carries the constructor default
(32, 48, 64, 96, 96, 96, 64, 64) as a class attribute so Res16UNet14/18/34 are constructible; and the
classifier takes PLANES[7] * BLOCK.expansion inputs (the reference passes PLANES[7], res16unet.py:343-351,
which only fits the BasicBlock variants)."""
import torch.nn as nn

from nerf_downstream_amd import gin_lite as gin

from .base_model import MinkowskiBaseModel
from .modules.common import conv, conv_tr, get_nonlinearity, get_norm
from .modules.resnet_block import BasicBlock, Bottleneck


class _Unit(nn.Sequential):
    """[conv, norm, nonlinearity] * n as the reference lays them out in an nn.Sequential; with a backend
    that fuses norm + ReLU (and takes the norm's statistics from the convolution's epilogue) each triple is
    two launches instead of four."""

    fused = False

    def forward(self, x):
        if not self.fused:
            return super().forward(x)
        mods = list(self)
        for j in range(0, len(mods), 3):
            x = mods[j + 1](mods[j](x, bn_stats=self.training), relu=True)
        return x


@gin.configurable
class Res16UNet(MinkowskiBaseModel):
    INSSEG = False
    BLOCK = BasicBlock
    PLANES = (32, 48, 64, 96, 96, 96, 64, 64)
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    NORM_TYPE = "BN"

    def __init__(self, in_channel, out_channel, PLANES=None, LAYERS=None, BLOCK=None, NORM_TYPE=None,
                 nonlinearity="MinkowskiReLU", bn_momentum=0.1, D=3, ME=None):
        super().__init__(D, ME=ME)
        ME = self._ME
        self.D, self.bn_momentum, self.nonlinearity = D, bn_momentum, nonlinearity
        self.PLANES, self.LAYERS = tuple(PLANES or self.PLANES), tuple(LAYERS or self.LAYERS)
        self.BLOCK, self.NORM_TYPE = BLOCK or self.BLOCK, NORM_TYPE or self.NORM_TYPE
        self._fused = bool(getattr(ME, "SUPPORTS_FUSED_NORM", False))
        P, exp = self.PLANES, self.BLOCK.expansion

        def unit(*convs):
            mods = []
            for c in convs:
                mods += [c, get_norm(self.NORM_TYPE, c.out_channels, D, bn_momentum=bn_momentum, ME=ME),
                         get_nonlinearity(nonlinearity, ME)()]
            u = _Unit(*mods)
            u.fused = self._fused
            return u

        def down(c):
            return unit(conv(c, c, kernel_size=2, stride=2, D=D, ME=ME))

        def up(cin, cout):
            return unit(conv_tr(cin, cout, kernel_size=2, upsample_stride=2, D=D, ME=ME))

        self.conv0p1s1 = unit(conv(in_channel, P[0], kernel_size=3, D=D, ME=ME), conv(P[0], P[0], kernel_size=3, D=D, ME=ME))
        self.conv1p1s2 = down(P[0])
        self.inplanes = P[0]
        self.block1 = self._make_layer(P[0], self.LAYERS[0])
        self.conv2p2s2 = down(self.inplanes)
        self.block2 = self._make_layer(P[1], self.LAYERS[1])
        self.conv3p4s2 = down(self.inplanes)
        self.block3 = self._make_layer(P[2], self.LAYERS[2])
        self.conv4p8s2 = down(self.inplanes)
        self.block4 = self._make_layer(P[3], self.LAYERS[3])
        self.convtr4p16s2 = up(self.inplanes, P[4])
        self.inplanes = P[4] + P[2] * exp
        self.block5 = self._make_layer(P[4], self.LAYERS[4])
        self.convtr5p8s2 = up(self.inplanes, P[5])
        self.inplanes = P[5] + P[1] * exp
        self.block6 = self._make_layer(P[5], self.LAYERS[5])
        self.convtr6p4s2 = up(self.inplanes, P[6])
        self.inplanes = P[6] + P[0] * exp
        self.block7 = self._make_layer(P[6], self.LAYERS[6])
        self.convtr7p2s2 = up(self.inplanes, P[7])
        self.inplanes = P[7] + P[0]
        self.block8 = self._make_layer(P[7], self.LAYERS[7])
        self.final = conv(P[7] * exp, out_channel, kernel_size=1, stride=1, bias=True, D=D, ME=ME)
        if self.INSSEG:
            raise NotImplementedError("the instance-segmentation offset head is out of scope")
        for m in self.modules():  # reference weight_initialization (res16unet.py:384-389)
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks):
        ME, out_planes = self._ME, planes * self.BLOCK.expansion
        shortcut = None
        if self.inplanes != out_planes:
            shortcut = nn.Sequential(
                conv(self.inplanes, out_planes, kernel_size=1, stride=1, D=self.D, ME=ME),
                get_norm(self.NORM_TYPE, out_planes, D=self.D, bn_momentum=self.bn_momentum, ME=ME),
            )
        mk = dict(norm_type=self.NORM_TYPE, nonlinearity_type=self.nonlinearity, bn_momentum=self.bn_momentum, D=self.D, ME=ME)
        seq = [self.BLOCK(self.inplanes, planes, stride=1, downsample=shortcut, **mk)]
        self.inplanes = out_planes
        seq += [self.BLOCK(self.inplanes, planes, stride=1, **mk) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        cat = self._ME.cat
        out_p1 = self.conv0p1s1(x.sparse())
        out_b1p2 = self.block1(self.conv1p1s2(out_p1))
        out_b2p4 = self.block2(self.conv2p2s2(out_b1p2))
        out_b3p8 = self.block3(self.conv3p4s2(out_b2p4))
        out = self.block4(self.conv4p8s2(out_b3p8))            # tensor stride 16
        out = self.block5(cat(self.convtr4p16s2(out), out_b3p8))  # 8
        out = self.block6(cat(self.convtr5p8s2(out), out_b2p4))   # 4
        out = self.block7(cat(self.convtr6p4s2(out), out_b1p2))   # 2
        out = self.block8(cat(self.convtr7p2s2(out), out_p1))     # 1
        return self.final(out).slice(x).F


class Res16UNet14(Res16UNet):
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class Res16UNet18(Res16UNet):
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class Res16UNet34(Res16UNet):
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet50(Res16UNet):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet101(Res16UNet):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 23, 2, 2, 2, 2)


class Res16UNet14A(Res16UNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet14A2(Res16UNet14A):
    LAYERS = (1, 1, 1, 1, 2, 2, 2, 2)


class Res16UNet14B(Res16UNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 128, 128)


class Res16UNet14B2(Res16UNet14B):
    LAYERS = (1, 1, 1, 1, 2, 2, 2, 2)


class Res16UNet14B3(Res16UNet14B):
    LAYERS = (2, 2, 2, 2, 1, 1, 1, 1)


class Res16UNet14C(Res16UNet14):
    PLANES = (32, 64, 128, 256, 192, 192, 128, 128)


class Res16UNet14D(Res16UNet14):
    PLANES = (32, 64, 128, 256, 384, 384, 384, 384)


class Res16UNet18A(Res16UNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet18B(Res16UNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 128, 128)


class Res16UNet18C(Res16UNet18):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


class Res16UNet18D(Res16UNet18):
    PLANES = (32, 64, 128, 256, 384, 384, 384, 384)


class Res16UNet34A(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 64)


class Res16UNet34B(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 32)


class Res16UNet34C(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
