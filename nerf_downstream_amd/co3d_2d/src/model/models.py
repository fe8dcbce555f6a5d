"""2-D comparison models (counterpart of the reference's co3d_2d/src/model/models.py:9-34): `ResNetBased` = a
torchvision ResNet (zero_init_residual=True) with its `fc` replaced by Identity, followed by Dropout and
Linear(in_features, 51).  The ResNet itself is defined here on the dense layers of model/dense.py with torchvision's
module and parameter names (`model.conv1.weight`, `model.layer1.0.bn2.running_var`, `model.layer2.0.downsample.0.weight`
...), so a torchvision checkpoint loads unchanged.  ViT variants (timm) are out of scope of the sparse-vs-dense
comparison row."""
import torch
import torch.nn as nn

from nerf_downstream_amd import gin_lite as gin

from . import dense


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = dense.Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = dense.BatchNorm2d(planes)
        self.conv2 = dense.Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = dense.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x, grid):
        identity = x
        st = self.training
        h, g, p = self.conv1(x, grid, bn_stats=st)
        h = self.bn1(h, relu=True, partial=p)
        h, g, p = self.conv2(h, g, bn_stats=st)
        if self.downsample is not None:
            identity, _, pd = self.downsample[0](x, grid, bn_stats=st)
            identity = self.downsample[1](identity, partial=pd)
        return self.bn2(h, relu=True, residual=identity, partial=p), g


class _Downsample(nn.Sequential):
    pass


class ResNet(nn.Module):
    def __init__(self, layers, num_classes=1000, zero_init_residual=False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = dense.Conv2d(3, 64, 7, 2, 3)
        self.bn1 = dense.BatchNorm2d(64)
        self.maxpool = dense.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.fc = nn.Linear(512, num_classes)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = _Downsample(dense.Conv2d(self.inplanes, planes, 1, stride, 0), dense.BatchNorm2d(planes))
        seq = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        seq += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.ModuleList(seq)

    def forward(self, images):
        x, grid = dense.to_rows(images)
        x, grid, p = self.conv1(x, grid, bn_stats=self.training)
        x = self.bn1(x, relu=True, partial=p)
        x, grid = self.maxpool(x, grid)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x, grid = blk(x, grid)
        x = dense.global_avg_pool(x, grid)
        return self.fc(x) if not isinstance(self.fc, nn.Identity) else x


_LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}


@gin.configurable
class ResNetBased(nn.Module):
    def __init__(self, model="resnet18", dropout_rate=0.2, pretrained=False):
        super().__init__()
        if model not in _LAYERS:
            raise NameError(f"Unknown model name : {model} (BasicBlock ResNets {sorted(_LAYERS)} are built here)")
        if pretrained:
            raise NotImplementedError("no network access for pretrained weights: load a torchvision state dict instead")
        net = ResNet(_LAYERS[model], zero_init_residual=True)
        in_features = net.fc.in_features
        net.fc = nn.Identity()
        self.fc = nn.Linear(in_features, 51, bias=True)
        self.dropout = nn.Dropout(dropout_rate)
        self.model = net

    def forward(self, x):
        return self.fc(self.dropout(self.model(x)))


def select_model(model_name):
    if model_name is None:
        raise NameError("Oops?")
    return ResNetBased(model_name)
