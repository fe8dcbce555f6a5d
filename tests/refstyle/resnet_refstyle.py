"""A Mink-ResNet written the way the reference composes its models -- `import MinkowskiEngine as ME` at the top,
layers taken straight from the ME namespace, `out += residual`, the reference's `get_norm` / `get_nonlinearity`
look-up tables built at import time from `ME.Minkowski*` classes and `MinkowskiFunctional` -- but written for
this test (it is NOT a copy of a reference file).  tests/test_gpu_compat.py imports it twice: against the HIP drop-in
(`nerf_downstream_amd.install_as_minkowski_engine()`) and against the CPU oracle's mini-ME, and compares the two."""
import MinkowskiEngine as ME
import MinkowskiEngine.MinkowskiFunctional as MEF
import torch.nn as nn

# resolved when the module is imported, as the reference's modules/common.py:36-51 does
ACTIVATIONS = {cls.__name__: cls for cls in (ME.MinkowskiReLU, ME.MinkowskiPReLU, ME.MinkowskiLeakyReLU, ME.MinkowskiELU,
                                             ME.MinkowskiCELU, ME.MinkowskiSELU, ME.MinkowskiGELU)}
NORMS = {"BN": lambda c: ME.MinkowskiBatchNorm(c, momentum=0.1), "IN": lambda c: ME.MinkowskiInstanceNorm(c)}


class Block(nn.Module):
    def __init__(self, cin, cout, stride, act="MinkowskiReLU"):
        super().__init__()
        self.conv1 = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=stride, dilation=1, bias=False, dimension=3)
        self.norm1 = NORMS["BN"](cout)
        self.conv2 = ME.MinkowskiConvolution(cout, cout, kernel_size=3, stride=1, dilation=1, bias=False, dimension=3)
        self.norm2 = NORMS["BN"](cout)
        self.act = ACTIVATIONS[act]()
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                ME.MinkowskiConvolution(cin, cout, kernel_size=1, stride=stride, bias=False, dimension=3), NORMS["BN"](cout))

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        out = self.act(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        out += residual
        return MEF.relu(out)


class TinyResNet(ME.MinkowskiNetwork):
    def __init__(self, in_channel, out_channel, D=3, act="MinkowskiReLU"):
        ME.MinkowskiNetwork.__init__(self, D)
        self.conv1 = ME.MinkowskiConvolution(in_channel, 32, kernel_size=3, stride=1, dimension=D)
        self.bn1 = ME.MinkowskiBatchNorm(32, momentum=0.1)
        self.relu = ME.MinkowskiReLU(inplace=True)
        self.pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=D)
        self.layer1 = Block(32, 32, 2, act)
        self.layer2 = Block(32, 64, 2, act)
        self.layer3 = Block(64, 64, 1, act)
        self.glob_avg = ME.MinkowskiGlobalAvgPooling()
        self.final = ME.MinkowskiConvolution(64, out_channel, kernel_size=1, bias=True, dimension=D)

    def forward(self, batch):
        x = ME.TensorField(coordinates=batch["coordinates"], features=batch["features"])
        out = self.pool(self.relu(self.bn1(self.conv1(x.sparse()))))
        out = self.layer3(self.layer2(self.layer1(out)))
        return self.final(self.glob_avg(out)).F
