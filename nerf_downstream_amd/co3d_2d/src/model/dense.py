"""Dense 2-D layers on the sparse-convolution kernels (BASELINE config #5: "co3d_2d/train.py 2D CNN on rendered
PeRFception RGB -- dense conv as HIP MFMA baseline").

A dense image batch is the degenerate case of a sparse tensor: every pixel is occupied.  Activations are kept as the
row matrix [B*H*W, C] (NHWC, exactly the feature layout of the sparse path) and a convolution is the same implicit GEMM
over a neighbour table -- here an ARITHMETIC one, nbr[(b, oy, ox)][(ky, kx)] = (b, oy*s - p + ky, ox*s - p + kx) or -1
outside the image, built once per (shape, kernel, stride, padding) and cached, since it does not depend on the data.
So the 3x3 / 1x1 convolutions, batch norm (+ fused ReLU / residual add) and the global average pooling run in the
hand-written gfx950 kernels of libmink_hip.so (`mink_conv_gather_gemm` / `mink_conv_wgrad`: LDS-staged tiles, fp32 or
bf16 MFMA); the one new kernel is the overlapping max pooling (`mink_pool_max_*`).  A kernel volume above 27 (the 7x7
stem) is applied as two offset groups whose outputs are added.

Parameter names and layouts are torch's (`weight` [Cout, Cin, kh, kw], `running_mean`, ...), so torchvision ResNet
checkpoints load unchanged (reference co3d_2d/src/model/models.py:18-23 builds torchvision's resnet18)."""
import math

import torch
import torch.nn as nn

from nerf_downstream_amd.minkowski import functional as Fn

_TABLES = {}


class Grid:
    """Shape of a dense activation [B*H*W, C]."""

    __slots__ = ("B", "H", "W")

    def __init__(self, B, H, W):
        self.B, self.H, self.W = int(B), int(H), int(W)

    @property
    def rows(self):
        return self.B * self.H * self.W


def _conv_tables(grid, k, stride, pad, device):
    """(out_grid, nbr [n_out, k*k] int32, nbr_t [n_in, k*k] int32, perm): the neighbour table of a k x k convolution
    over the dense grid (offsets enumerated kx fastest, like torch's weight[..., ky, kx]), its transpose, and for
    stride 2 the parity-class row order of the INPUT grid (every 128-row tile of the data-gradient GEMM then meets only
    the offsets its rows can have)."""
    key = (grid.B, grid.H, grid.W, k, stride, pad, device.index)
    ent = _TABLES.get(key)
    if ent is not None:
        return ent
    B, H, W = grid.B, grid.H, grid.W
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    b = torch.arange(B, device=device).view(B, 1, 1, 1)
    oy = torch.arange(Ho, device=device).view(1, Ho, 1, 1)
    ox = torch.arange(Wo, device=device).view(1, 1, Wo, 1)
    kk = torch.arange(k * k, device=device).view(1, 1, 1, k * k)
    iy, ix = oy * stride - pad + kk // k, ox * stride - pad + kk % k
    ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
    nbr = torch.where(ok, (b * H + iy) * W + ix, torch.full_like(iy + ix + b, -1)).reshape(B * Ho * Wo, k * k)
    nbr_t = torch.full((B * H * W, k * k), -1, dtype=torch.int64, device=device)
    o_idx = torch.arange(B * Ho * Wo, device=device).view(-1, 1).expand_as(nbr)
    k_idx = kk.reshape(1, k * k).expand_as(nbr)
    v = nbr >= 0
    nbr_t[nbr[v], k_idx[v]] = o_idx[v]
    perm = None
    if stride == 2:
        yy = torch.arange(H, device=device).view(1, H, 1)
        xx = torch.arange(W, device=device).view(1, 1, W)
        cls = (((yy + pad) % 2) * 2 + (xx + pad) % 2).expand(B, H, W).reshape(-1)
        parts = []
        for c in range(4):
            rows = torch.nonzero(cls == c).squeeze(1)
            fill = (-rows.numel()) % 128
            parts += [rows, torch.full((fill,), -1, dtype=torch.int64, device=device)]
        perm = torch.cat(parts).int().contiguous()
    ent = (Grid(B, Ho, Wo), nbr.int().contiguous(), nbr_t.int().contiguous(), perm)
    _TABLES[key] = ent
    return ent


class Conv2d(nn.Module):
    """torch.nn.Conv2d(bias=False) semantics on the row layout.  `forward(x, grid)` -> (y, out_grid)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        assert not bias, "the ResNet convolutions carry no bias"
        self.in_channels, self.out_channels, self.k, self.stride, self.padding = in_channels, out_channels, kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")  # torchvision's ResNet init

    def forward(self, x, grid, bn_stats=False):
        out_grid, nbr, nbr_t, perm = _conv_tables(grid, self.k, self.stride, self.padding, x.device)
        K = self.k * self.k
        w = self.weight.permute(2, 3, 1, 0).reshape(K, self.in_channels, self.out_channels)  # [ky*k+kx][cin][cout]
        same_map = self.stride == 1 and self.k % 2 == 1 and 2 * self.padding == self.k - 1
        holder = [] if bn_stats else None
        if K <= 27:
            tf = lambda transposed, t=(nbr, nbr_t, perm): t if transposed else (t[0], t[1], None)  # noqa: E731
            y = Fn.ConvolutionFunction.apply(x, w.contiguous(), tf, same_map, holder)
        else:  # 7x7 stem: offset groups of at most 27, outputs added
            y = None
            for s in range(0, K, 27):
                e = min(K, s + 27)
                tabs = (nbr[:, s:e].contiguous(), nbr_t[:, s:e].contiguous(), None)
                tf = lambda transposed, t=tabs: t  # noqa: E731
                part = Fn.ConvolutionFunction.apply(x, w[s:e].contiguous(), tf, False, None)
                y = part if y is None else Fn.AddFunction.apply(y, part)
        if holder:
            return y, out_grid, holder[0]
        return y, out_grid, None


class BatchNorm2d(nn.BatchNorm2d):
    """torch.nn.BatchNorm2d parameters / buffers; statistics over the rows (= N, H, W) in the fused HIP kernels.
    `forward(x, relu=, residual=, partial=)` computes relu(bn(x) + residual) in one pass."""

    def forward(self, x, relu=False, residual=None, partial=None):
        training = self.training or not self.track_running_stats
        if training and self.track_running_stats:
            self.num_batches_tracked += 1
        mom = self.momentum if self.momentum is not None else 1.0 / max(float(self.num_batches_tracked), 1.0)
        return Fn.BatchNormFunction.apply(x, self.weight, self.bias, self.running_mean, self.running_var, training, mom,
                                          self.eps, residual, bool(relu), partial if training else None)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size=3, stride=2, padding=1):
        super().__init__()
        self.k, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x, grid):
        out_grid, nbr, nbr_t, _ = _conv_tables(grid, self.k, self.stride, self.padding, x.device)
        return Fn.MaxPoolFunction.apply(x, nbr, nbr_t), out_grid


def global_avg_pool(x, grid):
    boff = _TABLES.get(("boff", grid.B, grid.H, grid.W, x.device.index))
    if boff is None:
        boff = (torch.arange(grid.B + 1, device=x.device) * (grid.H * grid.W)).int()
        _TABLES[("boff", grid.B, grid.H, grid.W, x.device.index)] = boff
    return Fn.GlobalAvgPoolFunction.apply(x, boff)


def to_rows(images):
    """[B, C, H, W] -> ([B*H*W, C] rows, Grid)."""
    B, C, H, W = images.shape
    return images.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().float(), Grid(B, H, W)


def kaiming_fan_out_std(conv):
    return math.sqrt(2.0 / (conv.out_channels * conv.k * conv.k))
