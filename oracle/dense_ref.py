"""oracle/dense_ref.py -- independent DENSE oracle that pins the CPU restatement.

TEST INFRASTRUCTURE.  A Minkowski sparse convolution equals a dense
``torch.nn.functional.conv3d`` on the densified grid (inactive sites = 0) sampled at
the active output sites:

  * stride 1, k=3          -> conv3d(padding=1) on the grid of step `ts`
  * stride 2, k=3 at ts    -> conv3d(stride=2, padding=1); out index q <-> coord 2*ts*q
  * stride 2, k=1          -> conv3d(kernel 1, stride=2)  (pure sub-sampling)
  * sum-pool k=2,s=2       -> avg_pool3d(2,2) * 8
  * stride 2, k=2          -> conv3d(kernel 2, stride=2, padding=0): even sizes use offsets {0,1}
  * transposed k=2, s=2    -> conv_transpose3d(stride=2)

Weight layout: W_dense[co,ci,dz,dy,dx] = kernel[(dx+1)+3(dy+1)+9(dz+1), ci, co];
coordinate columns (x,y,z) <-> dense dims (W,H,D).  None of this shares code with
the gather/GEMM/scatter restatement, so agreement pins the kernel-offset order, the
floor/stride conventions and the pairing direction of the kernel map.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _grid_index(coords, ts, align):
    """coords int [n,4] -> (b, iz, iy, ix) dense indices on the step-`ts` grid.

    The origin is shifted by a multiple of `align` so negative coordinates work and
    coarse lattices stay aligned with index 0.
    """
    c = np.asarray(coords).astype(np.int64)
    lo = c[:, 1:].min(0)
    shift = -(np.floor(lo / align).astype(np.int64) * align)
    p = (c[:, 1:] + shift) // ts
    return c[:, 0], p[:, 2], p[:, 1], p[:, 0], shift


def densify(coords, feats, ts, align, pad_to=2):
    b, iz, iy, ix, shift = _grid_index(coords, ts, align)
    B = int(b.max()) + 1
    dims = [int(v.max()) + 1 for v in (iz, iy, ix)]
    dims = [((d + pad_to - 1) // pad_to) * pad_to + pad_to for d in dims]
    g = torch.zeros(B, feats.shape[1], *dims, dtype=feats.dtype)
    g[torch.as_tensor(b), :, torch.as_tensor(iz), torch.as_tensor(iy), torch.as_tensor(ix)] = feats
    return g, shift


def sample(grid, out_coords, ts_out, shift):
    c = np.asarray(out_coords).astype(np.int64)
    p = (c[:, 1:] + shift) // ts_out
    return grid[torch.as_tensor(c[:, 0]), :, torch.as_tensor(p[:, 2]), torch.as_tensor(p[:, 1]), torch.as_tensor(p[:, 0])]


def to_dense_weight(kernel, k):
    cin, cout = kernel.shape[-2:]
    return kernel.reshape(k, k, k, cin, cout).permute(4, 3, 0, 1, 2).contiguous()


def conv(in_coords, feats, kernel, ksize, stride, ts_in, out_coords):
    """Dense evaluation of MinkowskiConvolution(k=ksize, stride) at `out_coords`."""
    ts_out = ts_in * stride
    g, shift = densify(in_coords, feats, ts_in, align=ts_out * 2)
    w = to_dense_weight(kernel.reshape(ksize**3, kernel.shape[-2], kernel.shape[-1]), ksize)
    y = F.conv3d(g, w, stride=stride, padding=(ksize - 1) // 2)
    return sample(y, out_coords, ts_out, shift)


def sum_pool(in_coords, feats, ts_in, out_coords):
    g, shift = densify(in_coords, feats, ts_in, align=ts_in * 4)
    y = F.avg_pool3d(g, 2, 2) * 8.0
    return sample(y, out_coords, ts_in * 2, shift)


def conv_transpose(in_coords, feats, kernel, ksize, stride, ts_in, out_coords):
    """Dense evaluation of MinkowskiConvolutionTranspose(k=ksize, stride=ksize) from the step-`ts_in`
    grid onto `out_coords` (step ts_in / stride): conv_transpose3d(stride) writes in[c] @ W[d] to the
    fine site stride*c + d -- the transposed pairing of the even-size region {0..k-1} of conv()."""
    assert ksize == stride, "non-overlapping up-sampling only (kernel_size == stride, as Res16UNet uses it)"
    ts_out = ts_in // stride
    g, shift = densify(in_coords, feats, ts_in, align=ts_in * 2)
    k = kernel.reshape(ksize, ksize, ksize, kernel.shape[-2], kernel.shape[-1])
    w = k.permute(3, 4, 0, 1, 2).contiguous()  # conv_transpose3d weight: [cin, cout, dz, dy, dx]
    y = F.conv_transpose3d(g, w, stride=stride)
    return sample(y, out_coords, ts_out, shift)
