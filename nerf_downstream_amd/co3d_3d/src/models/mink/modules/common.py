"""Layer factories of the Mink-ResNet family (counterpart of the reference's
co3d_3d/src/models/mink/modules/common.py:22-32,73-125,128-179, DENSE branch only).

Every factory takes the ME namespace to build from, so one model definition runs on the HIP
backend (default) and, in tests, on the CPU oracle."""
from nerf_downstream_amd import minkowski as _HIP_ME


def default_me():
    return _HIP_ME


def get_norm(norm_type, n_channels, D=3, bn_momentum=0.1, ME=None):
    ME = ME or _HIP_ME
    if norm_type != "BN":
        raise ValueError(f"Norm type: {norm_type} not supported (only BN is on the classification path)")
    return ME.MinkowskiBatchNorm(n_channels, momentum=bn_momentum)


def get_nonlinearity(nonlinearity_type, ME=None):
    ME = ME or _HIP_ME
    table = {"MinkowskiReLU": ME.MinkowskiReLU, 0: ME.MinkowskiReLU}
    if nonlinearity_type not in table:
        raise ValueError(f"nonlinearity {nonlinearity_type} not supported (only MinkowskiReLU is on the path)")
    return table[nonlinearity_type]


def conv(in_planes, out_planes, kernel_size, stride=1, dilation=1, bias=False, D=-1, conv_mode=0, ME=None):
    assert D > 0, "Dimension must be a positive integer"
    if int(getattr(conv_mode, "value", conv_mode)) != 0:
        raise ValueError("only SparseConvMode.DENSE (0) is implemented; weight-sparse inference is out of scope")
    ME = ME or _HIP_ME
    return ME.MinkowskiConvolution(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size,
                                   stride=stride, dilation=dilation, bias=bias, dimension=D)


def conv_tr(in_planes, out_planes, kernel_size, upsample_stride=1, dilation=1, bias=False, D=-1, conv_mode=0, ME=None):
    """Up-sampling (transposed) convolution of the segmentation family (reference common.py:128-179)."""
    assert D > 0, "Dimension must be a positive integer"
    if int(getattr(conv_mode, "value", conv_mode)) != 0:
        raise ValueError("only SparseConvMode.DENSE (0) is implemented; weight-sparse inference is out of scope")
    ME = ME or _HIP_ME
    return ME.MinkowskiConvolutionTranspose(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size,
                                            stride=upsample_stride, dilation=dilation, bias=bias, dimension=D)
