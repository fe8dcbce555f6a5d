"""Drop-in for the MinkowskiEngine Python surface used by the reference hot path
(`import MinkowskiEngine as ME`: train.py:13, resnet.py:6, common.py:9, base_model.py:1,
data/utils.py:1), backed by hand-written HIP kernels for gfx950 (libmink_hip.so)."""
from . import utils  # noqa: F401
from .coords import CoordinateManager, CoordinateMapKey  # noqa: F401
from .modules import (  # noqa: F401
    MinkowskiBatchNorm,
    MinkowskiConvolution,
    MinkowskiConvolutionTranspose,
    MinkowskiGlobalAvgPooling,
    MinkowskiNetwork,
    MinkowskiReLU,
    MinkowskiSumPooling,
    MinkowskiSyncBatchNorm,
    cat,
)
from .functional import set_conv_math  # noqa: F401
from .tensor import SparseTensor, TensorField  # noqa: F401

BACKEND = "hip-gfx950"
SUPPORTS_FUSED_NORM = True  # MinkowskiBatchNorm.forward(x, relu=, residual=)
SUPPORTS_PREPARE_AHEAD = True  # TensorField.prepare_ahead(plan) + CoordinateManager.trace
