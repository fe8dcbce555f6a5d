"""Drop-in for the MinkowskiEngine Python surface used by the reference hot path
(`import MinkowskiEngine as ME`: train.py:13, resnet.py:6, common.py:9, base_model.py:1,
data/utils.py:1), backed by hand-written HIP kernels for gfx950 (libmink_hip.so)."""
import enum

from . import MinkowskiFunctional, MinkowskiOps, utils  # noqa: F401
from .coords import CoordinateManager, CoordinateMapKey  # noqa: F401
from .modules import (  # noqa: F401
    MinkowskiBatchNorm,
    MinkowskiCELU,
    MinkowskiConvolution,
    MinkowskiConvolutionTranspose,
    MinkowskiDropout,
    MinkowskiELU,
    MinkowskiGELU,
    MinkowskiGlobalAvgPooling,
    MinkowskiInstanceNorm,
    MinkowskiLeakyReLU,
    MinkowskiLinear,
    MinkowskiNetwork,
    MinkowskiPReLU,
    MinkowskiReLU,
    MinkowskiSELU,
    MinkowskiSumPooling,
    MinkowskiSyncBatchNorm,
    cat,
)
from .functional import set_conv_math, set_conv_storage  # noqa: F401
from .tensor import SparseTensor, TensorField  # noqa: F401



class SparseTensorQuantizationMode(enum.Enum):
    """How `TensorField.sparse()` merges rows that fall into one voxel.  The reference uses the default (co3d.py hands
    distinct integer coordinates; res16unet.py passes UNWEIGHTED_AVERAGE explicitly); only that mode is implemented."""

    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3
    MAX_POOL = 4


BACKEND = "hip-gfx950"
SUPPORTS_FUSED_NORM = True  # MinkowskiBatchNorm.forward(x, relu=, residual=)
SUPPORTS_PREPARE_AHEAD = True  # TensorField.prepare_ahead(plan) + CoordinateManager.trace
