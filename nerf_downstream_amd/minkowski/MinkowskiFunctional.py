"""`import MinkowskiEngine.MinkowskiFunctional as MEF` (reference modules/common.py:10,56-71): the functional forms of
the pointwise activations on a SparseTensor."""
from . import functional as Fn
from .tensor import SparseTensor


def _wrap(input, out):
    return SparseTensor(out, input.coordinate_map_key, input.coordinate_manager)


def relu(input, inplace=False):
    return _wrap(input, Fn.ReLUFunction.apply(input.F))


def leaky_relu(input, negative_slope=0.01, inplace=False):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "leaky_relu", negative_slope, None))


def elu(input, alpha=1.0, inplace=False):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "elu", alpha, None))


def celu(input, alpha=1.0, inplace=False):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "celu", alpha, None))


def selu(input, inplace=False):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "selu", 0.0, None))


def gelu(input):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "gelu", 0.0, None))


def prelu(input, weight):
    return _wrap(input, Fn.ActivationFunction.apply(input.F, "prelu", 0.0, weight))
