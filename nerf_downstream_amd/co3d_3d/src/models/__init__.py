"""Model registry (counterpart of the reference's co3d_3d/src/models/__init__.py:18-20:
lookup by class name through the gin-configurable `get_model`)."""
from nerf_downstream_amd import gin_lite as gin

from .mink.resnet import ResNet14, ResNet18, ResNet34, ResNet50, ResNet101

MODELS = {c.__name__: c for c in (ResNet14, ResNet18, ResNet34, ResNet50, ResNet101)}


@gin.configurable
def get_model(name: str, in_channel, out_channel, sparse=None, ME=None):
    if name not in MODELS:
        raise KeyError(f"model {name!r} is not on the MI355X classification path; available: {sorted(MODELS)}")
    return MODELS[name](in_channel=in_channel, out_channel=out_channel, ME=ME)
