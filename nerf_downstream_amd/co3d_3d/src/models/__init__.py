"""Model registry (counterpart of the reference's co3d_3d/src/models/__init__.py:18-20:
lookup by class name through the gin-configurable `get_model`)."""
from nerf_downstream_amd import gin_lite as gin

from .mink.resnet import ResNet14, ResNet18, ResNet34, ResNet50, ResNet101

from .mink import res16unet as _unet

MODELS = {c.__name__: c for c in (ResNet14, ResNet18, ResNet34, ResNet50, ResNet101)}
MODELS.update({n: c for n, c in vars(_unet).items() if n.startswith("Res16UNet") and isinstance(c, type)})


@gin.configurable
def get_model(name: str, in_channel, out_channel, sparse=None, ME=None):
    if name not in MODELS:
        raise KeyError(f"model {name!r} is not implemented; available: {sorted(MODELS)}")
    return MODELS[name](in_channel=in_channel, out_channel=out_channel, ME=ME)
