"""`import MinkowskiEngine.MinkowskiOps as me` (reference res16unet.py:6): `me.cat`."""
from .modules import cat  # noqa: F401
