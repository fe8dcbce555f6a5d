"""MI355X-native sparse-3D-convolution classification path (PeRFception-CO3D plenoxels).

``nerf_downstream_amd.minkowski`` is the drop-in for the subset of the MinkowskiEngine
Python API that the reference hot path imports; all arithmetic runs in
hand-written HIP kernels behind the C ABI of ``include/mink_hip.h`` (libmink_hip.so).
"""
__version__ = "0.1.0"


def install_as_minkowski_engine():
    """Register the HIP backend under the module name ``MinkowskiEngine`` so reference-style
    code (`import MinkowskiEngine as ME`) binds to it unchanged."""
    import sys

    from . import minkowski

    sys.modules.setdefault("MinkowskiEngine", minkowski)
    for sub in ("MinkowskiFunctional", "MinkowskiOps", "utils"):  # `import MinkowskiEngine.MinkowskiFunctional as MEF`
        sys.modules.setdefault("MinkowskiEngine." + sub, getattr(minkowski, sub))
    return minkowski
