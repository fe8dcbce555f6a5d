"""Input boundary of the Minkowski models (counterpart of the reference's
co3d_3d/src/models/mink/base_model.py:6-13 and models/interface.py:4-9)."""
from abc import ABC, abstractmethod

from nerf_downstream_amd import minkowski as _HIP_ME


class InputInterface(ABC):
    @abstractmethod
    def process_input(self, batch):
        ...


class MinkowskiBaseModel(_HIP_ME.MinkowskiNetwork, InputInterface):
    def __init__(self, dimension=3, ME=None):
        _HIP_ME.MinkowskiNetwork.__init__(self, dimension)
        self._ME = ME or _HIP_ME

    def process_input(self, batch):
        """collated batch dict -> TensorField (float (b,x,y,z) coordinates + features)."""
        return self._ME.TensorField(coordinates=batch["coordinates"], features=batch["features"])
