"""Input boundary of the Minkowski models (counterpart of the reference's
co3d_3d/src/models/mink/base_model.py:6-13 and models/interface.py:4-9)."""
from abc import ABC, abstractmethod

from nerf_downstream_amd import minkowski as _HIP_ME


class InputInterface(ABC):
    @abstractmethod
    def process_input(self, batch):
        ...


class MinkowskiBaseModel(_HIP_ME.MinkowskiNetwork, InputInterface):
    """`process_input(batch)` -> TensorField, as in the reference.  On the HIP backend it also
    prepares, on a side HIP stream, every coordinate / kernel map the network asked for on the
    previous batch (hash build, stride maps, neighbour tables are pure functions of the
    coordinates), so calling it for batch i+1 while batch i is still in backward hides the map
    construction and its host synchronisations behind compute."""

    def __init__(self, dimension=3, ME=None):
        _HIP_ME.MinkowskiNetwork.__init__(self, dimension)
        self._ME = ME or _HIP_ME
        self._coord_plan = None
        self._recent_traces = []  # map-request traces of the last few fields (filled by their forward pass)
        self._side = None
        self.prepare_ahead = True

    def process_input(self, batch, defer=False):
        """`defer=True` (extension, HIP backend): only launch the coordinate pyramid of this batch
        on the side stream and return at once; `finish_input(field)` -- call it after the current
        batch's forward/backward have been queued -- reads the row counts back (already there by
        then, so the host never blocks) and builds the kernel maps."""
        ME = self._ME
        if "links" in batch:  # compact PeRFception batch: de-quantise + links -> coordinates on the device
            with self._prepare_stream_ctx(batch["links"]):
                coords, feats = ME.utils.decode_plenoxel_batch(batch)
            batch = dict(batch, coordinates=coords, features=feats)
        if "aug_params" in batch:  # augmentation programs drawn by the loader: applied to the whole batch here
            with self._prepare_stream_ctx(batch["coordinates"]):
                coords, feats = self._augment(batch)
            batch = dict(batch, coordinates=coords, features=feats)
        coords, feats = batch["coordinates"], batch["features"]
        if not (self.prepare_ahead and getattr(ME, "SUPPORTS_PREPARE_AHEAD", False) and coords.is_cuda):
            return ME.TensorField(coordinates=coords, features=feats)
        import torch

        for trace in self._recent_traces:  # the newest field may not have been through forward yet
            if trace:
                plan = ME.CoordinateManager.compile_plan(trace)
                if self._coord_plan is None or len(plan) > len(self._coord_plan):
                    self._coord_plan = plan
        if self._side is None:
            self._side = torch.cuda.Stream(device=coords.device)
        skew = getattr(getattr(ME, "functional", None), "skew", None)
        if skew is not None:
            skew(self._side)
        with torch.cuda.stream(self._side):
            tf = ME.TensorField(coordinates=coords, features=feats, plan=self._coord_plan or [], defer=defer)
        self._recent_traces = [tf.coordinate_manager.trace] + self._recent_traces[:2]
        return tf

    def _augment(self, batch):
        from nerf_downstream_amd.co3d_3d.src.data.transforms import raw_columns

        if not getattr(self._ME, "SUPPORTS_PREPARE_AHEAD", False):
            raise RuntimeError("augmentation programs are applied by the HIP backend (mink_augment_scenes)")
        if not batch["coordinates"].is_cuda:
            raise RuntimeError("augmentation runs on the GPU: move the batch to cuda first")
        return self._ME.utils.augment_batch(batch["coordinates"], batch["features"], batch["scene_offsets"],
                                            batch["aug_params"], batch["aug_streams"], batch["aug_seed"],
                                            raw_columns(batch["feature_names"]))

    def _prepare_stream_ctx(self, t):
        """The prepare stream as a context (made to wait for the current stream, where the H2D copies of
        the batch were queued), or a null context without prepare-ahead."""
        import contextlib

        import torch

        if not (self.prepare_ahead and t.is_cuda):
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream(device=t.device)
        self._side.wait_stream(torch.cuda.current_stream(t.device))
        return torch.cuda.stream(self._side)

    @staticmethod
    def finish_input(field):
        finish = getattr(field, "finish", None)
        if finish is not None:
            finish()
        return field
