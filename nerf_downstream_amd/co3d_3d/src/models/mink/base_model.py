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

    def process_input(self, batch, defer=False, _fence=True):
        """`defer=True` (extension, HIP backend): only launch the coordinate pyramid of this batch
        on the side stream and return at once; `finish_input(field)` -- call it after the current
        batch's forward/backward have been queued -- reads the row counts back (already there by
        then, so the host never blocks) and builds the kernel maps."""
        ME = self._ME
        if "links" in batch:  # compact PeRFception batch: de-quantise + links -> coordinates on the device
            with self._prepare_stream_ctx(batch["links"], fence=batch.get("h2d_event", True)):
                coords, feats = ME.utils.decode_plenoxel_batch(batch)
            batch = dict(batch, coordinates=coords, features=feats)
        if "aug_params" in batch:  # augmentation programs drawn by the loader: applied to the whole batch here
            with self._prepare_stream_ctx(batch["coordinates"], fence=batch.get("h2d_event", True)):
                coords, feats, pending = self._augment(batch, count_async=defer)
            batch = dict(batch, coordinates=coords, features=feats)
            if pending is not None:  # a scene drew dropout: the survivor count is still on its way to the host
                return _PendingField(self, batch, pending)
        coords, feats = batch["coordinates"], batch["features"]
        if not (self.prepare_ahead and getattr(ME, "SUPPORTS_PREPARE_AHEAD", False) and coords.is_cuda):
            return ME.TensorField(coordinates=coords, features=feats)
        import torch

        for trace in self._recent_traces:  # the newest field may not have been through forward yet
            if trace:
                plan = ME.CoordinateManager.compile_plan(trace)
                if self._coord_plan is None or len(plan) > len(self._coord_plan):
                    self._coord_plan = plan
        skew = getattr(getattr(ME, "functional", None), "skew", None)
        # a batch that was just copied to the device carries the event recorded after its H2D copies
        # (train.py `_to_device`): the prepare stream waits for exactly that, not for the compute stream
        with self._prepare_stream_ctx(coords, fence=batch.get("h2d_event", False) if _fence else False):
            if skew is not None:
                skew(self._side)
            tf = ME.TensorField(coordinates=coords, features=feats, plan=self._coord_plan or [], defer=defer)
        self._recent_traces = [tf.coordinate_manager.trace] + self._recent_traces[:2]
        return tf

    def _augment(self, batch, count_async=False):
        from nerf_downstream_amd.co3d_3d.src.data.transforms import raw_columns

        if not getattr(self._ME, "SUPPORTS_PREPARE_AHEAD", False):
            raise RuntimeError("augmentation programs are applied by the HIP backend (mink_augment_scenes)")
        if not batch["coordinates"].is_cuda:
            raise RuntimeError("augmentation runs on the GPU: move the batch to cuda first")
        out = self._ME.utils.augment_batch(batch["coordinates"], batch["features"], batch["scene_offsets"],
                                           batch["aug_params"], batch["aug_streams"], batch["aug_seed"],
                                           raw_columns(batch["feature_names"]), count_async=count_async)
        return out if count_async else (*out, None)

    def _prepare_stream_ctx(self, t, fence=True):
        """The prepare stream as a context, or a null context without prepare-ahead.  `fence`: True = wait for
        the current stream (where the batch's H2D copies were queued), an event = wait for it, False = none."""
        import contextlib

        import torch

        if not (self.prepare_ahead and t.is_cuda):
            return contextlib.nullcontext()
        if self._side is None:
            self._side = self._new_prepare_stream(t.device)
        if fence is True:
            self._side.wait_stream(torch.cuda.current_stream(t.device))
        elif fence:
            self._side.wait_event(fence)
        wait_gate = getattr(getattr(self._ME, "functional", None), "wait_prepare_gate", None)
        if wait_gate is not None:  # (MINK_PREPARE_GATE: one map build per step, beside its middle)
            wait_gate(self._side)
        return torch.cuda.stream(self._side)

    def _new_prepare_stream(self, device):
        new_stream = getattr(getattr(self._ME, "functional", None), "new_stream", None)
        import torch

        st = new_stream(device, "prepare") if new_stream is not None else torch.cuda.Stream(device=device)
        from nerf_downstream_amd.memory import reserve_on, reserved

        if reserved(device):  # (a process that reserved a segment for its compute stream -- bench.py, train.py -- gets one here too)
            reserve_on(st)
        return st

    @staticmethod
    def finish_input(field):
        if isinstance(field, _PendingField):
            field = field.materialise()
        finish = getattr(field, "finish", None)
        if finish is not None:
            finish()
        return field


class _PendingField:
    """A deferred batch whose augmentation dropped voxels: the transformed rows are on the device, their
    number arrives through pinned memory.  `finish_input` (called once the current batch's forward and
    backward are queued, so the count is there and the host does not wait) slices the buffers and
    builds the field -- pyramid and maps in one go -- on the prepare stream."""

    def __init__(self, model, batch, pending):
        self.model, self.batch, (self.count, self.event) = model, batch, pending

    def materialise(self):
        self.event.synchronize()
        k = int(self.count[0])
        b = {key: v for key, v in self.batch.items() if not key.startswith("aug_") and key != "links"}
        b["coordinates"], b["features"] = b["coordinates"][:k], b["features"][:k]
        return self.model.process_input(b, defer=False, _fence=False)  # the rows were produced on the prepare stream

    def sparse(self):
        return MinkowskiBaseModel.finish_input(self).sparse()
