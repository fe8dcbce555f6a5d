"""SparseTensor / TensorField of the HIP backend (subset of the ME classes used by the
reference: base_model.py:10-13, resnet.py:164,177, resnet_block.py:66, sparse_conv.py:387-425)."""
import torch

from . import functional as Fn
from .coords import CoordinateManager, CoordinateMapKey


class SparseTensor:
    """Feature matrix F [N,C] (ordinary autograd tensor in HBM) + an immutable coordinate map
    shared by reference through the coordinate manager."""

    def __init__(self, features, coordinate_map_key=None, coordinate_manager=None, coordinates=None, tensor_stride=1):
        if coordinate_map_key is None:
            assert coordinates is not None, "give either a coordinate_map_key or coordinates"
            assert coordinate_manager is None and _is_one(tensor_stride)
            coordinate_manager = CoordinateManager(D=coordinates.shape[1] - 1, device=coordinates.device)
            coordinate_map_key = coordinate_manager.insert_field(coordinates.int())
            if coordinate_manager.levels[1].n != coordinates.shape[0]:
                raise ValueError("duplicate coordinates: use ME.TensorField(...).sparse() to average them")
        self._F = features
        self.coordinate_map_key = coordinate_map_key
        self._manager = coordinate_manager

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self._manager.get_coordinates(self.coordinate_map_key)

    @property
    def coordinate_manager(self):
        return self._manager

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def D(self):
        return self._manager.D

    @property
    def shape(self):
        return self._F.shape

    @property
    def device(self):
        return self._F.device

    def __len__(self):
        return self._F.shape[0]

    def _check(self, other):
        if not (self._manager is other._manager and self.coordinate_map_key == other.coordinate_map_key):
            raise ValueError("SparseTensors must share the coordinate manager and the coordinate map key")

    def slice(self, field):
        """`out.slice(x)` (reference res16unet.py:435): the features of this tensor-stride-1 tensor read
        back at the rows of the TensorField it was quantised from (F[inverse mapping])."""
        if not (_is_one(self.tensor_stride) and field.coordinate_manager is self._manager):
            raise ValueError("slice: needs the tensor-stride-1 tensor and the field it came from")
        m = self._manager
        F = self._F if m.levels[1].n == field.F.shape[0] else self._F[m.field_inverse.long()]
        return TensorField(features=F, coordinates=field.C, _manager=m)

    def __iadd__(self, other):  # `out += residual`, reference resnet_block.py:66
        self._check(other)
        self._F = Fn.AddFunction.apply(self._F, other._F)
        return self

    def __add__(self, other):
        self._check(other)
        return SparseTensor(Fn.AddFunction.apply(self._F, other._F), self.coordinate_map_key, self._manager)

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, tensor_stride={self.tensor_stride})"


def _is_one(ts):
    return all(int(t) == 1 for t in (ts if isinstance(ts, (list, tuple)) else [ts]))


class TensorField:
    """ME.TensorField(coordinates=[N,1+D] float (batch,x,y,z), features=[N,C]).

    Owns a fresh coordinate manager; `.sparse()` floors the coordinates, inserts them into the
    hash map and averages the features of rows that collapse onto one voxel (A1, A2)."""

    def __init__(self, features=None, coordinates=None, plan=None, defer=False, **kwargs):
        """`plan` (extension): a compiled CoordinateManager request plan; all its maps are built
        right here, on the current (side) stream, and `.sparse()` makes the consumer stream wait.
        `defer=True` only launches the coordinate pyramid (no host synchronisation); `finish()`
        -- called explicitly once other work has been queued, or implicitly by `.sparse()` --
        reads the row counts back and builds the rest of the plan on the same stream."""
        assert features is not None and coordinates is not None
        if kwargs.get("_manager") is not None:  # a slice(): shares the manager of the field it came from
            self._F, self._C, self._manager, self._plan, self._ready = features, coordinates, kwargs["_manager"], None, None
            return
        qm = kwargs.get("quantization_mode")
        if qm is not None and getattr(qm, "name", str(qm)) != "UNWEIGHTED_AVERAGE":
            raise NotImplementedError(f"TensorField(quantization_mode={qm}): only UNWEIGHTED_AVERAGE (the ME default the "
                                      "reference relies on) is implemented")
        if not coordinates.is_cuda:
            raise RuntimeError("nerf_downstream_amd.minkowski runs on the GPU only: move the batch to cuda first")
        self._F, self._C = features, coordinates
        m = self._manager = CoordinateManager(D=coordinates.shape[1] - 1, device=coordinates.device)
        self._plan, self._ready = plan, None
        self._build_stream = Fn.current_stream(coordinates.device)
        defer = bool(defer) and plan is not None
        self.coordinate_field_map_key = m.insert_field(coordinates, CoordinateManager.plan_stride_chain(plan), defer=defer)
        if not defer:
            self.finish()

    def finish(self):
        """Second half of a deferred construction (no-op otherwise)."""
        m = self._manager
        if self._plan is None:
            m.finish_field()
            return
        plan, self._plan = self._plan, None
        Fn.skew(self._build_stream)
        Fn.wait_prepare_gate(self._build_stream)
        with torch.cuda.stream(self._build_stream):
            m.finish_field()
            m.replay(plan)
            F = self._F
            if (Fn._STORAGE_B16 and F.is_cuda and F.dtype == torch.float32 and F.dim() == 2 and F.shape[1] <= 32
                    and F.stride(1) == 1 and m.levels[1].n == F.shape[0]):
                # bf16 storage of the full-resolution stage: the bf16 copy of the input rows (no duplicate voxels: the
                # sparse tensor's features ARE these rows) is made here, beside the previous step, not at the head of this one
                m.xb = (F.data_ptr(), Fn.rows_to_bf16(F))
            self._ready = self._build_stream.record_event()

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self._C

    @property
    def coordinate_manager(self):
        return self._manager

    def sparse(self):
        m = self._manager
        self.finish()
        if self._ready is not None:  # maps were built ahead of time on another stream
            cur = Fn.current_stream()
            cur.wait_event(self._ready)
            m.hand_over(cur)
            for t in (self._F, self._C):  # may have been produced on the build stream (GPU-side decode)
                t.record_stream(cur)
            if m.xb is not None:
                m.xb[1].record_stream(cur)
            self._ready = None
        n_unique = m.levels[1].n
        F = self._F
        if n_unique == F.shape[0]:
            Fs = F.float()  # no duplicates: unique rows are the input rows, in order
        else:
            inv = m.field_inverse.long()
            order = torch.sort(inv, stable=True).indices.int()  # members of each voxel, input-row order
            seg = torch.zeros(n_unique + 1, dtype=torch.int32, device=F.device)
            seg[1:] = torch.cumsum(torch.bincount(inv, minlength=n_unique), 0).int()
            Fs = Fn.segment_mean(F, order, seg, n_unique)
        return SparseTensor(Fs, CoordinateMapKey(1), m)
