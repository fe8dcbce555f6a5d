"""Coordinate manager of the HIP backend (ME ``CoordinateManager`` counterpart).

One manager is created per input ``TensorField`` -- i.e. rebuilt every iteration and shared
by forward and backward, exactly like ME.  It owns, in HBM:

* per tensor stride ``ts``: the int32 coordinate rows, and the packed-key hash map over them;
* per (ts_in, ts_out): the stride map ``in2out``;
* per (ts_in, ts_out, kernel, dilation): the neighbour table ``nbr[n_out, K]`` and, on demand,
  its transpose ``nbr_t[n_in, K]`` (consumed by dgrad of strided convolutions).

Row order: FIRST OCCURRENCE in input-row order at every level.
Accessors mirror the ones the reference touches: ``stride(key, stride)``
(sparse_conv.py:403-405), ``kernel_map(in_key, out_key, stride, kernel_size, dilation)``
-> ``{k: IntTensor[2,n]}`` (sparse_conv.py:90-96,124-143), ``size(key)`` (sparse_conv.py:80).
"""
import os

import numpy as np
import torch

from .._lib import check, lib

ORIGIN_TS = 0
_STATUS_RANGE, _STATUS_UNSORTED, _STATUS_NOT_ASCENDING = 1, 2, 4


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _Fn():  # (functional imports this module: resolved at call time)
    from . import functional

    return functional


def _as_int(v):
    if isinstance(v, (list, tuple)):
        assert all(int(s) == int(v[0]) for s in v), "anisotropic strides/kernels are not supported"
        return int(v[0])
    return int(v)


_OFFSET_CACHE = {}

# Process-wide overflow sink of the block-index builds (include/mink_hip.h: mink_set_overflow_sink): a pinned host word the insert
# kernel sets when a table was given fewer slots than its map has blocks -- rows are then MISSING from the neighbour tables.  The
# capacities asked for here are exact when the pyramid holds the level at four times the stride, so this only fires on a broken
# invariant; it is looked at before every batch's table build (a numpy read of host memory: no copy, no synchronisation).
_OVERFLOW = {"word": None, "view": None}


def _overflow_sink():
    if _OVERFLOW["word"] is None:
        w = torch.zeros(1, dtype=torch.int32).pin_memory()
        check(lib().mink_set_overflow_sink(w.data_ptr()))
        _OVERFLOW["word"], _OVERFLOW["view"] = w, w.numpy()
    return _OVERFLOW["view"]


def check_block_index_overflow():
    """Raise if any block-index build of this process (so far executed on the device) ran out of slots."""
    v = _overflow_sink()
    if v[0] != 0:
        v[0] = 0
        raise RuntimeError(
            "a block index was built with fewer slots than its map has 4^3-cell blocks: rows are missing from the neighbour tables "
            "of an earlier batch (MinkKernelMapDesc.blk_cap must be mink_table_capacity(number of occupied blocks))")




def kernel_offsets(kernel_size, in_ts, dilation=1):
    """Kernel region (A4): x fastest, z slowest; odd sizes centred, even sizes {0..k-1}."""
    key = (int(kernel_size), int(in_ts), int(dilation))
    if key not in _OFFSET_CACHE:
        _OFFSET_CACHE[key] = _kernel_offsets(*key)
    return _OFFSET_CACHE[key]


_OFFSET_CT = {}


def _kernel_offsets_ct(kernel_size, in_ts, dilation):
    """`kernel_offsets` as the `offsets[81]` member of a MinkKernelMapDesc (filled once per shape: the per-batch table
    plan assigns it to ~14 descriptors)."""
    import ctypes

    key = (kernel_size, in_ts, dilation)
    v = _OFFSET_CT.get(key)
    if v is None:
        o = kernel_offsets(kernel_size, in_ts, dilation).ravel().tolist()
        v = _OFFSET_CT[key] = (ctypes.c_int32 * 81)(*(o + [0] * (81 - len(o))))
    return v


def _kernel_offsets(kernel_size, in_ts, dilation):
    k = int(kernel_size)
    r = np.arange(k) - (k - 1) // 2 if k % 2 == 1 else np.arange(k)
    dz, dy, dx = np.meshgrid(r, r, r, indexing="ij")
    off = np.stack([dx.ravel(), dy.ravel(), dz.ravel()], 1) * int(dilation) * int(in_ts)
    return np.ascontiguousarray(off, dtype=np.int32)


class CoordinateMapKey:
    def __init__(self, tensor_stride, name=""):
        self.ts = int(tensor_stride)
        self.name = name

    def get_tensor_stride(self):
        return [self.ts] * 3

    def get_coordinate_size(self):
        return 4

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and (self.ts, self.name) == (other.ts, other.name)

    def __hash__(self):
        return hash((self.ts, self.name))

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={self.get_tensor_stride()})"


class _Level:
    __slots__ = ("coords", "n", "tkeys", "tvals", "cap", "hash_empty")

    def __init__(self):
        self.hash_empty = False  # True: the pyramid found the rows strictly ascending and left this level's hash map empty


class _Arena:
    """Device memory of ONE coordinate manager as a few large blocks instead of ~25 small ones.

    The maps of a batch are built on the prepare stream and read on the compute / branch / weight-gradient streams, so
    every block carries `record_stream` marks -- and the caching allocator answers a block freed with such marks by
    recording an event on EACH of those streams (the compute stream among them: a barrier packet in the chain of
    dependent launches).  A manager dies once per training step: ~20 blocks made ~70 us of such packets between one
    step's optimizer and the next step's first kernel (bench.py --timeline).  Chunks are sized from what the previous
    manager used, so a steady-state batch takes one or two."""

    last_used = 1 << 20  # bytes the most recent manager has taken so far (class-wide: the next batch is about the same size)

    def __init__(self, device):
        self.device, self.chunks, self.off, self.used = device, [], 0, 0
        self.first = int(_Arena.last_used * 1.1) + 4096  # (read before this arena starts counting)
        self._typed = {}  # dtype -> the newest chunk seen as that type (a take is then one slice: ~20 takes per batch, host time)

    def take(self, shape, dtype):
        if isinstance(shape, (tuple, list)):
            shape = tuple(int(s) for s in shape)
            numel = 1
            for s in shape:
                numel *= s
        else:
            numel = int(shape)
            shape = None
        es = dtype.itemsize
        need = (max(numel, 1) * es + 255) // 256 * 256
        if not self.chunks or self.off + need > self.chunks[-1].numel():
            size = max(need, self.first if not self.chunks else self.chunks[-1].numel() // 2)
            self.chunks.append(torch.empty((size + 255) // 256 * 256, dtype=torch.uint8, device=self.device))
            self.off = 0
            self._typed.clear()
        cv = self._typed.get(dtype)
        if cv is None:
            cv = self._typed[dtype] = self.chunks[-1].view(dtype)
        o = self.off // es  # (offsets are multiples of 256 bytes)
        t = cv[o : o + numel]
        if shape is not None and len(shape) != 1:
            t = t.view(shape)
        self.off += need
        self.used += need
        _Arena.last_used = self.used
        return t


class CoordinateManager:
    def __init__(self, D=3, device=None):
        assert D == 3, "the HIP backend implements 3 spatial dimensions"
        self.D = D
        self.device = device
        self.levels = {}
        self.in2out = {}
        self.tables = {}
        self.field_inverse = None
        self.field_unique_index = None
        self._boff = {}
        self._batch_size = None
        self._batch_checked = False
        self.trace = []  # every map request, in order: lets the next batch be prepared ahead of use
        self._pending_field = None  # pyramid launched, row counts not read back yet (insert_field(defer=True))
        self.prepared = False
        self.xb = None  # (source pointer, bf16 copy [n, 32] of the field's features): TensorField.finish under bf16 storage
        self._lazy_marks = []  # (stream, event, tensors) of every map built on demand (outside replay)
        self._lazy_seen = {}   # stream -> number of marks that stream is already ordered after
        self._replaying = False
        self._arena = None
        self._blk_flags = {}
        self._arena_only = True  # False once a map was allocated outside the arena (the on-demand builders of stride() / tools)

    # ------------------------------------------------------------------ plan record / replay
    @staticmethod
    def compile_plan(trace):
        """De-duplicate a recorded request trace (keeping first-use order); a kernel table that is
        ever requested transposed is built transposed from the start."""
        want_t = {op[1:5] for op in trace if op[0] == "ktable" and op[5]}
        plan, seen = [], set()
        for op in trace:
            if op[0] == "ktable":
                op = op[:5] + (op[1:5] in want_t,)
            if op not in seen:
                seen.add(op)
                plan.append(op)
        return plan

    @staticmethod
    def plan_stride_chain(plan):
        """Strides of `plan` that form the chain ts 1 -> s0 -> s0*s1 ... (built with the field)."""
        chain, ts = [], 1
        for op in plan or ():
            if op[0] == "stride" and op[1] == ts:
                chain.append(op[2])
                ts *= op[2]
        return tuple(chain)

    def replay(self, plan):
        """Build every map of `plan` now (on the current stream) so the forward/backward pass that
        follows finds them cached."""
        self.prepared = bool(plan)  # the maps of the previous forward+backward exist before forward starts
        self._replaying = True  # built on the prepare stream; TensorField.sparse() hands them over as a whole
        try:
            self._replay_ops(plan)
        finally:
            self._replaying = False

    def _replay_ops(self, plan):
        self._build_tables_batched([op for op in plan if op[0] == "ktable"])
        self._build_perms_batched([op for op in plan if op[0] == "perm"])
        for op in plan:
            if op[0] == "stride":
                self.stride(CoordinateMapKey(op[1]), op[2])
            elif op[0] == "ktable":
                self.kernel_table(CoordinateMapKey(op[1]), CoordinateMapKey(op[2]), op[3], op[4], transposed=op[5])
            elif op[0] == "perm":
                self.class_perm(CoordinateMapKey(op[1]), op[2])
            elif op[0] == "boff":
                self.batch_offsets(CoordinateMapKey(op[1]))

    def _build_perms_batched(self, ops):
        """All parity-class permutations of a plan with one native call (four launches, not four per map)."""
        import ctypes

        from .._lib import ClassPartitionDesc

        L = lib()
        todo = [(ts, pad) for _, ts, pad in dict.fromkeys(ops) if ts in self.levels and ("perm", ts, pad) not in self.tables]
        for i0 in range(0, len(todo), 8):
            part = todo[i0 : i0 + 8]
            descs = (ClassPartitionDesc * len(part))()
            for d, (ts, pad) in zip(descs, part):
                lev = self.levels[ts]
                perm = self._take(int(L.mink_class_partition_rows(lev.n, pad)), torch.int32)
                ws = self._take(int(L.mink_class_partition_workspace_bytes(lev.n)), torch.uint8)
                d.coords, d.n, d.ts, d.pad, d.perm = lev.coords.data_ptr(), lev.n, ts, pad, perm.data_ptr()
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
                self.tables[("perm", ts, pad)] = perm
                self._note_lazy(perm)
            check(L.mink_class_partition_batch(len(part), ctypes.byref(descs), _stream()))

    def _build_tables_batched(self, ops):
        """All neighbour tables of a plan with one allocation and one native call."""
        import ctypes

        from .._lib import KernelMapDesc

        todo = []
        for _, ts_in, ts_out, ks, dil, transposed in ops:
            ent = self.tables.get((ts_in, ts_out, ks, dil))
            if ts_in in self.levels and ts_out in self.levels and (ent is None or (transposed and ent[1] is None)):
                todo.append((ts_in, ts_out, ks, dil, transposed))
        if not todo:
            return
        check_block_index_overflow()  # (an earlier batch's builds; arms the sink on first use)
        sizes = []
        for ts_in, ts_out, ks, dil, transposed in todo:
            K = ks ** 3
            sizes.append((self.levels[ts_out].n * K, self.levels[ts_in].n * K if transposed else 0))
        pool = self._take(sum(a + b for a, b in sizes) + 4, torch.int32)
        descs = (KernelMapDesc * len(todo))()
        # block index of every input map that is looked up (4^3-cell blocks: see MinkKernelMapDesc): one buffer set per
        # input tensor stride, built by the first table that uses it
        L = lib()
        blk, need = {}, 0
        for ts_in in dict.fromkeys(t[0] for t in todo):
            n_pad = (max(self.levels[ts_in].n, 8) + 1) // 2 * 2  # regions stay 16-byte aligned (and hold the scan scratch)
            # capacity for the BLOCKS of the map, not for its rows: a 4^3-cell block of the map at tensor stride ts is a cell of
            # the map at 4 ts, so a pyramid that holds that level knows the count exactly (B=16: 825 k rows in 37 k blocks --
            # a 2 MB table that an XCD's L2 holds, where the capacity for the rows was 32 MB to fill and to probe)
            coarse = self.levels.get(4 * ts_in)
            cap = int(L.mink_table_capacity(coarse.n if coarse is not None else self.levels[ts_in].n))
            blk[ts_in] = [cap, need, True, n_pad]  # capacity, int32 offset into the index pool, still to build
            need += 4 * cap + cap + 2 * n_pad + 4  # table (2 x int64 per slot), base, slot, rowids, counter (+pad)
        bpool = self._take(need + 4, torch.int32)
        bbase_ptr = bpool.data_ptr()
        off = 0
        pool_ptr, pool_so = pool.data_ptr(), pool.storage_offset()  # (as_strided counts from the start of the storage)
        ptrs = {ts: (lev.tkeys.data_ptr(), lev.tvals.data_ptr(), lev.coords.data_ptr()) for ts, lev in self.levels.items()}
        for d, (ts_in, ts_out, ks, dil, transposed), (na, nb) in zip(descs, todo, sizes):
            lin, lout = self.levels[ts_in], self.levels[ts_out]
            K = ks ** 3
            # (host time: one view per table -- as_strided -- and pointers by arithmetic; this loop runs once per batch)
            nbr = pool.as_strided((lout.n, K), (K, 1), pool_so + off)
            nbr_t = pool.as_strided((lin.n, K), (K, 1), pool_so + off + na) if transposed else None
            d.nbr, d.nbr_t, d.K = pool_ptr + 4 * off, ((pool_ptr + 4 * (off + na)) if transposed else None), K
            off += na + nb
            d.in_table_keys, d.in_table_vals, d.in_cap = ptrs[ts_in][0], ptrs[ts_in][1], lin.cap
            d.out_coords, d.n_out, d.n_in = ptrs[ts_out][2], lout.n, lin.n
            d.offsets = _kernel_offsets_ct(ks, ts_in, dil)
            cap, boff, build, n_pad = blk[ts_in]
            p0 = bbase_ptr + 4 * boff
            d.in_coords, d.in_ts, d.blk_build, d.blk_cap = ptrs[ts_in][2], ts_in, int(build), cap
            d.blk_table, d.blk_base = p0, p0 + 16 * cap
            d.blk_slot, d.blk_rowids, d.blk_counter = p0 + 20 * cap, p0 + 20 * cap + 4 * n_pad, p0 + 20 * cap + 8 * n_pad
            blk[ts_in][2] = False
            self.tables[(ts_in, ts_out, ks, dil)] = (nbr, nbr_t)
        check(lib().mink_kernel_map_batch(len(todo), ctypes.cast(descs, ctypes.c_void_p), _stream()))
        self._blk_pool = bpool  # (scratch of the call; kept until the manager goes so no stream bookkeeping is needed)
        for ts_in, (cap, boff, _, n_pad) in blk.items():  # the builds' overflow words (MinkKernelMapDesc.blk_counter)
            self._blk_flags[ts_in] = bpool[boff + 5 * cap + 2 * n_pad : boff + 5 * cap + 2 * n_pad + 1]

    def block_index_ok(self):
        """True when every block index built so far found a slot for every block (one host synchronisation; for tests and
        tools -- the capacities this manager asks for are exact, see `_build_tables_batched`)."""
        ok = all(int(f.item()) == -1 for f in self._blk_flags.values())
        if not ok:
            _overflow_sink()[0] = 0  # (reported here: the process-wide sink need not raise for it again)
        return ok

    def _take(self, shape, dtype):
        if self._arena is None:
            self._arena = _Arena(self.device)
        return self._arena.take(shape, dtype)

    def tensors(self):
        if self._arena is not None and not self._lazy_marks and self._arena_only:
            return list(self._arena.chunks)  # every map of a prepared manager lives in the arena's few blocks
        out = [self.field_inverse, self.field_unique_index]
        for lev in self.levels.values():
            out += [lev.coords, lev.tkeys, lev.tvals]
        out += list(self.in2out.values()) + list(self._boff.values())
        for v in self.tables.values():
            out += list(v) if isinstance(v, tuple) else [v]
        return [t for t in out if t is not None]

    def hand_over(self, stream):
        """Maps built on a side stream are about to be used on `stream`: tell the caching
        allocator so their memory is not recycled while kernels of `stream` still read it."""
        for t in self.tensors():
            t.record_stream(stream)

    # ------------------------------------------------------------------ internals
    def _unique(self, src, mode, n, out_ts):
        """keys -> hash map + first-occurrence rows.  Returns (_Level, unique_index, inverse)."""
        L, dev = lib(), self.device
        n = int(n)
        self._arena_only = False  # (on-demand path: plain allocations)
        keys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        meta = torch.zeros(2, dtype=torch.int32, device=dev)  # [n_unique, status]
        check(L.mink_coords_make_keys(src.data_ptr(), mode, n, out_ts, keys.data_ptr(), meta[1:].data_ptr(), _stream()))
        lev = _Level()
        lev.cap = int(L.mink_table_capacity(n))
        lev.tkeys = torch.empty(lev.cap, dtype=torch.int64, device=dev)
        lev.tvals = torch.empty(lev.cap, dtype=torch.int32, device=dev)
        coords = torch.empty(max(n, 1), 4, dtype=torch.int32, device=dev)
        uidx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(int(L.mink_unique_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        check(
            L.mink_coords_unique(
                keys.data_ptr(), n, lev.tkeys.data_ptr(), lev.tvals.data_ptr(), lev.cap, coords.data_ptr(),
                uidx.data_ptr(), inv.data_ptr(), meta.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
            )
        )
        nu, status = meta.tolist()  # the one host sync of this level
        if status & _STATUS_RANGE:
            raise ValueError(
                "coordinate outside the supported range (batch < 65535, |x|,|y|,|z| < 32768 after quantisation)"
            )
        lev.n = nu
        lev.coords = coords[:nu]
        return lev, uidx[:nu], inv[:n]

    # ------------------------------------------------------------------ construction
    def insert_field(self, fcoords, ahead_strides=(), defer=False):
        """A1+A2: floor-quantise the float field and insert it (tensor stride 1); optionally also
        build the chain of stride maps `ahead_strides` (e.g. (2,2,2,2,2) -> ts 2..32) in the same
        native call.  Row counts stay on the device until ONE read-back at the end; with
        `defer=True` that read-back (and everything that needs the counts) is left to
        `finish_field()`, so the caller can queue other work in between and never blocks."""
        assert fcoords.is_cuda and fcoords.dim() == 2 and fcoords.shape[1] == 4
        import ctypes

        self.device = dev = fcoords.device
        fc = fcoords.contiguous()
        mode = 1 if fc.dtype == torch.int32 else 0
        if mode == 0:
            fc = fc.float()
        n = fc.shape[0]
        ts_list = [1]
        for s in ahead_strides:
            ts_list.append(ts_list[-1] * _as_int(s))
        if n == 0:
            raise ValueError("empty coordinate field")
        L, nlev = lib(), len(ts_list)
        cap = int(L.mink_table_capacity(n))
        tkeys = self._take((nlev, cap), torch.int64)
        tvals = self._take((nlev, cap), torch.int32)
        coords = self._take((nlev, n, 4), torch.int32)
        index_b = self._take((nlev, n), torch.int32)
        index_a = self._take(n, torch.int32)
        meta = self._take(nlev + 2, torch.int32)
        ws = self._take(int(L.mink_levels_workspace_bytes(n)), torch.uint8)
        arr = lambda ptrs: (ctypes.c_void_p * nlev)(*ptrs)  # noqa: E731
        check(
            L.mink_coords_build_levels(
                fc.data_ptr(), mode, n, nlev, (ctypes.c_int32 * nlev)(*ts_list),
                arr([tkeys[l].data_ptr() for l in range(nlev)]), arr([tvals[l].data_ptr() for l in range(nlev)]), cap,
                arr([coords[l].data_ptr() for l in range(nlev)]), arr([index_a.data_ptr()] + [None] * (nlev - 1)),
                arr([index_b[l].data_ptr() for l in range(nlev)]), meta.data_ptr(), ws.data_ptr(), ws.numel(), _stream(),
            )
        )
        meta_host = torch.empty(nlev + 2, dtype=torch.int32, pin_memory=True)
        meta_host.copy_(meta, non_blocking=True)
        done = _Fn().current_stream(dev).record_event()
        self._pending_field = (n, ts_list, cap, tkeys, tvals, coords, index_a, index_b, meta_host, done, fc, ws)
        if not defer:
            self.finish_field()
        return CoordinateMapKey(1)

    def finish_field(self):
        """The one host synchronisation of the whole pyramid: read the row counts back and
        publish the levels.  No-op when nothing is pending."""
        if self._pending_field is None:
            return
        n, ts_list, cap, tkeys, tvals, coords, index_a, index_b, meta_host, done, _, _ = self._pending_field
        self._pending_field = None
        done.synchronize()
        m, nlev = meta_host.tolist(), len(ts_list)
        if m[nlev] & _STATUS_RANGE:
            raise ValueError(
                "coordinate outside the supported range (batch < 65535, |x|,|y|,|z| < 32768 after quantisation)"
            )
        self._batch_size = m[nlev + 1]
        n_prev = n
        L = lib()
        for l, ts in enumerate(ts_list):
            lev = _Level()
            # level l > 0 hashed the m[l-1] rows of the level above into a map of exactly their capacity (device-side count)
            cap_l = cap if l == 0 else int(L.mink_table_capacity(m[l - 1]))
            lev.n, lev.cap, lev.tkeys, lev.tvals = m[l], cap_l, tkeys[l, :cap_l], tvals[l, :cap_l]
            lev.coords = coords[l, : m[l]]
            self.levels[ts] = lev
            if l == 0:
                self.field_unique_index, self.field_inverse = index_a[: m[0]], index_b[0, :n]
                lev.hash_empty = not (m[nlev] & _STATUS_NOT_ASCENDING)  # (mink_coords_build_levels: no insert at level 0)
            else:
                self.in2out[(ts_list[l - 1], ts)] = index_b[l, :n_prev]
            n_prev = m[l]

    def hash_map(self, ts):
        """(keys, vals, capacity) of level `ts`'s per-voxel hash map (values = row ids), for callers that look coordinates up
        themselves (`mink_kernel_map`; the networks' tables come from the block index).  A field whose rows arrived strictly
        ascending skipped the insert at level 0 (`MINK_STATUS_NOT_ASCENDING` clear): the map is filled here, on demand."""
        lev = self.levels[ts]
        if lev.hash_empty:
            self._sync_lazy()
            full, _, _ = self._unique(lev.coords, 1, lev.n, 1)  # rows of a level are unique: row ids come out as they are
            lev.tkeys, lev.tvals, lev.cap, lev.hash_empty = full.tkeys, full.tvals, full.cap, False
            self._note_lazy(lev.tkeys, lev.tvals)
        return lev.tkeys, lev.tvals, lev.cap

    # ------------------------------------------------------------------ maps built on demand
    # A map requested for the first time is built on whatever stream asks for it.  With several
    # compute streams (shortcut branch, weight-gradient stream) another stream may hit the cached
    # entry next: it has to wait for the build and keep the memory alive for its own kernels.
    def _note_lazy(self, *tensors):
        if self._replaying or not torch.cuda.is_available():
            return
        cur = _Fn().current_stream(self.device)
        self._lazy_marks.append((cur, cur.record_event(), [t for t in tensors if t is not None]))
        self._lazy_seen[cur] = len(self._lazy_marks)

    def _sync_lazy(self):
        if not self._lazy_marks:
            return
        cur = _Fn().current_stream(self.device)
        seen = self._lazy_seen.get(cur, 0)
        for st, ev, tensors in self._lazy_marks[seen:]:
            if st != cur:
                cur.wait_event(ev)
                for t in tensors:
                    t.record_stream(cur)
        self._lazy_seen[cur] = len(self._lazy_marks)

    def stride(self, key, stride):
        s = _as_int(stride)
        if s == 1:
            return key
        self.trace.append(("stride", key.ts, s))
        ts_out = key.ts * s
        self._sync_lazy()
        if ts_out not in self.levels:
            src = self.levels[key.ts]
            lev, _, inv = self._unique(src.coords, 1, src.n, ts_out)
            self.levels[ts_out] = lev
            self.in2out[(key.ts, ts_out)] = inv
            self._note_lazy(lev.coords, lev.tkeys, lev.tvals, inv)
        return CoordinateMapKey(ts_out)

    def has_level(self, ts):
        self._sync_lazy()
        return int(ts) in self.levels

    def stride_map(self, in_key, out_key):
        k = (in_key.ts, out_key.ts)
        self._sync_lazy()
        if k not in self.in2out:
            raise KeyError(f"no stride map {k}: create the output map with stride() first")
        return self.in2out[k]

    def kernel_table(self, in_key, out_key, kernel_size, dilation=1, transposed=False):
        """Neighbour table nbr[n_out,K] (and nbr_t[n_in,K] when `transposed`)."""
        ks, dil = _as_int(kernel_size), _as_int(dilation)
        kk = (in_key.ts, out_key.ts, ks, dil)
        self.trace.append(("ktable",) + kk + (bool(transposed),))
        self._sync_lazy()
        ent = self.tables.get(kk)
        if ent is None or (transposed and ent[1] is None):
            # a table asked for outside a prepared plan (first batches, tools): the same block-index builder, for one table
            self._build_tables_batched([("ktable",) + kk + (bool(transposed),)])
            ent = self.tables[kk]
            self._note_lazy(*ent)
        return ent

    def kernel_map(self, in_key, out_key, stride=1, kernel_size=3, dilation=1, is_transpose=False, is_pool=False):
        """ME-format kernel map {k: IntTensor[2,n]} (row 0 = in rows, row 1 = out rows)."""
        assert not is_transpose, "transposed kernel maps are out of scope"
        nbr, _ = self.kernel_table(in_key, out_key, kernel_size, dilation)
        n_out, K = nbr.shape
        L = lib()
        counts = torch.empty(K + 1, dtype=torch.int32, device=self.device)
        ws = torch.empty(int(L.mink_rulebook_workspace_bytes(n_out, K)), dtype=torch.uint8, device=self.device)
        pin = torch.empty(max(n_out * K, 1), dtype=torch.int32, device=self.device)
        pout = torch.empty(max(n_out * K, 1), dtype=torch.int32, device=self.device)
        check(
            L.mink_rulebook(nbr.data_ptr(), n_out, K, counts.data_ptr(), pin.data_ptr(), pout.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
        )
        c = counts.tolist()
        return {k: torch.stack([pin[c[k] : c[k + 1]], pout[c[k] : c[k + 1]]]) for k in range(K) if c[k + 1] > c[k]}

    def identity_table(self, key):
        """nbr[n,1] = row index: lets the weight-gradient kernels serve the 1x1x1 convolutions."""
        ck = ("ident", key.ts)
        self._sync_lazy()
        if ck not in self.tables:
            n = self.size(key)
            self._arena_only = False
            self.tables[ck] = torch.arange(n, dtype=torch.int32, device=self.device).reshape(n, 1)
            self._note_lazy(self.tables[ck])
        return self.tables[ck]

    def class_perm(self, key, pad=128):
        """Parity-class row permutation of the map `key` (for dgrad of stride-2 convolutions)."""
        ck = ("perm", key.ts, pad)
        self.trace.append(ck)
        self._sync_lazy()
        if ck not in self.tables:
            L, lev = lib(), self.levels[key.ts]
            perm = self._take(int(L.mink_class_partition_rows(lev.n, pad)), torch.int32)
            ws = self._take(int(L.mink_class_partition_workspace_bytes(lev.n)), torch.uint8)
            check(L.mink_class_partition(lev.coords.data_ptr(), lev.n, key.ts, pad, perm.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
            self.tables[ck] = perm
            self._note_lazy(perm)
        return self.tables[ck]

    # ------------------------------------------------------------------ queries
    def size(self, key):
        return self.batch_size() if key.ts == ORIGIN_TS else self.levels[key.ts].n

    def batch_size(self):
        if not self._batch_checked:
            self._batch_checked = True
            self.batch_offsets(CoordinateMapKey(1))  # validates that the batch column is sorted
        return self._batch_size

    def batch_offsets(self, key):
        """int32[B+1] row ranges per batch index (rows of one batch are contiguous)."""
        self.trace.append(("boff", key.ts))
        self._sync_lazy()
        if key.ts not in self._boff:
            B = self._batch_size
            lev = self.levels[key.ts]
            boff = self._take(B + 1, torch.int32)
            status = self._take(1, torch.int32)
            status.zero_()
            check(lib().mink_batch_offsets(lev.coords.data_ptr(), lev.n, B, boff.data_ptr(), status.data_ptr(), _stream()))
            if key.ts == 1 and int(status.item()) & _STATUS_UNSORTED:
                raise ValueError("batch indices must be non-decreasing (use ME.utils.sparse_collate)")
            self._boff[key.ts] = boff
            self._note_lazy(boff)
        return self._boff[key.ts]

    def get_coordinates(self, key):
        if key.ts == ORIGIN_TS:
            c = torch.zeros(self.batch_size(), 4, dtype=torch.int32, device=self.device)
            c[:, 0] = torch.arange(self.batch_size(), device=self.device)
            return c
        return self.levels[key.ts].coords
