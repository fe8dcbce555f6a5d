"""Host -> device staging of collated batches for the trainer (co3d_3d/train.py).

The reference hands Lightning a DataLoader with `pin_memory=False` (co3d_3d/src/data/data_module.py:49-69) and lets it
move every tensor of a batch with its own pageable-memory copy on the compute stream.  At the step times of this
backend (3.4 ms for 16 scenes) that is the bottleneck twice over: a pageable copy is synchronous (the runtime stages it
through its own bounce buffer and the launching thread waits), and it sits on the stream the step runs on.  Here

* `pack()` (any thread: the trainer runs it in a helper thread, the memcpy releases the GIL) copies all tensors of a
  batch into ONE pinned host buffer out of a small ring, 256-byte aligned;
* `upload()` (trainer thread) is ONE asynchronous copy of that buffer into a device buffer out of a second ring, on a
  dedicated copy stream, and returns the batch as views of the device buffer plus `h2d_event` -- the event
  `MinkowskiBaseModel.process_input` already waits for on its prepare stream;
* `release()` marks the point on the compute stream behind which the batch's device buffer may be overwritten.

No allocator traffic per batch (the rings are allocated once and grown only when a batch is larger than any before),
no `record_stream` bookkeeping, one copy launch and two event operations per batch.
"""
import threading

import torch

_ALIGN = 256


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


class PackedBatch:
    __slots__ = ("slot", "nbytes", "layout", "extras")

    def __init__(self, slot, nbytes, layout, extras):
        self.slot, self.nbytes, self.layout, self.extras = slot, nbytes, layout, extras


class PinnedStager:
    HOST_KEEP = ("aug_params",)  # tensors that stay on the host (the augmentation kernel uploads its parameter rows itself)

    def __init__(self, device, host_slots=6, device_slots=4):
        import queue

        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._host = [None] * host_slots        # pinned uint8 buffers
        # free pinned buffers, each with the event of the upload that last read it: `acquire_host` BLOCKS until one is back, so a
        # producer may run as far ahead as it likes (a ring indexed modulo its size was overwritten under a batch the trainer
        # thread had taken but not yet queued for upload: labels of the wrong batch, found by the first 600-step run)
        self._host_free = queue.Queue()
        for i in range(host_slots):
            self._host_free.put((i, None))
        self._dev = [None] * device_slots       # device uint8 buffers
        self._dev_free = [None] * device_slots  # event on the compute stream: the step that last used the buffer is queued
        self._d = 0

    # ------------------------------------------------------------------ host side (helper thread)
    def acquire_host(self, nbytes):
        """(slot, pinned uint8 buffer of at least `nbytes`): a buffer no upload is reading any more (blocks until there is one)."""
        slot, ev = self._host_free.get()
        if ev is not None:
            ev.synchronize()
        buf = self._host[slot]
        if buf is None or buf.numel() < nbytes:
            buf = self._host[slot] = torch.empty(_round_up(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
        return slot, buf

    def pack(self, batch):
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v) and k not in self.HOST_KEEP}
        extras = {k: v for k, v in batch.items() if k not in tensors}
        layout, off = [], 0
        for k, t in tensors.items():
            nb = t.numel() * t.element_size()
            layout.append((k, off, nb, t.dtype, tuple(t.shape)))
            off += _round_up(max(nb, 1))
        slot, buf = self.acquire_host(off)
        jobs = []
        for k, o, nb, dt, shape in layout:
            if not nb:
                continue
            dst, src = buf[o : o + nb].view(dt).view(shape), tensors[k]
            if nb >= (4 << 20) and len(shape) >= 1 and shape[0] >= 2 * self.PACK_THREADS:
                # a large tensor goes in row chunks over a few threads (the copy releases the GIL): one thread moves ~8 GB/s,
                # and a 29 MB batch every 3.4 ms is 8.5 GB/s before anything else happens on that thread
                step = -(-shape[0] // self.PACK_THREADS)
                jobs += [(dst[i : i + step], src[i : i + step]) for i in range(0, shape[0], step)]
            else:
                dst.copy_(src)
        if jobs:
            list(self._pool().map(lambda a: a[0].copy_(a[1]), jobs))
        return PackedBatch(slot, off, layout, extras)

    PACK_THREADS = 3
    _POOL = None

    @classmethod
    def _pool(cls):
        if cls._POOL is None:
            from concurrent.futures import ThreadPoolExecutor

            cls._POOL = ThreadPoolExecutor(max_workers=cls.PACK_THREADS, thread_name_prefix="mink-pack")
        return cls._POOL

    # ------------------------------------------------------------------ device side (trainer thread)
    def upload(self, packed):
        slot = self._d
        self._d = (self._d + 1) % len(self._dev)
        dbuf = self._dev[slot]
        if dbuf is None or dbuf.numel() < packed.nbytes:
            if dbuf is not None:
                torch.cuda.current_stream(self.device).synchronize()  # (a larger batch than ever before: rare)
            dbuf = self._dev[slot] = torch.empty(_round_up(int(packed.nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=self.device)
        cs = self.copy_stream
        if self._dev_free[slot] is not None:
            cs.wait_event(self._dev_free[slot])
        with torch.cuda.stream(cs):
            dbuf[: packed.nbytes].copy_(self._host[packed.slot][: packed.nbytes], non_blocking=True)
            done = cs.record_event()
        self._host_free.put((packed.slot, done))
        out = dict(packed.extras)
        for k, o, nb, dt, shape in packed.layout:
            out[k] = dbuf[o : o + nb].view(dt).view(shape) if nb else torch.empty(shape, dtype=dt, device=self.device)
        out["h2d_event"], out["_stage_slot"] = done, slot
        return out

    def release(self, batch, stream=None):
        """The work that reads `batch` is queued on `stream` (default: the current stream): its buffer may be re-used behind it."""
        slot = batch.get("_stage_slot")
        if slot is not None:
            self._dev_free[slot] = (stream or torch.cuda.current_stream(self.device)).record_event()


class StagedLoader:
    """Iterates a DataLoader in a helper thread and packs every batch into pinned memory there (`depth` batches ahead);
    the trainer thread receives PackedBatch objects and uploads them.  One pass = one epoch."""

    _END = object()

    def __init__(self, loader_iter, stager, depth=2):
        import queue

        self.q = queue.Queue(maxsize=depth)
        self.err = None
        self.stager = stager

        def run():
            try:
                for b in loader_iter:
                    self.q.put(stager.pack(b))
            except BaseException as e:  # noqa: BLE001 -- handed to the consumer
                self.err = e
            self.q.put(self._END)

        self.thread = threading.Thread(target=run, daemon=True, name="mink-staging")
        self.thread.start()

    def next(self):
        item = self.q.get()
        if item is self._END:
            if self.err is not None:
                raise self.err
            return None
        return self.stager.upload(item)


class DirectCompactLoader:
    """One epoch of a COMPACT PeRFception-CO3D dataset (Co3DDatasetBase(compact=True): samples stay links / density / uint8 SH) read
    from the scene files STRAIGHT INTO the pinned staging buffers: a batch is collated by `os.preadv` calls -- one per array and
    scene, on a small thread pool, each a kernel copy out of the page cache with the GIL released -- into the slices of one pinned
    buffer where the batch's tensors live; the trainer uploads that buffer with one copy.  No DataLoader worker processes, no
    shared-memory hand-over, no intermediate tensor: the DataLoader path moves a batch's 29 MB three times on the host (map -> shared
    memory in a worker, page-faulting both sides; shared memory -> pinned in the trainer) and is bound by the workers' page faults
    (measured on the GPU box, 12 workers: 7.7 ms per batch; profiles/r06_train_e2e.txt), this path once.
    Used by co3d_3d/train.py when the dataset allows it (`usable`): data.npz scenes with stored members of the expected types and no
    per-scene row filter in the augmentation recipe (those need the density values on the host: the DataLoader path serves them).
    Yields PackedBatch objects with exactly the keys `collate_mink` produces for compact samples."""

    def __init__(self, dataset, index_iter, batch_size, stager, threads=8, depth=3, drop_last=True):
        import queue
        from concurrent.futures import ThreadPoolExecutor

        self.ds, self.stager, self.bs, self.drop_last = dataset, stager, int(batch_size), drop_last
        self.q = queue.Queue(maxsize=depth)
        self.err = None
        self.pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="mink-read")
        self._idx = index_iter
        self.thread = threading.Thread(target=self._run, daemon=True, name="mink-direct-loader")
        self.thread.start()

    _END = object()
    _LAYOUTS = {}  # scene file -> (n, offsets of links / density / sh, sh_scale [27], sh_min [27]) or None (not readable this way)

    @classmethod
    def usable(cls, dataset, probe=8):
        """Compact samples, no per-scene row filter, and the scenes are `data.npz` files this reader can serve: `probe` of them, spread
        over the file list, are looked at (0.3 ms each, cached) -- a tree of Plenoxel `last.ckpt` scenes, or of compressed archives, goes
        through the DataLoader path instead of failing at its first batch."""
        import os

        t = getattr(dataset, "transformations", None)
        if not (bool(getattr(dataset, "compact", False)) and hasattr(dataset, "files") and (t is None or not getattr(t, "prefilters", ()))):
            return False
        n = len(dataset.files)
        if n == 0:
            return False
        for i in sorted({(j * (n - 1)) // max(probe - 1, 1) for j in range(min(probe, n))}):
            path = os.path.join(dataset.data_root, f"plenoxel_co3d_{dataset.files[i][1]}", "data.npz")
            if cls._layout(path) is None:
                return False
        return True

    @classmethod
    def _layout(cls, path):
        import numpy as np

        ent = cls._LAYOUTS.get(path, False)
        if ent is not False:
            return ent
        from .co3d import _read_npz, npz_members

        ent = None
        try:
            mem, slow = npz_members(path)
            ok = (not slow and all(k in mem for k in ("links", "density", "sh", "sh_scale", "sh_min")) and mem["links"][1] == np.int32
                  and mem["density"][1] == np.float32 and mem["sh"][1] == np.uint8)
            if ok:
                n = int(mem["links"][2][0])
                ok = int(np.prod(mem["density"][2])) == n and int(np.prod(mem["sh"][2])) == 27 * n
            if ok:
                z = _read_npz(path)
                bc = lambda a: np.broadcast_to(np.asarray(a, np.float32).reshape(-1), (27,)).copy()  # noqa: E731
                ent = (n, mem["links"][0], mem["density"][0], mem["sh"][0], bc(z["sh_scale"]), bc(z["sh_min"]))
        except (OSError, ValueError, KeyError):
            ent = None
        cls._LAYOUTS[path] = ent
        return ent

    def _run(self):
        try:
            pending = []
            batch = []
            for i in self._idx:
                batch.append(int(i))
                if len(batch) == self.bs:
                    pending.append(self._submit(batch))
                    batch = []
                    if len(pending) > 1:
                        self._finish(pending.pop(0))
            if batch and not self.drop_last:
                pending.append(self._submit(batch))
            for p in pending:
                self._finish(p)
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            self.err = e
        self.q.put(self._END)

    def _finish(self, item):
        packed, futures = item
        for f in futures:
            f.result()
        self.q.put(packed)

    def _submit(self, indices):
        import os

        import numpy as np

        ds = self.ds
        scenes = []
        for i in indices:
            label, inst_id = ds.files[i]
            path = os.path.join(ds.data_root, f"plenoxel_co3d_{inst_id}", "data.npz")
            lay = self._layout(path)
            if lay is None:
                raise RuntimeError(f"{path}: not a stored-member data.npz of the expected types (use the DataLoader path: MINK_DIRECT_LOADER=0)")
            scenes.append((path, lay, ds.CLASS_LABELS.index(label)))
        B = len(scenes)
        n = [lay[0] for _, lay, _ in scenes]
        N = int(sum(n))
        specs = [("links", torch.int32, (N,)), ("density", torch.float32, (N,)), ("sh_q", torch.uint8, (N, 27)),
                 ("scene_offsets", torch.int32, (B + 1,)), ("sh_scale", torch.float32, (B, 27)), ("sh_min", torch.float32, (B, 27)),
                 ("labels", torch.int64, (B,))]
        progs = None
        if ds.transformations is not None:  # augmentation programs: drawn here, applied on the GPU (collate_mink / _with_programs)
            progs = [ds.transformations.sample() for _ in range(B)]
            specs.append(("aug_streams", torch.int32, (B,)))
        layout, off = [], 0
        for k, dt, shape in specs:
            nb = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
            layout.append((k, off, nb, dt, tuple(shape)))
            off += _round_up(max(nb, 1))
        slot, buf = self.stager.acquire_host(off)
        view = {k: (buf[o : o + nb].view(dt).view(shape) if nb else None) for k, o, nb, dt, shape in layout}
        offs = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        view["scene_offsets"].copy_(torch.from_numpy(offs.astype(np.int32)))
        view["labels"].copy_(torch.tensor([lab for _, _, lab in scenes], dtype=torch.int64))
        view["sh_scale"].copy_(torch.from_numpy(np.stack([lay[4] for _, lay, _ in scenes])))
        view["sh_min"].copy_(torch.from_numpy(np.stack([lay[5] for _, lay, _ in scenes])))
        extras = {"feature_names": tuple(ds.features), "reso": (128, 128, 128)}
        if progs is not None:
            view["aug_streams"].copy_(torch.from_numpy(np.array([p[1] for p in progs], dtype=np.uint32).view(np.int32).copy()))
            extras["aug_params"] = torch.stack([torch.from_numpy(p[0]) for p in progs])
            extras["aug_seed"] = int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
        raw = buf.numpy()  # (a view of the pinned bytes: preadv fills it in place)
        base = {k: o for k, o, _, _, _ in layout}

        def read_scene(j):
            path, lay, _ = scenes[j]
            nj, r0 = lay[0], int(offs[j])
            if nj == 0:
                return
            fd = os.open(path, os.O_RDONLY)
            try:
                for key, file_off, width in (("links", lay[1], 4), ("density", lay[2], 4), ("sh_q", lay[3], 27)):
                    dst = memoryview(raw[base[key] + r0 * width : base[key] + (r0 + nj) * width])
                    got = 0
                    while got < nj * width:
                        k = os.preadv(fd, [dst[got:]], file_off + got)
                        if k <= 0:
                            raise OSError(f"{path}: short read")
                        got += k
            finally:
                os.close(fd)

        futures = [self.pool.submit(read_scene, j) for j in range(B)]
        return PackedBatch(slot, off, layout, extras), futures

    def next(self):
        item = self.q.get()
        if item is self._END:
            self.pool.shutdown(wait=False)
            if self.err is not None:
                raise self.err
            return None
        return self.stager.upload(item)
