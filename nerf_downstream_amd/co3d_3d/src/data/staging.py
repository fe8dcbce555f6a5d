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

    def __init__(self, device, host_slots=5, device_slots=4):
        # host_slots: a helper thread that packs `depth` = 2 batches ahead (StagedLoader) touches, at one moment, the buffer it is
        # filling, two queued ones and the one the trainer thread has taken but not yet queued for upload: four, plus one to spare --
        # the buffer filled next was handed to `upload` five batches ago, so its `_host_free` event exists.
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._host = [None] * host_slots        # pinned uint8 buffers
        self._host_free = [None] * host_slots   # event: the upload that last read the buffer
        self._dev = [None] * device_slots       # device uint8 buffers
        self._dev_free = [None] * device_slots  # event on the compute stream: the step that last used the buffer is queued
        self._h = self._d = 0
        self._lock = threading.Lock()

    # ------------------------------------------------------------------ host side (helper thread)
    def pack(self, batch):
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v) and k not in self.HOST_KEEP}
        extras = {k: v for k, v in batch.items() if k not in tensors}
        layout, off = [], 0
        for k, t in tensors.items():
            nb = t.numel() * t.element_size()
            layout.append((k, off, nb, t.dtype, tuple(t.shape)))
            off += _round_up(max(nb, 1))
        with self._lock:
            slot = self._h
            self._h = (self._h + 1) % len(self._host)
        ev = self._host_free[slot]
        if ev is not None:
            ev.synchronize()  # (two batches old: done long ago)
        buf = self._host[slot]
        if buf is None or buf.numel() < off:
            buf = self._host[slot] = torch.empty(_round_up(int(off * 1.25), 1 << 20), dtype=torch.uint8).pin_memory()
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
        self._host_free[packed.slot] = done
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
