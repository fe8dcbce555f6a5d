"""Data parallelism for the sparse-conv classifier: one process per GPU, gradients averaged
with bucketed all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference uses PyTorch-Lightning DDP (co3d_3d/train.py:174-186): per-rank batches,
per-rank BatchNorm statistics, mean of gradients across ranks once per step.  Here:

* all parameter gradients live in ONE flat fp32 buffer (`.grad` tensors are views), laid out
  in REVERSE registration order so the buffer fills front-to-back as backward proceeds
  (layer4 -- 74 % of the bytes -- first);
* the buffer is cut into buckets of >= 32 MiB (one parameter tensor is never split; Mink-ResNet34's 63.5 M parameters = 254 MB
  are six messages of 25-54 MiB and the stem's, Mink-ResNet14's 14.4 M = 57.6 MB two and the stem's: few, large collectives
  are what xGMI's point-to-point links want, and every collective costs the host ~50 us of a step that is nearly host-bound
  under data parallelism -- 15 buckets of >= 8 MiB were 0.77 ms per ResNet34 step); the LAST bucket, which cannot overlap with
  anything, holds only the stem convolution + its batch norm (0.2 MB);
* a post-accumulate-grad hook counts ready parameters per bucket and launches the bucket's
  `all_reduce(async_op=True)` once it is complete -- RCCL runs it on its own HIP stream, overlapped
  with the remaining backward kernels.  The native trunk (one call for the whole backward pass) reports its blocks through
  a callback of the library at the point of the weight-gradient stream behind which a block's gradients are complete, and
  the bucket a block completes is all-reduced FROM that stream right there (round 5: no event per block, no stream of its
  own for the launches -- a fifth busy hardware queue is what made a step 1.5-3x slower at GPU_MAX_HW_QUEUES >= 8,
  hwqueues.py).  On the GPU (gradient-sink mode) a complete bucket is not
  launched from inside the backward of the small deep layers, where the host is what the GPU waits
  for, but at the next `flush()` point: behind the backward of a wide-and-shallow layer -- a static
  property of the layer, so every rank issues its collectives at the same points;
* `finish()` waits for all buckets (RCCL averages inside the collective; with gloo the sum is scaled by
  1/world here), before the optimizer step.
"""
import os

import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, force=False):
        """`force`: run the whole machinery (hooks, gradient sink, collectives) in a one-rank group too -- for measuring
        its overhead on a single GPU (bench.py BENCH_FORCE_REDUCER=1).

        With one rank (and no `force`) only the flat gradient buffer is kept: the backward kernels write the
        gradients straight into it (gradient sink), there are no hooks and no collectives -- one GPU and N GPUs run
        the same step, the collectives being the only difference."""
        self.group = process_group
        self.force = bool(force)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1 and dist.get_backend(process_group) == "gloo":
            # gloo (CPU rehearsals / tests) stalls on 32 MiB device tensors; RCCL wants them large
            bucket_bytes = min(bucket_bytes, 8 << 20)
        params = [p for p in module.parameters() if p.requires_grad][::-1]
        # every slice starts on a 256-byte boundary: the fused optimizer kernel only takes its 16-byte path when
        # all the pointers of a tensor are aligned (unaligned views cost it 3x: 149 us against 50 per step)
        ALIGN = 64
        total = sum(-(-p.numel() // ALIGN) * ALIGN for p in params)
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []  # (start, end, n_params)
        self._bucket_of = {}
        off, bstart, bcount = 0, 0, 0
        self._views = {}  # id(param) -> its .grad view (identity-checked in view_for)
        self._params = params  # keeps the ids alive
        self.offsets = []  # start of every parameter's slice, in the order of `_params` (FlatSGD lays weights and momentum out the same way)
        self.cleared = False  # the optimizer's step has already cleared the buffer (FlatSGD): zero_grad() skips its memset
        # The network's first layer is the last to get its gradient, and nothing is left to overlap its collective
        # with: keep that exposed message minimal.  The first registered weight tensor and the 1-D parameters that
        # follow it (the stem convolution and its batch norm: 0.2 MB for the ResNets) get a bucket of their own, so
        # the bucket before it (layer1) goes out beside the stem's weight-gradient kernel (0.9 ms at B=16) instead
        # of waiting for it.
        fwd = params[::-1]
        n_tail = 0
        if len(fwd) > 6 and fwd[0].dim() > 1:
            n_tail = 1
            while n_tail < len(fwd) and fwd[n_tail].dim() == 1:
                n_tail += 1
        tail_at = len(params) - n_tail if n_tail else -1
        for i, p in enumerate(params):
            if i == tail_at and bcount:
                self.buckets.append((bstart, off, bcount))
                bstart, bcount = off, 0
            self.offsets.append(off)
            p.grad = self.flat[off : off + p.numel()].view_as(p)
            self._views[id(p)] = p.grad
            self._bucket_of[id(p)] = len(self.buckets)  # keyed by id: Tensor.__hash__ is a Python-level call (~1 us) and
            # the per-step path would make half a dozen of them per parameter
            off += -(-p.numel() // ALIGN) * ALIGN
            bcount += 1
            if (off - bstart) * 4 >= bucket_bytes and not (0 <= tail_at <= i):
                self.buckets.append((bstart, off, bcount))
                bstart, bcount = off, 0
        if bcount:
            self.buckets.append((bstart, off, bcount))
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._next = 0  # buckets go out in ascending index order, whatever order they complete in (see _drain)
        # The native trunk queues the WHOLE backward pass in one call and then reports its blocks' gradients one block at a
        # time, each with the event (a raw hipEvent_t of libmink_hip.so) behind which that block's gradients are complete:
        # `stage_event` is the event of the block being reported, `_bucket_event[b]` the one that completed bucket b.  Its
        # all-reduce is then issued from a stream of its own that waits for that event only (`_launch`), not for the
        # tail of the compute streams -- which by then is the end of backward.
        self.stage_event = None
        self.stage_stream = None  # set while a block-done hook of the native trunk reports a block (see _launch)
        self._bucket_event = [None] * len(self.buckets)
        self._launch_stream = None
        # One rank: nothing is all-reduced, but a consumer (FlatSGD(in_backward=True)) may ask to be called with every bucket the
        # native trunk completes, from inside the backward call on its weight-gradient stream -- `_bucket_cb(b, start, end, stream)`
        self._bucket_cb = None
        self.launch_log = []  # (bucket, start, end) of every collective issued, in issue order (tests compare ranks)
        self._work = []
        self._hooks = []
        self._home = None  # the stream the step runs on (captured in zero_grad)
        # `defer` (switched on together with the gradient sink below): complete buckets wait for flush() / finish()
        # instead of being launched from inside the backward of the small deep layers
        self.defer = False
        self._written = set()  # parameters whose gradient was written in place this step
        self._counted = set()  # parameters already counted towards their bucket this step
        self._collect = self.world > 1 or self.force  # buckets are all-reduced
        if self._collect:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self._collect or dev.type == "cuda":
            if dev.type == "cuda":
                from .minkowski import functional as Fn

                if not self._collect:
                    Fn.set_grad_sink(self)
                elif os.environ.get("MINK_DP_MULTISTREAM", "1") == "0":  # conservative schedule: compute + prepare streams only
                    Fn.set_wgrad_overlap(False)
                    Fn.set_branch_fork(False)
                else:
                    # a shortcut-branch stream of its own stays off under data parallelism: with the process group's
                    # streams in the picture it gains nothing at six or seven hardware queues and costs a factor at
                    # eight (5.8 ms/step; DESIGN Appendix A).  The native trunk runs the branch on the weight-gradient
                    # stream instead, which is idle in forward and has room in backward: 4.00 -> 3.88 ms with a one-rank
                    # RCCL group (3.82 without the data-parallel machinery); fp32 math only (Fn.trunk_branch_mode)
                    Fn.set_branch_fork(False)
                    Fn.set_trunk_branch_on_side(True)
                    # convolution weight gradients are written straight into the flat buffer by the
                    # weight-gradient stream (no per-layer accumulate + join on the compute stream), batch-norm
                    # scale / shift gradients by their backward kernel (no accumulate launch per parameter)
                    Fn.set_grad_sink(self)
                    self.defer = os.environ.get("MINK_DP_DEFER_LAUNCH", "1") != "0"
        # RCCL averages inside the collective; gloo (CPU tests, rehearsals) sums and finish() scales
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._active = self._collect or dev.type == "cuda"

    # ---- gradient sink protocol (minkowski.functional.set_grad_sink)
    def view_for(self, p):
        """The slice of the flat buffer to write the gradient of `p` into -- once per step (a second
        gradient of the same parameter in one step must ADD, which autograd's accumulate does)."""
        k = id(p)
        g = self._views.get(k)
        if g is None or not self._active or k in self._written or p.grad is not g:
            return None  # not ours / second gradient this step / somebody replaced .grad: fall back to autograd
        self._written.add(k)
        return g

    def views_for(self, params):
        """`view_for` over a list, all or nothing, in one call (the native trunk claims ~40 / ~110 slices per step: the per-call
        overhead of doing so one at a time was 0.1 ms of a host-bound ResNet34 step)."""
        if not self._active:
            return None
        views, written, out = self._views, self._written, []
        for p in params:
            k = id(p)
            g = views.get(k)
            if g is None or k in written or p.grad is not g:
                return None
            out.append(g)
        written.update(id(p) for p in params)
        return out

    def ready_many(self, params):
        """`ready` for every parameter of a block (they share `stage_event`)."""
        if not self._collect and self._bucket_cb is None:
            return
        counted, bucket_of, ready, buckets = self._counted, self._bucket_of, self._ready, self.buckets
        done = False
        for p in params:
            k = id(p)
            if k in counted:
                continue
            counted.add(k)
            b = bucket_of[k]
            ready[b] += 1
            if ready[b] == buckets[b][2]:
                self._bucket_event[b] = self.stage_event
                done = True
        if done and (not self.defer or self.stage_stream is not None):
            self._drain()

    def gradients(self):
        """The parameter gradients in buffer order (reverse registration order) without the alignment padding."""
        return torch.cat([p.grad.flatten() for p in self._params])

    def release(self, p):
        """Undo a `view_for(p)` whose slice will not be written after all."""
        self._written.discard(id(p))

    def ready(self, p):
        self._on_grad(p)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        self.launch_log.append((b, s, e))
        if len(self.launch_log) > 4096:
            del self.launch_log[:2048]
        if self.flat.is_cuda:
            # The gradients of one bucket may have been produced on several HIP streams (the step's
            # own stream, the shortcut-branch stream, the weight-gradient stream); the collective
            # orders itself after the CURRENT stream only.  Launch it from the weight-gradient
            # stream (after making that one wait for the others): the compute stream is not held up.
            from .minkowski import functional as Fn

            dev = self.flat.device
            if self.stage_stream is not None:
                # in-call launch (mink_set_block_done_hook): the library has ordered `stage_stream` (its weight-gradient stream)
                # behind everything this bucket's blocks queued, and nothing later is queued on it yet: the collective waits for
                # exactly its own gradients, with no event and no stream of its own
                with torch.cuda.stream(self.stage_stream):
                    self._work.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))
                return
            ev = self._bucket_event[b]
            if ev is not None:
                from ._lib import check, lib

                if self._launch_stream is None:
                    self._launch_stream = self._make_launch_stream(dev)
                check(lib().mink_stream_wait_event(self._launch_stream.cuda_stream, ev))
                with torch.cuda.stream(self._launch_stream):
                    self._work.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))
                return
            cur = torch.cuda.current_stream(dev)
            side = Fn.side_stream_if_any(dev)
            launch_from = side if side is not None else cur
            seen = {launch_from}
            for st in [self._home, cur] + Fn.compute_streams(dev):
                if st is not None and st not in seen:
                    seen.add(st)
                    Fn.stream_wait(launch_from, st)
            with torch.cuda.stream(launch_from):
                self._work.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))
            return
        self._work.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))

    @staticmethod
    def _make_launch_stream(dev):
        """The stream the collectives are issued from (it carries event waits and records only).  MINK_DP_LAUNCH_STREAM=side:
        the weight-gradient stream instead of a stream of its own -- one busy hardware queue fewer (measurement hook for the
        hardware-queue cliff)."""
        if os.environ.get("MINK_DP_LAUNCH_STREAM") == "side":
            from .minkowski import functional as Fn

            return Fn._side_stream(dev)
        return torch.cuda.Stream(device=dev)

    def _on_grad(self, p):
        if not self._collect and self._bucket_cb is None:
            return
        k = id(p)
        if k in self._counted:  # a gradient written in place is reported by the kernel launcher AND (on
            return              # some torch versions) by the post-accumulate hook of the undefined grad
        self._counted.add(k)
        b = self._bucket_of[k]
        self._ready[b] += 1
        if self._ready[b] == self.buckets[b][2]:
            self._bucket_event[b] = self.stage_event
            if not self.defer or self.stage_stream is not None:  # (in-call launch: as ready_many)
                self._drain()

    def _drain(self):
        """Issue the collectives of the complete buckets IN ASCENDING BUCKET ORDER, stopping at the first incomplete one.
        Every rank must issue its collectives in the same order (RCCL matches them by issue order per communicator),
        and the order in which buckets COMPLETE is not the same on every rank: a rank whose batch is large enough for
        the native trunk reports a stage's parameters in registration order after the stage's backward call, a rank on
        the module path reports them in autograd order (norm2, conv2, shortcut, norm1, conv1) -- with a bucket boundary
        inside a stage the two complete neighbouring buckets in opposite orders.  The flat buffer is laid out in reverse
        registration order, so ascending bucket order is (nearly) completion order anyway: a bucket waits for its
        predecessor for a few kernels at most."""
        first = self._next
        while self._next < len(self.buckets) and self._ready[self._next] == self.buckets[self._next][2]:
            self._next += 1
        todo = [b for b in range(first, self._next) if not self._launched[b]]
        if not self._collect:  # one rank: hand the complete buckets to the consumer, if this is the in-call point of the native trunk
            if self._bucket_cb is not None and self.stage_stream is not None:
                for b in todo:
                    s, e, _ = self.buckets[b]
                    self._launched[b] = True
                    self._bucket_cb(b, s, e, self.stage_stream)
            else:
                self._next = first  # (not from here: the buckets stay pending for the consumer's own step)
            return
        if len(todo) > 1 and self.flat.is_cuda and all(self._bucket_event[b] is not None for b in todo):
            # the one-call backward reports every block after the whole pass is queued: several buckets go out at once, from
            # the launch stream, each behind its own event -- one stream switch for all of them
            from ._lib import check, lib

            if self._launch_stream is None:
                self._launch_stream = self._make_launch_stream(self.flat.device)
            wait, ls = lib().mink_stream_wait_event, self._launch_stream.cuda_stream
            with torch.cuda.stream(self._launch_stream):
                for b in todo:
                    s, e, _ = self.buckets[b]
                    self._launched[b] = True
                    self.launch_log.append((b, s, e))
                    check(wait(ls, self._bucket_event[b]))
                    self._work.append(dist.all_reduce(self.flat[s:e], op=self._op, group=self.group, async_op=True))
            if len(self.launch_log) > 4096:
                del self.launch_log[:2048]
            return
        for b in todo:
            self._launch(b)

    def flush(self):
        """Launch the collectives of the buckets completed so far.  With `defer` (gradient-sink mode) a complete
        bucket is not launched from inside the backward of the small deep layers -- there the host is what the
        GPU waits for, and a launch costs it 60 us -- but at the first point where the GPU has a long kernel
        queued (the convolution backward calls this after queuing the kernels of a large layer)."""
        if self._collect:
            self._drain()

    def set_bucket_callback(self, cb):
        """One rank only: `cb(bucket, start, end, stream)` is called for every bucket whose gradients the native trunk has completed,
        from inside its backward call, with the stream (the weight-gradient stream) that is ordered behind everything the bucket's
        blocks queued -- their data-gradient kernels, which read the weights, included (csrc/trunk.hip: mink_net_backward)."""
        if self._collect and cb is not None:
            raise ValueError("bucket callbacks are the one-rank path: with several ranks the buckets are all-reduced first")
        self._bucket_cb = cb

    def zero_grad(self):
        """Gradients accumulate into the flat buffer; clear it with one memset per step."""
        if self.flat.is_cuda:
            from .minkowski import functional as Fn

            self._home = Fn.current_stream(self.flat.device)

            if self._collect:  # last step's collectives were launched from the side streams (one rank: every gradient
                # write of the last backward pass was joined by its end-of-backward callback -- no barrier packets here)
                for st in Fn.compute_streams(self.flat.device):
                    if st != self._home:
                        Fn.stream_wait(self._home, st)
        if self.cleared:  # (FlatSGD.step cleared it behind its update: one pass less over the gradients)
            self.cleared = False
        else:
            self.flat.zero_()
        self._written.clear()
        self._counted.clear()
        self._next = 0
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._bucket_event = [None] * len(self.buckets)

    def finish(self):
        """Wait for the outstanding collectives (launching any bucket whose parameters did not
        all receive a gradient this step), then turn the sum into the mean."""
        if not self._collect:
            return
        self.flush()
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        if self._avg and self._work:
            # RCCL runs the collectives of one communicator on ONE stream in issue order: the current stream waiting for the last
            # of them has waited for all (each wait() is ~25 us of host time; gloo completes on the CPU and needs every one)
            # -- then the remaining Work objects are waited for too: by then each wait() returns at once (their collectives are
            # behind the last one on that stream), and a torch build that keeps the tensors of an async collective alive in its
            # Work until wait() releases them here instead of at garbage collection.  MINK_DP_WAIT_ALL=1: wait in issue order.
            if os.environ.get("MINK_DP_WAIT_ALL") == "1":
                for w in self._work:
                    w.wait()
            else:
                self._work[-1].wait()
                for w in self._work[:-1]:
                    w.wait()
        else:
            for w in self._work:
                w.wait()
        self._work.clear()
        if not self._avg:
            self.flat.mul_(1.0 / self.world)



class FlatSGD(torch.optim.Optimizer):
    """torch.optim.SGD (momentum, weight decay; dampening 0, no Nesterov -- the reference's recipe: co3d_3d/src/modules/optim.py:12-14,
    configs/co3d_cls.gin) as ONE kernel over flat buffers (`mink_sgd_step`).

    The reducer already keeps every gradient in one flat fp32 buffer; this optimizer re-homes the PARAMETERS into a second flat
    buffer of the same layout (`p.data` becomes a view: values are copied, the Parameter objects stay) and keeps the momentum in a
    third.  torch's fused SGD walks ~40 (ResNet14) / ~110 (ResNet34) tensors in multi-tensor chunks: 96 us / 314 us per step at
    1.3-3 TB/s; one pass over three flat buffers runs at the HBM rate, and clears the gradient buffer on its way (the memset of
    the next `reducer.zero_grad()`).  `param_groups`, `state[p]["momentum_buffer"]`, `state_dict()` / `load_state_dict()` and the
    torch LR schedulers work as with torch.optim.SGD."""

    def __init__(self, reducer, lr, momentum=0.0, weight_decay=0.0, clear_grads=True, in_backward=False):
        """`in_backward` (one rank, native trunk; round 6): the update of a bucket of parameters is launched from INSIDE the backward
        call, on the weight-gradient stream, as soon as that bucket's gradients are complete and the data-gradient kernels that read
        its weights are behind it -- the update of the deep layers (most of the bytes: 14.4 M parameters are 55 us of HBM traffic
        for Mink-ResNet14, 63.5 M are 280 us for Mink-ResNet34) then runs beside the rest of backward instead of behind the stem's
        weight gradient at the end of the step; `step()` updates what is left (the stem's bucket, and everything when the step did
        not go through the native trunk).  Same kernel on slices of the same buffers: the same arithmetic element by element.  Needs
        one backward pass per step (no gradient accumulation over several) and an lr that is final before backward: both hold for
        the reference's recipe (classification_training.py, optim.py)."""
        if not reducer.flat.is_cuda:
            raise ValueError("FlatSGD runs on the GPU (mink_sgd_step); use torch.optim.SGD on the CPU")
        # parameters in REGISTRATION order, like torch.optim.SGD(model.parameters()): optimizer checkpoints stay interchangeable
        # (the flat layout itself is the reducer's: reverse registration order)
        params = list(reducer._params)[::-1]
        self._offsets = list(reducer.offsets)[::-1]
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0.0, nesterov=False))
        self.reducer, self.clear_grads = reducer, bool(clear_grads)
        self.flat_w = torch.zeros_like(reducer.flat)
        self.flat_m = torch.zeros_like(reducer.flat)
        self._ptrs = []
        with torch.no_grad():
            for p, off in zip(params, self._offsets):
                w = self.flat_w[off : off + p.numel()].view_as(p)
                w.copy_(p.data)
                p.data = w  # same Parameter object, new home (plans that captured data_ptr() notice and refresh)
                self.state[p]["momentum_buffer"] = self.flat_m[off : off + p.numel()].view_as(p)
                self._ptrs.append(w.data_ptr())
        self._steps = 0
        self._stepped = []  # (start, end) of the slices already updated inside this step's backward pass
        self.in_backward = bool(in_backward) and not reducer._collect
        if self.in_backward:
            reducer.set_bucket_callback(self._bucket_ready)

    def _launch_slice(self, s, e, raw_stream):
        from ._lib import check, lib

        g = self.param_groups[0]
        flat = self.reducer.flat
        check(lib().mink_sgd_step(self.flat_w.data_ptr() + 4 * s, flat.data_ptr() + 4 * s, self.flat_m.data_ptr() + 4 * s, e - s, float(g["lr"]),
                                  float(g["momentum"]), float(g["weight_decay"]), int(self.clear_grads), raw_stream))

    def _bucket_ready(self, b, s, e, stream):
        self._launch_slice(s, e, stream.cuda_stream)
        self._stepped.append((s, e))

    @classmethod
    def like(cls, sgd, reducer, in_backward=False):
        """A FlatSGD with the hyper-parameters of an (unused) torch.optim.SGD."""
        if len(sgd.param_groups) != 1:
            raise ValueError(f"FlatSGD: one parameter group (one lr / momentum / weight decay for the flat buffer), got {len(sgd.param_groups)}")
        (g,) = sgd.param_groups
        if g.get("dampening", 0) or g.get("nesterov", False) or g.get("maximize", False):
            raise ValueError("FlatSGD: dampening / Nesterov / maximize are not implemented")
        return cls(reducer, lr=g["lr"], momentum=g.get("momentum", 0.0), weight_decay=g.get("weight_decay", 0.0), in_backward=in_backward)

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):  # (torch's constructor adds the first one through this method)
            raise ValueError("FlatSGD: one parameter group only -- the step is ONE kernel with one lr / momentum / weight decay over the flat buffer")
        super().add_param_group(param_group)

    def _check_homes(self, full=True):
        """Every step, every parameter: its data and its .grad still live where the kernel reads and writes (two pointer compares
        per parameter, ~20 us for ResNet34's 110 -- until round 4 the middle parameters were looked at every 64th step only)."""
        params = self.param_groups[0]["params"]
        views = self.reducer._views
        for i, p in enumerate(params):
            if p.data_ptr() != self._ptrs[i] or p.grad is not views[id(p)]:
                raise RuntimeError("FlatSGD: a parameter (or its .grad) no longer lives in the flat buffers (load_state_dict(assign=True), "
                                   ".to(), a replaced .data or .grad?); rebuild reducer and optimizer")

    @torch.no_grad()
    def step(self, closure=None):
        from ._lib import check, lib

        loss = closure() if closure is not None else None
        if len(self.param_groups) != 1:
            raise ValueError("FlatSGD: one parameter group only")
        self._check_homes()
        self._steps += 1
        g = self.param_groups[0]
        if g.get("dampening", 0) or g.get("nesterov", False) or g.get("maximize", False):
            raise ValueError("FlatSGD: dampening / Nesterov / maximize are not implemented")
        flat = self.reducer.flat
        raw = torch._C._cuda_getCurrentRawStream(flat.device.index if flat.device.index is not None else torch.cuda.current_device())
        if self._stepped:  # the slices the backward pass did not update itself (they are few: ascending, merged where adjacent)
            done, pos, todo = sorted(self._stepped), 0, []
            self._stepped = []
            for s_, e_ in done:
                if s_ > pos:
                    todo.append((pos, s_))
                pos = max(pos, e_)
            if pos < flat.numel():
                todo.append((pos, flat.numel()))
            for s_, e_ in todo:
                self._launch_slice(s_, e_, raw)
        else:
            check(lib().mink_sgd_step(self.flat_w.data_ptr(), flat.data_ptr(), self.flat_m.data_ptr(), flat.numel(), float(g["lr"]),
                                      float(g["momentum"]), float(g["weight_decay"]), int(self.clear_grads), raw))
        if self.clear_grads:
            self.reducer.cleared = True
        return loss

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)  # (deep-copies the momentum buffers: bring them home again)
        with torch.no_grad():
            for p, off in zip(self.param_groups[0]["params"], self._offsets):
                home = self.flat_m[off : off + p.numel()].view_as(p)
                buf = self.state[p].get("momentum_buffer")
                if buf is not None and buf.data_ptr() != home.data_ptr():
                    home.copy_(buf)
                self.state[p]["momentum_buffer"] = home
