"""Device-memory plumbing: one large segment for torch's caching allocator to carve from.

A training step allocates its activations, gradient scratch and the next batch's maps from torch's caching allocator.  The
batches differ in size (the reference's CO3D scenes have 30-90 k voxels each), so for tens of steps the allocator keeps
meeting a request no cached block fits and goes to `hipMalloc` -- 0.1 ms for a small block, 3-13 ms for one of a few hundred
megabytes, paid in the middle of a step (measured: ResNet34 at four scenes, one 13 ms call every ten steps while the pool was
still growing; a bench.py run of twenty steps that catches one reads 4.2-4.6 ms per step instead of 3.7).  An MI355X has 288 GB:
take one segment up front and hand it to the pool; every later request of >= 1 MB is a split of it.
"""
import logging
import os

import torch

_RESERVED = {}


def reserve(device, gigabytes=None):
    """Make torch's caching allocator own ONE free segment of `gigabytes` on `device` (default: $MINK_RESERVE_GB, else 24 GB,
    never more than a quarter of the memory that is free; 0 = do nothing).  Idempotent per device; returns the bytes reserved."""
    device = torch.device(device)
    if device.type != "cuda":
        return 0
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _RESERVED:
        return _RESERVED[idx]
    if gigabytes is None:
        gigabytes = float(os.environ.get("MINK_RESERVE_GB", "24"))
    free, _ = torch.cuda.mem_get_info(idx)
    nbytes = int(min(gigabytes * 2 ** 30, free // 4)) // (2 << 20) * (2 << 20)
    if nbytes > 0:
        block = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        del block  # (back to the allocator's pool of free LARGE blocks: split on demand, never returned to the driver)
    _RESERVED[idx] = nbytes
    logging.getLogger(__name__).info("reserved %.1f GB for the caching allocator on cuda:%d", nbytes / 2 ** 30, idx)
    return nbytes


def reserve_on(stream, gigabytes=None):
    """The same for allocations made under another stream: the caching allocator keeps its free blocks per stream, so the maps
    built on the prepare stream (one arena of a few hundred MB per batch, several alive at once when the host runs ahead) never
    come out of the segment `reserve` took on the default stream -- they grew the pool by 14 GB over the first hundreds of
    steps, 8-9 `hipMalloc`s inside a bench window of twenty.  Default: $MINK_RESERVE_SIDE_GB, else 16 GB (never more than an
    eighth of the free memory; 0 = do nothing).  ONE such segment per device and process: the first prepare stream gets it (a trainer
    has one model; a test process that builds many models must not reserve 16 GB for each)."""
    if stream is None or stream.device.type != "cuda":
        return 0
    key = ("stream", stream.device.index)
    if key in _RESERVED:
        return _RESERVED[key]
    if gigabytes is None:
        gigabytes = float(os.environ.get("MINK_RESERVE_SIDE_GB", "16"))
    free, _ = torch.cuda.mem_get_info(stream.device.index)
    nbytes = int(min(gigabytes * 2 ** 30, free // 8)) // (2 << 20) * (2 << 20)
    if nbytes > 0:
        with torch.cuda.stream(stream):
            block = torch.empty(nbytes, dtype=torch.uint8, device=stream.device)
            del block
    _RESERVED[key] = nbytes
    logging.getLogger(__name__).info("reserved %.1f GB for allocations under the prepare stream of cuda:%d", nbytes / 2 ** 30,
                                     stream.device.index)
    return nbytes


def reserved(device):
    """True once `reserve` has taken (or been asked for) a segment on `device` in this process."""
    device = torch.device(device)
    idx = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else None)
    return idx in _RESERVED


def reserved_bytes(device):
    """Bytes of the segments taken up front on `device` (compute stream + prepare stream)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return int(_RESERVED.get(idx, 0)) + int(_RESERVED.get(("stream", idx), 0))
