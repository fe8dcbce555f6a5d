"""parallel.BucketedGradAllReduce's one-rank bucket callback (what FlatSGD(in_backward=True) hangs on), host logic only: buckets are handed
over in ascending order, only from the in-call point of the native trunk (a stage stream is set), once each, and a new step starts clean."""
import torch
import torch.nn as nn

from nerf_downstream_amd.parallel import BucketedGradAllReduce


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(64, 256), nn.Linear(256, 256), nn.Linear(256, 256), nn.Linear(256, 8))


def test_buckets_reach_the_callback_in_order_and_only_in_call():
    m = _model()
    red = BucketedGradAllReduce(m, bucket_bytes=128 << 10)  # ~64 K floats per bucket: several buckets
    assert len(red.buckets) >= 3 and not red._collect
    seen = []
    red.set_bucket_callback(lambda b, s, e, st: seen.append((b, s, e, st)))
    params = list(red._params)  # flat-buffer order = reverse registration order = the order backward completes them
    bucket_of = [red._bucket_of[id(p)] for p in params]
    assert bucket_of == sorted(bucket_of)
    red.zero_grad()
    # outside the in-call point (no stage stream): counted, not handed over
    red.ready_many(params[:2])
    assert seen == []
    # the in-call point: every bucket that is complete goes out, lowest first
    red.stage_stream = "wgrad-stream"
    try:
        red.ready_many(params[2:])
    finally:
        red.stage_stream = None
    assert [b for b, *_ in seen] == list(range(len(red.buckets)))
    assert all(st == "wgrad-stream" for *_, st in seen)
    assert [(s, e) for _, s, e, _ in seen] == [(s, e) for s, e, _ in red.buckets]
    # once each: reporting the same parameters again changes nothing
    red.stage_stream = "wgrad-stream"
    red.ready_many(params)
    red.stage_stream = None
    assert len(seen) == len(red.buckets)
    # a new step starts clean; a bucket that is complete only partly stays back, and so does everything behind it
    red.zero_grad()
    seen.clear()
    last_of_bucket0 = max(i for i, b in enumerate(bucket_of) if b == 0)
    red.stage_stream = "wgrad-stream"
    red.ready_many(params[:last_of_bucket0] + params[last_of_bucket0 + 1:])  # everything but one parameter of bucket 0
    assert seen == []
    red.ready_many([params[last_of_bucket0]])
    red.stage_stream = None
    assert [b for b, *_ in seen] == list(range(len(red.buckets)))
    red.set_bucket_callback(None)
    red.zero_grad()
    red.stage_stream = "x"
    red.ready_many(params)  # no consumer: nothing to do, nothing raised
    red.stage_stream = None
