"""GPU box: host-side cost of the data-parallel machinery on ONE GPU (one-rank RCCL group, reducer forced on):
phase times with and without the reducer, and the time spent inside the reducer's own methods."""
import collections, os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist, torch.nn.functional as F
from bench import make_batches
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.parallel import BucketedGradAllReduce

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
T = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name, label=None):
    fn = getattr(cls, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = T[label or f"{cls.__name__}.{name}"]
            e[0] += time.perf_counter() - t0; e[1] += 1
    setattr(cls, name, timed)
for n in ("_on_grad", "_launch", "zero_grad", "finish", "view_for", "ready"):
    wrap(BucketedGradAllReduce, n)
wrap(dist, "all_reduce", "dist.all_reduce")
wrap(torch.cuda.Stream, "wait_stream", "Stream.wait_stream")

torch.manual_seed(0)
MODEL, BATCH = os.environ.get("MODEL", "ResNet14"), int(os.environ.get("BATCH", "16"))
model = get_model(MODEL, 28, 51).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
use = os.environ.get("DP", "1") == "1"
reducer = BucketedGradAllReduce(model, force=True) if use else None
batches = make_batches(2, BATCH, 0, 51, 128, 28)
batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
state = {"tf": model.process_input(batches[0])}
W = collections.defaultdict(float)
def step(i, rec):
    t0 = time.perf_counter()
    tf = state["tf"]
    nxt = model.process_input(batches[(i + 1) % 2], defer=True)
    reducer.zero_grad() if reducer else opt.zero_grad(set_to_none=True)
    ta = time.perf_counter(); out = model(tf); t1 = time.perf_counter()
    loss = F.cross_entropy(out, batches[i % 2]["labels"].long()); loss.backward(); t2 = time.perf_counter()
    state["tf"] = model.finish_input(nxt); t3 = time.perf_counter()
    if reducer: reducer.finish()
    t4 = time.perf_counter(); opt.step(); t5 = time.perf_counter()
    if rec:
        for k, v in (("launch_next+zero", ta - t0), ("forward", t1 - ta), ("backward", t2 - t1), ("finish_next", t3 - t2), ("reducer.finish", t4 - t3), ("opt", t5 - t4)):
            W[k] += v
for i in range(8): step(i, False)
torch.cuda.synchronize(); T.clear()
N = 30
t0 = time.perf_counter()
for i in range(8, 8 + N): step(i, True)
th = time.perf_counter() - t0
torch.cuda.synchronize(); tt = time.perf_counter() - t0
print("DP" if use else "plain", "host ms/step:", {k: round(v / N * 1e3, 3) for k, v in W.items()}, "host total", round(th / N * 1e3, 3), "wall", round(tt / N * 1e3, 3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:40s} {v[0] / N * 1e6:8.1f} us/step  {v[1] / N:5.1f} calls/step  {v[0] / max(v[1], 1) * 1e6:6.1f} us/call")
if reducer: print("buckets:", [(e - s) * 4 >> 20 for s, e, _ in reducer.buckets], "MiB; params per bucket", [c for _, _, c in reducer.buckets])
dist.destroy_process_group()
