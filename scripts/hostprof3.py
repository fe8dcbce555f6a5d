"""GPU box: host time spent INSIDE the Python forward/backward bodies of the autograd Functions (the autograd
engine runs the backward ones on its own thread, where cProfile does not look) against the whole pass."""
import collections, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from bench import make_batches
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.minkowski import functional as Fn

T = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    fn = getattr(cls, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = T[f"{cls.__name__}.{name}"]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(cls, name, staticmethod(timed))
for n in dir(Fn):
    c = getattr(Fn, n)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "forward"), wrap(c, "backward")

def wrapf(obj, name, label):
    fn = getattr(obj, name)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = T["  (inner) " + label]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(obj, name, timed)
if os.environ.get("HOSTPROF_INNER"):
    for nm in ("gather_gemm", "conv_wgrad", "_scratch", "skew", "_side_stream", "_bn_statistics"):
        wrapf(Fn, nm, nm)
    wrapf(torch.Tensor, "record_stream", "Tensor.record_stream")
    wrapf(torch.cuda.Stream, "wait_stream", "Stream.wait_stream")
    wrapf(torch.cuda.StreamContext, "__enter__", "StreamContext.__enter__")
    wrapf(torch.cuda.StreamContext, "__exit__", "StreamContext.__exit__")
    wrapf(torch, "empty", "torch.empty")
    wrapf(torch, "empty_like", "torch.empty_like")

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = get_model("ResNet14", 28, 51).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
batches = make_batches(2, 16, 0, 51, 128, 28)
batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
state = {"tf": model.process_input(batches[0])}
W = collections.defaultdict(float)
def step(i, rec):
    tf = state["tf"]
    nxt = model.process_input(batches[(i + 1) % 2], defer=True)
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); out = model(tf); t1 = time.perf_counter()
    loss = F.cross_entropy(out, batches[i % 2]["labels"].long()); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    state["tf"] = model.finish_input(nxt); opt.step()
    if rec:
        W["forward"] += t1 - t0; W["loss"] += t2 - t1; W["backward"] += t3 - t2
for i in range(6): step(i, False)
torch.cuda.synchronize(); T.clear()
N = 20
for i in range(6, 6 + N): step(i, True)
torch.cuda.synchronize()
print("whole passes, host ms/step:", {k: round(v / N * 1e3, 3) for k, v in W.items()})
fw = sum(v[0] for k, v in T.items() if k.endswith("forward")); bw = sum(v[0] for k, v in T.items() if k.endswith("backward"))
print(f"inside Function bodies: forward {fw / N * 1e3:.3f} ms/step, backward {bw / N * 1e3:.3f} ms/step")
for k, v in sorted(T.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:44s} {v[0] / N * 1e6:8.1f} us/step  {v[1] / N:5.1f} calls/step  {v[0] / max(v[1], 1) * 1e6:6.1f} us/call")
