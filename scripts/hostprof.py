"""Host-side timeline of one pipelined training step (run on the GPU box): how long the HOST needs to queue each
phase, against the wall time of the step.  usage: python scripts/hostprof.py [batch] [native_trunk 0/1] [model=ResNet14]"""
import collections, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from bench import make_batches
from nerf_downstream_amd.co3d_3d.src.models import get_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(0)
NAME = sys.argv[3] if len(sys.argv) > 3 else "ResNet14"
model = get_model(NAME, 28, 51).to(dev)
if len(sys.argv) > 2:
    model._native_trunk = sys.argv[2] != "0"
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
batches = make_batches(2, B, 0, 51, 128, 28)
batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
T = collections.defaultdict(float)
state = {"tf": model.process_input(batches[0])}

def step(i, rec=False):
    t0 = time.perf_counter()
    tf = state["tf"]
    nxt = model.process_input(batches[(i + 1) % 2], defer=True); ta = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = model(tf); t1 = time.perf_counter()
    loss = F.cross_entropy(out, batches[i % 2]["labels"].long()); loss.backward(); t2 = time.perf_counter()
    state["tf"] = model.finish_input(nxt); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    if rec:
        T["launch_next"] += ta - t0; T["forward"] += t1 - ta; T["backward"] += t2 - t1; T["finish_next"] += t3 - t2; T["opt"] += t4 - t3

for i in range(8): step(i)
import gc; gc.collect(); gc.freeze()
torch.cuda.synchronize()
N = 40
# (a) host only: the GPU is left to fall behind, then drained
t0 = time.perf_counter()
for i in range(8, 8 + N): step(i, True)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print(f"{NAME} B={B} native_trunk={model._native_trunk} host ms/step", {k: round(v / N * 1e3, 3) for k, v in T.items()}, "host total", round(th / N * 1e3, 3), "wall", round(tt / N * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for i in range(48, 58): step(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])

# (c) the library calls themselves: every mink_* entry point wrapped with a host timer (both the main thread and autograd's)
from nerf_downstream_amd._lib import lib
L = lib()
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(name, fn):
    def timed(*a):
        t = time.perf_counter()
        r = fn(*a)
        e = acc[name]; e[0] += 1; e[1] += time.perf_counter() - t
        return r
    return timed
from nerf_downstream_amd import _lib as _lm
for name in _lm.SIGNATURES:
    setattr(L, name, wrap(name, getattr(L, name)))
for i in range(60, 68): step(i)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for i in range(68, 68 + N): step(i)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host {th / N * 1e3:.3f} ms/step with the timers; library calls per step:")
tot = 0.0
for name, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:40s} {c / N:6.1f} calls  {t / N * 1e6:8.1f} us"); tot += t
print(f"  all library calls: {sum(t for _, t in acc.values()) / N * 1e6:.0f} us/step")
