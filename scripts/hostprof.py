"""Host-side timeline of one pipelined training step (run on the GPU box)."""
import collections, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from bench import make_batches
from nerf_downstream_amd.co3d_3d.src.models import get_model

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = get_model("ResNet14", 28, 51).to(dev)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
batches = make_batches(2, 16, 0, 51, 128, 28)
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

for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for i in range(6, 6 + N): step(i, True)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tt = time.perf_counter() - t0
print("host ms/step", {k: round(v / N * 1e3, 3) for k, v in T.items()}, "host total", round(th / N * 1e3, 3), "wall", round(tt / N * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for i in range(26, 36): step(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])
