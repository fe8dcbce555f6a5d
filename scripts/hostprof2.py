"""cProfile of process_input (prepare-ahead) and of forward/backward separately (GPU box)."""
import cProfile, io, os, pstats, sys, time
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
tf = model.process_input(batches[0])
for i in range(4):
    opt.zero_grad(set_to_none=True)
    out = model(tf); F.cross_entropy(out, batches[i % 2]["labels"].long()).backward()
    tf = model.process_input(batches[(i + 1) % 2]); opt.step()
torch.cuda.synchronize()
which = sys.argv[1] if len(sys.argv) > 1 else "prepare"
pr = cProfile.Profile()
for i in range(4, 14):
    opt.zero_grad(set_to_none=True)
    if which == "fwd": pr.enable()
    out = model(tf)
    if which == "fwd": pr.disable()
    loss = F.cross_entropy(out, batches[i % 2]["labels"].long())
    if which == "bwd": pr.enable()
    loss.backward()
    if which == "bwd": pr.disable()
    if which == "prepare": pr.enable()
    tf = model.process_input(batches[(i + 1) % 2])
    if which == "prepare": pr.disable()
    opt.step()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6500])
