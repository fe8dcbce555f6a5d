import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from bench import make_batches
from nerf_downstream_amd.co3d_3d.src.models import get_model
dev=torch.device('cuda',0)
torch.manual_seed(0)
model=get_model("ResNet14",28,51).to(dev)
opt=torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
batches=make_batches(2,16,0,51,128,28)
batches=[{k:(v.to(dev) if torch.is_tensor(v) else v) for k,v in b.items()} for b in batches]
def step(i, sync_free=False):
    b=batches[i%2]
    opt.zero_grad(set_to_none=True)
    out=model(model.process_input(b))
    loss=F.cross_entropy(out,b["labels"].long())
    loss.backward()
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
# phase timing
import collections
T=collections.defaultdict(float)
for i in range(10):
    b=batches[i%2]
    torch.cuda.synchronize(); t0=time.perf_counter()
    opt.zero_grad(set_to_none=True)
    tf=model.process_input(b); t1=time.perf_counter()
    out=model(tf); t2=time.perf_counter()
    loss=F.cross_entropy(out,b["labels"].long()); loss.backward(); t3=time.perf_counter()
    opt.step(); t4=time.perf_counter()
    torch.cuda.synchronize(); t5=time.perf_counter()
    T['process_input']+=t1-t0; T['forward(host)']+=t2-t1; T['backward(host)']+=t3-t2; T['opt(host)']+=t4-t3; T['drain']+=t5-t4; T['total']+=t5-t0
print({k:round(v/10*1e3,3) for k,v in T.items()})
pr=cProfile.Profile(); pr.enable()
for i in range(10): step(i)
torch.cuda.synchronize()
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(35); print(s.getvalue()[:6000])
