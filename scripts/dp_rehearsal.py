"""2-rank gloo rehearsal on ONE GPU of the bucketed all-reduce with a plain torch model
(no HIP kernels): isolates collective-ordering problems from the sparse backend."""
import faulthandler, os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(45, exit=True)
dist.init_process_group("gloo")
rank = dist.get_rank()
from nerf_downstream_amd.parallel import BucketedGradAllReduce
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(512, 2048), torch.nn.ReLU(), torch.nn.Linear(2048, 2048), torch.nn.ReLU(), torch.nn.Linear(2048, 10)).to(dev)
red = BucketedGradAllReduce(m, bucket_bytes=int(os.environ.get("BUCKET", 8 << 20)))
print(rank, "buckets", len(red.buckets), flush=True)
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
for step in range(4):
    red.zero_grad()
    m(torch.randn(64, 512, device=dev) + rank).sum().backward()
    print(rank, step, "launched", red._launched, flush=True)
    red.finish()
    opt.step()
    torch.cuda.synchronize()
    print(rank, step, "done", float(red.flat.abs().sum()), flush=True)
dist.barrier()
dist.destroy_process_group()
