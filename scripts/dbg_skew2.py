"""Race hunting at op level: weight gradient on the side stream behind a long sleep kernel."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from helpers import batch_scenes
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd.minkowski import functional as Fn

coords, feats = batch_scenes([31, 32, 33], grid=32, cin=64)
x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
m, k1 = x.coordinate_manager, x.coordinate_map_key
k2 = m.stride(k1, 2)
nbr, _ = m.kernel_table(k1, k2, 3, 1)
torch.manual_seed(0)
xin = torch.randn(x.F.shape[0], 64, device="cuda")
gy = torch.randn(nbr.shape[0], 64, device="cuda")
ref = Fn.conv_wgrad(xin, gy, nbr, (27, 64, 64))
torch.cuda.synchronize()
print("ref norm", float(ref.norm()), "rows", nbr.shape[0])
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for trial in range(4):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.cuda._sleep(3000000)
        out = Fn.conv_wgrad(xin, gy, nbr, (27, 64, 64))
    main.wait_stream(side)
    a = float(out.norm())
    torch.cuda.synchronize()
    print("trial", trial, "norm right after join", a, "after sync", float(out.norm()), "equal", bool(torch.equal(out, ref)))
