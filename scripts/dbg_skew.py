"""Race hunting: ResNet14 gradients with every auxiliary stream delayed (MINK_STREAM_SKEW) vs single-stream."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.nn.functional as F
from helpers import batch_scenes
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.minkowski import functional as Fn

torch.manual_seed(0)
hip = get_model("ResNet14", 28, 51).cuda()
coords, feats = batch_scenes([31, 32, 33], grid=32, cin=28)
batch = {"coordinates": coords.cuda(), "features": feats.cuda()}
labels = torch.tensor([1, 2, 3]).cuda()

def grads(overlap, prepared):
    old = Fn.set_wgrad_overlap(overlap)
    hip.prepare_ahead = prepared
    hip.zero_grad(set_to_none=True)
    F.cross_entropy(hip(hip.process_input(batch)), labels).backward()
    g = {k: p.grad.clone() for k, p in hip.named_parameters()}
    torch.cuda.synchronize()
    g2 = {k: p.grad.clone() for k, p in hip.named_parameters()}
    Fn.set_wgrad_overlap(old)
    return g, g2

ref, _ = grads(False, False)
for name, (ov, prep) in {"overlap lazy": (True, False), "overlap prepared#1": (True, True), "overlap prepared#2": (True, True), "no-overlap prepared": (False, True)}.items():
    g, g2 = grads(ov, prep)
    bad = [k for k in ref if not torch.equal(ref[k], g[k])]
    bad2 = [k for k in ref if not torch.equal(ref[k], g2[k])]
    print(name, "differs right after backward:", len(bad), bad[:6], "| after device sync:", len(bad2), bad2[:6])
    for k in bad[:3]:
        print("   ", k, "ref norm", float(ref[k].norm()), "got norm", float(g[k].norm()), "after sync", float(g2[k].norm()))
