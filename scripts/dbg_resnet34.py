"""Debug helper: find the gather_gemm call whose un-split (ksplit=1) result deviates from the planner's split (ResNet34, grid 40)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.nn.functional as F
from helpers import batch_scenes
from nerf_downstream_amd.co3d_3d.src.models import get_model
from nerf_downstream_amd.minkowski import functional as Fn
from nerf_downstream_amd._lib import lib

torch.manual_seed(0)
hip = get_model("ResNet34", 27, 51).cuda()
coords, feats = batch_scenes([11, 12, 13], grid=40, cin=27)
labels = torch.tensor([3, 17, 50])
orig = Fn.gather_gemm

def checked(x, w, nbr, cout, **kw):
    y = orig(x, w, nbr, cout, **kw)
    res = {}
    for zs in (1, 2, 3, 27):
        Fn._FORCE_KSPLIT = min(zs, nbr.shape[1])
        res[zs] = orig(x, w, nbr, cout, **kw)
    Fn._FORCE_KSPLIT = 0
    ref = res[27]
    errs = {zs: float((v - ref).abs().max()) / max(1e-6, float(ref.abs().max())) for zs, v in res.items()}
    errs["planner"] = float((y - ref).abs().max()) / max(1e-6, float(ref.abs().max()))
    flag = "  <<<<<< MISMATCH" if max(errs.values()) > 1e-4 else ""
    print(f"x={tuple(x.shape)} w={tuple(w.shape)} nbr={tuple(nbr.shape)} cout={cout} "
          f"{ {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()} } " + " ".join(f"{k}:{e:.1e}" for k, e in errs.items()) + flag)
    if flag:
        d = (res[1] - ref).abs()
        bad_cols = (d.max(dim=0).values > 1e-4 * float(ref.abs().max())).nonzero().flatten().tolist()
        bad_rows = (d.max(dim=1).values > 1e-4 * float(ref.abs().max())).nonzero().flatten()
        print("   bad cols", bad_cols[:40], "n bad rows", bad_rows.numel(), "first", bad_rows[:10].tolist(), "last", bad_rows[-5:].tolist())
    return y

Fn.gather_gemm = checked
out = hip(hip.process_input({"coordinates": coords.cuda(), "features": feats.cuda()}))
print("---- backward")
F.cross_entropy(out, labels.cuda()).backward()
