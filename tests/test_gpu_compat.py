"""-m gpu: the drop-in seam itself.  `install_as_minkowski_engine()` + a model file written in the reference's
composition style (`import MinkowskiEngine as ME` at the top, look-up tables built from `ME.Minkowski*` at import) must
import and run against the HIP backend and match the CPU oracle's mini-ME run of the SAME file; the ME classes that the
reference's layer factory touches at import time (modules/common.py:25-26,36-43,56-71) are checked against torch."""
import importlib.util
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

from helpers import batch_scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_refstyle(me_module, functional, name):
    """Import tests/refstyle/resnet_refstyle.py with `MinkowskiEngine` bound to `me_module`."""
    saved = {k: sys.modules.get(k) for k in ("MinkowskiEngine", "MinkowskiEngine.MinkowskiFunctional")}
    sys.modules["MinkowskiEngine"] = me_module
    sys.modules["MinkowskiEngine.MinkowskiFunctional"] = functional
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "refstyle", "resnet_refstyle.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


@pytest.mark.parametrize("act", ["MinkowskiReLU", "MinkowskiLeakyReLU", "MinkowskiGELU"])
def test_install_as_minkowski_engine_runs_reference_style_model(oracle_maps, act):
    import nerf_downstream_amd
    from oracle import me_cpu as OME

    for k in [k for k in sys.modules if k == "MinkowskiEngine" or k.startswith("MinkowskiEngine.")]:
        del sys.modules[k]
    hip_me = nerf_downstream_amd.install_as_minkowski_engine()
    import MinkowskiEngine as ME  # what a reference file sees
    import MinkowskiEngine.MinkowskiFunctional as MEF

    assert ME is hip_me and ME.BACKEND == "hip-gfx950" and MEF.relu is hip_me.MinkowskiFunctional.relu
    hip_mod = _load_refstyle(ME, MEF, "refstyle_hip")
    omef = types.ModuleType("MinkowskiEngine.MinkowskiFunctional")
    for fn in ("relu", "leaky_relu", "elu", "celu", "selu", "gelu", "prelu"):
        setattr(omef, fn, getattr(OME.MinkowskiFunctional, fn))
    cpu_mod = _load_refstyle(OME, omef, "refstyle_cpu")
    assert set(hip_mod.ACTIVATIONS) == set(cpu_mod.ACTIVATIONS) and len(hip_mod.ACTIVATIONS) == 7

    torch.manual_seed(1)
    ref = cpu_mod.TinyResNet(28, 11, act=act)
    hip = hip_mod.TinyResNet(28, 11, act=act).cuda()
    assert list(hip.state_dict()) == list(ref.state_dict())
    hip.load_state_dict(ref.state_dict())
    coords, feats = batch_scenes([3, 4, 5, 6], grid=32, cin=28)
    labels = torch.tensor([1, 5, 7, 10])
    out = hip({"coordinates": coords.cuda(), "features": feats.cuda()})
    oout = ref({"coordinates": coords, "features": feats})
    assert torch.allclose(out.cpu(), oout, atol=1e-3), (out.cpu() - oout).abs().max()
    F.cross_entropy(out, labels.cuda()).backward()
    F.cross_entropy(oout, labels).backward()
    for (k, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        rel = float((p.grad.cpu() - q.grad).norm() / q.grad.norm().clamp_min(1e-12))
        assert rel < 2e-2, (k, rel)


def _sparse(x, batch_sizes):
    from nerf_downstream_amd import minkowski as ME

    coords = torch.zeros(x.shape[0], 4)
    coords[:, 0] = torch.repeat_interleave(torch.arange(len(batch_sizes)), torch.tensor(batch_sizes)).float()
    coords[:, 1] = torch.arange(x.shape[0])
    field = ME.TensorField(coordinates=coords.cuda(), features=x.detach())
    return ME.SparseTensor(x, ME.CoordinateMapKey(1), field.coordinate_manager)


@pytest.mark.parametrize("name,torch_fn,args", [
    ("MinkowskiLeakyReLU", F.leaky_relu, (0.2,)), ("MinkowskiELU", F.elu, (0.7,)), ("MinkowskiCELU", F.celu, (1.3,)),
    ("MinkowskiSELU", F.selu, ()), ("MinkowskiGELU", F.gelu, ()),
])
def test_activation_modules_match_torch(name, torch_fn, args):
    from nerf_downstream_amd import minkowski as ME

    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1537, 37, generator=g) * 2.0  # odd sizes: no alignment assumptions
    w = torch.randn(1537, 37, generator=g)
    x = x0.clone().cuda().requires_grad_(True)
    y = getattr(ME, name)(*args)(_sparse(x, [1000, 537])).F
    (y * w.cuda()).sum().backward()
    xr = x0.clone().requires_grad_(True)
    yr = torch_fn(xr, *args)
    (yr * w).sum().backward()
    assert torch.allclose(y.detach().cpu(), yr.detach(), atol=2e-6, rtol=1e-5)
    assert torch.allclose(x.grad.cpu(), xr.grad, atol=2e-6, rtol=1e-5)
    fn = {"MinkowskiLeakyReLU": "leaky_relu", "MinkowskiELU": "elu", "MinkowskiCELU": "celu", "MinkowskiSELU": "selu",
          "MinkowskiGELU": "gelu"}[name]
    y2 = getattr(ME.MinkowskiFunctional, fn)(_sparse(x.detach(), [1000, 537]), *args).F
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("nparam", [1, 24])
def test_prelu_and_instance_norm_match_torch(nparam):
    from nerf_downstream_amd import minkowski as ME

    g = torch.Generator().manual_seed(1)
    C, sizes = 24, [700, 1, 413]  # a one-voxel sample: variance 0
    n = sum(sizes)
    x0, w = torch.randn(n, C, generator=g) * 1.5 + 0.2, torch.randn(n, C, generator=g)
    # PReLU
    mod = ME.MinkowskiPReLU(nparam, init=0.1).cuda()
    ref = torch.nn.PReLU(nparam, init=0.1)
    with torch.no_grad():
        ref.weight.copy_(torch.linspace(0.05, 0.4, nparam))
        mod.weight.copy_(ref.weight)
    x = x0.clone().cuda().requires_grad_(True)
    y = mod(_sparse(x, sizes)).F
    (y * w.cuda()).sum().backward()
    xr = x0.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * w).sum().backward()
    assert torch.allclose(y.detach().cpu(), yr.detach(), atol=1e-6)
    assert torch.allclose(x.grad.cpu(), xr.grad, atol=1e-6)
    assert torch.allclose(mod.weight.grad.cpu(), ref.weight.grad, atol=1e-3, rtol=1e-4)
    # InstanceNorm: each sample normalised by its own statistics
    inorm = ME.MinkowskiInstanceNorm(C).cuda()
    with torch.no_grad():
        inorm.weight.copy_(torch.linspace(0.5, 1.5, C)[None]), inorm.bias.copy_(torch.linspace(-0.2, 0.2, C)[None])
    x = x0.clone().cuda().requires_grad_(True)
    y = inorm(_sparse(x, sizes)).F
    (y * w.cuda()).sum().backward()
    xr = x0.clone().requires_grad_(True)
    parts, s = [], 0
    for k in sizes:
        seg = xr[s : s + k]
        mu, var = seg.mean(0, keepdim=True), seg.var(0, unbiased=False, keepdim=True)
        parts.append((seg - mu) / torch.sqrt(var + inorm.eps) * inorm.weight.detach().cpu() + inorm.bias.detach().cpu())
        s += k
    yr = torch.cat(parts)
    (yr * w).sum().backward()
    assert torch.allclose(y.detach().cpu(), yr.detach(), atol=2e-5, rtol=1e-5)
    assert torch.allclose(x.grad.cpu(), xr.grad, atol=2e-4, rtol=1e-4)


def test_even_kernel_stride1_input_gradient(oracle_maps):
    """kernel_size=2 (offsets {0,1}) with stride 1: the transposed neighbour table is NOT the flipped table (that
    identity needs a centred kernel), so the input gradient must come from the explicit transposed map."""
    from nerf_downstream_amd import minkowski as ME
    from oracle import me_cpu as OME

    coords, feats = batch_scenes([21, 22], grid=24, cin=16)
    torch.manual_seed(2)
    oc = OME.MinkowskiConvolution(16, 32, kernel_size=2, stride=1, dimension=3)
    hc = ME.MinkowskiConvolution(16, 32, kernel_size=2, stride=1, dimension=3).cuda()
    hc.load_state_dict(oc.state_dict())
    fo = feats.clone().requires_grad_(True)
    fh = feats.clone().cuda().requires_grad_(True)
    xo = OME.TensorField(coordinates=coords, features=fo).sparse()
    xh = ME.TensorField(coordinates=coords.cuda(), features=fh).sparse()
    yo, yh = oc(xo).F, hc(xh).F
    w = torch.randn(yo.shape, generator=torch.Generator().manual_seed(3))
    (yo * w).sum().backward()
    (yh * w.cuda()).sum().backward()
    assert torch.allclose(yh.detach().cpu(), yo.detach(), atol=2e-4)
    assert torch.allclose(fh.grad.cpu(), fo.grad, atol=2e-4), (fh.grad.cpu() - fo.grad).abs().max()
    assert torch.allclose(hc.kernel.grad.cpu(), oc.kernel.grad, atol=2e-3, rtol=1e-3)
