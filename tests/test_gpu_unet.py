"""-m gpu: the segmentation family on the HIP backend vs the CPU oracle: convolution k=2 s=2,
transposed convolution k=2 s=2 onto the encoder's map, cat, slice -- operator by operator (tight bounds),
then Res16UNet forward + backward (per-point logits within the north_star tolerance 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import batch_scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cmid,cout", [(28, 32, 48), (64, 96, 64), (20, 160, 36)])
def test_down_up_convolutions_match_oracle(oracle_maps, cin, cmid, cout):
    from nerf_downstream_amd import minkowski as ME
    from oracle import me_cpu as OME

    torch.manual_seed(0)
    coords, feats = batch_scenes([21, 22, 23], grid=32, cin=cin)
    rdown = OME.MinkowskiConvolution(cin, cmid, kernel_size=2, stride=2, dimension=3)
    rup = OME.MinkowskiConvolutionTranspose(cmid, cout, kernel_size=2, stride=2, bias=True, dimension=3)
    hdown = ME.MinkowskiConvolution(cin, cmid, kernel_size=2, stride=2, dimension=3).cuda()
    hup = ME.MinkowskiConvolutionTranspose(cmid, cout, kernel_size=2, stride=2, bias=True, dimension=3).cuda()
    assert hdown.kernel.shape == (8, cin, cmid) and hup.kernel.shape == (8, cmid, cout)
    assert float(hup.kernel.detach().abs().max()) <= (cout * 8) ** -0.5  # initialised from the OUT channels
    hdown.load_state_dict(rdown.state_dict()), hup.load_state_dict(rup.state_dict())

    rtf = OME.TensorField(coordinates=coords, features=feats)
    rx = rtf.sparse()
    rF = rx.F.detach().clone().requires_grad_(True)
    ry = rdown(OME.SparseTensor(rF, rx.coordinate_map_key, rx._manager))
    rz = rup(ry)
    rc = OME.cat(rz, rx).slice(rtf).F

    htf = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda())
    hx = htf.sparse()
    hF = hx.F.detach().clone().requires_grad_(True)
    hy = hdown(ME.SparseTensor(hF, hx.coordinate_map_key, hx.coordinate_manager))
    hz = hup(hy)
    hc = ME.cat(hz, hx).slice(htf).F
    assert hy.tensor_stride[0] == 2 and hz.tensor_stride[0] == 1 and hz.coordinate_map_key == hx.coordinate_map_key
    assert torch.equal(hx.C.cpu(), rx.C) and torch.equal(hy.C.cpu(), ry.C)  # canonical (first-occurrence) order on both sides
    assert torch.allclose(hy.F.cpu(), ry.F, atol=2e-5, rtol=1e-5)
    assert torch.allclose(hz.F.cpu(), rz.F, atol=2e-5, rtol=1e-5)
    assert torch.allclose(hc.cpu(), rc, atol=2e-5, rtol=1e-5) and hc.shape == (coords.shape[0], cout + cin)
    g = torch.randn_like(rz.F)
    rz.F.backward(g)
    hz.F.backward(g.cuda())
    assert torch.allclose(hF.grad.cpu(), rF.grad, atol=2e-5, rtol=1e-4)
    for h, r in ((hdown, rdown), (hup, rup)):
        scale = float(r.kernel.grad.abs().max())
        assert torch.allclose(h.kernel.grad.cpu(), r.kernel.grad, atol=2e-5 * max(scale, 1.0), rtol=1e-4)
    assert torch.allclose(hup.bias.grad.cpu(), rup.bias.grad, atol=1e-4, rtol=1e-5)
    # up-sampling onto a map that does not exist is refused (no generative up-sampling)
    with pytest.raises(RuntimeError, match="no coordinate map"):
        ME.MinkowskiConvolutionTranspose(cin, 8, kernel_size=2, stride=2, dimension=3).cuda()(hx)


@pytest.mark.parametrize("name,fused", [("Res16UNet14", True), ("Res16UNet14", False), ("Res16UNet18A", True)])
def test_res16unet_matches_oracle(oracle_maps, name, fused):
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    torch.manual_seed(0)
    ncls, cin = 20, 28
    ref = get_model(name, cin, ncls, ME=OME)
    hip = get_model(name, cin, ncls).cuda()
    hip.load_state_dict(ref.state_dict())
    if not fused:
        for m in hip.modules():
            for attr in ("_fused", "fused"):
                if hasattr(m, attr):
                    setattr(m, attr, False)
    coords, feats = batch_scenes([31, 32, 33, 34], grid=48, cin=cin)
    rng = torch.Generator().manual_seed(1)
    coords = coords.clone()
    coords[:, 1:] += torch.rand(coords.shape[0], 3, generator=rng) * 0.9  # float field: slice() maps voxels back to points
    extra = torch.randint(0, coords.shape[0], (coords.shape[0] // 5,), generator=rng).sort().values
    order = torch.argsort(torch.cat([coords[:, 0], coords[extra, 0]]), stable=True)
    coords, feats = torch.cat([coords, coords[extra]])[order], torch.cat([feats, feats[extra] * 0.5])[order]
    labels = torch.randint(0, ncls, (coords.shape[0],), generator=rng)
    labels[::17] = -100  # ignored points (ScanNet's unlabelled class)
    hfield = hip.process_input({"coordinates": coords.cuda(), "features": feats.cuda()})
    out = hip(hfield)
    oout = ref(ref.process_input({"coordinates": coords, "features": feats}))
    assert out.shape == (coords.shape[0], ncls) == oout.shape
    assert torch.allclose(out.cpu(), oout, atol=1e-3), (out.cpu() - oout).abs().max()
    loss, oloss = F.cross_entropy(out, labels.cuda(), ignore_index=-100), F.cross_entropy(oout, labels, ignore_index=-100)
    assert abs(loss.item() - oloss.item()) < 1e-3
    loss.backward()
    oloss.backward()
    hp, rp = dict(hip.named_parameters()), dict(ref.named_parameters())
    assert hp.keys() == rp.keys()
    rel = {k: float((hp[k].grad.cpu().double() - rp[k].grad.double()).norm() / rp[k].grad.double().norm().clamp_min(1e-12)) for k in hp}
    assert max(rel.values()) < 0.15, max(rel.items(), key=lambda kv: kv[1])
    errs = sorted(rel.values())
    assert errs[len(errs) // 2] < 2e-2, errs[len(errs) // 2]  # see test_gpu_resnet.py on ReLU flips
    flat_g = torch.cat([hp[k].grad.cpu().double().flatten() for k in hp])
    flat_o = torch.cat([rp[k].grad.double().flatten() for k in hp])
    assert float(torch.dot(flat_g, flat_o) / (flat_g.norm() * flat_o.norm())) > 0.999
    hb, rb = dict(hip.named_buffers()), dict(ref.named_buffers())
    for k in hb:
        assert torch.allclose(hb[k].float().cpu(), rb[k].float(), atol=1e-3, rtol=1e-3), k
