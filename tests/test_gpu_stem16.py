"""bf16 STORAGE of the full-resolution stage (nerf_downstream_amd/csrc/stem16.hip; BASELINE config "Mink-ResNet14 bf16
mixed precision"): every piece against what it replaces, then a whole training step.

* rows_to_bf16 / stem_conv_bf16s against a float64 gather-GEMM of the SAME rounded operands (numpy, test infrastructure) and
  against the fp32-storage bf16-math kernel;
* the bf16-reading forms of the fused bn+relu+pool forward, its parameter-gradient reduction and the fused stem weight
  gradient: BITWISE equal to the fp32-reading kernels fed the same (bf16-representable) values -- same arithmetic, same
  order, only the loads differ;
* a Mink-ResNet14 step with storage "bf16" against storage "fp32" (both bf16 math).
"""
import ctypes

import numpy as np
import pytest
import torch

from helpers import batch_scenes, trunk_node

pytestmark = pytest.mark.gpu


def _stem_inputs(seeds, grid):
    from nerf_downstream_amd import minkowski as ME

    coords, feats = batch_scenes(seeds, grid=grid, cin=28)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    m, k1, k2 = x.coordinate_manager, ME.CoordinateMapKey(1), ME.CoordinateMapKey(2)
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    m.stride(k1, 2)
    nbr_pool, _ = m.kernel_table(k1, k2, 2, 1)
    i2o = m.stride_map(k1, k2)
    return x.F.contiguous(), nbr, nbr_pool, i2o


def _conv_b16(xin, w, nbr):
    from nerf_downstream_amd._lib import check, lib

    L, n = lib(), xin.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    xb = torch.empty(n, 32, dtype=torch.bfloat16, device="cuda")
    check(L.mink_rows_to_bf16(xin.data_ptr(), n, xin.shape[1], xin.stride(0), xb.data_ptr(), st))
    rows = L.mink_stem_conv_bf16s_stats_rows()
    yb = torch.empty(n, 64, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(rows, 2, 64, dtype=torch.float64, device="cuda")
    check(L.mink_stem_conv_bf16s(xb.data_ptr(), n, w.data_ptr(), xin.shape[1], nbr.data_ptr(), n, 27, yb.data_ptr(), 64, part.data_ptr(), rows, st))
    return xb, yb, part


@pytest.mark.parametrize("seeds,grid,cin", [([1], 12, 28), ([2, 3], 24, 28), ([4, 5, 6], 40, 20), ([7], 9, 32)])
def test_stem_convolution_bf16_storage_against_float64(seeds, grid, cin):
    """y (bf16) = round(sum_k x_bf16[nbr[:, k]] @ w_bf16[k]): against the float64 sum of the same rounded operands the
    stored value is off by at most half a bf16 ulp of the result (2^-9 relative) plus the fp32 accumulation error of
    <= 756 products; ragged sizes (rows not a multiple of 32, fewer 32-row blocks than waves) included."""
    from nerf_downstream_amd import minkowski as ME

    coords, feats = batch_scenes(seeds, grid=grid, cin=cin)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    k1 = ME.CoordinateMapKey(1)
    nbr, _ = x.coordinate_manager.kernel_table(k1, k1, 3, 1)
    xin = x.F.contiguous()
    torch.manual_seed(grid)
    w = torch.randn(27, cin, 64, device="cuda") * 0.1
    xb, yb, part = _conv_b16(xin, w, nbr)
    assert torch.equal(xb[:, :cin], xin.to(torch.bfloat16)) and not bool(xb[:, cin:].any())
    xr = xb[:, :cin].double().cpu().numpy()
    wr = w.to(torch.bfloat16).double().cpu().numpy()
    tab = nbr.cpu().numpy()
    ref = np.zeros((tab.shape[0], 64))
    mag = np.zeros_like(ref)
    for k in range(27):
        has = tab[:, k] >= 0
        ref[has] += xr[tab[has, k]] @ wr[k]
        mag[has] += np.abs(xr[tab[has, k]]) @ np.abs(wr[k])
    got = yb.double().cpu().numpy()
    bound = np.abs(ref) * 2.0 ** -8 + mag * 1e-6 + 1e-30
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) / bound).max())
    # statistics: column sums / sums of squares of the STORED values
    s = part.sum(0).cpu().numpy()
    assert np.allclose(s[0], got.sum(0), rtol=1e-6, atol=1e-6 * np.abs(got).sum(0).max())
    assert np.allclose(s[1], (got ** 2).sum(0), rtol=1e-6)
    again = _conv_b16(xin, w, nbr)
    assert torch.equal(again[1], yb) and torch.equal(again[2], part)  # deterministic, statistics included


def test_stem_convolution_bf16_storage_against_bf16_math_kernel():
    """Same operands, same products as the fp32-storage kernel under set_conv_math("bf16"); only the order of the fp32
    accumulation differs: after rounding to bf16 the two agree to one bf16 ulp."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.minkowski import functional as Fn

    xin, nbr, _, _ = _stem_inputs([11, 12, 13, 14, 15, 16], 64)
    assert nbr.shape[0] > 45000
    w = torch.randn(27, 28, 64, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) * 0.05
    old = ME.set_conv_math("bf16")
    try:
        y_ref = Fn.gather_gemm(xin, w, nbr, 64)
    finally:
        ME.set_conv_math(old)
    _, yb, _ = _conv_b16(xin, w, nbr)
    d = (yb.float() - y_ref).abs()
    assert bool((d <= y_ref.abs() * 2.0 ** -7 + 1e-4).all()), float(d.max())


def test_bf16_reading_stem_kernels_equal_the_fp32_reading_ones_bitwise():
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import check, lib

    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    xin, nbr, nbr_pool, i2o = _stem_inputs([21, 22, 23, 24, 25, 26], 64)
    n, npool, C = xin.shape[0], nbr_pool.shape[0], 64
    g = torch.Generator("cuda").manual_seed(2)
    w = torch.randn(27, 28, 64, device="cuda", generator=g) * 0.05
    xb, yb, _ = _conv_b16(xin, w, nbr)
    yf = yb.float().contiguous()              # the same values, stored as fp32
    xf = xb[:, :28].float().contiguous()
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt().contiguous()
    gamma = (torch.rand(C, device="cuda", generator=g) + 0.5).contiguous()
    beta = (torch.rand(C, device="cuda", generator=g) - 0.5).contiguous()
    # ---- forward: pool(relu(bn(y)))
    out_f, out_b = torch.empty(npool, C, device="cuda"), torch.empty(npool, C, device="cuda")
    check(L.mink_bn_relu_pool_fwd(yf.data_ptr(), C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), nbr_pool.data_ptr(), npool, 8, out_f.data_ptr(), st))
    check(L.mink_bn_relu_pool_fwd_b16(yb.data_ptr(), C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), nbr_pool.data_ptr(), npool, 8, out_b.data_ptr(), st))
    assert torch.equal(out_f, out_b) and float(out_f.abs().max()) > 0
    # ---- backward: parameter gradients of the norm, then the fused weight gradient
    gp = torch.randn(npool, C, device="cuda", generator=g)
    wsb = L.mink_bn_workspace_bytes(n, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dg_f, db_f, dg_b, db_b = (torch.empty(C, device="cuda") for _ in range(4))
    check(L.mink_bn_relu_pool_bwd(gp.data_ptr(), yf.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), i2o.data_ptr(), None, dg_f.data_ptr(), db_f.data_ptr(), ws.data_ptr(), wsb, st))
    check(L.mink_bn_relu_pool_bwd_b16(gp.data_ptr(), yb.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), i2o.data_ptr(), dg_b.data_ptr(), db_b.data_ptr(), ws.data_ptr(), wsb, st))
    assert torch.equal(dg_f, dg_b) and torch.equal(db_f, db_b) and float(dg_f.abs().max()) > 0
    old = ME.set_conv_math("bf16")
    try:
        assert L.mink_conv_wgrad_bn_relu_pool_supported(n, 28, 28, n, 27, C)
        need = L.mink_conv_wgrad_workspace_bytes(n, 27, 28, C)
        slabs = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        dw_f, dw_b = torch.empty(27, 28, C, device="cuda"), torch.empty(27, 28, C, device="cuda")
        check(L.mink_conv_wgrad_bn_relu_pool(xf.data_ptr(), n, 28, 28, yf.data_ptr(), C, gp.data_ptr(), npool, i2o.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dg_f.data_ptr(), db_f.data_ptr(), nbr.data_ptr(), n, 27, dw_f.data_ptr(), slabs.data_ptr(), need, st))
        check(L.mink_conv_wgrad_bn_relu_pool_b16(xb.data_ptr(), n, 28, yb.data_ptr(), C, gp.data_ptr(), npool, i2o.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dg_f.data_ptr(), db_f.data_ptr(), nbr.data_ptr(), n, 27, dw_b.data_ptr(), slabs.data_ptr(), need, st))
    finally:
        ME.set_conv_math(old)
    torch.cuda.synchronize()
    assert torch.equal(dw_f, dw_b) and float(dw_f.abs().max()) > 0
    # without bf16 math the bf16-storage weight gradient refuses (there is no fp32-MFMA kernel reading bf16 tensors)
    rc = L.mink_conv_wgrad_bn_relu_pool_b16(xb.data_ptr(), n, 28, yb.data_ptr(), C, gp.data_ptr(), npool, i2o.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dg_f.data_ptr(), db_f.data_ptr(), nbr.data_ptr(), n, 27, dw_b.data_ptr(), slabs.data_ptr(), need, st)
    assert rc != 0 and b"bf16 math" in L.mink_last_error()


def test_training_step_with_bf16_storage_against_fp32_storage():
    """One Mink-ResNet14 step on the native trunk, bf16 math both times: storing the input and the stem output as bf16
    rounds ONE more tensor (the stem output, 2^-9 relative; the input is rounded by the bf16 MFMA either way), so the
    two runs differ like two bf16 runs do: logits within 2e-2 of their scale, gradient cosine > 0.99.  The storage switch
    must really be taken (the stem output buffer is half the size) and must leave fp32 math alone."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.co3d_3d.src.models import get_model

    coords, feats = batch_scenes([31, 32, 33, 34, 35, 36], grid=64, cin=28)
    labels = torch.arange(6, device="cuda")
    res = {}
    for math, storage in (("bf16", "fp32"), ("bf16", "bf16"), ("fp32", "bf16")):
        torch.manual_seed(5)
        net = get_model("ResNet14", 28, 51).cuda()
        old_m, old_s = ME.set_conv_math(math), ME.set_conv_storage(storage)
        try:
            out = net(net.process_input({"coordinates": coords.cuda(), "features": feats.cuda()}))
            torch.nn.functional.cross_entropy(out, labels).backward()
            torch.cuda.synchronize()
        finally:
            ME.set_conv_math(old_m), ME.set_conv_storage(old_s)
        node = trunk_node(out)
        assert node is not None, "the native trunk was not taken"
        res[(math, storage)] = (out.detach().clone(), torch.cat([p.grad.flatten() for p in net.parameters()]), node.saved[0][7],
                                node.saved[0][2].numel())
    (o_f, g_f, b_f, sz_f), (o_b, g_b, b_b, sz_b), (_, _, b_x, sz_x) = res[("bf16", "fp32")], res[("bf16", "bf16")], res[("fp32", "bf16")]
    assert (b_f, b_b, b_x) == (False, True, False) and sz_b < sz_f and sz_x == sz_f
    scale = float(o_f.abs().max())
    err = float((o_b - o_f).abs().max())
    cos = float(torch.dot(g_b.double(), g_f.double()) / (g_b.double().norm() * g_f.double().norm()))
    print(f"bf16 storage vs fp32 storage (bf16 math): logits {err:.3e} of scale {scale:.2f}, gradient cosine {cos:.6f}")
    assert err < 2e-2 * max(scale, 1.0) and cos > 0.99, (err, scale, cos)
    assert torch.isfinite(g_b).all()
