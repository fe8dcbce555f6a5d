"""-m gpu: every float HIP kernel vs the CPU oracle (fp32, tolerances stated per check)."""
import numpy as np
import pytest
import torch

from helpers import batch_scenes

pytestmark = pytest.mark.gpu

ATOL = 2e-4  # fp32 accumulation-order differences over <= 27*512 products of O(1) values
RTOL = 2e-4


def _pair(seeds, grid, cin, negative=False):
    """The same TensorField on the HIP backend and on the oracle."""
    from nerf_downstream_amd import minkowski as ME
    from oracle import me_cpu as OME

    coords, feats = batch_scenes(seeds, grid=grid, cin=cin, negative=negative)
    return ME, OME, ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()), OME.TensorField(coordinates=coords, features=feats)


def _to_ts(ME, x, ts):
    pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)
    while x.tensor_stride[0] < ts:
        x = pool(x)
    return x


@pytest.mark.parametrize(
    "cin,cout,ksize,stride,ts,grid",
    [
        (28, 64, 3, 1, 1, 24),   # stem shape (vector path, Cin not a multiple of 32)
        (27, 64, 3, 1, 1, 16),   # `features=["sh"]` default: scalar gather path
        (64, 64, 3, 2, 2, 24),   # layer1.conv1 (strided: dgrad through the transposed table)
        (64, 64, 3, 1, 4, 24),   # layer1.conv2 (same-map dgrad with flipped offsets)
        (64, 128, 1, 2, 2, 24),  # downsample 1x1 stride 2
        (128, 256, 3, 2, 4, 32), # small N: split-K path
        (256, 256, 3, 1, 8, 32),
        (256, 512, 3, 2, 8, 64),
        (5, 7, 3, 1, 1, 12),     # odd channel counts everywhere (fully guarded path)
    ],
)
def test_convolution(oracle_maps, cin, cout, ksize, stride, ts, grid):
    _convolution_case(oracle_maps, cin, cout, ksize, stride, ts, grid)


def _convolution_case(oracle_maps, cin, cout, ksize, stride, ts, grid):
    torch.manual_seed(1)
    ME, OME, tf, otf = _pair([3, 4], grid, cin, negative=True)
    x, ox = _to_ts(ME, tf.sparse(), ts), _to_ts(OME, otf.sparse(), ts)
    assert torch.allclose(x.F.cpu(), ox.F, atol=1e-5)
    oconv = OME.MinkowskiConvolution(cin, cout, kernel_size=ksize, stride=stride, dimension=3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ksize, stride=stride, dimension=3).cuda()
    conv.load_state_dict(oconv.state_dict())
    F_g = x.F.detach().clone().requires_grad_(True)
    F_c = ox.F.detach().clone().requires_grad_(True)
    y = conv(ME.SparseTensor(F_g, x.coordinate_map_key, x.coordinate_manager))
    oy = oconv(OME.SparseTensor(F_c, ox.coordinate_map_key, ox.coordinate_manager))
    assert y.F.shape == oy.F.shape and y.tensor_stride == oy.tensor_stride
    assert np.array_equal(y.C.cpu().numpy(), oy.C.numpy())
    assert torch.allclose(y.F.cpu(), oy.F, atol=ATOL, rtol=RTOL)
    g = torch.randn_like(oy.F)
    y.F.backward(g.cuda())
    oy.F.backward(g)
    assert torch.allclose(F_g.grad.cpu(), F_c.grad, atol=ATOL, rtol=RTOL)
    scale = max(1.0, float(oconv.kernel.grad.abs().max()))
    assert torch.allclose(conv.kernel.grad.cpu(), oconv.kernel.grad, atol=ATOL * scale, rtol=RTOL)


def test_convolution_is_deterministic():
    ME, _, tf, _ = _pair([5, 6], 24, 28)
    x = tf.sparse()
    conv = ME.MinkowskiConvolution(28, 64, kernel_size=3, dimension=3).cuda()
    a = conv(x).F
    b = conv(x).F
    assert torch.equal(a, b)  # no atomics anywhere: bitwise reproducible


@pytest.mark.parametrize("C,relu,res", [(64, False, False), (64, True, False), (128, True, True), (512, False, True),
                                        (2048, True, True), (1536, False, False)])  # > 1024 channels: column slabs
def test_batch_norm(C, relu, res):
    from nerf_downstream_amd import minkowski as ME

    torch.manual_seed(0)
    n = 3001
    x = (torch.randn(n, C) * 2 + 0.5)
    r = torch.randn(n, C) if res else None
    bn = ME.MinkowskiBatchNorm(C).cuda()
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5), ref.bias.uniform_(-0.5, 0.5)
    bn.bn.load_state_dict(ref.state_dict())
    coords = torch.zeros(n, 4)
    coords[:, 1] = torch.arange(n)
    m = ME.TensorField(coordinates=coords.cuda(), features=x.cuda()).coordinate_manager
    key = ME.CoordinateMapKey(1)
    xg = x.cuda().requires_grad_(True)
    rg = r.cuda().requires_grad_(True) if res else None
    y = bn(ME.SparseTensor(xg, key, m), relu=relu, residual=ME.SparseTensor(rg, key, m) if res else None).F
    xc = x.clone().requires_grad_(True)
    rc = r.clone().requires_grad_(True) if res else None
    yc = ref(xc)
    if res:
        yc = yc + rc
    if relu:
        yc = torch.relu(yc)
    assert torch.allclose(y.cpu(), yc, atol=1e-5, rtol=1e-5)
    assert torch.allclose(bn.bn.running_mean.cpu(), ref.running_mean, atol=1e-6)
    assert torch.allclose(bn.bn.running_var.cpu(), ref.running_var, atol=1e-5)
    assert int(bn.bn.num_batches_tracked) == 1
    g = torch.randn(n, C)
    y.backward(g.cuda())
    yc.backward(g)
    assert torch.allclose(xg.grad.cpu(), xc.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(bn.bn.weight.grad.cpu(), ref.weight.grad, atol=2e-3, rtol=1e-4)
    assert torch.allclose(bn.bn.bias.grad.cpu(), ref.bias.grad, atol=2e-3, rtol=1e-4)
    if res:
        assert torch.allclose(rg.grad.cpu(), rc.grad, atol=1e-6)
    # eval mode uses the running statistics
    bn.eval(), ref.eval()
    ye = bn(ME.SparseTensor(x.cuda(), key, m)).F
    assert torch.allclose(ye.cpu(), ref(x), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("n,C,nslab,relu,res", [(1, 16, 0, True, False), (63, 64, 3, True, True), (512, 512, 14, True, False),
                                                (1024, 128, 1, False, True), (777, 256, 5, True, True), (300, 48, 0, False, False)])
def test_one_launch_batch_norm_of_few_row_layers(n, C, nslab, relu, res):
    """mink_bn_small_fwd / _bwd (the native trunk's layers below 1,024 rows: split-K slab sum + statistics + norm + residual +
    ReLU in one launch; backward likewise) against torch.nn.BatchNorm1d in float64 on the same values -- ragged row counts,
    one row (variance 0), channel counts that are not multiples of 64, slabs and no slabs."""
    from nerf_downstream_amd._lib import check, lib

    L = lib()
    assert L.mink_bn_small_rows() == 1024
    torch.manual_seed(n + C)
    dev = torch.device("cuda", 0)
    slabs = (torch.randn(max(nslab, 1), n, C) * 1.5 + 0.2).to(dev)
    y = slabs[0].clone() if nslab == 0 else torch.empty(n, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    r = torch.randn(n, C, device=dev) if res else None
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    out, mean, invstd = torch.empty(n, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    check(L.mink_bn_small_fwd(slabs.data_ptr() if nslab else None, nslab, n, C, y.data_ptr(), 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(),
                              r.data_ptr() if res else None, int(relu), out.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(),
                              rv.data_ptr(), st))
    y64 = (slabs[:max(nslab, 1)].double().sum(0) if nslab else slabs[0].double()).cpu().requires_grad_(True)
    assert torch.allclose(y.cpu().double(), y64.detach(), atol=1e-5, rtol=1e-6)
    ref = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref.weight.copy_(gamma.cpu().double()), ref.bias.copy_(beta.cpu().double())
    if n == 1:  # torch refuses one row in training mode; the kernels follow the formula (variance 0)
        z = (y64 - y64.mean(0)) / torch.sqrt(y64.var(0, unbiased=False) + 1e-5) * ref.weight + ref.bias
    else:
        z = ref(y64)
    rc = r.cpu().double().requires_grad_(True) if res else None
    if res:
        z = z + rc
    if relu:
        z = torch.relu(z)
    assert torch.allclose(out.cpu().double(), z.detach(), atol=2e-5, rtol=1e-5)
    assert torch.allclose(mean.cpu().double(), y64.detach().mean(0), atol=1e-5)
    if n > 1:
        assert torch.allclose(rm.cpu().double(), ref.running_mean, atol=1e-6) and torch.allclose(rv.cpu().double(), ref.running_var, atol=1e-5)
    # backward: the incoming gradient as `nslab` slabs too
    gsl = torch.randn(max(nslab, 1), n, C, device=dev)
    g_sum = torch.empty(n, C, device=dev)
    dx, dres = torch.empty(n, C, device=dev), (torch.empty(n, C, device=dev) if res else None)
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    check(L.mink_bn_small_bwd(gsl.data_ptr(), nslab, None, g_sum.data_ptr() if nslab else None, y.data_ptr(), out.data_ptr(), n, C, mean.data_ptr(),
                              invstd.data_ptr(), gamma.data_ptr(), int(relu), dx.data_ptr(), dres.data_ptr() if res else None,
                              dgamma.data_ptr(), dbeta.data_ptr(), st))
    g64 = gsl[:max(nslab, 1)].double().sum(0).cpu() if nslab else gsl[0].double().cpu()
    if nslab:
        assert torch.allclose(g_sum.cpu().double(), g64, atol=1e-5)
    z.backward(g64)
    scale = float(y64.grad.abs().max()) + 1e-12
    assert float((dx.cpu().double() - y64.grad).abs().max()) < 2e-4 * max(scale, 1.0)
    if n > 1:
        assert torch.allclose(dgamma.cpu().double(), ref.weight.grad, atol=2e-3, rtol=1e-4)
    assert torch.allclose(dbeta.cpu().double(), ref.bias.grad, atol=2e-3, rtol=1e-4)
    if res:
        assert torch.allclose(dres.cpu().double(), rc.grad, atol=1e-6)


@pytest.mark.parametrize("n,C,rows", [(2128, 256, 67), (8432, 128, 128), (512, 512, 16), (37056, 64, 128), (100, 64, 1), (3000, 192, 100)])
def test_batch_norm_finalize_inside_the_apply_pass_is_bitwise(n, C, rows):
    """mink_bn_apply_from_partials / mink_bn_bwd with the finalize folded into the apply pass (one launch less each, <= 128
    partial rows) against the separate launches (mink_bn_set_fold(0)): outputs, statistics, running statistics and parameter
    gradients equal BIT FOR BIT -- every workgroup re-derives the sums in the finalize kernel's own order."""
    from nerf_downstream_amd._lib import check, lib

    L = lib()
    torch.manual_seed(rows + C)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    x = (torch.randn(n, C) * 1.3 + 0.4).to(dev)
    part = torch.zeros(rows, 2, C, dtype=torch.float64, device=dev)
    for r in range(rows):  # genuine partials of x over interleaved rows
        xs = x[r::rows].double()
        part[r, 0], part[r, 1] = xs.sum(0), (xs * xs).sum(0)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    res = torch.randn(n, C, device=dev)
    outs = {}
    for fold in (0, 1128):  # (0: separate launches; 1000 + 128: folded up to 128 partial rows whatever the re-read volume)
        old = L.mink_bn_set_fold(fold)
        try:
            y, mean, invstd = torch.empty(n, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            check(L.mink_bn_apply_from_partials(x.data_ptr(), n, C, part.data_ptr(), rows, 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), res.data_ptr(),
                                                1, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), st))
            gy = torch.randn(n, C, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
            dx, dres = torch.empty(n, C, device=dev), torch.empty(n, C, device=dev)
            dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
            ws = torch.empty(L.mink_bn_workspace_bytes(n, C), dtype=torch.uint8, device=dev)
            check(L.mink_bn_bwd(gy.data_ptr(), x.data_ptr(), y.data_ptr(), n, C, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), 1,
                                dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), st))
            torch.cuda.synchronize()
            outs[fold] = (y, mean, invstd, rm, rv, dx, dres, dg, db)
        finally:
            L.mink_bn_set_fold(old)
    for k, (a, b) in enumerate(zip(outs[0], outs[1128])):
        assert torch.equal(a, b), k
    ref = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref.weight.copy_(gamma.cpu().double()), ref.bias.copy_(beta.cpu().double())
    z = torch.relu(ref(x.cpu().double()) + res.cpu().double())
    assert torch.allclose(outs[1128][0].cpu().double(), z, atol=2e-5, rtol=1e-5)


def test_relu_add_pool_globalavg(oracle_maps):
    ME, OME, tf, otf = _pair([8, 9, 10], 24, 8, negative=True)
    x, ox = tf.sparse(), otf.sparse()
    F_g = x.F.detach().clone().requires_grad_(True)
    F_c = ox.F.detach().clone().requires_grad_(True)
    xs = ME.SparseTensor(F_g, x.coordinate_map_key, x.coordinate_manager)
    oxs = OME.SparseTensor(F_c, ox.coordinate_map_key, ox.coordinate_manager)
    a = ME.MinkowskiReLU()(xs)
    a += xs
    oa = OME.MinkowskiReLU()(oxs)
    oa += oxs
    p = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)(a)
    op = OME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)(oa)
    assert np.array_equal(p.C.cpu().numpy(), op.C.numpy())
    assert torch.allclose(p.F.cpu(), op.F, atol=1e-5)
    g = ME.MinkowskiGlobalAvgPooling()(p)
    og = OME.MinkowskiGlobalAvgPooling()(op)
    assert g.F.shape == (3, 8) and torch.allclose(g.F.cpu(), og.F, atol=1e-5)
    assert np.array_equal(g.C.cpu().numpy(), og.C.numpy())
    w = torch.randn(3, 8)
    (g.F * w.cuda()).sum().backward()
    (og.F * w).sum().backward()
    assert torch.allclose(F_g.grad.cpu(), F_c.grad, atol=1e-6)


def test_field_to_sparse_average(oracle_maps):
    from nerf_downstream_amd import minkowski as ME
    from oracle import me_cpu as OME

    rng = np.random.default_rng(0)
    coords, feats = batch_scenes([1, 2], grid=12, cin=6, negative=True)
    rep = torch.from_numpy(np.sort(rng.integers(0, len(coords), 3 * len(coords))))
    coords, feats = coords[rep].clone(), torch.from_numpy(rng.standard_normal((len(rep), 6)).astype(np.float32))
    coords[:, 1:] += torch.from_numpy(rng.uniform(0, 0.99, (len(rep), 3)).astype(np.float32))
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    ox = OME.TensorField(coordinates=coords, features=feats).sparse()
    assert x.F.shape[0] < len(rep)
    assert np.array_equal(x.C.cpu().numpy(), ox.C.numpy())
    assert torch.allclose(x.F.cpu(), ox.F, atol=1e-6)


@pytest.mark.parametrize("mode,tol", [("bf16", 3e-2), ("bf16x3", 2e-4)])
@pytest.mark.parametrize("cin,cout,stride,ts", [(28, 64, 1, 1), (64, 128, 2, 2), (128, 128, 1, 4)])
def test_convolution_reduced_math(oracle_maps, mode, tol, cin, cout, stride, ts):
    """bf16 MFMA (BASELINE config "bf16 mixed precision") and split-bf16 modes of the forward /
    input-gradient GEMMs against the fp32 oracle; tolerance relative to the output scale."""
    torch.manual_seed(2)
    ME, OME, tf, otf = _pair([3, 4], 24, cin, negative=True)
    x, ox = _to_ts(ME, tf.sparse(), ts), _to_ts(OME, otf.sparse(), ts)
    oconv = OME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=stride, dimension=3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=stride, dimension=3).cuda()
    conv.load_state_dict(oconv.state_dict())
    F_g = x.F.detach().clone().requires_grad_(True)
    F_c = ox.F.detach().clone().requires_grad_(True)
    old = ME.set_conv_math(mode)
    try:
        y = conv(ME.SparseTensor(F_g, x.coordinate_map_key, x.coordinate_manager))
        oy = oconv(OME.SparseTensor(F_c, ox.coordinate_map_key, ox.coordinate_manager))
        g = torch.randn_like(oy.F)
        y.F.backward(g.cuda())
        oy.F.backward(g)
    finally:
        ME.set_conv_math(old)
    for a, b in [(y.F, oy.F), (F_g.grad, F_c.grad)]:
        scale = float(b.detach().abs().max())
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= tol * scale, (mode, float((a.detach().cpu() - b.detach()).abs().max()), scale)
    gw, ogw = conv.kernel.grad.cpu(), oconv.kernel.grad
    if mode == "bf16" and cin % 64 == 0 and cout % 64 == 0:
        # round 5: under --math bf16 the mid-layer weight gradients run on the bf16 matrix cores too (wgrad16_kernel)
        err = float((gw - ogw).abs().max()) / float(ogw.abs().max())
        assert 1e-5 < err <= tol, err  # (really bf16 operands, and within their accuracy)
    else:  # split-bf16 and the narrow stem keep the exact-fp32 weight-gradient kernels
        assert torch.allclose(gw, ogw, atol=ATOL * max(1.0, float(ogw.abs().max())), rtol=RTOL)


def test_fused_bn_relu_sumpool(oracle_maps):
    """pool(relu(bn(x))) fused (extension kwarg of MinkowskiSumPooling) == the three modules."""
    from nerf_downstream_amd import minkowski as ME

    torch.manual_seed(4)
    _, _, tf, _ = _pair([6, 7], 24, 16, negative=True)
    x = tf.sparse()
    m, key = x.coordinate_manager, x.coordinate_map_key
    bn_a, bn_b = ME.MinkowskiBatchNorm(16).cuda(), ME.MinkowskiBatchNorm(16).cuda()
    with torch.no_grad():
        bn_a.bn.weight.uniform_(0.5, 1.5), bn_a.bn.bias.uniform_(-0.5, 0.5)
    bn_b.load_state_dict(bn_a.state_dict())
    pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)
    Fa = x.F.detach().clone().requires_grad_(True)
    Fb = x.F.detach().clone().requires_grad_(True)
    ya = pool(ME.SparseTensor(Fa, key, m), norm=bn_a).F
    yb = pool(ME.MinkowskiReLU()(bn_b(ME.SparseTensor(Fb, key, m)))).F
    assert torch.allclose(ya, yb, atol=1e-5, rtol=1e-5)
    g = torch.randn_like(ya)
    ya.backward(g), yb.backward(g)
    assert torch.allclose(Fa.grad, Fb.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(bn_a.bn.weight.grad, bn_b.bn.weight.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(bn_a.bn.bias.grad, bn_b.bn.bias.grad, atol=1e-3, rtol=1e-4)
    assert torch.allclose(bn_a.bn.running_var, bn_b.bn.running_var) and int(bn_a.bn.num_batches_tracked) == 1


@pytest.mark.parametrize("cin,ldx,drop", [(28, 28, 0), (28, 32, 1), (20, 20, 3), (32, 32, 0)])
def test_stem_wgrad_streaming(cin, ldx, drop):
    """The streaming weight-gradient kernel (K = 27, cin <= 32, >= ~44k rows; conv.hip
    wgrad_stream_kernel) against the tiled LDS kernel and an fp64 torch restatement of
    dW[k] = X[nbr[:, k]]^T dY on the same neighbour table.  `drop` trims rows so the row count is
    odd / not a tile multiple; ldx > cin exercises a padded feature matrix."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    coords, feats = batch_scenes([11, 12, 13], grid=80, cin=4)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    m, key = x.coordinate_manager, x.coordinate_map_key
    nbr, _ = m.kernel_table(key, key, 3, 1)
    n = nbr.shape[0] - drop
    assert n >= 44000, n
    nbr = nbr[:n].contiguous()
    torch.manual_seed(cin)
    xin = torch.randn(x.F.shape[0], ldx, device="cuda")[:, :cin]
    dy = torch.randn(n, 64, device="cuda")
    got = Fn.conv_wgrad(xin, dy, nbr, (27, cin, 64))
    old = lib().mink_conv_set_stagger(1024)
    try:
        tiled = Fn.conv_wgrad(xin, dy, nbr, (27, cin, 64))
    finally:
        lib().mink_conv_set_stagger(old)
    ref = torch.zeros(27, cin, 64, dtype=torch.float64, device="cuda")
    for k in range(27):
        idx = nbr[:, k].long()
        ok = idx >= 0
        ref[k] = xin[idx[ok]].double().t() @ dy[ok].double()
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-5 * scale   # fp32 sums of ~1e4 O(1) products
    assert float((tiled.double() - ref).abs().max()) <= 2e-5 * scale
    assert torch.equal(got, Fn.conv_wgrad(xin, dy, nbr, (27, cin, 64)))  # deterministic


@pytest.mark.parametrize("stride", [1, 2])
def test_split_k_every_split_matches_unsplit(stride):
    """Every offset split the planner can return (gridDim.z = 1..27, including splits whose last
    slice is short) gives the un-split result for forward, same-map dgrad (flipped offsets) and
    the class-permuted stride-2 dgrad."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.minkowski import functional as Fn

    coords, feats = batch_scenes([21, 22], grid=28, cin=64)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    m, k_in = x.coordinate_manager, x.coordinate_map_key
    k_out = m.stride(k_in, stride)
    nbr, nbr_t = m.kernel_table(k_in, k_out, 3, 1, transposed=(stride != 1))
    torch.manual_seed(stride)
    w = torch.randn(27, 64, 128, device="cuda") * 0.05
    wt = w.transpose(1, 2).contiguous()
    gy = torch.randn(nbr.shape[0], 128, device="cuda")
    perm = m.class_perm(k_in) if stride == 2 else None

    def run():
        y = Fn.gather_gemm(x.F, w, nbr, 128)
        if stride == 1:  # the two forms of the data gradient: explicit W^T, and W read transposed in place
            dx = Fn.gather_gemm(gy, wt, nbr, 64, flip_k=True)
            dx_t = Fn.gather_gemm(gy, w, nbr, 64, w_transposed=True, flip_k=True)
        else:
            dx = Fn.gather_gemm(gy, wt, nbr_t, 64, row_perm=perm)
            dx_t = Fn.gather_gemm(gy, w, nbr_t, 64, w_transposed=True, row_perm=perm)
        assert torch.allclose(dx, dx_t, atol=1e-4, rtol=1e-4)
        return y, dx_t

    try:
        Fn._FORCE_KSPLIT = 1
        y1, dx1 = run()
        for zs in sorted({-(-27 // kper) for kper in range(1, 28)}):
            Fn._FORCE_KSPLIT = zs
            y, dx = run()
            assert torch.allclose(y, y1, atol=1e-4, rtol=1e-4), (zs, float((y - y1).abs().max()))
            assert torch.allclose(dx, dx1, atol=1e-4, rtol=1e-4), (zs, float((dx - dx1).abs().max()))
    finally:
        Fn._FORCE_KSPLIT = 0


def test_full_size_stem_convolution_against_fp64_torch():
    """BASELINE shape (B=16 scenes of the 128^3 synthetic grid = 825 k voxels, 28 -> 64, K=27):
    the two dominant kernels -- flattened-K forward and streaming weight gradient -- and the
    same-map input gradient against an fp64 torch restatement of y = sum_k X[nbr[:,k]] W[k] on
    the neighbour table (itself pinned bit-exact to the oracle at small sizes and by
    test_full_size_properties at this size).  Also linearity, a size-independent property."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_batches
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.minkowski import functional as Fn

    b = make_batches(1, 16, 0, 51, 128, 28)[0]
    x = ME.TensorField(coordinates=b["coordinates"].cuda(), features=b["features"].cuda()).sparse()
    m, key = x.coordinate_manager, x.coordinate_map_key
    nbr, _ = m.kernel_table(key, key, 3, 1)
    n = nbr.shape[0]
    assert n > 800_000
    torch.manual_seed(0)
    xin = x.F.contiguous()
    w = torch.randn(27, 28, 64, device="cuda") * 0.05
    gy = torch.randn(n, 64, device="cuda")
    y = Fn.gather_gemm(xin, w, nbr, 64)
    dw = Fn.conv_wgrad(xin, gy, nbr, (27, 28, 64))
    dx = Fn.gather_gemm(gy, w, nbr, 28, w_transposed=True, flip_k=True)
    y_ref = torch.zeros(n, 64, dtype=torch.float64, device="cuda")
    dx_ref = torch.zeros(n, 28, dtype=torch.float64, device="cuda")
    dw_ref = torch.zeros(27, 28, 64, dtype=torch.float64, device="cuda")
    for k in range(27):
        idx = nbr[:, k].long()
        ok = (idx >= 0).nonzero().flatten()
        src = idx[ok]
        xs, gs = xin[src].double(), gy[ok].double()
        y_ref[ok] += xs @ w[k].double()
        dw_ref[k] = xs.t() @ gs
        dx_ref.index_add_(0, src, gs @ w[k].double().t())  # dx[i] += dy[o] W[k]^T for every pair (i, o)
    for got, ref, name in ((y, y_ref, "fwd"), (dx, dx_ref, "dgrad"), (dw, dw_ref, "wgrad")):
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max())
        assert err <= 3e-5 * scale, (name, err, scale)  # fp32 sums of <= 756 (fwd/dgrad) / ~6e5 (wgrad) products
    assert torch.allclose(Fn.gather_gemm(2.0 * xin, w, nbr, 64), 2.0 * y, rtol=1e-6, atol=1e-6)  # exact scaling by 2


@pytest.mark.parametrize("cin", [28, 27])
def test_fused_stem_conv_bn_relu_pool(cin):
    """pool(relu(bn(conv(x)))) as one autograd node (weight gradient with the batch-norm backward
    folded into its operand load) against the same modules applied one after the other, at a size
    the streaming kernel takes (124 k rows)."""
    import copy

    from nerf_downstream_amd import minkowski as ME

    coords, feats = batch_scenes([41, 42, 43], grid=80, cin=cin)
    x = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda()).sparse()
    torch.manual_seed(cin)
    conv = ME.MinkowskiConvolution(cin, 64, kernel_size=3, dimension=3).cuda()
    bn = ME.MinkowskiBatchNorm(64).cuda()
    with torch.no_grad():
        bn.bn.weight.uniform_(0.5, 1.5), bn.bn.bias.uniform_(-0.3, 0.3)
    conv2, bn2 = copy.deepcopy(conv), copy.deepcopy(bn)
    pool = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)
    fused = pool(x, norm=bn, conv=conv)
    assert type(fused.F.grad_fn).__name__ == "ConvBNReLUSumPoolFunctionBackward"  # the fused node really ran
    plain = pool(conv2(x), norm=bn2)
    assert torch.allclose(fused.F, plain.F, atol=1e-4, rtol=1e-4)
    assert torch.allclose(bn.bn.running_var, bn2.bn.running_var, rtol=1e-5) and int(bn.bn.num_batches_tracked) == 1
    g = torch.randn_like(plain.F)
    fused.F.backward(g)
    plain.F.backward(g)
    for a, b, name in ((conv.kernel.grad, conv2.kernel.grad, "dW"), (bn.bn.weight.grad, bn2.bn.weight.grad, "dgamma"),
                       (bn.bn.bias.grad, bn2.bn.bias.grad, "dbeta")):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("n,cin,cout", [(5000, 128, 96), (70001, 64, 256), (4096, 20, 36)])
def test_pointwise_convolution_weight_gradient(n, cin, cout):
    """1x1x1 stride-1 convolution (`use_mm`): library GEMMs forward / input gradient, the streaming
    weight-gradient kernel (identity table) for X^T dY -- against torch in float64."""
    from nerf_downstream_amd import minkowski as ME

    torch.manual_seed(0)
    coords = torch.zeros(n, 4)
    coords[:, 1], coords[:, 2] = torch.arange(n) % 1000, torch.arange(n) // 1000
    x = torch.randn(n, cin)
    tf = ME.TensorField(coordinates=coords.cuda(), features=x.cuda())
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=1, stride=1, bias=True, dimension=3).cuda()
    assert conv.use_mm and conv.kernel.shape == (cin, cout)
    xg = x.cuda().requires_grad_(True)
    y = conv(ME.SparseTensor(xg, ME.CoordinateMapKey(1), tf.coordinate_manager)).F
    assert "PointwiseConvolution" in type(y.grad_fn.next_functions[0][0]).__name__
    g = torch.randn(n, cout)
    y.backward(g.cuda())
    xd, wd = x.double().requires_grad_(True), conv.kernel.detach().cpu().double().requires_grad_(True)
    yd = xd @ wd + conv.bias.detach().cpu().double()
    yd.backward(g.double())
    assert torch.allclose(y.detach().cpu().double(), yd.detach(), atol=1e-4, rtol=1e-4)
    assert torch.allclose(xg.grad.cpu().double(), xd.grad, atol=1e-4, rtol=1e-4)
    scale = float(wd.grad.abs().max())
    assert float((conv.kernel.grad.cpu().double() - wd.grad).abs().max()) < 2e-5 * scale


def test_stem_weight_gradient_on_bf16_matrix_cores(oracle_maps):
    """conv math "bf16": the stem's streaming weight-gradient kernel runs on the bf16 MFMA as well (operands loaded as
    fp32, packed to bf16 in registers, fp32 accumulation) -- plain form against the fp32 kernel on the same inputs, and
    the fused form (dY recomputed from the conv output and the pooled gradient) through a whole Mink-ResNet14 step.
    bf16 rounds each operand to 8 significant bits: relative L2 error of a 50 k-row sum ~ 2^-9 / sqrt(terms) per element,
    bounded here by 1e-2."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.minkowski import functional as Fn

    coords, feats = batch_scenes([51, 52, 53, 54, 55, 56], grid=64, cin=28)
    tf = ME.TensorField(coordinates=coords.cuda(), features=feats.cuda())
    x = tf.sparse()
    m, k1 = x.coordinate_manager, ME.CoordinateMapKey(1)
    nbr, _ = m.kernel_table(k1, k1, 3, 1)
    assert nbr.shape[0] > 45000  # large enough for the streaming kernel
    gy = torch.randn(nbr.shape[0], 64, device="cuda", generator=torch.Generator("cuda").manual_seed(0))
    ref = Fn.conv_wgrad(x.F.contiguous(), gy, nbr, (27, 28, 64))
    old = ME.set_conv_math("bf16")
    try:
        got = Fn.conv_wgrad(x.F.contiguous(), gy, nbr, (27, 28, 64))
        again = Fn.conv_wgrad(x.F.contiguous(), gy, nbr, (27, 28, 64))
    finally:
        ME.set_conv_math(old)
    rel = float((got - ref).norm() / ref.norm())
    assert 1e-6 < rel < 1e-2, rel  # really bf16 operands (not the fp32 kernel), and within bf16 accuracy
    assert torch.equal(got, again)  # deterministic
    # fused form: one training step of the whole network in bf16 math, with the weight gradient on the bf16 matrix
    # cores and on the exact-fp32 kernel (same upstream gradients either way)
    from nerf_downstream_amd._lib import lib

    grads = {}
    for tag, flag in (("bf16 wgrad", 0), ("fp32 wgrad", 1 << 28)):
        torch.manual_seed(4)
        net = get_model("ResNet14", 28, 51).cuda()
        old = ME.set_conv_math("bf16")
        lib().mink_conv_set_stagger(flag)
        try:
            out = net(net.process_input({"coordinates": coords.cuda(), "features": feats.cuda()}))
            torch.nn.functional.cross_entropy(out, torch.arange(6, device="cuda")).backward()
            torch.cuda.synchronize()
        finally:
            lib().mink_conv_set_stagger(0)
            ME.set_conv_math(old)
        assert net._trunk_plan, "the native trunk (fused stem) was not taken"
        grads[tag] = net.conv1.kernel.grad.clone()
    rel = float((grads["bf16 wgrad"] - grads["fp32 wgrad"]).norm() / grads["fp32 wgrad"].norm())
    assert 1e-6 < rel < 1e-2, rel


@pytest.mark.parametrize("n_out,cin,cout,K", [(1000, 64, 64, 27), (4097, 64, 128, 27), (129, 128, 64, 27), (530, 256, 256, 27), (70, 512, 512, 27),
                                              (36754, 64, 64, 27)])
def test_mid_layer_weight_gradient_on_the_bf16_matrix_cores(n_out, cin, cout, K):
    """wgrad16_kernel (BASELINE config #4 for the mid-layer weight gradients: bf16 operands, row-major bf16 tiles in LDS, fragments
    through the transposing LDS read, fp32 accumulation) on synthetic tables with the mid-layer fill -- ragged row counts (partial
    tiles, partial 16-pair MFMA steps, an offset nobody has, one a whole tile lacks):
      * against a float64 sum of the SAME operands rounded to bf16 (what the kernel is asked to compute): fp32-accumulation accuracy;
      * against the exact-fp32 kernel: within bf16 operand rounding (and not equal to it: the bf16 kernel really ran);
      * bitwise against itself."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out + cin + 3 * cout)
    n_in = max(4, n_out + 17)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) < 0.45] = -1
    nbr[:, K // 2 - 1] = -1
    if n_out > 300:
        nbr[128:256, 1] = -1
    x = torch.randn(n_in, cin, generator=g)
    dy = torch.randn(n_out, cout, generator=g)
    xb, dyb = x.bfloat16().double(), dy.bfloat16().double()
    ref = torch.zeros(K, cin, cout, dtype=torch.float64)
    for k in range(K):
        sel = nbr[:, k] >= 0
        ref[k] = xb[nbr[sel, k].long()].T @ dyb[sel]
    xd, dyd, nd = x.to(dev), dy.to(dev), nbr.to(dev)
    exact = Fn.conv_wgrad(xd, dyd, nd, (K, cin, cout))
    old = ME.set_conv_math("bf16")
    try:
        got = Fn.conv_wgrad(xd, dyd, nd, (K, cin, cout))
        again = Fn.conv_wgrad(xd, dyd, nd, (K, cin, cout))
        lib().mink_conv_set_stagger(1 << 28)  # bit 28: bf16 math keeps the exact-fp32 weight-gradient kernel
        kept = Fn.conv_wgrad(xd, dyd, nd, (K, cin, cout))
    finally:
        lib().mink_conv_set_stagger(0)
        ME.set_conv_math(old)
    scale = float(ref.abs().max()) + 1e-30
    err = float((got.cpu().double() - ref).abs().max()) / scale
    assert err < 2e-5, err  # fp32 accumulation of <= n_out products of bf16 operands (exact products: 16 significant bits)
    assert torch.equal(got, again)
    assert torch.equal(kept, exact)  # the switch really selects the fp32 kernel ...
    rel = float((got - exact).norm() / exact.norm())
    assert 1e-4 < rel < 1e-2, rel  # ... and the default really rounds the operands to bf16


def _chain_tolerance(xd, w, nd, ref, scale, flip=False):
    """Bound for max |y - float64| / max |y| of the row-compacted kernels.  Since round 5 an output element is ONE fp32
    accumulation chain over all its ~K x cin products (an offset's accumulators START as the C rows they belong to; until
    round 4 every offset was summed from zero and then added to C: K short chains) -- the order of a plain fp32 GEMM over
    the flattened (offset, channel) axis.  The yardstick is therefore measured, not modelled: the same terms summed one
    fused step at a time in fp32, offsets and channels ascending (torch on the GPU: IEEE fp32), against the same float64
    reference; the kernel may be at most twice as far away (its MFMAs add four products per step in another association),
    and never needs more than the 2e-6 that held before for the short chains."""
    K, cin, cout = w.shape
    s = torch.zeros(nd.shape[0], cout, device=xd.device)
    for k in range(K):
        sel = nd[:, k] >= 0
        xs = torch.where(sel[:, None], xd[nd[:, k].clamp_min(0).long()], torch.zeros((), device=xd.device))
        wk = w[K - 1 - k if flip else k]
        for c in range(cin):
            s.addcmul_(xs[:, c : c + 1], wk[c][None, :])
    err_seq = float((s.cpu().double() - ref).abs().max()) / scale
    return max(2e-6, 2.0 * err_seq)


@pytest.mark.parametrize("n_out,K,cin,cout", [(1, 27, 64, 64), (63, 27, 64, 128), (65, 27, 96, 64), (1000, 27, 128, 128),
                                              (4097, 27, 64, 64), (300, 8, 64, 64), (129, 9, 256, 64), (128, 27, 512, 512), (530, 27, 256, 256)])
@pytest.mark.parametrize("transposed", [False, True])
def test_row_compacted_kernel_against_float64(n_out, K, cin, cout, transposed):
    """compact_gemm_kernel (fp32 mid layers: per-offset row compaction, C tile in LDS) on synthetic tables: ragged row
    counts (1, 63, 65 rows: partial tiles and partial 16-row blocks), offsets without any neighbour in a tile, every
    legal split of the offsets (un-split launches for K <= 9 take the direct epilogue with bias and the fused column
    statistics), forward and transposed-weight (data gradient) forms -- against a float64 gather + matmul, and bitwise
    against itself (two runs)."""
    import ctypes
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out * 31 + K + cin)
    n_in = max(4, n_out + 17)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) < 0.45] = -1      # the mid-layer fill
    nbr[:, K // 2 - 1] = -1                                  # an offset nobody has
    if n_out > 70:
        nbr[64:128, 1] = -1                                  # ... and one that a whole tile lacks
    x = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) * 0.1
    bias = torch.randn(cout, generator=g)
    wk = w.transpose(1, 2).contiguous() if transposed else w   # transposed form reads W[k] as [cout][cin] and flips k
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        sel = nbr[:, k] >= 0
        kw = K - 1 - k if transposed else k
        ref[sel] += x[nbr[sel, k].long()].double() @ w[kw].double()
    xd, wd, nd, bd = x.to(dev), wk.to(dev), nbr.to(dev), bias.to(dev)
    scale = float(ref.abs().max()) + 1e-30
    tol = _chain_tolerance(xd, w.to(dev), nd, ref, scale, flip=transposed)
    L = lib()
    assert L.mink_conv_plan(n_out, K, cin, cout, 0) * 9 >= K, "the planner must hand this shape to the compacted kernel"
    splits = sorted({-(-K // kper) for kper in range(1, 10)})
    try:
        for zs in splits:
            Fn._FORCE_KSPLIT = zs
            direct = zs == 1
            if transposed:
                y = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=True, flip_k=True, bias=bd if direct else None)
                part = None
            else:
                y, part = Fn.gather_gemm(xd, wd, nd, cout, bias=bd if direct else None, stats=True)
            want = ref + (bias.double() if direct else 0.0)
            err = float((y.cpu().double() - want).abs().max()) / scale
            assert err < tol, (zs, err, tol)
            if part is not None:
                s = part.sum(0).cpu()
                assert torch.allclose(s[0], want.sum(0), rtol=1e-4, atol=1e-3 * scale), zs
                assert torch.allclose(s[1], (want * want).sum(0), rtol=1e-4, atol=1e-3 * scale * scale), zs
            y2 = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=transposed, flip_k=transposed, bias=bd if direct else None)
            assert torch.equal(y, y2), zs
            # and the output-stationary kernel it replaces agrees to rounding
            L.mink_conv_set_stagger(1 << 30)
            y3 = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=transposed, flip_k=transposed, bias=bd if direct else None)
            L.mink_conv_set_stagger(0)
            assert float((y3 - y).abs().max()) / scale < tol + 2e-6, zs  # (each within its own bound of float64)
    finally:
        Fn._FORCE_KSPLIT = 0
        L.mink_conv_set_stagger(0)


@pytest.mark.parametrize("n_out,K,cin,cout,pure", [(1, 27, 64, 64, True), (130, 27, 64, 64, True), (1000, 27, 128, 64, True),
                                                   (517, 27, 256, 128, False), (4097, 27, 64, 64, False), (300, 9, 512, 64, True)])
@pytest.mark.parametrize("transposed", [False, True])
def test_row_compacted_kernel_over_permuted_rows_against_float64(n_out, K, cin, cout, pure, transposed):
    """compact_gemm_kernel<.., PERM> (fp32 data gradient of a strided convolution: rows through a class permutation, ALL
    offsets per tile, split over 32-channel chunks) on synthetic tables: a permutation with -1 padding between its
    segments; `pure` segments whose rows share at most eight live offsets (the parity classes) and mixed tiles with
    more live offsets than the rulebook's nine slots (several rounds); every channel split; both weight layouts --
    against a float64 gather + matmul, bitwise against itself, and against the dense kernel it replaces."""
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out * 7 + K + cin + int(pure))
    n_in = max(4, n_out // 3 + 5)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    cls = torch.randint(0, 8, (n_out,), generator=g)
    if pure:  # class c keeps a fixed set of <= 8 offsets, each present with probability 0.7
        keep = torch.zeros(8, K, dtype=torch.bool)
        for c in range(8):
            keep[c, torch.randperm(K, generator=g)[: 1 + c]] = True
        nbr[~keep[cls]] = -1
        nbr[torch.rand(n_out, K, generator=g) < 0.3] = -1
    else:
        nbr[torch.rand(n_out, K, generator=g) < 0.5] = -1
    # permutation: rows grouped by class, every segment padded with -1 to a multiple of 128 (mink_class_partition's form)
    segs = []
    for c in range(8):
        rows = torch.nonzero(cls == c).flatten().to(torch.int32)
        pad = (-len(rows)) % 128
        segs += [rows, torch.full((pad,), -1, dtype=torch.int32)]
    perm = torch.cat(segs)
    x = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) * 0.1
    wk = w.transpose(1, 2).contiguous() if transposed else w
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        sel = nbr[:, k] >= 0
        ref[sel] += x[nbr[sel, k].long()].double() @ w[k].double()
    xd, wd, nd, pd = x.to(dev), wk.to(dev), nbr.to(dev), perm.to(dev)
    scale = float(ref.abs().max()) + 1e-30
    tol = _chain_tolerance(xd, w.to(dev), nd, ref, scale)
    L = lib()
    ncc = cin // 32
    assert 1 <= L.mink_conv_plan(perm.numel(), K, cin, cout, 1) <= ncc, "the planner must hand this shape to the permuted compacted kernel"
    try:
        for zs in sorted({1, 2, ncc // 2 or 1, ncc, min(K, ncc + 3)}):
            Fn._FORCE_KSPLIT = zs
            y = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=transposed, row_perm=pd)
            err = float((y.cpu().double() - ref).abs().max()) / scale
            assert err < tol, (zs, err, tol)
            assert torch.equal(y, Fn.gather_gemm(xd, wd, nd, cout, w_transposed=transposed, row_perm=pd)), zs
            L.mink_conv_set_stagger(-(1 << 31))  # bit 31: the dense class-permuted kernel
            Fn._FORCE_KSPLIT = 1
            y3 = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=transposed, row_perm=pd)
            L.mink_conv_set_stagger(0)
            assert float((y3 - y).abs().max()) / scale < 2 * tol, zs  # (two fp32 summation orders, each within tol of float64)
    finally:
        Fn._FORCE_KSPLIT = 0
        L.mink_conv_set_stagger(0)


@pytest.mark.parametrize("n_out,K,cin,cout,pure", [(130, 27, 64, 64, True), (1000, 27, 128, 64, True), (517, 27, 256, 128, False), (300, 9, 512, 64, True)])
def test_permuted_rows_kernel_on_the_bf16_matrix_cores(n_out, K, cin, cout, pure):
    """--math bf16: the class-permuted data gradient on compact_gemm_kernel<.., MATH = 1> (one v_mfma_f32_16x16x32_bf16 per block and
    item, operands rounded to bf16 in registers) against a float64 sum of the SAME operands rounded to bf16, bitwise against itself,
    and against the dense bf16 kernel it replaces (set_stagger bit 27)."""
    from nerf_downstream_amd import minkowski as ME
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out * 7 + K + cin + int(pure))
    n_in = max(4, n_out // 3 + 5)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    cls = torch.randint(0, 8, (n_out,), generator=g)
    if pure:
        keep = torch.zeros(8, K, dtype=torch.bool)
        for c in range(8):
            keep[c, torch.randperm(K, generator=g)[: 1 + c]] = True
        nbr[~keep[cls]] = -1
        nbr[torch.rand(n_out, K, generator=g) < 0.3] = -1
    else:
        nbr[torch.rand(n_out, K, generator=g) < 0.5] = -1
    segs = []
    for c in range(8):
        rows = torch.nonzero(cls == c).flatten().to(torch.int32)
        segs += [rows, torch.full(((-len(rows)) % 128,), -1, dtype=torch.int32)]
    perm = torch.cat(segs)
    x = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) * 0.1
    wk = w.transpose(1, 2).contiguous()  # the data-gradient form reads W[k] as [cout][cin]
    xb, wb = x.bfloat16().double(), w.bfloat16().double()
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        sel = nbr[:, k] >= 0
        ref[sel] += xb[nbr[sel, k].long()] @ wb[k]
    xd, wd, nd, pd = x.to(dev), wk.to(dev), nbr.to(dev), perm.to(dev)
    scale = float(ref.abs().max()) + 1e-30
    L = lib()
    old = ME.set_conv_math("bf16")
    try:
        for zs in (1, 2):
            Fn._FORCE_KSPLIT = zs
            y = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=True, row_perm=pd)
            err = float((y.cpu().double() - ref).abs().max()) / scale
            assert err < 1e-5, (zs, err)  # exact products of bf16 operands, fp32 accumulation
            assert torch.equal(y, Fn.gather_gemm(xd, wd, nd, cout, w_transposed=True, row_perm=pd)), zs
        L.mink_conv_set_stagger(1 << 27)  # the dense bf16 kernel
        Fn._FORCE_KSPLIT = 1
        Fn._PLAN_CACHE.clear()
        y3 = Fn.gather_gemm(xd, wd, nd, cout, w_transposed=True, row_perm=pd)
        assert float((y3 - y).abs().max()) / scale < 2e-5
        assert not torch.equal(y3, y)  # (another kernel, another summation order: the switch really switches)
    finally:
        Fn._FORCE_KSPLIT = 0
        L.mink_conv_set_stagger(0)
        Fn._PLAN_CACHE.clear()
        ME.set_conv_math(old)


@pytest.mark.parametrize("n_out,cin,cout", [(36754, 64, 64), (8355, 128, 128), (20000, 64, 128), (16000, 128, 64)])
def test_stream_k_launch_of_the_mid_layer_kernel(n_out, cin, cout):
    """Round 6's second structural variant (compact_gemm_kernel<.., SK>, mink_conv_set_pipeline bit 3): ONE resident round of 1,024 workers,
    each taking an equal run of the launch's (tile, offset) pairs in tile-major order -- a run crosses tile boundaries, so a worker
    processes up to three segments (tile, offset range), writes each into the slab (worker - first worker of the tile), and the worker that
    finishes a tile zeroes the slabs nobody wrote.  Checked here at the row counts where the plan takes it (a run of >= 5 offsets): against
    a float64 sum of the same products (the bound of the grid form: one fp32 chain per output element and slab), against the grid form
    (another partition of a tile's offsets into slabs: not bitwise), repeatable bit for bit, forward and same-map data gradient, on
    a ragged table with an empty offset and rows without neighbours; and that the plan really switched (slab counts may differ).
    It measured within 2 % of the grid form (profiles/r06_stream_k.txt) and stays off."""
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out + cin)
    K = 27
    n_in = n_out
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) < 0.48] = -1
    nbr[:, 13] = -1
    nbr[n_out // 5 : n_out // 5 + 70] = -1  # a whole tile and more without neighbours
    x = torch.randn(n_in, cin, generator=g).to(dev)
    w = (torch.randn(K, cin, cout, generator=g) * 0.1).to(dev)
    gy = torch.randn(n_out, cout, generator=g).to(dev)
    nd = nbr.to(dev)
    L = lib()
    cases = {"fwd": lambda: Fn.gather_gemm(x, w, nd, cout), "dgrad": lambda: Fn.gather_gemm(gy, w, nd, cin, w_transposed=True, flip_k=True)}
    sel = torch.arange(0, n_out, max(1, n_out // 257))  # float64 on a sample of rows (every 64-row tile phase, both ends)
    sel = torch.cat([sel, torch.tensor([0, 63, 64, n_out - 1])]).unique()
    try:
        for name, fn in cases.items():
            out = {}
            for mode in (0, 8):
                L.mink_conv_set_pipeline(mode)
                Fn._PLAN_CACHE.clear()
                out[mode] = fn()
                assert torch.equal(out[mode], fn()), (name, mode)
            src, wk, co = (x, w, cout) if name == "fwd" else (gy, w.transpose(1, 2), cin)
            ref = torch.zeros(len(sel), co, dtype=torch.float64, device=dev)
            for k in range(K):
                kk = k if name == "fwd" else K - 1 - k
                col = nd[sel.to(dev), k].long()
                ok = col >= 0
                ref[ok] += src[col[ok]].double() @ wk[kk].double()
            scale = float(ref.abs().max())
            for mode in (0, 8):
                err = float((out[mode][sel.to(dev)].double() - ref).abs().max()) / scale
                assert err < 3e-6, (name, mode, err)
            assert float((out[0] - out[8]).abs().max()) / scale < 3e-6, name
            assert not torch.equal(out[0], out[8]) or True  # (equal only by accident: another summation order)
        L.mink_conv_set_pipeline(8)
        Fn._PLAN_CACHE.clear()
        ks_sk = Fn._plan_ksplit(L, n_out, K, cin, cout, 0)
        assert 2 <= ks_sk <= 8, ks_sk  # ceil(27 / run) + 1 with a run of >= 5 offsets
    finally:
        L.mink_conv_set_pipeline(0)
        Fn._PLAN_CACHE.clear()


@pytest.mark.parametrize("n_out,K,cin,cout,perm_rows", [(130, 27, 64, 64, False), (1000, 27, 128, 128, False), (517, 27, 256, 64, False),
                                                         (64, 27, 64, 64, False), (700, 27, 128, 64, True), (300, 9, 512, 64, True),
                                                         (2100, 27, 64, 64, True)])
def test_three_stage_pipeline_of_the_mid_layer_kernel_is_bitwise_the_two_stage_one(n_out, K, cin, cout, perm_rows):
    """Round 6's structural variant of compact_gemm_kernel (csrc/conv.hip, template parameter P3; mink_conv_set_pipeline): three LDS
    stages of the gathered-row tile, the MFMA operands of an item read one step ahead into a second register set, gather stage four
    and weight stage three items ahead on two iterators, three workgroups per CU.  Same products added in the same order: forward,
    same-map data gradient (weights read transposed in place, offsets flipped) and the class-permuted strided data gradient must
    equal the two-stage form BIT FOR BIT, un-split and split, on ragged tables (rows without neighbours, offsets without rows,
    a tile of exactly 64 rows, item counts that are not multiples of the six unrolled step bodies).  It measured 4-17 % slower
    (profiles/r06_three_stage_negative.txt) and stays off; this test is what keeps the measurement honest."""
    from nerf_downstream_amd._lib import lib
    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(n_out * 3 + K + cin + cout)
    n_in = max(4, n_out // 2 + 3)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) < 0.45] = -1
    nbr[:, K // 2 + 1] = -1  # an offset nobody has
    nbr[n_out // 3] = -1     # a row without neighbours
    x = (torch.randn(n_in, cin, generator=g)).to(dev)
    w = (torch.randn(K, cin, cout, generator=g) * 0.1).to(dev)
    nd = nbr.to(dev)
    cases = []
    if perm_rows:
        cls = torch.randint(0, 8, (n_out,), generator=g)
        segs = []
        for c in range(8):
            rows = torch.nonzero(cls == c).flatten().to(torch.int32)
            segs += [rows, torch.full(((-len(rows)) % 128,), -1, dtype=torch.int32)]
        pd = torch.cat(segs).to(dev)
        wk = w.transpose(1, 2).contiguous()
        cases.append(("perm", lambda: Fn.gather_gemm(x, wk, nd, cout, w_transposed=True, row_perm=pd)))
    else:
        cases.append(("fwd", lambda: Fn.gather_gemm(x, w, nd, cout)))
        gy = torch.randn(n_out, cout, generator=g).to(dev)
        nd2 = torch.randint(0, n_out, (n_out, K), generator=g, dtype=torch.int32)
        nd2[torch.rand(n_out, K, generator=g) < 0.5] = -1
        nd2 = nd2.to(dev)
        cases.append(("dgrad", lambda: Fn.gather_gemm(gy, w, nd2, cin, w_transposed=True, flip_k=True)))
    L = lib()
    try:
        for zs in (0, 1, 3):
            Fn._FORCE_KSPLIT = zs
            for name, fn in cases:
                L.mink_conv_set_pipeline(0)
                a = fn()
                L.mink_conv_set_pipeline(3)
                b = fn()
                assert torch.equal(a, b), (name, zs, float((a - b).abs().max()))
                assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    finally:
        Fn._FORCE_KSPLIT = 0
        L.mink_conv_set_pipeline(0)


@pytest.mark.parametrize("B,C,ncls,rows", [(16, 512, 51, 33), (3, 2048, 40, 5), (5, 64, 7, 1), (2, 96, 130, 40)])
def test_classifier_head_matches_torch(B, C, ncls, rows):
    """mink_head_forward/backward (global average pooling + the kernel-volume-1 `final` convolution with bias, reference
    resnet.py:175-177) and mink_softmax_ce_* (F.cross_entropy, classification_training.py:33) against float64 torch,
    including an EMPTY batch element (zero rows -> pooled 0, logits = bias) and ragged row counts."""
    import torch.nn.functional as F

    from nerf_downstream_amd.minkowski import functional as Fn

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(B * 1000 + C + ncls)
    counts = torch.randint(1, 2 * rows + 1, (B,), generator=g)
    if B > 2:
        counts[1] = 0
    boff = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).to(torch.int32)
    n = int(boff[-1])
    x = torch.randn(n, C, generator=g)
    w = torch.randn(C, ncls, generator=g) * 0.05
    bias = torch.randn(1, ncls, generator=g) * 0.1
    labels = torch.randint(0, ncls, (B,), generator=g)
    xd, wd, bd = (t.to(dev).requires_grad_() for t in (x, w, bias))
    logits = Fn.global_avg_linear(xd, boff.to(dev), wd, bd)
    loss = Fn.cross_entropy(logits, labels.to(dev))
    loss.backward()
    x64, w64, b64 = (t.double().requires_grad_() for t in (x, w, bias))
    pooled = torch.stack([x64[boff[b]:boff[b + 1]].mean(0) if counts[b] > 0 else torch.zeros(C, dtype=torch.float64) for b in range(B)])
    logits64 = pooled @ w64 + b64
    loss64 = F.cross_entropy(logits64, labels)
    loss64.backward()
    assert float((logits.detach().cpu().double() - logits64.detach()).abs().max()) < 1e-5
    assert abs(float(loss.detach()) - float(loss64.detach())) < 1e-5
    for got, want, name in ((xd.grad, x64.grad, "dx"), (wd.grad, w64.grad, "dw"), (bd.grad, b64.grad, "db")):
        err = float((got.cpu().double() - want).norm() / want.norm().clamp_min(1e-30))
        assert err < 1e-5, (name, err)
    # an upstream factor on the loss, and bitwise repeatability
    xd.grad = None
    logits2 = Fn.global_avg_linear(xd, boff.to(dev), wd, bd)
    assert torch.equal(logits2, logits)
    (3.0 * Fn.cross_entropy(logits2, labels.to(dev))).backward()
    assert float((xd.grad.cpu().double() - 3.0 * x64.grad).norm() / x64.grad.norm()) < 1e-5
    # a label outside [0, classes) must not pass silently
    bad = labels.clone()
    bad[0] = ncls
    assert torch.isnan(Fn.cross_entropy(logits.detach(), bad.to(dev)))


def test_shortcut_data_gradient_pieces():
    """mink_dense_xwt (y = x @ W^T on the fp32 matrix cores, ragged shapes) and mink_rows_scatter_add (distinct
    destination rows, -1 entries skipped) against float64 torch; together they must equal, BIT FOR BIT, what the
    gather-GEMM's accumulate form computed for the 1x1x1 strided shortcut (same accumulation chain per element)."""
    from nerf_downstream_amd._lib import check, lib

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for n, Kd, N in ((1000, 128, 64), (37, 64, 64), (513, 512, 256), (70, 20, 12)):
        x = torch.randn(n, Kd, generator=g)
        w = torch.randn(N, Kd, generator=g) * 0.1
        y = torch.empty(n, N, device=dev)
        xd, wd = x.to(dev), w.to(dev)
        check(lib().mink_dense_xwt(xd.data_ptr(), wd.data_ptr(), n, Kd, N, y.data_ptr(), None))
        ref = x.double() @ w.double().t()
        assert float((y.cpu().double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max())), (n, Kd, N)
        # scatter: every other source row goes to a distinct destination row, the rest nowhere
        n_dst = 3 * n + 5
        idx = torch.full((n,), -1, dtype=torch.int32)
        sel = torch.randperm(n, generator=g)[: n // 2]
        idx[sel] = torch.randperm(n_dst, generator=g)[: sel.numel()].to(torch.int32)
        if N % 4 == 0:
            dst0 = torch.randn(n_dst, N, generator=g)
            dst, idxd = dst0.to(dev), idx.to(dev)
            check(lib().mink_rows_scatter_add(y.data_ptr(), idxd.data_ptr(), n, N, dst.data_ptr(), None))
            want = dst0.clone()
            want[idx[sel].long()] += y.cpu()[sel]
            assert torch.equal(dst.cpu(), want)
