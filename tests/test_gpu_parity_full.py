"""-m gpu: the parity cases the hot-path claims rest on, at full size and for every configuration BASELINE.json names:

* whole-model Mink-ResNet14 under set_conv_math("bf16x3") / ("bf16") against the fp32 CPU oracle;
* the BASELINE batch itself -- 16 scenes x 128^3, ~825 k voxels -- through the whole network (logits within the
  north_star 1e-3) with the tensor-stride 1 / 2 coordinate maps and kernel maps bit-exact against oracle/mink_maps.c;
* SURVEY 8d's fixed split (512 train / 128 val, 51 classes) on a task that does NOT saturate: validation top-1 of a
  HIP-trained and an oracle-trained network (same init, data order, recipe) within 0.1 points.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import batch_scenes, host_threads, trunk_node

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "nerf_downstream_amd", "co3d_3d", "configs")


def _pair(name, cin, ncls, seed=0):
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    torch.manual_seed(seed)
    ref = get_model(name, cin, ncls, ME=OME)
    hip = get_model(name, cin, ncls).cuda()
    hip.load_state_dict(ref.state_dict())
    return hip, ref


@pytest.mark.parametrize("math,logit_tol,grad_cos", [("bf16x3", 1e-3, 0.9999), ("bf16", 2.5e-2, 0.97)])
def test_whole_model_reduced_precision_matrix_math(oracle_maps, math, logit_tol, grad_cos):
    """BASELINE config "Mink-ResNet14 bf16 mixed precision (MFMA bf16 on rulebook GEMM)": the whole network, forward and
    backward, with the convolution GEMMs on the bf16 matrix cores (fp32 accumulate, fp32 tensors in HBM).

    bf16x3 (x = hi + lo, three products) carries ~2^-17 relative error per product: it has to meet the north_star fp32
    tolerance, 1e-3 on the logits.  Plain bf16 rounds both GEMM operands to 8 significant bits (2^-9 relative each):
    over ten convolution layers with batch norm re-normalising in between, errors add like a random walk to ~1e-2 of
    the logit scale (measured 7.5e-3 on logits of scale 1.06; gradient cosine 0.9896): 2.5e-2 absolute / cosine 0.97
    is 3x that and still far from the O(1) error / cosine ~0 of a wrong kernel.  Gradients: cosine against the fp32
    oracle's gradient over all parameters."""
    from nerf_downstream_amd.minkowski import functional as Fn

    hip, ref = _pair("ResNet14", 28, 51)
    coords, feats = batch_scenes([31, 32, 33, 34, 35, 36], grid=48, cin=28)
    labels = (torch.arange(6) * 17 + 5) % 51
    old = Fn.set_conv_math(math)
    try:
        out = hip(hip.process_input({"coordinates": coords.cuda(), "features": feats.cuda()}))
        F.cross_entropy(out, labels.cuda()).backward()
        torch.cuda.synchronize()
    finally:
        Fn.set_conv_math(old)
    oout = ref(ref.process_input({"coordinates": coords, "features": feats}))
    F.cross_entropy(oout, labels).backward()
    err = float((out.detach().cpu() - oout.detach()).abs().max())
    scale = float(oout.detach().abs().max())
    print(f"[{math}] max |logit error| {err:.3e} (logit scale {scale:.2f})")
    assert err < logit_tol, (math, err)
    hp, rp = dict(hip.named_parameters()), dict(ref.named_parameters())
    g = torch.cat([hp[k].grad.cpu().double().flatten() for k in hp])
    og = torch.cat([rp[k].grad.double().flatten() for k in hp])
    cos = float(torch.dot(g, og) / (g.norm() * og.norm()))
    print(f"[{math}] gradient cosine {cos:.6f}")
    assert cos > grad_cos, (math, cos)


def _baseline_batch(batch=16, grid=128, cin=28):
    from nerf_downstream_amd.co3d_3d.src.data.synthetic import SparseVoxelDataset
    from nerf_downstream_amd.co3d_3d.src.data.utils import collate_mink

    ds = SparseVoxelDataset(phase="train", num_samples=1 << 20, num_classes=51, grid=grid, features=["density", "sh"])
    return collate_mink([ds[i] for i in range(batch)])  # exactly bench.py's first batch


@pytest.mark.long
@pytest.mark.timeout(30)
def test_baseline_batch_forward_and_maps_match_oracle(oracle_maps):
    """BASELINE config #2's own batch (B=16, 128^3, ~825 k voxels x 28 features)."""
    from nerf_downstream_amd import minkowski as ME

    torch.set_num_threads(host_threads(16))
    oracle_maps.set_threads(host_threads(16))
    b = _baseline_batch()
    coords, feats = b["coordinates"], b["features"]
    assert coords.shape[0] > 800_000
    hip, ref = _pair("ResNet14", 28, 51, seed=777)
    with torch.no_grad():
        field = hip.process_input({"coordinates": coords.cuda(), "features": feats.cuda()})
        out = hip(field)
        oout = ref(ref.process_input({"coordinates": coords, "features": feats}))
    err = float((out.cpu() - oout).abs().max())
    print(f"B=16 x 128^3: {coords.shape[0]} voxels, max |logit error| vs the CPU oracle {err:.3e}")
    assert out.shape == (16, 51) and err < 1e-3, err
    # integer maps, bit for bit (first-occurrence row order on both sides)
    m = field.coordinate_manager
    q = oracle_maps.quantize(coords.numpy())
    ui, inv = oracle_maps.unique(q)
    c1 = q[ui]
    assert np.array_equal(m.levels[1].coords.cpu().numpy(), c1)
    c2, i2o = oracle_maps.stride_map(c1, 2)
    k1, k2 = ME.CoordinateMapKey(1), ME.CoordinateMapKey(2)
    assert np.array_equal(m.levels[2].coords.cpu().numpy(), c2)
    assert np.array_equal(m.stride_map(k1, k2).cpu().numpy(), i2o)
    c4, _ = oracle_maps.stride_map(c2, 4)
    k4 = ME.CoordinateMapKey(4)
    assert np.array_equal(m.levels[4].coords.cpu().numpy(), c4)
    for (kin, cin_), (kout, cout_), ks in [((k1, c1), (k1, c1), 3), ((k1, c1), (k2, c2), 2), ((k2, c2), (k4, c4), 3),
                                         ((k2, c2), (k4, c4), 1), ((k4, c4), (k4, c4), 3)]:
        want = oracle_maps.kernel_map_table(cin_, cout_, oracle_maps.kernel_offsets(ks, kin.ts))
        got, _ = m.kernel_table(kin, kout, ks, 1)
        assert np.array_equal(got.cpu().numpy(), want), (kin.ts, kout.ts, ks)


def _rel_err(g, g64):
    """Plain relative L2 error of one tensor -- every element counts (until round 4 the worst 1 % of output channels was
    trimmed here; with the ReLU branches of the HIP run imposed on the yardstick there is nothing for a trim to excuse)."""
    return float((g - g64).norm() / g64.norm().clamp_min(1e-300))


def _relu_masks_of_hip_run(out):
    """(z > 0) of every ReLU of the residual stages in forward order, as the native trunk decided it: the trunk keeps
    h1 = relu(norm1(conv1 x)) and out = relu(norm2(conv2 h1) + shortcut) of each stage for its backward pass
    (minkowski/trunk.py: arena = [y1 | h1 | y2 | out | ...]); a ReLU output is positive exactly where its branch was."""
    from helpers import trunk_node

    node = trunk_node(out)
    assert node is not None, "no native-trunk node in the autograd graph"
    masks = []
    for st, sv in zip(node.plan.stages, node.saved[1:]):
        arena, n_out = sv[0], sv[7]
        cnt = n_out * st.C
        masks.append((arena[cnt : 2 * cnt].view(n_out, st.C) > 0).cpu())
        masks.append((arena[3 * cnt : 4 * cnt].view(n_out, st.C) > 0).cpu())
    return masks


def _stem_masks_of_hip_run(out, model):
    """The stem ReLU's branch decisions as the HIP backward kernels take them.  The trunk keeps only the POOLED sum of the
    stem's ReLU output, so its backward kernels recompute z = gamma * xhat + beta from the kept convolution output y and
    the batch's (mean, 1 / std) -- and the two kernels that do so round xhat differently (both are valid fp32):
      * gamma / beta gradients (csrc/elementwise.hip, colreduce / bn_relu_pool_bwd_kernel; also the forward's own
        decision):                      xhat = fl(fl(y - mean) * invstd),        z = fma(xhat, gamma, beta)
      * the weight gradient (csrc/conv.hip, wgrad_stream_kernel<.., FUSE>): xhat = fma(y, invstd, fl(-mean * invstd)), z = fma(xhat, gamma, beta)
    Both are reproduced here in exact fp32 arithmetic on the CPU (an fma through float64: the product of two floats is exact
    there).  -> (mask for bn1.*, mask for conv1.kernel), each [n0, C0] bool."""
    node = trunk_node(out)
    x, w0, arena0, nbr0, nbr_pool, i2o, pad, b16, _ = node.saved[0]
    n0, n1, C0 = x.shape[0], nbr_pool.shape[0], w0.shape[-1]
    a = arena0.detach().cpu()
    ny = a.numel() - n1 * C0 - 2 * C0  # floats of the y (+ bf16 input copy) region (minkowski/trunk.py)
    if b16:  # bf16 storage: y is kept as bf16 [n0][C0] at the head of the arena; the kernels widen it and go on in fp32
        y = a[: n0 * C0 // 2].view(torch.bfloat16).view(n0, C0).float()
    else:
        assert ny == n0 * C0
        y = a[: n0 * C0].view(n0, C0)
    mean, invstd = a[ny + n1 * C0 : ny + n1 * C0 + C0], a[ny + n1 * C0 + C0 : ny + n1 * C0 + 2 * C0]
    ga, be = model.bn1.bn.weight.detach().cpu(), model.bn1.bn.bias.detach().cpu()

    def fma(p, q, r):
        return (p.double() * q.double() + r.double()).float()

    xh_bn = (y - mean) * invstd
    xh_w = fma(y, invstd, -mean * invstd)
    return fma(xh_bn, ga, be) > 0, fma(xh_w, ga, be) > 0


def _bf16_operands(storage):
    """oracle.me_cpu.OPERAND_HOOK for BASELINE config #4: every convolution GEMM operand rounded to bf16 (round to nearest even, what
    `(__bf16)v` and the MFMA packers of csrc/conv.hip / stem16.hip do) exactly where the HIP path rounds it -- forward: rows and
    weights; data gradient: dY and weights, EXCEPT the 1x1x1 strided shortcut, whose data gradient is the exact-fp32 dense GEMM
    mink_dense_xwt (csrc/trunk.hip); weight gradient: rows and dY.  `storage`: the stem's convolution output is also STORED as
    bf16 (set_conv_storage("bf16")): its forward value is rounded, the gradient passes unchanged."""

    def rnd(t):
        return t.to(torch.bfloat16).to(t.dtype)

    def hook(t, role, shape):
        K, cin = shape[0], shape[1]
        if role == "fwd_y":
            return rnd(t) if storage and K == 27 and cin <= 32 else t
        if role.startswith("dgrad") and K == 1:
            return t
        return rnd(t)

    return hook


def _float64_grads_under_hip_branches(name, state, masks, coords, feats, labels, dtype=torch.float64, stem_mask=None,
                                      operand_hook=None):
    """One forward + backward of the CPU oracle in `dtype` at the weights `state`, with the ReLU branch decisions of the HIP
    run (`masks`, forward order, every ReLU behind the stem's; `stem_mask`: the stem's, or None = it runs naturally)
    imposed through oracle.me_cpu.RELU_HOOK.  -> (logits, {name: gradient}, [(relu index, elements whose
    branch differs from this run's own, largest |z| / rms(z) among them, where)])."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    m = get_model(name, 28, 51, ME=OME)
    if dtype == torch.float64:
        m = m.double()
    m.load_state_dict({k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in state.items()})
    m.train()
    calls, flips = [0], []

    def hook(z):
        k = calls[0]
        calls[0] += 1
        if k == 0 and stem_mask is None:
            return torch.relu(z)
        mk = stem_mask if k == 0 else masks[k - 1]
        assert mk.shape == z.shape, (k, mk.shape, z.shape)
        diff = (z > 0) != mk
        nf = int(diff.sum())
        zd = z.detach()
        where = [tuple(int(v) for v in ix) for ix in diff.nonzero()[:4]] if nf else []
        flips.append((k, nf, float(zd[diff].abs().max() / zd.std()) if nf else 0.0, where))
        return z * mk.to(z.dtype)

    OME.RELU_HOOK, OME.OPERAND_HOOK = hook, operand_hook
    try:
        o = m(m.process_input({"coordinates": coords, "features": feats.to(dtype)}))
        loss = F.cross_entropy(o, labels)
        loss.backward()
    finally:
        OME.RELU_HOOK = OME.OPERAND_HOOK = None
    assert calls[0] == len(masks) + 1, (calls[0], len(masks) + 1)
    return o.detach(), float(loss), {k: p.grad for k, p in m.named_parameters()}, flips


def _assert_gradients_match_float64(tag, name, hip_grads, state, masks, coords, feats, labels, stem_masks=None, verbose=False,
                                    operand_hook=None, grad_bound=2e-5, flip_bound=1e-4, cos_bound=0.999999):
    """The rigorous form of "the HIP gradient is the reference's gradient" (no trim, no flip allowance):
      (1) wherever a ReLU branch of the HIP run differs from the float64 run's own, the float64 pre-activation is zero to
          fp32 rounding: |z| <= 1e-4 of the tensor's standard deviation -- the flips are legitimate, and each is NAMED in
          the output (layer, element);
      (2) with the HIP run's branches imposed on the float64 run, every parameter gradient matches it to rounding: plain
          relative L2 <= 2e-5 per tensor (2e-4 until the stem's ReLU was imposed too), the stem's three included: `stem_masks` = _stem_masks_of_hip_run -- bn1.* against
          the float64 run under the decisions of the kernel that computes them, conv1.kernel under those of the
          weight-gradient kernel (a second float64 run, only when the two masks differ somewhere)."""
    sm_bn, sm_w = stem_masks if stem_masks is not None else (None, None)
    out64, loss64, g64, flips = _float64_grads_under_hip_branches(name, state, masks, coords, feats, labels, stem_mask=sm_bn,
                                                                  operand_hook=operand_hook)
    if sm_w is not None:
        ndiff = int((sm_bn != sm_w).sum())
        print(f"[{tag}] stem ReLU: the weight-gradient kernel and the norm-gradient kernels decide {ndiff} element(s) of {sm_bn.numel()} differently")
        if ndiff:
            _, _, g64w, flips_w = _float64_grads_under_hip_branches(name, state, masks, coords, feats, labels, stem_mask=sm_w,
                                                                    operand_hook=operand_hook)
            g64 = dict(g64, **{"conv1.kernel": g64w["conv1.kernel"]})
            flips = flips + [f for f in flips_w if f[0] == 0]
    nflip = sum(f[1] for f in flips)
    zmax = max(f[2] for f in flips)
    named = "; ".join(f"relu {f[0]}: {f[1]} element(s) e.g. {f[3][:2]} |z|/sd {f[2]:.1e}" for f in flips if f[1])
    print(f"[{tag}] ReLU branches of the HIP run that differ from the float64 run's own: {nflip} element(s) in "
          f"{sum(1 for f in flips if f[1])} of {len(masks) + (stem_masks is not None)} layers, largest |z|/sd(z) there {zmax:.2e}" + (f" -- {named}" if named else ""))
    assert zmax <= flip_bound, flips
    worst, bad, table = (None, 0.0), [], []
    for k, g in hip_grads.items():
        e = _rel_err(g.detach().cpu().double(), g64[k])
        # (measured with every branch imposed, rounds 4-5: <= 4e-6 on every tensor of ResNet14 at 16 scenes, ResNet34 at 4 and
        #  the four training probes -- 2e-5 leaves 5x for another summation order and still sits 100x under a dropped row)
        bound = grad_bound if stem_masks is not None else (max(grad_bound, 2e-4) if not (k.startswith("conv1") or k.startswith("bn1")) else 1e-3)
        table.append(f"{k:34s} hip {e:.2e}  bound {bound:.0e}")
        if e > worst[1]:
            worst = (k, e)
        if not e <= bound:
            bad.append((k, e))
    if bad or verbose or os.environ.get("MINK_TEST_VERBOSE"):
        print(f"[{tag}] per-tensor gradient error vs float64 (HIP's ReLU branches imposed):\n  " + "\n  ".join(table))
    assert not bad, (tag, bad)
    flat_g = torch.cat([hip_grads[k].detach().cpu().double().flatten() for k in hip_grads])
    flat_o = torch.cat([g64[k].flatten() for k in hip_grads])
    cos = float(torch.dot(flat_g, flat_o) / (flat_g.norm() * flat_o.norm()))
    tot = float((flat_g - flat_o).norm() / flat_o.norm())
    print(f"[{tag}] worst per-tensor gradient error vs float64: {worst[0]} {worst[1]:.2e}; all parameters: relative L2 {tot:.2e}, cosine {cos:.10f}")
    assert cos > cos_bound, cos
    return out64, loss64, nflip, worst, tot


def _bench_like_step(hip, batch, labels, passes=3):
    """The training step exactly as bench.py queues it: flat gradient buffer as the gradient sink of the backward kernels,
    coordinate pyramid launched ahead (defer=True) and finished after the previous pass, map plan replayed on the prepare
    stream (so the shortcut branch forks onto its own stream and the weight gradients run on theirs).  The first pass
    records the plan, the second runs while the third batch's maps are replayed from it (the pyramid of pass p+1 is
    launched before forward p, as in bench.py, so it takes two passes to reach the steady state); gradients and logits
    are those of the LAST pass (train-mode batch norm: neither depends on the
    running statistics that move between passes)."""
    from nerf_downstream_amd.parallel import BucketedGradAllReduce

    reducer = BucketedGradAllReduce(hip)
    tf = hip.process_input(batch)
    out = field = None
    for p in range(passes):
        nxt = hip.process_input(batch, defer=True) if p + 1 < passes else None
        reducer.zero_grad()
        field = tf
        out = hip(tf)
        F.cross_entropy(out, labels).backward()
        if nxt is not None:
            tf = hip.finish_input(nxt)
        reducer.finish()
    torch.cuda.synchronize()
    return out, field, reducer


def _check_every_map(oracle_maps, field, coords, plan_ops):
    """Every coordinate level, stride map, neighbour table (+ transposed tables) and class permutation the step built,
    bit for bit against oracle/mink_maps.c (first-occurrence row order on both sides)."""
    m = field.coordinate_manager
    q = oracle_maps.quantize(coords.numpy())
    ui, inv = oracle_maps.unique(q)
    want = {1: q[ui]}
    assert np.array_equal(m.field_inverse.cpu().numpy(), inv)
    ts = 1
    while 2 * ts in m.levels:
        want[2 * ts], i2o = oracle_maps.stride_map(want[ts], 2 * ts)
        assert np.array_equal(m.in2out[(ts, 2 * ts)].cpu().numpy(), i2o), ts
        ts *= 2
    assert sorted(want) == sorted(m.levels), (sorted(want), sorted(m.levels))
    for t, c in want.items():
        assert m.levels[t].n == c.shape[0] and np.array_equal(m.levels[t].coords.cpu().numpy(), c), t
    n_tables = n_perm = n_t = 0
    for key, ent in m.tables.items():
        if key[0] == "perm":
            _, t, pad = key
            assert np.array_equal(ent.cpu().numpy(), oracle_maps.class_partition(want[t], t, pad)), key
            n_perm += 1
        elif key[0] == "ident":
            continue
        else:
            ts_in, ts_out, ks, dil = key
            nbr = oracle_maps.kernel_map_table(want[ts_in], want[ts_out], oracle_maps.kernel_offsets(ks, ts_in, dil))
            assert np.array_equal(ent[0].cpu().numpy(), nbr), key
            n_tables += 1
            if ent[1] is not None:
                assert np.array_equal(ent[1].cpu().numpy(), oracle_maps.transpose_table(nbr, want[ts_in].shape[0])), key
                n_t += 1
    planned = {op[1:5] for op in plan_ops if op[0] == "ktable"}
    assert planned <= set(k for k in m.tables if k[0] not in ("perm", "ident")), "a planned table was not built"
    return len(want), n_tables, n_t, n_perm


@pytest.mark.long
@pytest.mark.timeout(45)
@pytest.mark.parametrize("name,batch,math", [("ResNet14", 16, "fp32"), ("ResNet34", 4, "fp32"), ("ResNet14", 16, "bf16"),
                                             ("ResNet14", 16, "bf16s")])
def test_baseline_batch_backward_and_every_map_match_oracle(oracle_maps, name, batch, math):
    """BASELINE config #2's own batch (B=16 x 128^3, ~825 k voxels) and config #3's per-GPU shape (ResNet34, B=4) through
    forward AND backward at the launch shapes bench.py runs -- the planners pick kernels, split factors and launch orders
    by row count, so small-grid parity says nothing about these launches.  fp32: logits 1e-3 (north_star), every
    parameter gradient against a float64 run of the oracle with the criterion of tests/test_gpu_resnet.py (relative L2
    <= max(1e-3, 8x the oracle's own fp32 error, 3/sqrt(rows x channels) for a ReLU flip)), every map of the plan bit for
    bit.  bf16 (BASELINE config #4 at full size): EVERY parameter gradient against a float64 run of the oracle on operands rounded to bf16
    where the HIP kernels round theirs, under the HIP run's ReLU branches, bound 3 x 2^-8 per tensor (derived where it is asserted, below);
    logits 2.5e-2 against the fp32 oracle (test_whole_model_reduced_precision_matrix_math) and 1.2e-2 against that float64 run; bf16s: the
    same with the input features and the stem output STORED as bf16 (set_conv_storage, csrc/stem16.hip)."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from nerf_downstream_amd.minkowski import functional as Fn
    from oracle import me_cpu as OME

    threads = host_threads(32)
    torch.set_num_threads(threads)
    oracle_maps.set_threads(threads)
    b = _baseline_batch(batch=batch)
    coords, feats, labels = b["coordinates"], b["features"], b["labels"].long()
    hip, ref = _pair(name, 28, 51, seed=777)
    old, old_st = Fn.set_conv_math("bf16" if math == "bf16s" else math), Fn.set_conv_storage("bf16" if math == "bf16s" else "fp32")
    try:
        out, field, reducer = _bench_like_step(hip, {"coordinates": coords.cuda(), "features": feats.cuda()}, labels.cuda())
    finally:
        Fn.set_conv_math(old), Fn.set_conv_storage(old_st)
    assert hip._trunk_plan, "the native trunk (bench.py's path) was not taken"
    if math == "bf16s":
        assert trunk_node(out).saved[0][7] is True, "bf16 storage was not taken"
        assert trunk_node(out).saved[0][8] is not None, "the bf16 copy of the input was not made ahead on the prepare stream"
    assert field.coordinate_manager.prepared, "the second pass must run on maps prepared ahead"
    oout = ref(ref.process_input({"coordinates": coords, "features": feats}))
    err = float((out.detach().cpu() - oout.detach()).abs().max())
    hp, rp = dict(hip.named_parameters()), dict(ref.named_parameters())
    assert hp.keys() == rp.keys()
    print(f"[{name} B={batch} {math}] {coords.shape[0]} voxels, max |logit error| vs the CPU oracle {err:.3e}")
    if math != "fp32":
        # BASELINE config #4, tensor by tensor (round 6; until then ONE cosine over all 14.4 M parameters, which layer 4's 74 % of the
        # bytes dominate).  Yardstick: a float64 run of the oracle whose convolution operands are rounded to bf16 exactly where the
        # HIP kernels round theirs (_bf16_operands), under the HIP run's ReLU branches -- what a bf16-operand / fp32-accumulate
        # network is ASKED to compute.  What the two can agree to: rounding to bf16 is a step function, so an operand element whose
        # two values (fp32 accumulation here, float64 there: 1e-6 apart) straddle a tie goes to different neighbours, 2^-8 apart.
        # With the runs d apart (relative), a fraction d / 2^-8 of the elements does that, which leaves them sqrt(d * 2^-8) apart
        # behind the layer: 1e-6 -> 6e-5 -> 5e-4 -> 1.4e-3 -> 2.4e-3 -> ... -> the fixed point 2^-8 = 3.9e-3, reached within five
        # layers in either direction.  So ~4e-3 per tensor is the floor of ANY whole-network comparison of two correct bf16-operand
        # implementations (measured: 1.4e-3 at the head .. 5.0e-3 at layer 1, profiles/r06_parity_bf16_per_tensor.txt), against
        # ~1.2e-1 per tensor for the comparison with the fp32 network that this replaces (cosine 0.9926) and O(1) for a wrong tensor:
        # bound 1.2e-2 = 3 x 2^-8 for EVERY tensor; the kernels themselves are held to 2e-5 against float64 on their own operands
        # (tests/test_gpu_ops.py, test_gpu_stem16.py).
        assert err < 2.5e-2, err
        masks = _relu_masks_of_hip_run(out)
        out64, _, _, worst, tot = _assert_gradients_match_float64(
            f"{name} B={batch} {math}", name, {k: hp[k].grad for k in hp}, ref.state_dict(), masks, coords, feats, labels,
            stem_masks=_stem_masks_of_hip_run(out, hip), operand_hook=_bf16_operands(storage=math == "bf16s"),
            grad_bound=BF16_GRAD_BOUND, flip_bound=BF16_FLIP_BOUND, cos_bound=0.9999, verbose=True)
        err64 = float((out.detach().cpu().double() - out64).abs().max())
        print(f"[{name} B={batch} {math}] max |logit error| vs float64 on bf16-rounded operands {err64:.3e}")
        assert err64 < 1.2e-2, err64  # (the same floor: logits of scale ~1 at 3 x 2^-8)
        # and the old yardstick, kept as information: cosine against the fp32 oracle's gradient
        F.cross_entropy(oout, labels).backward()
        g = torch.cat([hp[k].grad.cpu().double().flatten() for k in hp])
        og = torch.cat([rp[k].grad.double().flatten() for k in hp])
        cos = float(torch.dot(g, og) / (g.norm() * og.norm()))
        print(f"[{name} B={batch} {math}] gradient cosine vs the fp32 oracle {cos:.6f}")
        assert cos > 0.97, cos
        return
    assert err < 1e-3, err
    # Gradients.  Yardstick: a float64 run of the oracle.  A ReLU whose pre-activation is zero to rounding takes either
    # branch with no effect on the loss but a finite one on the gradient, and at this size such elements exist in every
    # stage (~1e-6 relative differences in z between two fp32 summation orders x 0.5-50 M activations per stage): the
    # oracle's own fp32 run sits 9e-4 from the float64 gradient upstream of one.  So the float64 run is made with the
    # BRANCH DECISIONS OF THE HIP RUN imposed (h > 0 of every ReLU output the trunk kept for its backward pass); see
    # _assert_gradients_match_float64 for the two assertions (legitimate flips; every tensor to rounding, every element
    # counted, no allowance).
    masks = _relu_masks_of_hip_run(out)
    out64, _, _, _, _ = _assert_gradients_match_float64(f"{name} B={batch}", name, {k: hp[k].grad for k in hp}, ref.state_dict(), masks,
                                                        coords, feats, labels, stem_masks=_stem_masks_of_hip_run(out, hip))
    assert float((out.detach().cpu().double() - out64).abs().max()) < 1e-3
    # the gradients sit in the reducer's flat buffer (the bench's layout): every .grad is a view of it
    lo, hi = reducer.flat.data_ptr(), reducer.flat.data_ptr() + 4 * reducer.flat.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in hip.parameters())
    nlev, ntab, ntr, nperm = _check_every_map(oracle_maps, field, coords, hip._coord_plan)
    print(f"[{name} B={batch}] bit-exact: {nlev} coordinate levels, {ntab} neighbour tables ({ntr} transposed), {nperm} class permutations")
    assert nlev == 6 and ntab >= 14 and ntr >= 4 and nperm == 4, (nlev, ntab, ntr, nperm)


BF16_GRAD_BOUND, BF16_FLIP_BOUND = 1.2e-2, 5e-2  # 3 x 2^-8 per tensor; a differing ReLU branch within 5e-2 sd of zero (measured 1.0e-2)


from top1_recipe import N_VAL_STAT, SPLIT, fit as _fit, recipe_hash, stat_predictions, stat_val_batches, val_logits as _val_logits  # noqa: E402

TOP1_FIXTURE = os.path.join(ROOT, "tests", "golden", "top1_oracle_v1.npz")


@pytest.mark.long
@pytest.mark.timeout(40)
def test_fixed_split_top1_matches_oracle(oracle_maps):
    """north_star: "top-1 on a fixed synthetic split matching reference +-0.1 %".  SURVEY 8d's split (512 training / 128
    validation scenes, 51 classes, 300 steps, same seeds; tests/top1_recipe.py) on a task that does NOT saturate: the class
    signal is weakened (class_sep) and a per-scene offset added (scene_sigma), so validation top-1 lands near 78 %.

    What can be asserted depends on what is well posed.  fp32 training of this network is chaotic: the CPU oracle run
    twice with different thread counts (nothing but its own summation order changes) drifts apart from 5e-7 in the loss
    at step 2 to 1e-3 at step 8 and 7e-2 at step 12 (test_reference_training_does_not_reproduce_itself), so NO second
    implementation -- not even the reference against itself -- can reproduce a 300-step trajectory, and the trained top-1
    of two runs differs by a few validation scenes.  Hence three assertions:
      1. every training step is the reference's step: at steps 0 / 100 / 200 / 299 of the HIP run, the oracle evaluated
         at the SAME weights and batch gives the same loss (1e-4), and a float64 run of the oracle with the HIP step's
         ReLU branch decisions imposed gives the same gradient, tensor by tensor (2e-4, stem included), at ALL FOUR probes;
         every branch that differs from the float64 run's own is named and has |z| <= 1e-4 sd (zero to fp32 rounding).
         (Round 4 had loosened this to "1e-2 everywhere, 3 of 4 close" after a red run at step 200; that run's 2.3e-3 is
         what ONE flipped element upstream of ~100 k voxels does to an un-imposed comparison.)
      2. top-1 of a given network is the reference's top-1: the HIP-trained weights evaluated on the whole validation
         split by the HIP path and by the oracle agree within 0.1 points (in fact scene by scene) -- the reference's
         "+-0.1 %" in the only form that is well posed;
      3. the trained accuracy is statistically the reference's: `test_fixed_split_top1_statistics` below."""
    from nerf_downstream_amd.co3d_3d.src.models import get_model
    from oracle import me_cpu as OME

    torch.set_num_threads(host_threads(16))
    oracle_maps.set_threads(host_threads(16))
    dev = torch.device("cuda", 0)
    probes = {}

    def probe(step, model, batch, loss):
        # the oracle (fp32) at the same weights and batch: the loss
        ref = get_model("ResNet14", 28, 51, ME=OME)
        ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
        ref.train()
        cb = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
        with torch.no_grad():
            oloss = F.cross_entropy(ref(ref.process_input(cb)), cb["labels"].long())
        # the gradient: against float64 with THIS step's ReLU branches of the HIP run imposed (the trunk's kept
        # activations are still alive behind loss.grad_fn)
        node = trunk_node(loss)
        assert node is not None, "the probe step did not take the native trunk"
        masks = _relu_masks_of_hip_run(loss)
        _, loss64, nflip, worst, tot = _assert_gradients_match_float64(
            f"probe step {step}", "ResNet14", {k: p.grad for k, p in model.named_parameters()}, model.state_dict(), masks,
            cb["coordinates"], cb["features"], cb["labels"].long(), stem_masks=_stem_masks_of_hip_run(loss, model))
        probes[step] = (abs(float(loss) - float(oloss)), abs(float(loss) - loss64), nflip, worst, tot)

    hip, lh = _fit(None, dev, probe_steps=(0, 100, 200, SPLIT["steps"] - 1), probe=probe)
    print("per-step parity along the HIP trajectory (|loss - oracle fp32|, |loss - float64 under HIP's branches|, flipped ReLU elements, "
          "worst tensor, all-parameter relative L2):", probes)
    assert len(probes) == 4
    # Every probe, no exception: the loss is the oracle's (1e-4), and the gradient is the float64 gradient under the HIP run's
    # own (legitimate, named) branch decisions to 2e-4 per tensor (stem 1e-3) -- asserted inside the probe.
    for step, (dl, dl64, _, _, _) in probes.items():
        assert dl < 1e-4 and dl64 < 1e-4, (step, dl, dl64)
    # 2. evaluation parity of the trained network
    logits_h, labels = _val_logits(hip, dev)
    ref = get_model("ResNet14", 28, 51, ME=OME)
    ref.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
    logits_o, _ = _val_logits(ref, torch.device("cpu"))
    acc_h = 100.0 * float((logits_h.argmax(1) == labels).float().mean())
    acc_o = 100.0 * float((logits_o.argmax(1) == labels).float().mean())
    print(f"HIP-trained weights on the validation split: top-1 HIP {acc_h:.3f} %, oracle {acc_o:.3f} %, "
          f"max |logit difference| {float((logits_h - logits_o).abs().max()):.2e}")
    assert 100.0 / 51 * 5 < acc_o < 95.0, "the task must neither sit at chance nor saturate"
    assert abs(acc_h - acc_o) <= 0.1 and torch.equal(logits_h.argmax(1), logits_o.argmax(1))
    assert float((logits_h - logits_o).abs().max()) < 1e-3


HIP_SEEDS = tuple(range(24))


@pytest.mark.long
@pytest.mark.timeout(45)
def test_fixed_split_top1_statistics():
    """north_star: "top-1 on a fixed synthetic split matching reference +-0.1 %" as the statistical statement it can only
    be (fp32 training of this network is chaotic: test_reference_training_does_not_reproduce_itself).

    Oracle side: tests/golden/top1_oracle_v1.npz -- the CPU oracle trained with seeds 0..15 of the recipe and evaluated BY
    THE ORACLE on the 1,024-scene validation split, generated once in the build container by oracle/make_top1_fixture.py
    (~6 minutes per seed on eight cores: that work does not belong on the GPU box inside the driver's limit; in round 3 it
    was there and the run was killed).  The fixture carries a hash of recipe and data which is recomputed here first.
    HIP side: TWENTY-FOUR seeds trained and evaluated here on the HIP path (seconds each).  That the HIP path and the oracle
    give the same top-1 for the same weights is asserted scene by scene in the test above.

    Asserted: |mean top-1 (HIP runs) - mean top-1 (oracle runs)| <= 2 x the standard error of that difference,
    sqrt(s_h^2 / n_h + s_o^2 / n_o) with s the standard deviation over seeds, each term floored by the binomial resolution
    of the split (p (1 - p) / 1024 per run).  Two, not one: a one-sigma band rejects a third of all pairs of IDENTICAL
    implementations.  The result line is also appended to gpurun_out/top1_statistics.txt on the GPU box (kept as profiles/r04_top1_statistics.txt)."""
    fx = np.load(TOP1_FIXTURE)
    assert str(fx["recipe"]) == recipe_hash(), "tests/golden/top1_oracle_v1.npz was made for another recipe: rerun oracle/make_top1_fixture.py"
    acc_o, labels = fx["top1"].astype(np.float64), fx["labels"].astype(np.int64)
    assert len(acc_o) >= 5 and fx["preds"].shape == (len(acc_o), N_VAL_STAT)
    assert np.allclose(acc_o, 100.0 * (fx["preds"] == labels[None]).mean(1))
    dev = torch.device("cuda", 0)
    val = stat_val_batches()
    assert np.array_equal(torch.cat([y for _, y in val]).numpy(), labels)
    val = [({k: v.to(dev) for k, v in b.items()}, y) for b, y in val]
    acc_h = []
    for sd in HIP_SEEDS:
        hip, _ = _fit(None, dev, seed=sd)
        acc_h.append(100.0 * float((stat_predictions(hip, val, dev) == labels).mean()))
    acc_h = np.array(acc_h)
    diff = float(acc_h.mean() - acc_o.mean())
    p = float(np.concatenate([acc_h, acc_o]).mean()) / 100.0
    floor = 100.0 ** 2 * p * (1.0 - p) / N_VAL_STAT
    se = float(np.sqrt(max(acc_h.var(ddof=1), floor) / len(acc_h) + max(acc_o.var(ddof=1), floor) / len(acc_o)))
    line = (f"top-1 on {N_VAL_STAT} validation scenes (recipe {recipe_hash()}): HIP seeds {list(HIP_SEEDS)} {np.round(acc_h, 2).tolist()} "
            f"(mean {acc_h.mean():.2f}, sd {acc_h.std(ddof=1):.2f}); oracle seeds {fx['seeds'].tolist()} {np.round(acc_o, 2).tolist()} "
            f"(mean {acc_o.mean():.2f}, sd {acc_o.std(ddof=1):.2f}); difference of means {diff:+.2f} points, standard error {se:.2f}")
    print(line)
    for d in (os.path.join(ROOT, "gpurun_out"),):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            with open(os.path.join(d, "top1_statistics.txt"), "a") as f:
                f.write(line + "\n")
    assert 100.0 / 51 * 5 < 100.0 * p < 95.0, "the task must neither sit at chance nor saturate"
    assert se <= 1.0, se
    assert abs(diff) <= 2.0 * se, (diff, se)


@pytest.mark.long
@pytest.mark.timeout(90)
def test_reference_training_does_not_reproduce_itself(oracle_maps, monkeypatch):
    """Why the fixed-split criterion above is not "identical trajectories": the CPU oracle alone, run with two thread
    counts (only its summation order changes), leaves its own trajectory within a dozen steps of the same recipe."""
    from oracle import me_cpu as OME

    monkeypatch.setitem(SPLIT, "grid", 32)
    monkeypatch.setitem(SPLIT, "steps", 30)
    traj = {}
    hi = max(2, host_threads(8))
    for threads in (hi, max(1, hi // 2 - 1) if hi > 2 else 1):
        torch.set_num_threads(threads)
        oracle_maps.set_threads(threads)
        traj[threads] = _fit(OME, torch.device("cpu"))[1]
    torch.set_num_threads(host_threads(16))
    oracle_maps.set_threads(host_threads(16))
    a, b = traj.values()
    d = np.abs(a - b)
    print("oracle vs oracle |loss difference| per step:", np.array2string(d, precision=6))
    assert d[0] < 1e-5 and d.max() > 1e-4
