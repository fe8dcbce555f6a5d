"""bf16-storage stem kernels against the fp32-storage bf16-math kernels on one BASELINE-shape batch (run on the GPU box).
usage: python scripts/stem16_bench.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import make_batches
from nerf_downstream_amd import minkowski as ME
from nerf_downstream_amd._lib import lib, check
from nerf_downstream_amd.minkowski import functional as Fn

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
b = make_batches(1, 16, 0, 51, 128, 28)[0]
tf = ME.TensorField(coordinates=b["coordinates"].to(dev), features=b["features"].to(dev))
x = tf.sparse()
m = x.coordinate_manager
k1 = ME.CoordinateMapKey(1)
nbr, _ = m.kernel_table(k1, k1, 3, 1)
torch.manual_seed(0)
w = torch.randn(27, 28, 64, device=dev) * 0.05
xin = x.F.contiguous()
n = xin.shape[0]
L = lib()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


ME.set_conv_math("bf16")
y_ref, part_ref = Fn.gather_gemm(xin, w, nbr, 64, stats=True)
t_ref = timeit(lambda: Fn.gather_gemm(xin, w, nbr, 64, stats=True))

xb = torch.empty(n, 32, dtype=torch.bfloat16, device=dev)
check(L.mink_rows_to_bf16(xin.data_ptr(), n, 28, xin.stride(0), xb.data_ptr(), st))
assert torch.equal(xb[:, :28], xin.to(torch.bfloat16)) and not xb[:, 28:].any()
rows = L.mink_stem_conv_bf16s_stats_rows()
yb = torch.empty(n, 64, dtype=torch.bfloat16, device=dev)
part = torch.empty(rows, 2, 64, dtype=torch.float64, device=dev)
run = lambda: check(L.mink_stem_conv_bf16s(xb.data_ptr(), n, w.data_ptr(), 28, nbr.data_ptr(), n, 27, yb.data_ptr(), 64, part.data_ptr(), rows, st))
run()
torch.cuda.synchronize()
d = (yb.float() - y_ref).abs()
tol = y_ref.abs() * 2.0 ** -7 + 1e-3  # one bf16 ulp of the stored value + accumulation-order slack
print("max |yb - y_ref|", float(d.max()), "violations of 1 ulp(bf16)", int((d > tol).sum()), "of", d.numel())
s_chk = yb.double().sum(0), (yb.double() ** 2).sum(0)
s_got = part.sum(0)
print("stats rel err", float(((s_got[0] - s_chk[0]).abs() / (s_chk[0].abs() + 1)).max()), float(((s_got[1] - s_chk[1]).abs() / s_chk[1]).max()))
t_cast = timeit(lambda: check(L.mink_rows_to_bf16(xin.data_ptr(), n, 28, xin.stride(0), xb.data_ptr(), st)))
t_new = timeit(run)
pairs = int((nbr >= 0).sum())
print(f"stem fwd: fp32 storage / bf16 math {t_ref:7.1f} us | bf16 storage {t_new:7.1f} us (+ cast {t_cast:5.1f} us) | gathers {pairs * 64 / t_new / 1e6:6.2f} TB/s")

# ---- fused stem weight gradient: fp32 storage (bf16 math) / bf16 storage with 2-byte gathers / with LDS-transposed operands
k2 = ME.CoordinateMapKey(2)
m.stride(k1, 2)
nbr_pool, _ = m.kernel_table(k1, k2, 2, 1)
i2o = m.stride_map(k1, k2)
npool, C = nbr_pool.shape[0], 64
yf = yb.float().contiguous()
xf = xb[:, :28].float().contiguous()
mean, invstd = yf.mean(0).contiguous(), (yf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
gamma, beta = (torch.rand(C, device=dev) + 0.5), (torch.rand(C, device=dev) - 0.5)
gp = torch.randn(npool, C, device=dev)
dg, db = torch.randn(C, device=dev), torch.randn(C, device=dev)
need = L.mink_conv_wgrad_workspace_bytes(n, 27, 28, C)
slabs = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
dws = [torch.empty(27, 28, C, device=dev) for _ in range(4)]
f32 = lambda o: check(L.mink_conv_wgrad_bn_relu_pool(xf.data_ptr(), n, 28, 28, yf.data_ptr(), C, gp.data_ptr(), npool, i2o.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dg.data_ptr(), db.data_ptr(), nbr.data_ptr(), n, 27, o.data_ptr(), slabs.data_ptr(), need, st))
b16 = lambda o: check(L.mink_conv_wgrad_bn_relu_pool_b16(xb.data_ptr(), n, 28, yb.data_ptr(), C, gp.data_ptr(), npool, i2o.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dg.data_ptr(), db.data_ptr(), nbr.data_ptr(), n, 27, o.data_ptr(), slabs.data_ptr(), need, st))
t0 = timeit(lambda: f32(dws[0]))
L.mink_conv_set_stagger(1 << 11)
t1 = timeit(lambda: b16(dws[1]))
L.mink_conv_set_stagger(0)
t2 = timeit(lambda: b16(dws[2]))
torch.cuda.synchronize()
print(f"stem wgrad (fused): fp32 storage {t0:7.1f} us | bf16 storage, 2-byte gathers {t1:7.1f} us | LDS-transposed {t2:7.1f} us")
print("bitwise: 2-byte gathers == fp32 storage", bool(torch.equal(dws[0], dws[1])), "| LDS-transposed == 2-byte gathers", bool(torch.equal(dws[1], dws[2])))
