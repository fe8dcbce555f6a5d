import torch
torch.manual_seed(0)
for (m, k, n) in [(25, 2048, 512), (200, 512, 2048), (5000, 256, 64), (25, 512, 512)]:
    a, b = torch.randn(m, k), torch.randn(k, n)
    ref = a.double() @ b.double()
    cpu = (a @ b).double()
    gpu = (a.cuda() @ b.cuda()).cpu().double()
    gpu_t = (a.cuda().t().contiguous().t() @ b.cuda()).cpu().double()
    print((m, k, n), "rel err cpu %.2e gpu %.2e gpu(strided A) %.2e" % (float((cpu - ref).norm() / ref.norm()), float((gpu - ref).norm() / ref.norm()), float((gpu_t - ref).norm() / ref.norm())))
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "fp32 precision", torch.get_float32_matmul_precision())
