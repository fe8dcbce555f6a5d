"""BASELINE config #5 on the GPU box: the dense 2-D comparison network (ResNet18, 224^2 synthetic renders, batch 32,
LitModel recipe: CE + weight-decay term + SGD) -- images/s forward+backward+update with the convolutions on the bf16
matrix cores and in exact fp32, beside the same network on torch's own dense operators (MIOpen) for orientation.
usage: python scripts/bench_2d.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from nerf_downstream_amd.co3d_2d.src.modules.classification import LitModel
from nerf_downstream_amd.minkowski import functional as Fn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(B, 3, 224, 224, device=dev)
y = torch.randint(0, 51, (B,), device=dev)


def bench(model, opt, step_fn, tag):
    for _ in range(5):
        step_fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag}: {ms:.2f} ms/step, {B / ms * 1e3:.0f} images/s")


lit = LitModel("resnet18").to(dev)
opt = lit.configure_optimizers()


def hip_step():
    opt.zero_grad(set_to_none=True)
    loss, _ = lit.training_step({"images": x, "labels": y})
    loss.backward()
    opt.step()


# (the second 25-step window of a fresh process runs 30-100 % slower than the ones before and after it, whatever the math mode -- allocator /
#  clock settling, not the kernels: 8.8 / 11.5 / 8.9 ms in fp32 -- so the process is warmed past it before anything is timed)
for _ in range(60):
    hip_step()
for math in ("bf16", "fp32"):
    Fn.set_conv_math(math)
    bench(lit, opt, hip_step, f"HIP sparse-kernel ResNet18 B={B} 224^2, conv math {math}")
Fn.set_conv_math("fp32")

from test_gpu_dense2d import _TorchResNet18
ref = _TorchResNet18().to(dev).to(memory_format=torch.channels_last)
ropt = torch.optim.SGD(ref.parameters(), 0.1, momentum=0.9)
xr = x.to(memory_format=torch.channels_last)


def torch_step(amp):
    def f():
        ropt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = torch.nn.functional.cross_entropy(ref(xr), y, label_smoothing=0.005)
        loss.backward()
        ropt.step()
    return f


bench(ref, ropt, torch_step(True), "torch (MIOpen) ResNet18 channels_last, bf16 autocast")
bench(ref, ropt, torch_step(False), "torch (MIOpen) ResNet18 channels_last, fp32")
