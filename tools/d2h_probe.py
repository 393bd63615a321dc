"""how does this platform move 29 MB from the device to pinned host memory, and
what does a kernel running beside the copy pay?  (hipMemcpyAsync shows up as a
blit kernel, __amd_rocclr_copyBuffer, in a rocprofv3 kernel trace)
python tools/d2h_probe.py"""
import time

import torch

n = 100000
d = torch.randn((n, 36), dtype=torch.float64, device="cuda")
h = torch.empty((n, 36), dtype=torch.float64, pin_memory=True)
x = torch.randn(64 * 1024 * 1024 // 8, device="cuda", dtype=torch.float64)
side = torch.cuda.Stream()
for _ in range(3):
    h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("D2H 28.8 MB: %.3f ms = %.1f GB/s" % (dt * 1e3, 28.8e-3 / dt))


def busy():
    for _ in range(10):
        x.mul_(1.0000001)


for _ in range(2):
    busy()
torch.cuda.synchronize()
t0 = time.perf_counter()
busy()
torch.cuda.synchronize()
alone = time.perf_counter() - t0
t0 = time.perf_counter()
with torch.cuda.stream(side):
    for _ in range(6):
        h.copy_(d, non_blocking=True)
busy()
torch.cuda.synchronize()
both = time.perf_counter() - t0
print("10 in-place scalings of 64 MB: alone %.3f ms; with 6 copies on another stream %.3f ms "
      "(the copies alone: %.3f ms)" % (alone * 1e3, both * 1e3, 6 * dt * 1e3))
