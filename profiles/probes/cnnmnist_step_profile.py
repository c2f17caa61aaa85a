#!/usr/bin/env python3
"""torch.profiler table of one CNNMNIST training step (GPU box): which kernels the step's time is."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bitnetmcu_amd import qat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.manual_seed(0)
m = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym").cuda()
x = torch.randn(n, 1, 16, 16, device="cuda")
gy = torch.randn(n, 10, device="cuda")


def step():
    m.zero_grad(set_to_none=True)
    (m(x) * gy).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]
total = sum(e.device_time_total for e in prof.key_averages())
print(f"{n} images, 3 steps: {total / 3e3:.2f} ms of kernels per step")
for e in rows:
    print(f"{e.device_time_total / 3e3:8.3f} ms/step  {e.count // 3:4d} x  {e.key[:110]}")
