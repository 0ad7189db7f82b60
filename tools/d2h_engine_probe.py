#!/usr/bin/env python3
"""Which engine carries a device -> page-locked host copy here, and what it costs a kernel that runs beside it:
run under `rocprofv3 --kernel-trace --memory-copy-trace --stats` (a shader copy shows up as a kernel)."""
import time

import torch

torch.cuda.set_device(0)
n = 3840 * 2160
dev = torch.zeros(n, dtype=torch.int32, device="cuda")
host = torch.zeros(n, dtype=torch.int32).pin_memory()
a = torch.randn(4096, 4096, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    host.copy_(dev, non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    host.copy_(dev, non_blocking=True)
torch.cuda.synchronize()
print(f"copy alone: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
with torch.cuda.stream(s1):
    for _ in range(3):
        b = a @ a
torch.cuda.synchronize()
t = time.perf_counter()
with torch.cuda.stream(s1):
    for _ in range(10):
        b = a @ a
torch.cuda.synchronize()
alone = (time.perf_counter() - t) / 10 * 1e3
t = time.perf_counter()
with torch.cuda.stream(s2):
    for _ in range(4):
        host.copy_(dev, non_blocking=True)
with torch.cuda.stream(s1):
    for _ in range(10):
        b = a @ a
s1.synchronize()
beside = (time.perf_counter() - t) / 10 * 1e3
torch.cuda.synchronize()
print(f"4096^3 matmul alone {alone:.3f} ms, beside copies {beside:.3f} ms")
