#!/usr/bin/env python3
"""tools/hbm_write_rate.py -- what a pure store stream reaches on this box (torch fill_ / zero_ of a 6.5 GB buffer, the size of config 3's
matrix values), beside a copy of the same size: the roof the assembly's write-out stands under."""
import time
import torch
n = int(6.5e9 // 8)
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
for name, fn, nbytes in (("fill_", lambda: x.fill_(1.5), 8 * n), ("zero_", lambda: x.zero_(), 8 * n), ("copy_ (read + write)", lambda: y.copy_(x), 16 * n)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.3f} ms, {nbytes / dt / 1e12:.2f} TB/s")
