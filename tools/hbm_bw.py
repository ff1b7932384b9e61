#!/usr/bin/env python3
"""Achievable HBM rates on the box (context for the roofline): fill of 4 GiB (write only), copy of 2 GiB (read + write)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 29                                             # doubles: 4 GiB
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n // 2, dtype=torch.float64, device=dev)
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: x.fill_(1.0))
print("fill 4 GiB: %.3f ms -> %.0f GB/s written" % (t, n * 8 / t / 1e6))
t = timed(lambda: y.copy_(x[: n // 2]))
print("copy 2 GiB: %.3f ms -> %.0f GB/s read + %.0f GB/s written" % (t, n * 4 / t / 1e6, n * 4 / t / 1e6))
