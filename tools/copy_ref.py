#!/usr/bin/env python3
"""Reference point for the step kernel's memory roofline: a plain device copy moving the same number of bytes."""
import torch

for mb in (42.4, 64, 256, 1024):
    n = int(mb * 1e6) // 16 * 16
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"copy {mb} MB -> {us:.2f} us per copy, {2 * n / us / 1e6:.2f} TB/s (read + write)", flush=True)
