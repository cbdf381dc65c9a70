#!/usr/bin/env python3
"""Time the non-step kernels at the bench size (observation expand, get_state export, reset, pool packing)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
for t in range(5):
    env.step(env.synthetic_actions(t), observe=False)


def timeit(name, fn, bytes_moved, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"n={n} {name:28s} {us:9.1f} us   {bytes_moved / us / 1e6:6.2f} TB/s", flush=True)


obs32 = torch.empty((n, 217), dtype=torch.float32, device=env.device)
obs16 = torch.empty((n, 217), dtype=torch.bfloat16, device=env.device)
mask = (torch.arange(n, device=env.device) % 3 == 0).to(torch.uint8)
timeit("expand_obs f32", lambda: env.observe(out=obs32), n * (32 + 868))
timeit("expand_obs bf16", lambda: env.observe(out=obs16), n * (32 + 434))
timeit("get_state export", lambda: env.packed_state(), n * (32 + 46))
timeit("reset (all)", lambda: env.reset(), n * (64 + 32))
timeit("reset (masked 1/3)", lambda: env.reset(mask), n * (32 + 1 + (64 + 32) / 3))
timeit("load_configs (pack)", lambda: env.load_configs(rows, pieces), n * (40 + 41 + 64))
