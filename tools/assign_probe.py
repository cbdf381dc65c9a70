#!/usr/bin/env python3
"""Step time at the bench configuration under the two assignment modes (hash: every reset gathers from a random record;
sequential: boards that reset together take adjacent records) and with a small pool (every gather hits the caches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

dev = torch.device("cuda", 0)
n, L, M, K = 1 << 20, 10, 40, 600
for assign, pool in (("hash", n), ("sequential", n), ("hash", 4096), ("sequential", 4096), ("hash", n)):
    env = T.BatchedTetris(L, M, n, device=dev, seed=0, auto_reset=True, assign=assign)
    rows, pieces = env.synthetic_configs(pool)
    env.load_configs(rows, pieces)
    env.reset()
    acts = torch.empty((64, n), dtype=torch.uint8, device=dev)
    for t in range(64):
        env.synthetic_actions(t, out=acts[t])
    r = torch.empty(n, dtype=torch.float32, device=dev)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for t in range(100):
        env.step_into(acts[t % 64], r, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(K):
        env.step_into(acts[t % 64], r, d)
    e1.record()
    torch.cuda.synchronize()
    print(f"assign={assign:10s} pool={pool:8d}: {e0.elapsed_time(e1) / K * 1e3:.2f} us/step, episodes {env.stats()['episodes']}", flush=True)
    env.terminate()
