#!/usr/bin/env python3
"""Launch period of tpl_step at a small batch (a shard of a multi-GPU job, or BASELINE configs[1]):
    python tools/small_step_probe.py [boards=131072] [L=10] [M=40]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
M = int(sys.argv[3]) if len(sys.argv) > 3 else 40
env = T.BatchedTetris(L, M, n, auto_reset=True)
rows, pieces = env.synthetic_configs(1 << 20)
env.load_configs(rows, pieces)
env.reset()
S = 400
actions = torch.empty((S, n), dtype=torch.uint8, device=env.device)
for t in range(S):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=env.device)
done = torch.empty(n, dtype=torch.uint8, device=env.device)
for t in range(50):
    env.step_into(actions[t], reward, done)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(S):
    env.step_into(actions[t], reward, done)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / S
print(f"tpl_step, {n} boards (L={L}, M={M}): {us:.2f} us per step = {n / us / 1e3:.2f} G env-steps/s, {96 * n / us / 1e3 / 8000:.3f} of the HBM roofline")
