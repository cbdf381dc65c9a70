#!/usr/bin/env python3
"""Duration of the bf16 policy kernel at the bench size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = 200 if n <= 262144 else 12
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
image = T.actor.policy_image(T.PolicyMLP(), env.device)
act = torch.empty(n, dtype=torch.uint8, device=env.device)
for _ in range(5):
    env.policy_act(image, out=act)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    env.policy_act(image, out=act)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
tf = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14) * n / (ms * 1e-3) / 1e12
print(f"policy_bf16 {n} boards: {ms * 1e3:.2f} us = {tf:.0f} TFLOP/s = {tf / 2500:.3f} of the bf16 MFMA peak")
