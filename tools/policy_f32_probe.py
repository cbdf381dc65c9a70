#!/usr/bin/env python3
"""Duration of the float32 policy kernel at the bench size, and its rate against the f32 MFMA peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
image = T.actor.policy_image(T.PolicyMLP(), env.device, f32=True)
act = torch.empty(n, dtype=torch.uint8, device=env.device)
for _ in range(3):
    env.policy_act(image, out=act)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    env.policy_act(image, out=act)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
tf = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14) * n / (ms * 1e-3) / 1e12
print(f"policy_f32 {n} boards: {ms * 1e3:.1f} us = {tf:.1f} TFLOP/s = {tf / 157.3:.3f} of the f32 MFMA peak")
