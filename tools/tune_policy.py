#!/usr/bin/env python3
"""Time the policy kernel and the actor megakernel on the bench's config-5 workload (median of interleaved rounds)."""
import os
import statistics
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
image = T.actor.policy_image(T.PolicyMLP(), env.device)
out = torch.empty(n, dtype=torch.uint8, device=env.device)
policy, mega = [], []
for rnd in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        env.policy_act(image, out=out)
    e1.record()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    env.actor_rollout(image, 40, record=True)
    f1.record()
    torch.cuda.synchronize()
    if rnd:
        policy.append(e0.elapsed_time(e1) / 50 * 1e3)
        mega.append(f0.elapsed_time(f1) / 40 * 1e3)
print(f"n={n}: policy kernel {statistics.median(policy):.1f} us   megakernel {statistics.median(mega):.1f} us/iteration", flush=True)
