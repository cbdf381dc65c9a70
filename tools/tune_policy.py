#!/usr/bin/env python3
"""A/B the policy kernels' tuning knobs in one process: standalone policy kernel and the actor megakernel."""
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
combos = [(1, 0)]
res = {c: ([], []) for c in combos}
for rnd in range(4):
    for c in combos:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            env.policy_act(image, out=out)
        e1.record()
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if c[0] == 1:
            f0.record()
            env.actor_rollout(image, 40, record=True)
            f1.record()
            torch.cuda.synchronize()
        if rnd:
            res[c][0].append(e0.elapsed_time(e1) / 50 * 1e3)
            if c[0] == 1:
                res[c][1].append(f0.elapsed_time(f1) / 40 * 1e3)
for c, (a, b) in res.items():
    line = f"n={n} variant={c[0]} stagger={c[1]}: policy {statistics.median(a):.1f} us"
    if b:
        line += f"   megakernel {statistics.median(b):.1f} us/iteration"
    print(line, flush=True)
