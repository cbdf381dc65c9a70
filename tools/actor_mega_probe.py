#!/usr/bin/env python3
"""Per-step time of the actor megakernels (T iterations of policy -> epsilon-greedy -> step in one launch) at the bench size:
bf16 operands (tpl_actor_rollout) and, where the library has it, float32 operands (tpl_actor_rollout_f32).
    python tools/actor_mega_probe.py [boards] [steps per launch]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
out = []
for name, f32, k, reps in (("bf16", False, iters, 6), ("f32", True, max(1, iters // 5), 3), ("split", "split", max(1, iters // 5), 3)):
    if (f32 is True and not hasattr(env._lib, "tpl_actor_rollout_f32")) or (f32 == "split" and not hasattr(env._lib, "tpl_actor_rollout_split")):
        continue
    image = T.actor.policy_image(T.PolicyMLP(), env.device, f32=f32)
    try:
        env.actor_rollout(image, k)
    except Exception as e:                                   # an older tree: no float32 megakernel
        out.append(f"{name}: n/a ({type(e).__name__})")
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.actor_rollout(image, k)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * k)
    out.append(f"{name}: {us:.2f} us/step = {n / us / 1e3:.2f} G env-steps/s ({k} steps per launch)")
print(f"actor megakernel, {n} boards: " + "; ".join(out))
