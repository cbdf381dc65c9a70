#!/usr/bin/env python3
"""The policy path's kernels at the bench size (BASELINE configs[4]: 262,144 boards, Model(217, 14) of model/model.py:9-20), one
timing line each -- what tools/profile_policy.sh runs under rocprofv3 and what the same-box A/Bs of profiles/r0*_policy ran:

    python tools/policy_probe.py [boards] [--kernels bf16,f32,split,mega_bf16,mega_f32,mega_split] [--steps T]

bf16 / f32 / split = tpl_policy_act with the bf16, float32 and three-bf16-pieces image (policy_kernel, policy_f32_kernel,
policy_split_kernel); mega_* = tpl_actor_rollout with the same images, T iterations of policy -> explore -> move per launch
(actor_rollout_kernel, actor_rollout_f32_kernel, actor_rollout_split_kernel; T = --steps for bf16, a fifth of it for the others)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

ap = argparse.ArgumentParser()
ap.add_argument("boards", nargs="?", type=int, default=262144)
ap.add_argument("--kernels", default="bf16,f32,split,mega_bf16,mega_f32,mega_split")
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
n = args.boards
FLOPS = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14)
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
IMAGES = {"bf16": False, "f32": True, "split": "split"}


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for kernel in args.kernels.split(","):
    kind = kernel.replace("mega_", "")
    image = T.actor.policy_image(T.PolicyMLP(), env.device, f32=IMAGES[kind])
    if kernel.startswith("mega_"):
        k = args.steps if kind == "bf16" else max(1, args.steps // 5)
        env.actor_rollout(image, k)
        us = timed(lambda: env.actor_rollout(image, k), 6 if kind == "bf16" else 3) * 1e3 / k
        print(f"{kernel} {n} boards: {us:.2f} us/step = {n / us / 1e3:.2f} G env-steps/s ({k} steps per launch)", flush=True)
    else:
        act = torch.empty(n, dtype=torch.uint8, device=env.device)
        for _ in range(5):
            env.policy_act(image, out=act)
        ms = timed(lambda: env.policy_act(image, out=act), (200 if kind == "bf16" else 20) if n <= 262144 else 12)
        tf = FLOPS * n / (ms * 1e-3) / 1e12
        peak = {"bf16": 2500.0, "f32": 157.3}.get(kind)
        print(f"policy_{kind} {n} boards: {ms * 1e3:.2f} us = {tf:.1f} model TFLOP/s"
              + (f" = {tf / peak:.3f} of the {kind} MFMA peak" if peak else " (float32 accuracy from three bf16 pieces per operand)"), flush=True)
