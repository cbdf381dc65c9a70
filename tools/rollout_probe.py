#!/usr/bin/env python3
"""A few fused-rollout launches at the bench size, for counting under rocprofv3 (--pmc) or timing:
    python tools/rollout_probe.py [--steps-per-launch 100] [--launches 12] [--outputs 0|1|2]
(outputs: 0 none, 1 reward f32 + done u8 per step, 2 the compact trajectory: one byte per board-step)"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--steps-per-launch", type=int, default=100)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--outputs", type=int, default=0)
    ap.add_argument("--random", action="store_true", help="the uniform random policy drawn on the device (tpl_rollout_random), no outputs")
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    n, K = args.boards, args.steps_per_launch
    env = T.BatchedTetris(10, 40, n, auto_reset=True)
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    actions = torch.empty((K, n), dtype=torch.uint8, device=env.device)
    for t in range(K):
        env.synthetic_actions(t, out=actions[t])
    for _ in range(2):
        env.rollout_into(actions, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    traj = torch.empty(((K + 3) // 4, n), dtype=torch.int32, device=env.device)
    rs = torch.empty((K, n), dtype=torch.float32, device=env.device)
    ds = torch.empty((K, n), dtype=torch.uint8, device=env.device)
    import ctypes
    for _ in range(args.launches):
        if args.random:
            env.rollout_random(K, seed=0)
        elif args.outputs == 2:
            env.rollout_trajectory(actions, out=traj)
        elif args.outputs:
            T._lib.check(env._lib.tpl_rollout(env._h, ctypes.c_void_p(actions.data_ptr()), actions.stride(0), K,
                                              ctypes.c_void_p(rs.data_ptr()), ctypes.c_void_p(ds.data_ptr()), None, None, env._stream()))
        else:
            env.rollout_into(actions, K)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (args.launches * K)
    print(f"rollout {K} steps/launch, outputs={'device-drawn policy' if args.random else args.outputs}: {us:.3f} us/step = {n / us / 1e3:.1f} G env-steps/s")
    env.terminate()


if __name__ == "__main__":
    main()
