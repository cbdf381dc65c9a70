#!/usr/bin/env python3
"""A/B the step kernel's tuning knobs in ONE process (interleaved rounds, median + min) on the bench workload."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--M", type=int, default=40)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--bpl", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--threads", type=int, nargs="+", default=[256])
    ap.add_argument("--auto-reset", type=int, default=1)
    ap.add_argument("--action-period", type=int, default=0,
                    help="reuse the first P action rows cyclically (small P: the actions stay cache-resident)")
    args = ap.parse_args()
    import torch
    import tetris_piclim as T

    n = args.boards
    env = T.BatchedTetris(args.L, args.M, n, auto_reset=bool(args.auto_reset))
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    K = args.steps
    actions = torch.empty((K, n), dtype=torch.uint8, device=env.device)
    for t in range(K):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=env.device)
    done = torch.empty(n, dtype=torch.uint8, device=env.device)
    P = args.action_period or K
    combos = [(b, th) for b in args.bpl for th in args.threads]
    res = {c: [] for c in combos}
    for r in range(args.rounds + 1):
        for b in combos:
            env.set_tuning(*b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(K):
                env.step_into(actions[t % P], reward, done)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[b].append(e0.elapsed_time(e1) / K * 1e3)
    # fused rollout: K steps per launch
    for chunk in (8, 32, K):
        v = []
        for r in range(args.rounds + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(0, K - chunk + 1, chunk):
                env.rollout_into(actions[t:t + chunk], chunk)
            e1.record()
            torch.cuda.synchronize()
            if r:
                v.append(e0.elapsed_time(e1) / ((K // chunk) * chunk) * 1e3)
        print(f"n={n} rollout chunk={chunk}: median {statistics.median(v):.2f} us/step  -> "
              f"{n / statistics.median(v) / 1e3:.1f} G steps/s", flush=True)
    for b, v in res.items():
        print(f"n={n} bpl,threads={b}: median {statistics.median(v):.2f} us  min {min(v):.2f} us  -> "
              f"{n / statistics.median(v) / 1e3:.1f} G steps/s", flush=True)


if __name__ == "__main__":
    main()
