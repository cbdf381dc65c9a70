#!/usr/bin/env python3
"""Per-launch duration of the step kernel from a cold start: why do the first launches of a timed region take longer?

    python tools/launch_probe.py [--boards N] [--launches K] [--idle SECONDS]

After the environment is set up the GPU is left idle for `--idle` seconds, then K step launches are issued with a HIP
event after each (and, for comparison, the same K launches between two events only).  Prints the per-launch times of
the first launches, chunk medians, and the totals of the two forms (what the per-launch events themselves cost)."""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--idle", type=float, default=1.0)
    ap.add_argument("--d2h", type=int, default=0, help="read the statistics back (a device-to-host copy) right before each round")
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    dev = torch.device("cuda", 0)
    n, K = args.boards, args.launches
    env = T.BatchedTetris(10, 40, n, device=dev, seed=0, auto_reset=True, assign="hash")
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    S = 64
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    out = {"boards": n, "launches": K}
    for name, idle in (("after_idle", args.idle), ("back_to_back", 0.0), ("after_idle_again", args.idle)):
        time.sleep(idle)
        if args.d2h:
            env.stats()
            torch.cuda.synchronize(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        ev[0].record()
        for t in range(K):
            env.step_into(actions[t % S], reward, done)
            ev[t + 1].record()
        torch.cuda.synchronize(dev)
        us = [ev[t].elapsed_time(ev[t + 1]) * 1e3 for t in range(K)]
        time.sleep(idle)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(K):
            env.step_into(actions[t % S], reward, done)
        e1.record()
        torch.cuda.synchronize(dev)
        chunks = [round(statistics.median(us[i:i + 25]), 2) for i in range(0, K, 25)]
        out[name] = {"idle_s": idle, "first_40_us": [round(x, 1) for x in us[:40]], "median_per_25": chunks,
                     "mean_us_event_per_launch": sum(us) / K, "mean_us_two_events": e0.elapsed_time(e1) * 1e3 / K,
                     "median_us": statistics.median(us)}
    print(json.dumps(out))
    env.terminate()


if __name__ == "__main__":
    main()
