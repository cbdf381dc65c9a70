#!/usr/bin/env python3
"""The step loop at the bench size with PoolRefresher carving beside it, by the generator's footprint: how many
persistent waves share its queue, which stream they run on (plain / lowest priority / CU-masked), and the step kernel's
geometry.  Same box, one process, alternating -- step time alone, step time with the supply, fresh configurations per
second, pool reuse factor.
    python tools/live_supply_probe.py [--steps 6000] [--count 65536]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--count", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default="", help="comma-separated subset of the case keys")
    ap.add_argument("--grid", default="", help="instead of the named cases: WAVES,..xLOG2COUNT,..  e.g. 256,512,1024x18,19,20")
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    import bench_side as bench
    n, dev = args.boards, torch.device("cuda", 0)
    env = T.BatchedTetris(10, 40, n, device=dev, auto_reset=True)
    S = 64
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    cases = [("all", "1024 waves (one configuration per lane: the round-2 form)", dict(waves=1024), 2),
             ("256", "256 waves", dict(waves=256), 2), ("128", "128 waves", dict(waves=128), 2),
             ("64", "64 waves", dict(waves=64), 2), ("32", "32 waves", dict(waves=32), 2),
             ("128b4", "128 waves, step kernel with 4 boards per lane", dict(waves=128), 4),
             ("256b4", "256 waves, step kernel with 4 boards per lane", dict(waves=256), 4),
             ("128low", "128 waves, lowest-priority stream", dict(waves=128, low_priority=True), 2),
             ("128cu32", "128 waves on a 32-CU stream", dict(waves=128, reserved_cus=32), 2)]
    cases += [("1M1024", "1024 waves, batches of 2^20", dict(waves=1024, count=1 << 20), 2),
              ("1M2048", "2048 waves, batches of 2^20", dict(waves=2048, count=1 << 20), 2),
              ("1M4096", "4096 waves, batches of 2^20", dict(waves=4096, count=1 << 20), 2)]
    if args.grid:
        waves, counts = args.grid.split("x")
        cases = [(f"w{w}c{c}", f"{w} waves, batches of 2^{c}", dict(waves=int(w), count=1 << int(c)), 2)
                 for c in counts.split(",") for w in waves.split(",")]
    for r in range(args.rounds):
        for key, name, kw, bpl in cases:
            if args.only and key not in args.only.split(","):
                continue
            env.set_tuning(bpl, 256)
            out = bench.measure_live_supply(torch, T, env, actions, reward, done, 0, **{'count': args.count, 'min_steps': args.steps, **kw})
            print(json.dumps({"round": r, "generator": name, **{k: out[k] for k in (
                "ms_per_step", "ms_per_step_without_refresher", "slowdown", "pool_swaps", "configurations_supplied_per_s",
                "pool_reuse_factor")}}), flush=True)
    env.terminate()


if __name__ == "__main__":
    main()
