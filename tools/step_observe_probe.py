#!/usr/bin/env python3
"""tpl_step_observe against tpl_step + tpl_expand_obs at a given batch size, for timing or under rocprofv3:
    python tools/step_observe_probe.py [--boards 65536] [--L 5] [--M 20] [--bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=65536)
    ap.add_argument("--L", type=int, default=5)
    ap.add_argument("--M", type=int, default=20)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    import bench
    n, dev = args.boards, torch.device("cuda", 0)
    dtype = torch.bfloat16 if args.bf16 else torch.float32
    env = T.BatchedTetris(args.L, args.M, n, device=dev, auto_reset=True)
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    S = 64
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    obs = torch.empty((n, 217), dtype=dtype, device=dev)
    step = iter(range(10 ** 9))

    def two():
        env.step_into(actions[next(step) % S], reward, done)
        env.observe(out=obs)

    def one():
        env.step_observe_into(actions[next(step) % S], reward, done, obs)

    for f in (two, one):
        for _ in range(20):
            f()
    torch.cuda.synchronize()
    for rnd in range(3):
        t2 = bench.timed(torch, dev, two, args.steps) * 1e3
        t1 = bench.timed(torch, dev, one, args.steps) * 1e3
        gbs = (70 + 217 * obs.element_size()) * n / (t1 * 1e-6) / 1e9
        print(f"{n} boards, {dtype}: step + observe {t2:.2f} us, step_observe {t1:.2f} us ({gbs:.0f} GB/s)", flush=True)
    env.terminate()


if __name__ == "__main__":
    main()
