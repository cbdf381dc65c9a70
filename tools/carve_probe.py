#!/usr/bin/env python3
"""A few launches of the device carving generator (csrc/carve_device.hip), for timing or for counters under rocprofv3:
    python tools/carve_probe.py [--count 65536] [--launches 4] [--L 10] [--M 40]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=65536)
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--M", type=int, default=40)
    ap.add_argument("--waves", type=int, default=0, help="waves sharing the generator's queue (0 = automatic)")
    args = ap.parse_args()
    import torch
    import tetris_piclim as T
    env = T.BatchedTetris(args.L, args.M, 64)
    env.carved_configs(4096)
    torch.cuda.synchronize()
    for k in range(args.launches):
        t0 = time.perf_counter()
        env.carved_configs(args.count, first=k * args.count, waves=args.waves)   # returns after its status check (host sync)
        dt = time.perf_counter() - t0
        print(f"carve_device: {args.count} configurations (L={args.L}, M={args.M}, waves={args.waves or 'auto'}) in {dt * 1e3:.2f} ms = {args.count / dt / 1e6:.2f} M/s")
    env.terminate()


if __name__ == "__main__":
    main()
