#!/usr/bin/env python3
"""What does ANY kernel running on another stream cost the step loop?  The main stream steps 2^20 boards while a side
stream runs torch's one-thread spin kernel (no memory traffic, one wave on one SIMD of the chip) for the whole time.
    python tools/coexist_probe.py"""
import os
import sys

sys.path.insert(0, os.getcwd())


def main():
    import torch
    import tetris_piclim as T
    import bench
    n, dev = 1 << 20, torch.device("cuda", 0)
    env = T.BatchedTetris(10, 40, n, device=dev, auto_reset=True)
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    S = 64
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    step = lambda: env.step_into(actions[0], reward, done)
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    # a fresh torch stream each round: HIP deals its streams out over a few hardware queues, and a side stream that
    # lands on the stepping stream's queue does not run BESIDE it
    for rnd in range(10):
        side = torch.cuda.Stream(dev)
        alone = bench.timed(torch, dev, step, 2000)
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(1e8))                      # tens of milliseconds of one spinning thread
        both = bench.timed(torch, dev, step, 2000)
        torch.cuda.synchronize()
        print(f"stream {rnd} ({side.cuda_stream:#x}): step alone {alone * 1e3:.2f} us, with a one-thread spin kernel on the side stream {both * 1e3:.2f} us", flush=True)
    env.terminate()


if __name__ == "__main__":
    main()
