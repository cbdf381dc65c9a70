#!/usr/bin/env python3
"""Would two half-batches on two streams (no ordering between them) step faster than one batch on one stream?
Each half's launches are ordered on its own stream; the halves' launch gaps and read/write phases can overlap."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

dev = torch.device("cuda", 0)
L, M, K = 10, 40, 400


def make(n, offset):
    env = T.BatchedTetris(L, M, n, device=dev, seed=0, global_offset=offset, auto_reset=True)
    rows, pieces = env.synthetic_configs(n, first=offset)
    env.load_configs(rows, pieces)
    env.reset()
    acts = torch.empty((64, n), dtype=torch.uint8, device=dev)
    for t in range(64):
        env.synthetic_actions(t, out=acts[t])
    return env, acts, torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)


def run(envs, streams, tag):
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_event(e0)
        calls = []
        for (env, acts, r, d), st in zip(envs, streams):
            calls.append((env._lib.tpl_step, env._h, [acts[t].data_ptr() for t in range(64)], r.data_ptr(), d.data_ptr(), st.cuda_stream))
        for t in range(K):
            for fn, h, aptr, rp, dp, sh in calls:            # straight through the C ABI: the host must not be the limit
                fn(h, aptr[t % 64], 0, rp, dp, sh)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K * 1e3)
    print(f"{tag}: {best:.2f} us per full-batch step (best of 3)", flush=True)


# HIP deals its streams out over four hardware queues and two streams on one queue do not overlap (profiles/NOTES.md, round 3):
# the pairs are taken from consecutive streams of torch's pool, so most of them sit on different queues
for parts, bpl in ((1, 2), (2, 1), (2, 2), (4, 1)):
    n = (1 << 20) // parts
    envs = [make(n, k * n) for k in range(parts)]
    for e in envs:
        e[0].set_tuning(bpl, 256)
    pool = [torch.cuda.Stream(dev) for _ in range(8)]
    torch.cuda.synchronize()
    for first in range(0, 8 - parts + 1, 1 if parts > 1 else 8):
        run(envs, pool[first: first + parts], f"{parts} stream(s) x {n} boards, {bpl} board(s) per lane, pool streams {first}..{first + parts - 1}")
    for env, *_ in envs:
        env.terminate()
