#!/usr/bin/env python3
"""Stream launches vs HIP-graph replay of the same step sequence: per-step time at the bench size."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
actions = torch.empty((K, n), dtype=torch.uint8, device=env.device)
for t in range(K):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=env.device)
done = torch.empty(n, dtype=torch.uint8, device=env.device)


def run_stream():
    for t in range(K):
        env.step_into(actions[t], reward, done)


def timed(fn, reps=6):
    out = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            out.append(e0.elapsed_time(e1) / K * 1e3)
    return statistics.median(out), min(out)


run_stream()
torch.cuda.synchronize()
print("stream launches: median %.2f us/step (min %.2f)" % timed(run_stream), flush=True)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    run_stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    run_stream()
torch.cuda.synchronize()
print("graph replay   : median %.2f us/step (min %.2f)" % timed(g.replay), flush=True)
