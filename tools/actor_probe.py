#!/usr/bin/env python3
"""Run the fused actor loop a few hundred times (for rocprofv3 --kernel-trace --stats) and time eager vs graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
for use_graph in (False, True):
    act = T.Actor(env, T.PolicyMLP(), use_graph=use_graph, fused=True)
    act.run(20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    act.run(200)
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n} graph={use_graph}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per iteration", flush=True)
