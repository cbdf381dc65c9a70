#!/usr/bin/env python3
"""Cost of the batched Tetris.get_state() (bool board [N,20,10] + scalars) at the bench size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
for name, fn in (("get_state()", env.get_state), ("packed_state()", env.packed_state)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n} {name:16s} {e0.elapsed_time(e1) / 10 * 1e3:9.1f} us", flush=True)
