#!/usr/bin/env python3
"""What the step kernel's small output streams cost: tpl_step with reward / done present or null (same process,
interleaved rounds)."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
K = 300
actions = torch.empty((K, n), dtype=torch.uint8, device=env.device)
for t in range(K):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=env.device)
done = torch.empty(n, dtype=torch.uint8, device=env.device)
lib = T._lib.lib()
stream = torch.cuda.current_stream().cuda_stream
cases = {"reward+done": (reward.data_ptr(), done.data_ptr()), "reward only": (reward.data_ptr(), None),
         "done only": (None, done.data_ptr()), "neither": (None, None)}
res = {k: [] for k in cases}
for r in range(6):
    for name, (rp, dp) in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(K):
            lib.tpl_step(env._h, C.c_void_p(actions[t].data_ptr()), 0, C.c_void_p(rp), C.c_void_p(dp), C.c_void_p(stream))
        e1.record()
        torch.cuda.synchronize()
        if r:
            res[name].append(e0.elapsed_time(e1) / K * 1e3)
for name, v in res.items():
    print(f"n={n} {name:12s}: median {statistics.median(v):.2f} us  min {min(v):.2f} us", flush=True)
