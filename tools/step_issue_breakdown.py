#!/usr/bin/env python3
"""Where the host's microseconds per step_into() call go at a batch small enough for the device to keep up (the period of
a shard of a multi-GPU job is the HOST's, tools/step_issue_rate.py): each part timed alone over 20,000 repetitions."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(1 << 18)
env.load_configs(rows, pieces)
env.reset()
actions = torch.empty((64, n), dtype=torch.uint8, device=env.device)
for t in range(64):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=env.device)
done = torch.empty(n, dtype=torch.uint8, device=env.device)
R = 20000


def timed(label, f, sync=False):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:58s} {(t1 - t0) * 1e6 / R:6.2f} us")


a0 = actions[0]
lib, h = env._lib, env._h
pa, pr, pd = a0.data_ptr(), reward.data_ptr(), done.data_ptr()
st = env._stream()
k = [0]


def index():
    k[0] = (k[0] + 1) & 63
    return actions[k[0]]


timed("actions[t] (a row view of a [64, n] tensor)", index)
timed("three data_ptr()", lambda: (a0.data_ptr(), reward.data_ptr(), done.data_ptr()))
timed("env._stream()", env._stream)
timed("lib.tpl_step with ready-made integers (ctypes + C + launch)", lambda: lib.tpl_step(h, pa, 0, pr, pd, st))
timed("env.step_into(row, reward, done) on one row", lambda: env.step_into(a0, reward, done))
timed("env.step_into(actions[t], reward, done)", lambda: env.step_into(index(), reward, done))
env.terminate()
