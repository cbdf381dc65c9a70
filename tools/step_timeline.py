#!/usr/bin/env python3
"""Where a step-kernel launch spends its time (diagnostic build): run with TPL_DIAG_CLOCK=1.

Lane 0 of every wave stamps the 100 MHz real-time counter at: 0 entry, 1 state words and actions arrived,
2 moves computed, 3 pool records of finished boards arrived, 4 state stores issued, 5 stores acknowledged.
Printed per launch: when waves enter (dispatch ramp), and the distribution of each phase over the waves, in us
relative to the first wave's entry."""
import ctypes as C
import os
import sys

assert os.environ.get("TPL_DIAG_CLOCK") == "1", "run as TPL_DIAG_CLOCK=1 python tools/step_timeline.py"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
bpl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
env.set_tuning(bpl, 256)
K = 120
actions = torch.empty((K, n), dtype=torch.uint8, device=env.device)
for t in range(K):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=env.device)
done = torch.empty(n, dtype=torch.uint8, device=env.device)
waves = (n + 64 * bpl - 1) // (64 * bpl)
lib = T._lib.lib()
lib.tpl_dev_set_step_diag.argtypes = [C.c_void_p, C.c_void_p]
for t in range(K - 3):                       # steady state first, unstamped
    env.step_into(actions[t], reward, done)
stamps = torch.zeros((3, waves, 6), dtype=torch.int64, device=env.device)
for j in range(3):
    lib.tpl_dev_set_step_diag(env._h, C.c_void_p(stamps[j].data_ptr()))
    env.step_into(actions[K - 3 + j], reward, done)
lib.tpl_dev_set_step_diag(env._h, None)
torch.cuda.synchronize()
names = ["entry", "state+actions in", "moves computed", "pool records in", "state stores out", "stores acked"]
for j in range(3):
    s = stamps[j].cpu().double() / 100.0     # us
    t0 = s[:, 0].min()
    s = s - t0
    print(f"launch {j}: {waves} waves, bpl={bpl}; first entry -> last ack {s[:, 5].max():.2f} us")
    if j:
        prev_end = (stamps[j - 1, :, 5].max().item()) / 100.0
        print(f"   gap from the previous launch's last ack to this launch's first entry: {t0.item() - prev_end:.2f} us")
    for k, name in enumerate(names):
        q = torch.quantile(s[:, k], torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.double))
        print(f"   {name:18s} min {q[0]:6.2f}  p10 {q[1]:6.2f}  median {q[2]:6.2f}  p90 {q[3]:6.2f}  max {q[4]:6.2f}")
    d = s[:, 1:] - s[:, :-1]
    for k in range(5):
        q = torch.quantile(d[:, k], torch.tensor([0.1, 0.5, 0.9], dtype=torch.double))
        print(f"   per wave: {names[k]:18s} -> {names[k + 1]:18s} p10 {q[0]:5.2f}  median {q[1]:5.2f}  p90 {q[2]:5.2f}")
