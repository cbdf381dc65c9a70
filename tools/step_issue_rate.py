#!/usr/bin/env python3
"""Who sets the period of a host-driven step loop, by batch size: the device's time per launch (HIP events around 2000 steps)
against the time the HOST needs to issue the 2000 step_into() calls (perf_counter around the loop, no synchronize inside),
and how long the device still ran after the last call returned.  Below ~200,000 boards the two periods coincide and the device is
idle when the loop ends: the period is the host's -- 3.5 us of it hipLaunchKernel itself (tools/launch_paths.hip), the rest
ctypes, the wrapper's argument checks and (first form) the `actions[t]` view built per step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import tetris_piclim as T
for n in (1024, 16384, 131072, 262144, 1048576):
    env = T.BatchedTetris(10, 40, n, auto_reset=True)
    rows, pieces = env.synthetic_configs(1 << 18)
    env.load_configs(rows, pieces)
    env.reset()
    S = 2000
    actions = torch.empty((64, n), dtype=torch.uint8, device=env.device)
    for t in range(64):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=env.device)
    done = torch.empty(n, dtype=torch.uint8, device=env.device)
    for t in range(50):
        env.step_into(actions[t % 64], reward, done)
    torch.cuda.synchronize()
    rows_of = list(actions.unbind(0))
    for form in ("actions[t]", "row views made once"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if form == "actions[t]":
            for t in range(S):
                env.step_into(actions[t % 64], reward, done)
        else:
            for t in range(S):
                env.step_into(rows_of[t % 64], reward, done)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{n:8d} boards, {form:19s}: device period {e0.elapsed_time(e1) * 1e3 / S:6.2f} us/step; host issued the {S} calls in "
              f"{(t1 - t0) * 1e6 / S:6.2f} us/step, waited {(t2 - t1) * 1e3:.2f} ms for the device after the last one")
    env.terminate()
