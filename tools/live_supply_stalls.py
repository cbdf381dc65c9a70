#!/usr/bin/env python3
"""Does the generator's stream run BESIDE the stepping stream?  Repetitions of: a fresh PoolRefresher, 120 groups of 32 steps
with an event after each group, the distribution of the groups' per-step times.  A stream that shares the stepping stream's
hardware queue shows as one group of ~30 ms per swap (the generator's launch serialised with the steps).
    python tools/live_supply_stalls.py ['{"reserved_cus": 256}'] [repetitions]      # first argument: PoolRefresher keywords (JSON)
    TORCH_POOL=1 ...   take the side stream straight from torch's pool, as round 2 did (every fourth one stalls)
    FRESH=1 ...        forget the tested stream between repetitions (select again each time)"""
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, tetris_piclim as T
n, dev = 1 << 20, torch.device("cuda", 0)
KW = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
print('KW', KW, flush=True)
env = T.BatchedTetris(10, 40, n, device=dev, auto_reset=True)
S = 100
actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
for t in range(S):
    env.synthetic_actions(t, out=actions[t])
reward = torch.empty(n, dtype=torch.float32, device=dev)
done = torch.empty(n, dtype=torch.uint8, device=dev)
rows, pieces = T.generate_configs(10, 40, 4096, seed=0)
env.load_configs(rows, pieces); env.reset()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    for t in range(100):
        env.step_into(actions[t % S], reward, done)
    torch.cuda.synchronize()
    feeder = T.PoolRefresher(env, 65536, seed=0, first=4096 + rep * (1 << 22), **KW)
    if os.environ.get('TORCH_POOL'):
        feeder.close(); feeder.side = torch.cuda.Stream(dev); feeder.start()
    G = 120
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(G + 1)]
    evs[0].record()
    swaps = 0
    for g in range(G):
        for t in range(32):
            env.step_into(actions[t % S], reward, done)
        swaps += bool(feeder.poll())
        evs[g + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[g].elapsed_time(evs[g + 1]) / 32 * 1e3 for g in range(G))
    print(f"rep {rep} side stream {feeder.side.cuda_stream:#x}: us/step min {per[0]:.1f} p10 {per[12]:.1f} median {per[60]:.1f} p90 {per[108]:.1f} max {per[-1]:.1f}; swaps {swaps}", flush=True)
    feeder.close()
    if os.environ.get('FRESH'):
        import importlib; T.pool._CONCURRENT.clear()
    # the guard: let M + 1 steps pass so that the next feeder can swap
env.terminate()
