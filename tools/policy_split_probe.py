#!/usr/bin/env python3
"""Duration of the split policy kernel (float32 accuracy from three bf16 pieces) beside the float32 MFMA kernel."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
out = []
for name, kind in (("float32 MFMA", True), ("three bf16 pieces", "split")):
    image = T.actor.policy_image(T.PolicyMLP(), env.device, f32=kind)
    act = torch.empty(n, dtype=torch.uint8, device=env.device)
    for _ in range(3):
        env.policy_act(image, out=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.policy_act(image, out=act)
    e1.record()
    torch.cuda.synchronize()
    out.append((name, e0.elapsed_time(e1) / 20 * 1e3))
print(f"{n} boards: " + "; ".join(f"{k}: {v:.1f} us" for k, v in out) + f"; ratio {out[0][1] / out[1][1]:.2f}")
