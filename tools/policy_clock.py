#!/usr/bin/env python3
"""In-kernel clock of the policy kernel (diagnostic build): run with TPL_DIAG_CLOCK=1.

Each wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its tile loop; the ratio is the clock
the chip actually held while the matrix pipe was loaded."""
import os
import statistics
import sys

assert os.environ.get("TPL_DIAG_CLOCK") == "1", "run as TPL_DIAG_CLOCK=1 python tools/policy_clock.py"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_piclim as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
env = T.BatchedTetris(10, 40, n, auto_reset=True)
rows, pieces = env.synthetic_configs(n)
env.load_configs(rows, pieces)
env.reset()
torch.manual_seed(0)
image = T.actor.policy_image(T.PolicyMLP(), env.device)
out = torch.empty(n, dtype=torch.uint8, device=env.device)
stamps = torch.zeros((n, 14), dtype=torch.float32, device=env.device)     # reused as the stamp buffer
for _ in range(200):                                                      # let the clock settle under load
    env.policy_act(image, out=out, logits=stamps)
torch.cuda.synchronize()
waves = 256 * 8
d = stamps.view(torch.int64).flatten()[: 4 * waves].view(waves, 4).cpu()
rows = [r for r in d.tolist() if r[1] > 0]
clk = [100e6 * float(a) / float(b) / 1e9 for a, b, _, _ in rows]
dur = [float(b) / 100.0 for _, b, _, _ in rows]
pro = [float(c) / 100.0 for _, _, c, _ in rows]
first = min(e for _, _, _, e in rows)
start = [(e - first) / 100.0 for _, _, _, e in rows]
end = [(e - first + c + b) / 100.0 for _, b, c, e in rows]
print(f"waves {len(clk)}: in-kernel clock median {statistics.median(clk):.3f} GHz (min {min(clk):.3f}, max {max(clk):.3f}); "
      f"tile loop median {statistics.median(dur):.1f} us (min {min(dur):.1f}, max {max(dur):.1f}); entry -> weights in LDS median "
      f"{statistics.median(pro):.1f} us (max {max(pro):.1f}); waves start within {max(start):.1f} us of the first; last wave out "
      f"{max(end):.1f} us after the first wave started")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    env.policy_act(image, out=out, logits=stamps)
e1.record()
torch.cuda.synchronize()
print(f"launch period {e0.elapsed_time(e1) * 10:.1f} us")
