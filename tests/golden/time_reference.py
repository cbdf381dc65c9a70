#!/usr/bin/env python3
"""Rate of the REFERENCE's Tetris.move in this container, beside the two CPU restatements (the C oracle and the NumPy
per-board port) on the same workload, one core each.  Writes tests/golden/reference_timing.json (numbers only), which
bench.py's cpu_baseline leg quotes so that the GPU box's figures can be read in reference-equivalent terms.

Runs only where /root/reference exists.  Workload: the synthetic set of bench.py (L=10, M=40), move; reset when
finished -- the loop of game/performance_test.py:13-17.  The ratio lets the GPU box's cpu_baseline (the oracle)
be read in reference-equivalent terms."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/game")
sys.path.insert(0, ROOT)
os.chdir("/tmp")
import tetris as ref  # noqa: E402
from oracle import numpy_port as NP  # noqa: E402
from oracle import oracle as O  # noqa: E402

L, M, n, seed = 10, 40, 256, 0
rows = O.synth_boards(seed, 0, n, L)
pieces = O.synth_pieces(seed, 0, n, M)
acts = [O.synth_actions(seed, 0, n, t) for t in range(400)]

g = ref.Tetris(1, 1, warm_reset=False)
g.L, g.M = L, M
moves = 0
t0 = time.perf_counter()
for b in range(n):
    cfg = b
    g.board = ((rows[cfg][:, None] >> np.arange(10)) & 1).astype(bool)
    g.pieces = pieces[cfg].tolist()
    g.lines_cleared, g.moves_used, g.state = 0, 0, None
    for t in range(400):
        a = int(acts[t][b])
        g.move(a // 10, a % 10)
        moves += 1
        if g.state is not None:                       # reset from the same pool, like the batched environment
            cfg = (cfg + 1) % n
            g.board = ((rows[cfg][:, None] >> np.arange(10)) & 1).astype(bool)
            g.pieces = pieces[cfg].tolist()
            g.lines_cleared, g.moves_used, g.state = 0, 0, None
ref_rate = moves / (time.perf_counter() - t0)
done, sec = O.bench_run(seed, 65536, L, M, 200, 1)
npb = NP.bench(seed, n, L, M, 3.0)
print(f"reference Tetris.move: {ref_rate:,.0f} moves/s on one core ({moves} moves)")
print(f"NumPy per-board port:  {npb['moves_per_s']:,.0f} moves/s on one core")
print(f"oracle (C port):       {done / sec:,.0f} env-steps/s on one core")
print(f"ratio oracle / reference = {done / sec / ref_rate:,.0f}; numpy port / reference = {npb['moves_per_s'] / ref_rate:.2f}")
out = {"where": "build container (not the GPU box)", "cores": 1, "workload": f"synthetic L={L} M={M}, move; reset when finished",
       "reference_moves_per_s": ref_rate, "numpy_port_moves_per_s": npb["moves_per_s"], "c_port_env_steps_per_s": done / sec,
       "c_port_over_reference": done / sec / ref_rate, "numpy_port_over_reference": npb["moves_per_s"] / ref_rate}
with open(os.path.join(ROOT, "tests", "golden", "reference_timing.json"), "w") as fh:
    json.dump(out, fh, indent=1)
