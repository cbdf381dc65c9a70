#!/usr/bin/env python3
"""Rate of the REFERENCE's Tetris.move in this container, beside the CPU oracle on the same workload, one core each.

Runs only where /root/reference exists.  Workload: the synthetic set of bench.py (L=10, M=40), move; reset when
finished -- the loop of game/performance_test.py:13-17.  The ratio lets the GPU box's cpu_baseline (the oracle)
be read in reference-equivalent terms."""
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
print(f"reference Tetris.move: {ref_rate:,.0f} moves/s on one core ({moves} moves)")
print(f"oracle (C port):       {done / sec:,.0f} env-steps/s on one core")
print(f"ratio oracle / reference = {done / sec / ref_rate:,.0f}")
