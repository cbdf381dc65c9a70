#!/usr/bin/env python3
"""Golden fixtures for the prescribed-configuration supply (the reference's carving generator).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_carving.py

The reference's `game/tetris.py` is imported as-is; only its source of randomness is replaced: the module
global `random` becomes a recorder that forwards to a seeded `random.Random` and logs every decision as
(lo, hi, value).  `shuffle` is CPython's own algorithm (for i = n-1..1: j = randint(0, i); swap) expressed through
the recorded randint, so the tape alone determines the run.  For each game the fixture holds the tape and what
the reference produced from it: board, piece list, solution.  Driving a restatement of the generator with the
tape must reproduce them exactly -- that pins the generator's logic (carve, backtracking, padding), not just
its statistics.

  carving_L5_M20.npz  64 games     carving_L10_M40.npz  32 games     carving_L15_M40.npz  8 games
"""
import os
import random as _random
import sys
import time

import numpy as np

REF = os.environ.get("TPL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "game"))
os.chdir("/tmp")

import tetris as ref  # noqa: E402


class TapeRandom:
    def __init__(self, seed):
        self._r = _random.Random(seed)
        self.tape = []

    def randint(self, lo, hi):
        v = self._r.randint(lo, hi)
        self.tape.append((lo, hi, v))
        return v

    def shuffle(self, x):
        for i in reversed(range(1, len(x))):
            j = self.randint(0, i)
            x[i], x[j] = x[j], x[i]


def rows_of(board):
    return (board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(axis=1).astype(np.uint16)


def make(L, M, count):
    rows = np.zeros((count, 20), np.uint16)
    pieces = np.zeros((count, M + 1), np.uint8)
    sol = np.zeros((count, M, 2), np.uint8)
    sol_len = np.zeros(count, np.int32)
    tapes, offsets = [], [0]
    t0 = time.time()
    for k in range(count):
        shim = TapeRandom(5000 * L + k)
        ref.random = shim                      # the reference now draws every decision through the recorder
        g = ref.Tetris(L, M, warm_reset=False, debug=True)
        rows[k], pieces[k] = rows_of(g.board), g.pieces
        sol_len[k] = len(g.solution)
        sol[k, : len(g.solution)] = g.solution
        tapes.append(np.array(shim.tape, dtype=np.uint8).reshape(-1, 3))
        offsets.append(offsets[-1] + len(shim.tape))
        # the reference's own property (game/main.py:49-57)
        for r, l in g.solution:
            g.move(r, l)
        assert g.state is True
    ref.random = _random
    tape = np.concatenate(tapes)
    np.savez_compressed(os.path.join(HERE, f"carving_L{L}_M{M}.npz"), L=L, M=M, rows=rows, pieces=pieces, sol=sol,
                        sol_len=sol_len, tape=tape, offsets=np.array(offsets, np.int64))
    print(f"L={L} M={M}: {count} games, {offsets[-1]} decisions, {time.time() - t0:.1f}s")


if __name__ == "__main__":
    make(5, 20, 64)
    make(10, 40, 32)
    make(15, 40, 8)
