#!/usr/bin/env python3
"""Golden fixtures for the reference's forward generator + solver (game/tetris_algo_main/).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_forward.py

For each seed the reference's own TetrisGameGenerator(seed, goal, tetrominoes, initial_height_max=4) and
TetrisSolver(board, sequence, goal, max_attempts=1000).solve() are run as generate_batch does (main.py:35-74) and
their outputs stored: board, sequence (letters as indices into 'IJLOSTZ'), verdict, failed-attempt count and the
solver's stack of (letter, rotation, column).

  forward_L5_M20.npz  seeds 0..99 (the reference's own range)   forward_L3_M20.npz  seeds 0..49
  forward_L10_M40.npz seeds 0..39
"""
import os
import sys
import time

import numpy as np

REF = os.environ.get("TPL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "game"))
os.chdir("/tmp")

from tetris_algo_main.TetrisGameGenerator import TetrisGameGenerator  # noqa: E402
from tetris_algo_main.TetrisSolver import TetrisSolver  # noqa: E402

LETTERS = "IJLOSTZ"


def make(L, M, seeds):
    n = len(seeds)
    rows = np.zeros((n, 20), np.uint16)
    seq = np.zeros((n, M), np.uint8)
    winnable = np.zeros(n, np.uint8)
    failed = np.zeros(n, np.int32)
    stack = np.zeros((n, M, 3), np.uint8)
    stack_len = np.zeros(n, np.int32)
    t0 = time.time()
    for k, seed in enumerate(seeds):
        g = TetrisGameGenerator(seed=seed, goal=L, tetrominoes=M, initial_height_max=4)
        rows[k] = (g.board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1)
        seq[k] = [LETTERS.index(c) for c in g.sequence]
        res, st, fa = TetrisSolver(g.board, g.sequence, g.goal, max_attempts=1000).solve()
        winnable[k], failed[k] = bool(res), fa
        if res:
            stack_len[k] = len(st)
            stack[k, : len(st)] = [(LETTERS.index(c), r, col) for c, r, col in st]
    np.savez_compressed(os.path.join(HERE, f"forward_L{L}_M{M}.npz"), L=L, M=M, seeds=np.array(seeds, np.uint64), rows=rows,
                        sequence=seq, winnable=winnable, failed_attempts=failed, stack=stack, stack_len=stack_len)
    print(f"L={L} M={M}: {int(winnable.sum())}/{n} winnable in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    make(5, 20, list(range(100)))
    make(3, 20, list(range(50)))
    make(10, 40, list(range(40)))
