"""NumPy per-board restatement of the reference's `Tetris.move` -- TEST INFRASTRUCTURE ONLY.

The second CPU leg SURVEY.md 8(d)(ii) names: one board as a 20x10 bool array, advanced with the same NumPy
operation sequence the reference uses (per-column np.where for the tops, a slice |= for the lock, np.all on the
piece's rows, a row gather + vstack for the compaction), so its speed on a host core is the reference's speed on that
core to within noise -- the reference itself cannot travel to the GPU box.  `bench.py`'s cpu_baseline leg times it
beside the C port (oracle/tetris_oracle.c); tests/test_oracle_golden.py pins it to the same golden vectors.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Citations are lines of game/tetris.py in the upstream repo.
"""
from __future__ import annotations

import json
import sys
import time

import numpy as np

ROWS, COLS = 20, 10

# `tetrominos` (:23-57) as (row masks top->bottom with bit x = mask column x, reverse topography); ids I0 L1 J2 T3 S4
# Z5 O6 (:8-16).  Written from the table in SURVEY.md 8(a); tests check it against the shapes.npz fixture.
_TABLE = (
    (((15,), (0, 0, 0, 0)), ((1, 1, 1, 1), (3,))),
    (((4, 7), (1, 1, 1)), ((3, 2, 2), (0, 2)), ((7, 1), (1, 0, 0)), ((1, 1, 3), (2, 2))),
    (((1, 7), (1, 1, 1)), ((2, 2, 3), (2, 2)), ((7, 4), (0, 0, 1)), ((3, 1, 1), (2, 0))),
    (((2, 7), (1, 1, 1)), ((2, 3, 2), (1, 2)), ((7, 2), (0, 1, 0)), ((1, 3, 1), (2, 1))),
    (((6, 3), (1, 1, 0)), ((1, 3, 2), (1, 2))),
    (((3, 6), (0, 1, 1)), ((2, 3, 1), (2, 1))),
    (((3, 3), (1, 1)),),
)


def _entry(masks, topo):
    w = len(topo)
    cells = np.array([[(m >> x) & 1 for x in range(w)] for m in masks], dtype=bool)
    return cells, np.array(topo)


SHAPES = tuple(tuple(_entry(m, t) for m, t in rots) for rots in _TABLE)


def get_tetromino(piece: int, rotations: int):
    """get_tetromino (:60-61): the rotation count is taken modulo the piece's number of rotations."""
    rots = SHAPES[piece]
    return rots[rotations % len(rots)]


class Board:
    """One game with the reference's attributes (:143-151, 186-187): board bool[20,10], pieces list, counters, state."""

    def __init__(self, L, M, board=None, pieces=(), lines_cleared=0, moves_used=0):
        self.L, self.M = L, M
        self.board = np.zeros((ROWS, COLS), dtype=bool) if board is None else np.array(board, dtype=bool)
        self.pieces = list(pieces)
        self.lines_cleared, self.moves_used, self.state = lines_cleared, moves_used, None

    @staticmethod
    def cells_of(rows) -> np.ndarray:
        return ((np.asarray(rows, dtype=np.uint16)[:, None] >> np.arange(COLS)) & 1).astype(bool)

    def rows(self) -> np.ndarray:
        return (self.board.astype(np.uint16) << np.arange(COLS, dtype=np.uint16)).sum(1).astype(np.uint16)

    def move(self, rotations: int, location: int) -> None:
        """Tetris.move (:354-422) with calculate_drop_deltas (:427-433) and calculate_drop (:424-425)."""
        piece = self.pieces.pop(0)                                          # :356 consumed before anything can fail
        cells, topo = get_tetromino(piece, rotations)                       # :359-360
        h, w = cells.shape
        location = min(location, COLS - w)                                  # :363-364 right clamp only
        tops = []
        for x in range(location, location + w):                             # :429-431 first filled row, 20 if none
            filled = np.where(self.board[:, x])[0]
            tops.append(filled[0] if filled.size else ROWS)
        drop = int(np.min(np.array(tops) - topo)) - 1                       # :433, :424-425
        if drop < 0:                                                        # :372-374 top-out, nothing else changes
            self.state = False
            return
        self.board[drop:drop + h, location:location + w] |= cells          # :377-378
        self.moves_used += 1                                                # :379
        full = np.all(self.board[drop:drop + h, :], axis=1)                 # :382-383 only the piece's rows
        cleared = int(np.count_nonzero(full))
        if cleared == 0:                                                    # :389-394
            if self.moves_used >= self.M:
                self.state = False
            return
        gone = set((drop + np.where(full)[0]).tolist())
        keep = [r for r in range(ROWS) if r not in gone]                    # :397-407 order-preserving compaction
        self.board = np.vstack((np.zeros((cleared, COLS), dtype=bool), self.board[keep]))
        self.lines_cleared += cleared                                       # :409
        if self.lines_cleared >= self.L:                                    # :415-417 win before the move limit
            self.state = True
        elif self.moves_used >= self.M:                                     # :420-422
            self.state = False


def bench(seed: int, boards: int, L: int, M: int, seconds: float) -> dict:
    """The loop of game/performance_test.py:13-17 (move; when finished, take the next configuration) on the synthetic
    workload of bench.py, for about `seconds` of wall time on ONE core.  Inputs come from the C oracle's generator."""
    from oracle import oracle as O
    rows = O.synth_boards(seed, 0, boards, L)
    pieces = O.synth_pieces(seed, 0, boards, M)
    steps = 64
    acts = np.stack([O.synth_actions(seed, 0, boards, t) for t in range(steps)])
    cells = [Board.cells_of(r) for r in rows]
    lists = [p.tolist() for p in pieces]
    moves, t0, b = 0, time.perf_counter(), 0
    deadline = t0 + seconds
    while time.perf_counter() < deadline:
        cfg = b % boards
        g = Board(L, M, cells[cfg], lists[cfg])
        for t in range(steps):
            a = int(acts[t, cfg])
            g.move(a // 10, a % 10)
            moves += 1
            if g.state is not None:
                cfg = (cfg + 1) % boards
                g = Board(L, M, cells[cfg], lists[cfg])
        b += 1
    sec = time.perf_counter() - t0
    return {"moves": moves, "seconds": sec, "moves_per_s": moves / sec}


if __name__ == "__main__":
    # python -m oracle.numpy_port SEED BOARDS L M SECONDS  -> one JSON line (bench.py starts one of these per host core)
    a = sys.argv[1:]
    print(json.dumps(bench(int(a[0]), int(a[1]), int(a[2]), int(a[3]), float(a[4]))), flush=True)
