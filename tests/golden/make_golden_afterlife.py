#!/usr/bin/env python3
"""afterlife.npz -- what the REFERENCE does with a game after it has ended, and with the counters across reset().

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_afterlife.py

Two behaviours of game/tetris.py that the batched environment deliberately does not have (it freezes a finished board and
zeroes the counters at a reset) and that `tetris_piclim.Tetris(..., reference_quirks=True)` reproduces:
  * move() never looks at `state` (:354-422): a finished game keeps popping pieces, locking them, counting moves and lines,
    and `state` is overwritten only where the code assigns it -- a won game can turn lost, a lost one won;
    when the piece list is empty `self.pieces.pop(0)` raises IndexError (:356);
  * reset() (:438-449) replaces board and pieces and leaves lines_cleared / moves_used / state as they were.
Every case: a sequence of games (board, pieces) handed to the reference's own reset() through its warm-reset queue (a stand-in
queue object: load_warm_reset() just calls queue.get(), :445-447), each played with seeded random moves for `plays` moves or
until the pieces run out, state recorded after every event.  Data only: inputs and the reference's outputs."""
import os
import random
import sys

import numpy as np

REF = os.environ.get("TPL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "game"))
sys.path.insert(0, ROOT)
os.chdir("/tmp")

import tetris as ref  # noqa: E402  (the reference module)
from oracle import oracle as O  # noqa: E402  (synthetic boards / piece lists only)

STATE = {None: 0, True: 1, False: 2}
RESET, MOVE, MOVE_RAISED = 0, 1, 2


def rows_of(board):
    return (board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(axis=1).astype(np.uint16)


def board_of(rows):
    return ((np.asarray(rows, dtype=np.uint16)[:, None] >> np.arange(10)) & 1).astype(bool)


class Handed:
    """Stands in for the multiprocessing queue of the warm reset: get() hands out the next prepared game."""

    def __init__(self, games):
        self.games = list(games)

    def get(self):
        return self.games.pop(0)


def case(L, M, games, plays, rng):
    """games: [(rows u16[20], pieces u8[M+1])]; the reference object is built once and reset() through its queue.  plays[k]: how many
    seeded random moves game k gets, or the list of (rotations, location) to play."""
    g = ref.Tetris(1, 1, warm_reset=False)
    g.L, g.M = L, M
    g.lines_cleared, g.moves_used, g.state = 0, 0, None
    g.warm_reset, g.queue = True, Handed([(board_of(r).copy(), [int(p) for p in ps]) for r, ps in games])
    events = []
    for k in range(len(games)):
        g.reset()                                                    # the reference's own reset(): counters untouched
        events.append((RESET, 0, 0, rows_of(g.board), g.lines_cleared, g.moves_used, STATE[g.state], len(g.pieces)))
        for step in range(plays[k] if isinstance(plays[k], int) else len(plays[k])):
            rot, loc = (rng.randint(0, 8), rng.randint(0, 10)) if isinstance(plays[k], int) else plays[k][step]
            try:
                g.move(rot, loc)
                kind = MOVE
            except IndexError:                                       # :356 pop from an empty list
                kind = MOVE_RAISED
            events.append((kind, rot, loc, rows_of(g.board), g.lines_cleared, g.moves_used, STATE[g.state], len(g.pieces)))
    g.warm_reset = False                                             # nothing to terminate
    return events


def main():
    rng = random.Random(20261005)
    out = {}
    cases = []
    # (L, M, number of games, moves played per game): past the end of the piece list (M + 1 pieces) in several of them
    for ci, (L, M, n_games, per_game) in enumerate([(1, 4, 3, 7), (2, 6, 4, 9), (3, 12, 3, 15), (5, 20, 3, 24), (10, 40, 2, 44),
                                                    (1, 1, 5, 3), (4, 8, 6, 5), (2, 30, 2, 33)]):
        rows = O.synth_boards(700 + ci, 0, n_games, L)
        pieces = O.synth_pieces(700 + ci, 0, n_games, M)
        if ci % 2 == 0:                                              # an empty board now and then: long afterlives, line clears
            rows[0] = 0
        ev = case(L, M, list(zip(rows, pieces)), [per_game] * n_games, rng)
        out[f"c{ci}_L"], out[f"c{ci}_M"] = np.int32(L), np.int32(M)
        out[f"c{ci}_rows0"], out[f"c{ci}_pieces"] = rows.astype(np.uint16), pieces.astype(np.uint8)
        out[f"c{ci}_kind"] = np.array([e[0] for e in ev], np.uint8)
        out[f"c{ci}_action"] = np.array([(e[1], e[2]) for e in ev], np.int32)
        out[f"c{ci}_rows"] = np.array([e[3] for e in ev], np.uint16)
        for name, col in (("lines", 4), ("moves", 5), ("state", 6), ("pieces_left", 7)):
            out[f"c{ci}_{name}"] = np.array([e[col] for e in ev], np.int32)
        cases.append(ev)
    # composed games: a WON game that goes on and turns lost at the move limit; a game LOST at the limit that goes on and wins;
    # a top-out after a win; each followed by a reset() that inherits the counters
    I, O_ = 0, 6
    almost = np.zeros(20, np.uint16)
    almost[19] = 0x3FE                                               # bottom row full but column 0
    two = almost.copy()
    two[18] = 0x3FE
    tall = np.zeros(20, np.uint16)
    tall[1:] = 0x001                                                 # column 0 filled up to row 1: the next piece there tops out
    composed = [
        (1, 3, [(almost, [I, O_, O_, O_]), (almost, [I, I, I, I])],
         [[(1, 0), (0, 2), (0, 4), (0, 6)], [(1, 0), (1, 0), (1, 5)]]),          # win at move 1, lost at move 3; then win again with carried counters
        (2, 1, [(two, [I, O_]), (two, [I, I])], [[(0, 4), (1, 0)], [(1, 0), (1, 3)]]),   # lost at the limit after one move, then a double clear: won
        (1, 5, [(almost, [I, I, O_, O_, O_, O_]), (tall, [O_, O_, O_, O_, O_, O_])],
         [[(1, 0), (1, 0), (0, 3)], [(0, 0), (0, 4), (0, 0)]]),                  # won; reset keeps True; a top-out turns it False
    ]
    for L, M, games, plays in composed:
        ci = len(cases)
        rows = np.array([g[0] for g in games], np.uint16)
        pieces = np.array([g[1] for g in games], np.uint8)
        assert pieces.shape[1] == M + 1
        ev = case(L, M, list(zip(rows, pieces)), plays, rng)
        out[f"c{ci}_L"], out[f"c{ci}_M"] = np.int32(L), np.int32(M)
        out[f"c{ci}_rows0"], out[f"c{ci}_pieces"] = rows, pieces
        out[f"c{ci}_kind"] = np.array([e[0] for e in ev], np.uint8)
        out[f"c{ci}_action"] = np.array([(e[1], e[2]) for e in ev], np.int32)
        out[f"c{ci}_rows"] = np.array([e[3] for e in ev], np.uint16)
        for name, col in (("lines", 4), ("moves", 5), ("state", 6), ("pieces_left", 7)):
            out[f"c{ci}_{name}"] = np.array([e[col] for e in ev], np.int32)
        cases.append(ev)
    out["n"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(HERE, "afterlife.npz"), **out)
    # what the fixture exercises (printed, so that a regeneration shows it still does)
    flips = raised = carried = past_m = 0
    for ev in cases:
        for a, b in zip(ev, ev[1:]):
            if b[0] == RESET and (a[4] or a[5] or a[6]):
                carried += 1
            if b[0] == MOVE and a[6] and b[6] and a[6] != b[6]:
                flips += 1
            raised += b[0] == MOVE_RAISED
    for ci, ev in enumerate(cases):
        past_m += sum(1 for e in ev if e[5] > int(out[f"c{ci}_M"]))
    print(f"afterlife.npz: {len(cases)} cases, {sum(len(e) for e in cases)} events; terminal state flipped {flips} times, "
          f"IndexError {raised} times, counters carried across {carried} resets, moves_used beyond M in {past_m} events")


if __name__ == "__main__":
    main()
