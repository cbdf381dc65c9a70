#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE implementation.

Run only where the reference checkout exists (it never travels to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

The reference's `game/tetris.py` is imported as-is; every expected value stored in the
fixtures is an output of its `Tetris.move` / `get_tetromino` / carving generator.  The fixtures are data
only: inputs (boards, piece lists, actions) and the reference's outputs.  The synthetic inputs of F3 come
from the oracle's counter-based generator (oracle/tetris_oracle.c), and are stored alongside the outputs
so the fixtures are self-contained.

Fixtures (all .npz, little-endian, a few hundred KB in total):
  F0 shapes.npz          the 7x4 (piece, rotations) table incl. the modulo behaviour of get_tetromino
  F1 plumbing.npz        BASELINE config 1: empty board, fixed pieces, 40 seeded moves; plus an all-O tiling
  F2 carved_L*_M*.npz    carved (board, pieces, solution) triples and the per-step replay
  F3 synthetic_*.npz     first 1024 boards of the synthetic sets stepped for M moves (freeze when finished)
  F4 edges.npz           edge cases (overhang top-out, clamp, rotation modulo, win on last move, ...)
  F5 random_moves.npz    4096 random single moves on random boards
  F6 afterlife.npz       (make_golden_afterlife.py) finished games played on, counters carried across reset()
"""
import os
import random
import sys
import time

import numpy as np

REF = os.environ.get("TPL_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "game"))
sys.path.insert(0, ROOT)
os.chdir("/tmp")  # the reference tree is read-only; keep any stray output away from it

import tetris as ref  # noqa: E402  (the reference module)
from oracle import oracle as O  # noqa: E402  (synthetic inputs + self-check only)

STATE = {None: 0, True: 1, False: 2}


def rows_of(board):
    """bool 20x10 -> u16[20], bit x = column x."""
    return (board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(axis=1).astype(np.uint16)


def board_of(rows):
    return ((np.asarray(rows, dtype=np.uint16)[:, None] >> np.arange(10)) & 1).astype(bool)


_template = None


def ref_game(L, M, rows, pieces, lines=0, moves=0, state=None):
    """A reference Tetris object forced into a given position (SURVEY 8c recipe)."""
    global _template
    g = ref.Tetris(1, 1, warm_reset=False)  # cheap carve, then overwrite everything
    g.L, g.M = L, M
    g.board = board_of(rows).copy()
    g.pieces = [int(p) for p in pieces]
    g.lines_cleared, g.moves_used, g.state = lines, moves, state
    return g


def snap(g):
    return rows_of(g.board), g.lines_cleared, g.moves_used, STATE[g.state], len(g.pieces)


def trace(g, actions):
    """Apply the actions with Tetris.move, recording the state after every move."""
    T = len(actions)
    out = dict(rows=np.zeros((T, 20), np.uint16), lines=np.zeros(T, np.int32), moves=np.zeros(T, np.int32),
               state=np.zeros(T, np.uint8), pieces_left=np.zeros(T, np.int32))
    for t, (rot, loc) in enumerate(actions):
        g.move(int(rot), int(loc))
        r, li, mo, st, pl = snap(g)
        out["rows"][t], out["lines"][t], out["moves"][t], out["state"][t], out["pieces_left"][t] = r, li, mo, st, pl
    return out


def fnv(rows):
    h = 0xcbf29ce484222325
    for r in rows:
        h = ((h ^ int(r)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


# ----------------------------------------------------------------------------------------------- F0
def make_shapes():
    h = np.zeros((7, 4), np.uint8)
    w = np.zeros((7, 4), np.uint8)
    mask = np.zeros((7, 4, 4), np.uint8)
    topo = np.zeros((7, 4, 4), np.uint8)
    nrot = np.array([len(t) for t in ref.tetrominos], np.uint8)
    for p in range(7):
        for r in range(4):
            m, rt = ref.get_tetromino(p, r)
            h[p, r], w[p, r] = m.shape
            for i in range(m.shape[0]):
                mask[p, r, i] = sum(int(m[i, x]) << x for x in range(m.shape[1]))
            topo[p, r, : len(rt)] = rt
    # the modulo behaviour for large rotation counts
    big = np.array([[ref.get_tetromino(p, r)[0].shape for r in range(4, 12)] for p in range(7)], np.uint8)
    np.savez_compressed(os.path.join(HERE, "shapes.npz"), h=h, w=w, mask=mask, revtopo=topo, nrot=nrot, big_hw=big)


# ----------------------------------------------------------------------------------------------- F1
def make_plumbing():
    L, M = 10, 40
    pieces = [i % 7 for i in range(M + 1)]
    rng = random.Random(123)
    actions = [(rng.randint(0, 3), rng.randint(0, 9)) for _ in range(M)]
    g = ref_game(L, M, np.zeros(20, np.uint16), pieces)
    # config 1 stops at the first terminal state like a caller would; record the full 40 anyway but note
    # how many moves were made before the game finished
    tr = {}
    T = 0
    recs = []
    for a in actions:
        if g.state is not None:
            break
        g.move(*a)
        recs.append(snap(g))
        T += 1
    tr["rows"] = np.array([r[0] for r in recs], np.uint16)
    tr["lines"] = np.array([r[1] for r in recs], np.int32)
    tr["moves"] = np.array([r[2] for r in recs], np.int32)
    tr["state"] = np.array([r[3] for r in recs], np.uint8)
    tr["pieces_left"] = np.array([r[4] for r in recs], np.int32)

    # all-O tiling: loc cycles 0,2,4,6,8 -> every 5 pieces clear 2 lines; survives all 40 moves
    L2, M2 = 20, 40
    pieces2 = [6] * (M2 + 1)
    actions2 = [(0, (2 * i) % 10) for i in range(M2)]
    g2 = ref_game(L2, M2, np.zeros(20, np.uint16), pieces2)
    tr2 = trace(g2, actions2)
    np.savez_compressed(os.path.join(HERE, "plumbing.npz"),
                        L=L, M=M, pieces=np.array(pieces, np.uint8), actions=np.array(actions[:T], np.uint8),
                        all_actions=np.array(actions, np.uint8),
                        **{"t_" + k: v for k, v in tr.items()},
                        o_L=L2, o_M=M2, o_pieces=np.array(pieces2, np.uint8), o_actions=np.array(actions2, np.uint8),
                        **{"o_" + k: v for k, v in tr2.items()})
    print(f"F1: random trace finished after {T} moves in state {tr['state'][-1]}; "
          f"O tiling: lines={tr2['lines'][-1]} state={tr2['state'][-1]}")


# ----------------------------------------------------------------------------------------------- F2
def make_carved(L, M, count):
    rows = np.zeros((count, 20), np.uint16)
    pieces = np.zeros((count, M + 1), np.uint8)
    sol_len = np.zeros(count, np.int32)
    sol = np.zeros((count, M, 2), np.uint8)
    # replay (reference's test_carving_invertability, game/main.py:49-57), recorded per step
    r_rows = np.zeros((count, M, 20), np.uint16)
    r_lines = np.zeros((count, M), np.uint8)
    r_moves = np.zeros((count, M), np.uint8)
    r_state = np.zeros((count, M), np.uint8)
    t0 = time.time()
    for k in range(count):
        random.seed(1000 * L + k)
        g = ref.Tetris(L, M, warm_reset=False, debug=True)
        assert len(g.pieces) == M + 1
        rows[k] = rows_of(g.board)
        pieces[k] = g.pieces
        s = list(g.solution)
        sol_len[k] = len(s)
        sol[k, : len(s)] = s
        g.lines_cleared, g.moves_used, g.state = 0, 0, None
        for t, (rot, loc) in enumerate(s):
            g.move(rot, loc)
            r_rows[k, t], r_lines[k, t], r_moves[k, t], r_state[k, t] = rows_of(g.board), g.lines_cleared, g.moves_used, STATE[g.state]
        assert g.state is True, "reference property: replaying a carved solution wins"
    np.savez_compressed(os.path.join(HERE, f"carved_L{L}_M{M}.npz"), L=L, M=M, rows=rows, pieces=pieces,
                        sol_len=sol_len, sol=sol, r_rows=r_rows, r_lines=r_lines, r_moves=r_moves, r_state=r_state)
    print(f"F2: {count} carved games L={L} M={M} in {time.time() - t0:.1f}s, mean solution {sol_len.mean():.1f} moves")


# ----------------------------------------------------------------------------------------------- F3
def make_synthetic(name, L, M, count, seed=0):
    rows = O.synth_boards(seed, 0, count, L)
    pieces = O.synth_pieces(seed, 0, count, M)
    actions = np.stack([O.synth_actions(seed, 0, count, t) for t in range(M)])  # [M][count]
    f_rows = np.zeros((count, 20), np.uint16)
    f_lines = np.zeros(count, np.uint8)
    f_moves = np.zeros(count, np.uint8)
    f_state = np.zeros(count, np.uint8)
    f_left = np.zeros(count, np.uint8)
    hashes = np.zeros((M, count), np.uint64)
    s_state = np.zeros((M, count), np.uint8)
    s_lines = np.zeros((M, count), np.uint8)
    s_moves = np.zeros((M, count), np.uint8)
    for b in range(count):
        g = ref_game(L, M, rows[b], pieces[b])
        for t in range(M):
            if g.state is None:  # freeze once finished (build rule); the reference is only ever stepped while running
                a = int(actions[t, b])
                g.move(a // 10, a % 10)
            r = rows_of(g.board)
            hashes[t, b] = fnv(r)
            s_state[t, b], s_lines[t, b], s_moves[t, b] = STATE[g.state], g.lines_cleared, g.moves_used
        f_rows[b], f_lines[b], f_moves[b], f_state[b], f_left[b] = snap(g)
    np.savez_compressed(os.path.join(HERE, f"synthetic_{name}.npz"), L=L, M=M, seed=seed, rows=rows, pieces=pieces,
                        actions=actions, f_rows=f_rows, f_lines=f_lines, f_moves=f_moves, f_state=f_state,
                        f_pieces_left=f_left, hashes=hashes, s_state=s_state, s_lines=s_lines, s_moves=s_moves)
    print(f"F3 {name}: states {np.bincount(f_state, minlength=3)} mean lines {f_lines.mean():.3f} mean moves {f_moves.mean():.2f}")


# ----------------------------------------------------------------------------------------------- F4
def make_edges():
    cases = []

    def case(name, L, M, rows, pieces, actions, lines=0, moves=0):
        g = ref_game(L, M, rows, pieces, lines, moves)
        tr = trace(g, actions)
        cases.append(dict(name=name, L=L, M=M, rows0=np.array(rows, np.uint16), pieces=np.array(pieces, np.uint8),
                          actions=np.array(actions, np.uint8), lines0=lines, moves0=moves, **tr))

    Z = [0] * 20
    # overhang: a cell in row 0 under a piece column forces top-out even when the cells would not overlap
    r = list(Z); r[0] = 1
    case("overhang_topout_J1", 10, 40, r, [2, 0, 0], [(1, 0)])
    # a piece cannot slide under an overhang: I horizontal over a roof at row 10
    r = list(Z); r[10] = 0b0000001111
    case("roof_blocks", 10, 40, r, [0, 0, 0], [(0, 0), (0, 0)])
    # full row outside the piece's rows survives
    r = list(Z); r[19] = 0x3FF; r[18] = 0x3FE
    case("full_row_elsewhere_survives", 10, 40, r, [6, 0, 0], [(0, 4)])
    # and is not counted when another row clears
    r = list(Z); r[19] = 0x3FF; r[18] = 0x3FE; r[17] = 0
    case("clear_above_full_row", 10, 40, r, [0, 0, 0], [(1, 0)])
    # clamp: location 10 (and 9 for wide pieces) clamps to 10 - w
    case("clamp_I_loc10", 10, 40, Z, [0, 0, 0], [(0, 10), (0, 9)])
    case("clamp_O_loc9", 10, 40, Z, [6, 6, 6], [(0, 9), (0, 10)])
    case("clamp_T_loc8", 10, 40, Z, [3, 3, 3], [(0, 8), (1, 9)])
    # rotation modulo: I rot 3 == rot 1, O any rot, S rot 2 == rot 0
    case("rot_mod", 10, 40, Z, [0, 6, 4, 5, 1, 2, 3, 0], [(3, 0), (3, 2), (2, 4), (3, 7), (7, 4), (6, 0), (5, 8)])
    # win on the last allowed move is a win (win tested before the move limit)
    r = list(Z); r[19] = 0x3FF & ~0b1111
    case("win_on_last_move", 1, 1, r, [0, 0], [(0, 0)])
    # move limit without a clear loses
    case("lose_on_move_limit", 1, 2, Z, [6, 6, 6], [(0, 0), (0, 2)])
    # clear that does not reach L at the move limit loses
    r = list(Z); r[19] = 0x3FF & ~0b1111
    case("clear_but_move_limit", 2, 1, r, [0, 0], [(0, 0)])
    # top-out consumes the piece but not a move
    r = [0x001] * 20
    case("topout_consumes_piece", 10, 40, r, [0, 1, 2], [(1, 0)])
    # four-line clear with the vertical I
    r = list(Z)
    for y in range(16, 20):
        r[y] = 0x3FE
    case("tetris_4_lines", 4, 40, r, [0, 0], [(1, 0)])
    # non-adjacent double clear: rows 17 and 19 complete, row 18 not
    r = list(Z); r[19] = 0x3FE; r[18] = 0x1FE; r[17] = 0x3FE; r[16] = 0x3FE
    case("split_clear", 10, 40, r, [0, 0], [(1, 0)])
    # piece resting exactly at the top row (drop == 0) is legal
    r = [0] * 20
    for y in range(4, 20):
        r[y] = 0x001
    case("drop_zero_ok", 10, 40, r, [0, 0], [(1, 0)])
    # moves on a full-height stack next to an empty well
    r = [0x1FF] * 20
    for y in range(0, 16):
        r[y] = 0
    case("deep_well", 10, 40, r, [0, 0, 0], [(1, 9), (0, 0)])
    # counters carried in from a mid-game position
    r = list(Z); r[19] = 0x3FE
    case("midgame_counters", 5, 10, r, [0, 0, 0], [(1, 0)], lines=4, moves=9)

    out = {"n": len(cases), "names": np.array([c["name"] for c in cases])}
    for i, c in enumerate(cases):
        for k, v in c.items():
            if k != "name":
                out[f"c{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "edges.npz"), **out)
    for c in cases:
        print(f"F4 {c['name']:32s} state={c['state'].tolist()} lines={c['lines'].tolist()} moves={c['moves'].tolist()}")


# ----------------------------------------------------------------------------------------------- F5
def random_move_cases(count, seed):
    rng = np.random.default_rng(seed)
    rows = np.zeros((count, 20), np.uint16)
    for b in range(count):
        height = int(rng.integers(0, 21))
        dens = rng.uniform(0.2, 0.95)
        cells = rng.random((20, 10)) < dens
        cells[: 20 - height] = False
        # a few nearly-full rows so clears actually happen
        for y in range(20 - height, 20):
            if rng.random() < 0.5:
                cells[y] = True
                cells[y, rng.integers(0, 10, size=int(rng.integers(1, 3)))] = False
        rows[b] = rows_of(cells)
    piece = rng.integers(0, 7, count).astype(np.uint8)
    rot = rng.integers(0, 8, count).astype(np.uint8)
    loc = rng.integers(0, 11, count).astype(np.uint8)
    L = rng.integers(1, 6, count).astype(np.uint8)
    M = rng.integers(1, 6, count).astype(np.uint8)
    lines0 = rng.integers(0, 3, count).astype(np.uint8)
    moves0 = np.minimum(rng.integers(0, 5, count), M - 1).astype(np.uint8)
    return rows, piece, rot, loc, L, M, lines0, moves0


def run_reference_moves(rows, piece, rot, loc, L, M, lines0, moves0):
    n = len(piece)
    o_rows = np.zeros((n, 20), np.uint16)
    o_lines = np.zeros(n, np.uint8); o_moves = np.zeros(n, np.uint8); o_state = np.zeros(n, np.uint8)
    g = ref_game(1, 1, rows[0], [0, 0])
    for b in range(n):
        g.L, g.M = int(L[b]), int(M[b])
        g.board = board_of(rows[b]).copy()
        g.pieces = [int(piece[b]), 0]
        g.lines_cleared, g.moves_used, g.state = int(lines0[b]), int(moves0[b]), None
        g.move(int(rot[b]), int(loc[b]))
        o_rows[b], o_lines[b], o_moves[b], o_state[b], _ = snap(g)
    return o_rows, o_lines, o_moves, o_state


def make_random_moves():
    inp = random_move_cases(4096, 7)
    out = run_reference_moves(*inp)
    keys_in = ("rows", "piece", "rot", "loc", "L", "M", "lines0", "moves0")
    np.savez_compressed(os.path.join(HERE, "random_moves.npz"), **dict(zip(keys_in, inp)),
                        o_rows=out[0], o_lines=out[1], o_moves=out[2], o_state=out[3])
    print(f"F5: states {np.bincount(out[3], minlength=3)} cleared-any {(out[1] > inp[6]).mean():.3f}")


def selfcheck_oracle(count=100_000):
    """the oracle against the imported reference on >= 1e5 random tuples."""
    inp = random_move_cases(count, 99)
    ref_out = run_reference_moves(*inp)
    rows, piece, rot, loc, L, M, lines0, moves0 = inp
    bad = 0
    for b in range(count):
        g = O.Game(int(L[b]), int(M[b]), rows[b], [int(piece[b]), 0], int(lines0[b]), int(moves0[b]))
        g.move(int(rot[b]), int(loc[b]))
        ok = (np.array_equal(g.rows, ref_out[0][b]) and g.lines_cleared == ref_out[1][b]
              and g.moves_used == ref_out[2][b] and g.state == ref_out[3][b])
        bad += not ok
    print(f"oracle self-check vs reference: {count - bad}/{count} random single moves identical")
    assert bad == 0


if __name__ == "__main__":
    t0 = time.time()
    make_shapes()
    make_plumbing()
    make_edges()
    make_random_moves()
    make_synthetic("L5_M20", 5, 20, 1024)
    make_synthetic("L10_M40", 10, 40, 1024)
    make_carved(5, 20, 256)
    make_carved(10, 40, 256)
    if "--no-selfcheck" not in sys.argv:
        selfcheck_oracle()
    print(f"done in {time.time() - t0:.1f}s")
