"""The NumPy per-board restatement (oracle/numpy_port.py, the second cpu_baseline leg) against the golden vectors
generated from the imported reference: F0 shape table, F1 plumbing traces, F4 edge cases, F5 random single moves, and
F2 carved replays (a subset; the C oracle covers all of them).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden

from oracle import numpy_port as NP

_STATE = {None: 0, True: 1, False: 2}


def _play(L, M, rows, pieces, actions, lines=0, moves=0):
    g = NP.Board(int(L), int(M), NP.Board.cells_of(rows), [int(p) for p in pieces], int(lines), int(moves))
    out = []
    for rot, loc in actions:
        g.move(int(rot), int(loc))
        out.append((g.rows(), g.lines_cleared, g.moves_used, _STATE[g.state], len(g.pieces)))
    return out


def test_f0_shape_table():
    f = load_golden("shapes.npz")
    for p in range(7):
        assert len(NP.SHAPES[p]) == f["nrot"][p]
        for r in range(12):
            cells, topo = NP.get_tetromino(p, r)
            rr = r % int(f["nrot"][p])
            h, w = int(f["h"][p, rr]), int(f["w"][p, rr])
            assert cells.shape == (h, w)
            masks = (cells.astype(int) << np.arange(w)).sum(1)
            assert masks.tolist() == f["mask"][p, rr, :h].tolist() and topo.tolist() == f["revtopo"][p, rr, :w].tolist()


@pytest.mark.parametrize("prefix", ["t_", "o_"])
def test_f1_plumbing(prefix):
    f = load_golden("plumbing.npz")
    if prefix == "t_":
        L, M, pieces, actions = f["L"], f["M"], f["pieces"], f["actions"]
    else:
        L, M, pieces, actions = f["o_L"], f["o_M"], f["o_pieces"], f["o_actions"]
    for t, (rows, lines, moves, state, left) in enumerate(_play(L, M, np.zeros(20, np.uint16), pieces, actions)):
        assert np.array_equal(rows, f[prefix + "rows"][t]), t
        assert (lines, moves, state, left) == (f[prefix + "lines"][t], f[prefix + "moves"][t], f[prefix + "state"][t],
                                               f[prefix + "pieces_left"][t]), t


def test_f4_edges():
    f = load_golden("edges.npz")
    for i in range(int(f["n"])):
        g = lambda k: f[f"c{i}_{k}"]
        got = _play(g("L"), g("M"), g("rows0"), g("pieces"), g("actions"), g("lines0"), g("moves0"))
        for t, (rows, lines, moves, state, left) in enumerate(got):
            name = str(f["names"][i])
            assert np.array_equal(rows, g("rows")[t]), (name, t)
            assert (lines, moves, state, left) == (g("lines")[t], g("moves")[t], g("state")[t], g("pieces_left")[t]), (name, t)


def test_f5_random_moves():
    f = load_golden("random_moves.npz")
    for b in range(f["rows"].shape[0]):
        (rows, lines, moves, state, _), = _play(f["L"][b], f["M"][b], f["rows"][b], [f["piece"][b], 0],
                                                [(f["rot"][b], f["loc"][b])], f["lines0"][b], f["moves0"][b])
        assert np.array_equal(rows, f["o_rows"][b]), b
        assert (lines, moves, state) == (f["o_lines"][b], f["o_moves"][b], f["o_state"][b]), b


def test_f2_carved_replay_wins():
    f = load_golden("carved_L10_M40.npz")
    L, M = int(f["L"]), int(f["M"])
    for k in range(0, f["rows"].shape[0], 8):
        n = int(f["sol_len"][k])
        got = _play(L, M, f["rows"][k], f["pieces"][k], f["sol"][k, :n])
        for t, (rows, lines, moves, state, _) in enumerate(got):
            assert np.array_equal(rows, f["r_rows"][k, t]), (k, t)
            assert (lines, moves, state) == (f["r_lines"][k, t], f["r_moves"][k, t], f["r_state"][k, t]), (k, t)
        assert got[-1][3] == 1 and got[-1][1] >= L


def test_bench_loop_runs():
    out = NP.bench(0, 64, 10, 40, 0.2)
    assert out["moves"] > 0 and out["moves_per_s"] > 1000
