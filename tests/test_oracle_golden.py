"""The CPU oracle against every golden vector generated from the imported reference (tests/golden/).

These run without a GPU.  They are what pins the oracle: F0 shape table, F1 plumbing traces, F2 carved
triples (the reference's own test_carving_invertability property, game/main.py:49-57), F3 synthetic sets,
F4 edge cases, F5 random single moves.
"""
import numpy as np
import pytest

from conftest import board_hashes, load_golden


def _play(O, L, M, rows, pieces, actions, lines=0, moves=0):
    g = O.Game(int(L), int(M), rows, pieces, int(lines), int(moves))
    out = []
    for rot, loc in actions:
        g.move(int(rot), int(loc))
        out.append((g.rows.copy(), g.lines_cleared, g.moves_used, g.state, len(g.pieces)))
    return out


def test_f0_shape_table(oracle):
    f = load_golden("shapes.npz")
    for p in range(7):
        assert oracle.num_rotations(p) == f["nrot"][p]
        for r in range(4):
            h, w, mask, topo = oracle.get_tetromino(p, r)
            assert (h, w) == (f["h"][p, r], f["w"][p, r])
            assert mask == f["mask"][p, r, :h].tolist()
            assert topo == f["revtopo"][p, r, :w].tolist()
        for k, r in enumerate(range(4, 12)):  # rotations % len
            h, w, _, _ = oracle.get_tetromino(p, r)
            assert (h, w) == tuple(f["big_hw"][p, k])


@pytest.mark.parametrize("prefix", ["t_", "o_"])
def test_f1_plumbing(oracle, prefix):
    f = load_golden("plumbing.npz")
    if prefix == "t_":
        L, M, pieces, actions = f["L"], f["M"], f["pieces"], f["actions"]
    else:
        L, M, pieces, actions = f["o_L"], f["o_M"], f["o_pieces"], f["o_actions"]
    got = _play(oracle, L, M, np.zeros(20, np.uint16), pieces, actions)
    for t, (rows, lines, moves, state, left) in enumerate(got):
        assert np.array_equal(rows, f[prefix + "rows"][t]), t
        assert (lines, moves, state, left) == (f[prefix + "lines"][t], f[prefix + "moves"][t], f[prefix + "state"][t],
                                               f[prefix + "pieces_left"][t]), t


@pytest.mark.parametrize("name", ["carved_L5_M20.npz", "carved_L10_M40.npz"])
def test_f2_carved_replay_wins(oracle, name):
    f = load_golden(name)
    L, M = int(f["L"]), int(f["M"])
    for k in range(f["rows"].shape[0]):
        n = int(f["sol_len"][k])
        got = _play(oracle, L, M, f["rows"][k], f["pieces"][k], f["sol"][k, :n])
        for t, (rows, lines, moves, state, _) in enumerate(got):
            assert np.array_equal(rows, f["r_rows"][k, t]), (k, t)
            assert (lines, moves, state) == (f["r_lines"][k, t], f["r_moves"][k, t], f["r_state"][k, t]), (k, t)
        assert got[-1][3] == 1 and got[-1][1] >= L  # replaying a carved solution wins


@pytest.mark.parametrize("name", ["synthetic_L5_M20.npz", "synthetic_L10_M40.npz"])
def test_f3_synthetic(oracle, name):
    f = load_golden(name)
    L, M, seed = int(f["L"]), int(f["M"]), int(f["seed"])
    n = f["rows"].shape[0]
    # the generator reproduces the stored inputs
    assert np.array_equal(oracle.synth_boards(seed, 0, n, L), f["rows"])
    assert np.array_equal(oracle.synth_pieces(seed, 0, n, M), f["pieces"])
    for t in range(M):
        assert np.array_equal(oracle.synth_actions(seed, 0, n, t), f["actions"][t])
    # batched oracle env (freeze rule) against the reference's per-step fingerprints
    env = oracle.Env(n, L, M, 0, seed)
    env.set_pool(f["rows"], f["pieces"])
    env.set_options(auto_reset=False, assign_mode=1)
    env.reset()
    for t in range(M):
        env.step(f["actions"][t])
        s = env.get_state()
        hashes = board_hashes(s["rows"])
        assert np.array_equal(hashes, f["hashes"][t]), t
        assert np.array_equal(s["state"], f["s_state"][t]) and np.array_equal(s["lines"], f["s_lines"][t])
        assert np.array_equal(s["moves"], f["s_moves"][t])
    s = env.get_state()
    assert np.array_equal(s["rows"], f["f_rows"]) and np.array_equal(s["pieces_left"], f["f_pieces_left"])
    assert np.array_equal(s["state"], f["f_state"])


def test_board_hash_helper_matches_c(oracle):
    rows = oracle.synth_boards(4, 0, 64, 10)
    assert np.array_equal(board_hashes(rows), np.array([oracle.board_hash(r) for r in rows], dtype=np.uint64))


def test_f4_edges(oracle):
    f = load_golden("edges.npz")
    for i in range(int(f["n"])):
        g = lambda k: f[f"c{i}_{k}"]
        got = _play(oracle, g("L"), g("M"), g("rows0"), g("pieces"), g("actions"), g("lines0"), g("moves0"))
        for t, (rows, lines, moves, state, left) in enumerate(got):
            name = str(f["names"][i])
            assert np.array_equal(rows, g("rows")[t]), (name, t)
            assert (lines, moves, state, left) == (g("lines")[t], g("moves")[t], g("state")[t], g("pieces_left")[t]), (name, t)


def test_f5_random_moves(oracle):
    f = load_golden("random_moves.npz")
    for b in range(f["rows"].shape[0]):
        g = oracle.Game(int(f["L"][b]), int(f["M"][b]), f["rows"][b], [int(f["piece"][b]), 0], int(f["lines0"][b]),
                        int(f["moves0"][b]))
        g.move(int(f["rot"][b]), int(f["loc"][b]))
        assert np.array_equal(g.rows, f["o_rows"][b]), b
        assert (g.lines_cleared, g.moves_used, g.state) == (f["o_lines"][b], f["o_moves"][b], f["o_state"][b]), b


def test_env_rules_freeze_autoreset_stats(oracle):
    """Build-defined batched rules: freeze, auto-reset, reward, statistics."""
    L, M, n = 5, 20, 512
    rows, pieces = oracle.synth_boards(3, 0, n, L), oracle.synth_pieces(3, 0, n, M)
    frozen = oracle.Env(n, L, M, 0, 3)
    frozen.set_pool(rows, pieces)
    frozen.set_options(auto_reset=False, assign_mode=1, per_line=2.0, win=10.0, lose=-1.0)
    frozen.reset()
    auto = oracle.Env(n, L, M, 0, 3)
    auto.set_pool(rows, pieces)
    auto.set_options(auto_reset=True, assign_mode=0, per_line=2.0, win=10.0, lose=-1.0)
    auto.reset()
    assert np.array_equal(frozen.get_state()["rows"], rows)        # sequential, episode 0 -> config i
    total_done = 0
    for t in range(3 * M):
        a = oracle.synth_actions(3, 0, n, t)
        before = frozen.get_state()
        r, d = frozen.step(a)
        after = frozen.get_state()
        was_done = before["state"] != 0
        assert np.all(d[was_done] == 1) and np.all(r[was_done] == 0)
        assert np.array_equal(after["rows"][was_done], before["rows"][was_done])
        newly = (~was_done) & (after["state"] != 0)
        assert np.all(d == (after["state"] != 0))
        lost = newly & (after["state"] == 2)
        cleared = after["lines"].astype(int) - before["lines"].astype(int)
        expect = 2.0 * cleared + np.where(newly & (after["state"] == 1), 10.0, 0.0) + np.where(lost, -1.0, 0.0)
        assert np.array_equal(r[~was_done], expect[~was_done].astype(np.float32))
        r2, d2 = auto.step(a)
        total_done += int(d2.sum())
        assert np.all(auto.get_state()["state"] == 0)              # finished boards restart in the same step
    st = auto.stats()
    assert st["episodes"] == total_done and st["episodes"] > n
    assert frozen.stats()["episodes"] == n and np.all(frozen.get_state()["state"] != 0)
    # masked reset starts the next episode of exactly the masked boards
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    frozen.reset(mask)
    s = frozen.get_state()
    assert np.all(s["state"][mask == 1] == 0) and np.all(s["state"][mask == 0] != 0)
    assert frozen.clock == 3 * M                                    # one tick per lockstep step since the full reset
    want = np.array([frozen.assign(b, frozen.clock) for b in range(n)])   # the next episode begins at the next step
    assert all(frozen.birth(b) == (frozen.clock if mask[b] else 0) for b in range(n))
    assert np.array_equal(s["rows"][mask == 1], rows[want[mask == 1]])


def test_obs_layout(oracle):
    L, M, n = 5, 20, 8
    rows, pieces = oracle.synth_boards(1, 0, n, L), oracle.synth_pieces(1, 0, n, M)
    env = oracle.Env(n, L, M)
    env.set_pool(rows, pieces)
    env.set_options(assign_mode=1)
    env.reset()
    obs = env.expand_obs()
    assert obs.shape == (n, 217)
    cells = ((rows[:, :, None] >> np.arange(10)) & 1).reshape(n, 200).astype(np.float32)
    assert np.array_equal(obs[:, :200], cells)
    assert np.array_equal(obs[:, 200:207].argmax(1), pieces[:, 0]) and np.array_equal(obs[:, 207:214].argmax(1), pieces[:, 1])
    assert np.all(obs[:, 214] == L) and np.all(obs[:, 215] == M) and np.all(obs[:, 216] == 0)


def afterlife_events(f, ci):
    """The events of case `ci` of afterlife.npz: (kind, rotations, location, rows, lines, moves, state, pieces_left)."""
    return list(zip(f[f"c{ci}_kind"], f[f"c{ci}_action"][:, 0], f[f"c{ci}_action"][:, 1], f[f"c{ci}_rows"], f[f"c{ci}_lines"],
                    f[f"c{ci}_moves"], f[f"c{ci}_state"], f[f"c{ci}_pieces_left"]))


def test_afterlife_of_finished_games_and_counters_across_reset(oracle):
    """What the reference does with a finished game (tests/golden/make_golden_afterlife.py): move() never reads `state`
    (game/tetris.py:354-422), so the game goes on -- pieces popped, moves and lines counted past M and L, `state` overwritten only
    where the code assigns it (won -> lost and lost -> won both occur in the fixture) -- until pop(0) raises IndexError on the
    empty list (:356); and reset() (:438-449) swaps board and pieces but keeps lines_cleared / moves_used / state.  The oracle's
    to_move is the reference's move line for line, so it has the same afterlife; the reset is restated here as the reference
    has it (a new board and piece list under the old counters)."""
    f = load_golden("afterlife.npz")
    flips = 0
    for ci in range(int(f["n"])):
        L, M = int(f[f"c{ci}_L"]), int(f[f"c{ci}_M"])
        game, k, lines, moves, state = None, -1, 0, 0, 0
        for kind, rot, loc, rows, li, mo, st, left in afterlife_events(f, ci):
            if kind == 0:                                             # reset(): the next prepared game, counters carried
                k += 1
                game = oracle.Game(L, M, f[f"c{ci}_rows0"][k], f[f"c{ci}_pieces"][k], lines, moves, state)
            elif len(game.pieces) == 0:
                assert kind == 2                                      # the reference raised IndexError: nothing changes
            else:
                assert kind == 1
                before = game.state
                game.move(int(rot), int(loc))
                flips += before != 0 and game.state != before
            lines, moves, state = game.lines_cleared, game.moves_used, game.state
            assert np.array_equal(game.rows, rows) and (lines, moves, state, len(game.pieces)) == (li, mo, st, left), (ci, kind)
    assert flips >= 6
