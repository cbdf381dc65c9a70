"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle, bit-exact.

Everything here is integer/byte work, so the bar is equality.  The only floating-point outputs are the reward
(a product and at most one sum of small integers and the configured constants, computed in the same order on
both sides) and the observation (0/1 and small integers) -- also compared for equality.
"""
import numpy as np
import pytest

from conftest import board_hashes, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import tetris_piclim
    return tetris_piclim


def _np(t):
    return t.cpu().numpy()


def _state(env):
    s = {k: _np(v) for k, v in env.packed_state().items()}
    s["rows"] = s["rows"].view(np.uint16)
    return s


def _assert_state_equal(got, want, ctx=""):
    for k in ("rows", "cur", "nxt", "lines", "moves", "state", "pieces_left"):
        if k in want:
            assert np.array_equal(got[k], want[k]), f"{ctx}: {k} differs at {np.argwhere(got[k] != want[k])[:5].tolist()}"


# ------------------------------------------------------------------------------------------------- F1
@pytest.mark.parametrize("prefix", ["t_", "o_"])
def test_f1_plumbing_with_reference_style_object(T, prefix):
    """BASELINE config 1: one board, fixed piece sequence; reads like the reference's own test loop."""
    f = load_golden("plumbing.npz")
    if prefix == "t_":
        L, M, pieces, actions = int(f["L"]), int(f["M"]), f["pieces"], f["actions"]
    else:
        L, M, pieces, actions = int(f["o_L"]), int(f["o_M"]), f["o_pieces"], f["o_actions"]
    game = T.Tetris(L, M, configs=(np.zeros((1, 20), np.uint16), pieces[None]))
    ref_state = {0: None, 1: True, 2: False}
    for t, (rot, loc) in enumerate(actions):
        game.move(int(rot), int(loc))
        rows = (game.board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1)
        assert np.array_equal(rows, f[prefix + "rows"][t]), t
        assert game.lines_cleared == f[prefix + "lines"][t] and game.moves_used == f[prefix + "moves"][t]
        assert game.state is ref_state[int(f[prefix + "state"][t])]
        assert len(game.pieces) == f[prefix + "pieces_left"][t]
    game.terminate()


@pytest.mark.parametrize("L,M", [(15, 40), (5, 20)])
def test_carved_solution_replays_to_a_win_and_reset_loads_the_next_game(T, L, M):
    """Property behind the reference's test_carving_invertability (game/main.py:49-57), on the single-board class: the
    recorded solution of a carved game, replayed from move 0, ends in `state is True` with at least L lines cleared,
    within the M moves; reset() then loads the NEXT prescribed configuration with the counters zeroed."""
    game = T.Tetris(L, M, warm_reset=False, debug=True)
    for episode in range(2):
        assert game.state is None and game.lines_cleared == 0 and game.moves_used == 0 and len(game.pieces) == M + 1
        assert 1 <= len(game.solution) <= M
        for used, (rotations, location) in enumerate(game.solution, start=1):
            assert game.state is None, "won or lost before the solution was played out"
            game.move(rotations, location)
            assert game.moves_used == used
        assert game.state is True and game.lines_cleared >= L
        board, cur, nxt, l_rem, m_rem, state = game.get_state()
        assert board.shape == (20, 10) and board.dtype == bool and l_rem <= 0 and m_rem >= 0 and state is True
        game.reset()
    game.terminate()


def test_recarving_a_solution_rebuilds_the_board(T):
    """Property behind the reference's test_carving_repeatability (game/main.py:32-47): take a full stack of L rows and
    carve a game's pieces out of it in reverse order at the solution's (rotation, location) -- every carve succeeds and
    what is left is that game's prescribed board, cell for cell."""
    L, M = 15, 40
    game = T.Tetris(L, M, warm_reset=False, debug=True)
    stack = T.Tetris(L, M, warm_reset=False, debug=True)
    stack.board[:] = False
    stack.board[20 - L:, :] = True
    last = len(game.solution) - 1
    for k in range(last, -1, -1):
        rotations, location = game.solution[k]
        assert stack.carve(game.pieces[k], rotations, location, k == last), f"carve {k} refused"
    # every carve takes four cells out, except the first one made (the LAST piece to fall), which may stick out of the stack
    removed = 10 * L - int(stack.board.sum())
    assert 4 * last + 1 <= removed <= 4 * (last + 1)
    assert np.array_equal(stack.board, game.board)
    game.terminate()
    stack.terminate()


# ------------------------------------------------------------------------------------------------- F2
@pytest.mark.parametrize("name", ["carved_L5_M20.npz", "carved_L10_M40.npz"])
def test_f2_carved_solutions_win(T, name):
    """The reference's test_carving_invertability (game/main.py:49-57) on 256 boards at once."""
    f = load_golden(name)
    L, M = int(f["L"]), int(f["M"])
    n = f["rows"].shape[0]
    env = T.BatchedTetris(L, M, n, assign="sequential", config_pool=(f["rows"], f["pieces"]))
    env.reset()
    sol_len = f["sol_len"]
    last = dict(rows=f["rows"].copy(), lines=np.zeros(n, np.uint8), moves=np.zeros(n, np.uint8), state=np.zeros(n, np.uint8))
    for t in range(int(sol_len.max())):
        active = t < sol_len
        rot = np.where(active, f["sol"][:, min(t, M - 1), 0], 0).astype(np.int64)
        loc = np.where(active, f["sol"][:, min(t, M - 1), 1], 0).astype(np.int64)
        env.move(rot, loc)
        s = _state(env)
        last["rows"][active] = f["r_rows"][active, t]
        last["lines"][active] = f["r_lines"][active, t]
        last["moves"][active] = f["r_moves"][active, t]
        last["state"][active] = f["r_state"][active, t]
        _assert_state_equal(s, last, f"{name} step {t}")     # finished boards are frozen at their last state
    s = _state(env)
    assert np.all(s["state"] == T.WON) and np.all(s["lines"] >= L)
    env.terminate()


# ------------------------------------------------------------------------------------------------- F3
@pytest.mark.parametrize("name", ["synthetic_L5_M20.npz", "synthetic_L10_M40.npz"])
def test_f3_synthetic_against_reference_fingerprints(T, oracle, name):
    f = load_golden(name)
    L, M, seed = int(f["L"]), int(f["M"]), int(f["seed"])
    n = f["rows"].shape[0]
    env = T.BatchedTetris(L, M, n, seed=seed, assign="sequential")
    # device generator == fixture inputs (which came from the CPU generator)
    rows_d, pieces_d = env.synthetic_configs(n)
    assert np.array_equal(_np(rows_d).view(np.uint16), f["rows"]) and np.array_equal(_np(pieces_d), f["pieces"])
    env.load_configs(rows_d, pieces_d)
    env.reset()
    for t in range(M):
        a = env.synthetic_actions(t)
        assert np.array_equal(_np(a), f["actions"][t])
        env.step(a, observe=False)
        s = _state(env)
        hashes = board_hashes(s["rows"])
        assert np.array_equal(hashes, f["hashes"][t]), t
        assert np.array_equal(s["state"], f["s_state"][t]) and np.array_equal(s["lines"], f["s_lines"][t])
        assert np.array_equal(s["moves"], f["s_moves"][t])
    s = _state(env)
    _assert_state_equal(s, dict(rows=f["f_rows"], lines=f["f_lines"], moves=f["f_moves"], state=f["f_state"],
                                pieces_left=f["f_pieces_left"]), name)
    env.terminate()


# ------------------------------------------------------------------------------------------------- F4
def _inject_midgame(env, lines0, moves0, window):
    """A game in progress, written into the resident planes (the test's own restatement of the
    layout): lines_cleared -> B.z[27:20], moves_used -> A.y[31:28] (low nibble) and A.w[31:28] (high nibble), the piece
    window (3-bit ids, entry 0 falls next) -> B.w and B.z[31:28].  The reference's counterpart is plain assignment to
    game.lines_cleared / game.moves_used / game.pieces (game/tetris.py:149-150, 187)."""
    import torch
    a, b = (x.cpu().numpy().view(np.uint32).copy() for x in env.raw_planes())
    lines0, moves0, window = (np.asarray(v).astype(np.uint64) for v in (lines0, moves0, window))
    a[:, 1] = (a[:, 1] & 0x0FFFFFFF) | ((moves0 & 15) << 28).astype(np.uint32)
    a[:, 3] = (a[:, 3] & 0x0FFFFFFF) | (((moves0 >> 4) & 15) << 28).astype(np.uint32)
    b[:, 2] = (b[:, 2] & 0x000FFFFF) | ((lines0 & 0xFF) << 20).astype(np.uint32) | (((window >> 32) & 15) << 28).astype(np.uint32)
    b[:, 3] = (window & 0xFFFFFFFF).astype(np.uint32)
    env.write_raw_planes(torch.from_numpy(a.view(np.int32)), torch.from_numpy(b.view(np.int32)))


def _window_of(pieces_abs, cursor, M):
    """The 36-bit piece window of a board whose next piece is pieces_abs[cursor]: what is left of piece word
    cursor // 10 (entries [10w, 10w + 12), ids past M read 7) after cursor % 10 pops."""
    w0 = cursor // 10 * 10
    ids = [(int(pieces_abs[k]) if k <= M else 7) for k in range(w0, w0 + 12)]
    return sum(v << (3 * j) for j, v in enumerate(ids[cursor - w0:]))


def test_f4_midgame_counters_on_the_hip_path(T):
    """The F4 cases that begin in mid-game (lines_cleared / moves_used non-zero: game/tetris.py:409-422 with counters
    carried in) run on the device: the position is written into the resident planes, the step clock set to the moves
    already made (so that the episode's birth step is 0 and its pool record is found again by the window refill)."""
    f = load_golden("edges.npz")
    ran = 0
    for i in range(int(f["n"])):
        g = lambda k: f[f"c{i}_{k}"]
        name = str(f["names"][i])
        lines0, moves0 = int(g("lines0")), int(g("moves0"))
        if not (lines0 or moves0):
            continue
        L, M = int(g("L")), int(g("M"))
        given = g("pieces")
        pieces_abs = np.zeros(M + 1, np.uint8)                    # absolute positions: given[0] falls at move moves0
        fit = min(len(given), M + 1 - moves0)
        pieces_abs[moves0: moves0 + fit] = given[:fit]
        env = T.BatchedTetris(L, M, 1, assign="sequential", config_pool=(g("rows0")[None], pieces_abs[None]))
        env.reset()
        _inject_midgame(env, [lines0], [moves0], [_window_of(pieces_abs, moves0, M)])
        env._clocks().fill_(moves0)
        for t, (rot, loc) in enumerate(g("actions")):
            env.move(np.array([rot]), np.array([loc]))
            s = _state(env)
            assert np.array_equal(s["rows"][0], g("rows")[t]), (name, t)
            assert (int(s["lines"][0]), int(s["moves"][0]), int(s["state"][0])) == (g("lines")[t], g("moves")[t], g("state")[t]), (name, t)
            # pieces consumed so far (the fixture's list may be shorter than what M + 1 - moves0 would leave)
            assert (M + 1 - moves0) - int(s["pieces_left"][0]) == len(given) - int(g("pieces_left")[t]), (name, t)
        env.terminate()
        ran += 1
    assert ran >= 1


def _dealing_order(kinds):
    """Which pool entry tetris_piclim.Tetris deals at each reset, by the rule its docstring states: entry `steps so far` mod pool
    size, a step being a move() that reached the device or a reset() that follows no move.  -> (entry per game, pool size)."""
    steps = birth = 0
    entries = []
    for i, kind in enumerate(kinds):
        if kind == 0:                                   # reset (the first one is the constructor's deal: entry 0)
            if i > 0:
                if steps == birth:
                    steps += 1
                birth = steps
            entries.append(steps)
        elif kind == 1:
            steps += 1                                  # (a move that raised IndexError never reached the device)
    return entries, steps + 2


def test_afterlife_of_finished_games_with_reference_quirks_on_the_hip_path(T):
    """tests/golden/afterlife.npz (recorded from the reference) through `Tetris(..., reference_quirks=True)`: finished games
    that go on -- counters past M and L, terminal state flipping both ways, IndexError when the M + 1 pieces are used up -- and
    reset() keeping lines_cleared / moves_used / state (game/tetris.py:354-422, 438-449).  Every move's board mechanics run on
    the device; board, counters, state and the piece list after every event are the reference's."""
    from test_oracle_golden import afterlife_events
    f = load_golden("afterlife.npz")
    ref_state = {0: None, 1: True, 2: False}
    raised = flips = 0
    for ci in range(int(f["n"])):
        L, M = int(f[f"c{ci}_L"]), int(f[f"c{ci}_M"])
        events = afterlife_events(f, ci)
        entries, pool = _dealing_order([e[0] for e in events])
        rows = np.repeat(f[f"c{ci}_rows0"][:1], pool, axis=0)
        pieces = np.repeat(f[f"c{ci}_pieces"][:1], pool, axis=0)
        for k, entry in enumerate(entries):
            rows[entry], pieces[entry] = f[f"c{ci}_rows0"][k], f[f"c{ci}_pieces"][k]
        game = T.Tetris(L, M, configs=(rows, pieces), reference_quirks=True)
        k = -1
        for i, (kind, rot, loc, want_rows, li, mo, st, left) in enumerate(events):
            before = game.state
            if kind == 0:
                k += 1
                if i > 0:
                    game.reset()
                assert np.array_equal(np.array(game.pieces), f[f"c{ci}_pieces"][k]), (ci, k)
            elif kind == 2:
                with pytest.raises(IndexError, match="pop from empty list"):
                    game.move(int(rot), int(loc))
                raised += 1
            else:
                game.move(int(rot), int(loc))
                flips += before is not None and game.state is not before
            got = (game.board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1)
            assert np.array_equal(got, want_rows), (ci, i)
            assert (game.lines_cleared, game.moves_used, len(game.pieces)) == (li, mo, left), (ci, i)
            assert game.state is ref_state[int(st)], (ci, i)
            if left >= 2:
                board, cur, nxt, l_rem, m_rem, state = game.get_state()
                assert (cur, nxt, l_rem, m_rem, state) == (game.pieces[0], game.pieces[1], L - li, M - mo, ref_state[int(st)])
            else:
                with pytest.raises(IndexError):
                    game.get_state()                    # self.pieces[1] of the reference's get_state (:436)
        game.terminate()
    assert raised >= 40 and flips >= 6


def test_reference_quirks_against_the_oracle_on_random_afterlives(T, oracle):
    """The same mode against the oracle's move (the reference's, line for line: it never reads `state`) on games the fixture
    does not hold: other (L, M), every rotation count 0..8 and location 0..10, thirty games in a row through reset()."""
    rng = np.random.default_rng(5)
    for L, M in ((1, 2), (3, 9), (6, 25), (10, 40), (250, 254)):
        n = 30
        rows = oracle.synth_boards(900 + L, 0, n, min(L, 16))
        rows[::3] = 0
        pieces = oracle.synth_pieces(900 + L, 0, n, M)
        plays = [int(rng.integers(1, M + 4)) for _ in range(n)]
        kinds = [x for p in plays for x in [0] + [1] * min(p, M + 1)]
        entries, pool = _dealing_order(kinds)
        big_rows, big_pieces = np.repeat(rows[:1], pool, axis=0), np.repeat(pieces[:1], pool, axis=0)
        for k, entry in enumerate(entries):
            big_rows[entry], big_pieces[entry] = rows[k], pieces[k]
        game = T.Tetris(L, M, configs=(big_rows, big_pieces), reference_quirks=True)
        lines = moves = state = 0
        for k in range(n):
            if k:
                game.reset()
            cpu = oracle.Game(L, M, rows[k], pieces[k], lines, moves, state)
            for t in range(plays[k]):
                rot, loc = int(rng.integers(0, 9)), int(rng.integers(0, 11))
                if t >= M + 1:
                    with pytest.raises(IndexError):
                        game.move(rot, loc)
                    continue
                game.move(rot, loc)
                cpu.move(rot, loc)
                got = (game.board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1)
                assert np.array_equal(got, cpu.rows), (L, M, k, t)
                assert (game.lines_cleared, game.moves_used, game.pieces) == (cpu.lines_cleared, cpu.moves_used, cpu.pieces), (L, M, k, t)
                assert game.state is {0: None, 1: True, 2: False}[cpu.state], (L, M, k, t)
            lines, moves, state = cpu.lines_cleared, cpu.moves_used, cpu.state
        game.terminate()


def test_f4_edge_cases(T):
    f = load_golden("edges.npz")
    for i in range(int(f["n"])):
        g = lambda k: f[f"c{i}_{k}"]
        name = str(f["names"][i])
        if int(g("lines0")) or int(g("moves0")):
            continue      # mid-game cases: test_f4_midgame_counters_on_the_hip_path
        L, M = int(g("L")), int(g("M"))
        pieces = np.full(M + 1, 0, np.uint8)
        given = g("pieces")
        pieces[: len(given)] = given[: M + 1]
        game = T.Tetris(L, M, configs=(g("rows0")[None], pieces[None]))
        for t, (rot, loc) in enumerate(g("actions")):
            game.move(int(rot), int(loc))
            rows = (game.board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1)
            assert np.array_equal(rows, g("rows")[t]), (name, t)
            assert (game.lines_cleared, game.moves_used) == (g("lines")[t], g("moves")[t]), (name, t)
            assert game.state is {0: None, 1: True, 2: False}[int(g("state")[t])], (name, t)
            # the fixture's piece list may be shorter than M+1 (the reference does not care); compare consumption
            assert (M + 1) - len(game.pieces) == len(given) - int(g("pieces_left")[t]), (name, t)
        game.terminate()


# ------------------------------------------------------------------------------------------------- F5
def test_f5_random_single_moves(T):
    """All 4096 reference-generated single moves, the 3590 that start in mid-game included (lines_cleared and
    moves_used carried in -- up to moves_used > M, which the reference plays on from: game/tetris.py:389-394, 415-422):
    the counters and the falling piece are written into the resident planes."""
    f = load_golden("random_moves.npz")
    checked = 0
    for L in np.unique(f["L"]):
        for M in np.unique(f["M"]):
            idx = np.nonzero((f["L"] == L) & (f["M"] == M))[0]
            if len(idx) == 0:
                continue
            pieces = np.zeros((len(idx), int(M) + 1), np.uint8)
            pieces[:, 0] = f["piece"][idx]
            env = T.BatchedTetris(int(L), int(M), len(idx), assign="sequential", config_pool=(f["rows"][idx], pieces))
            env.reset()
            # window: the fixture's piece falls next, nothing behind it (moves0 <= 4: no refill is due)
            _inject_midgame(env, f["lines0"][idx], f["moves0"][idx], f["piece"][idx].astype(np.uint64))
            env.move(f["rot"][idx].astype(np.int32), f["loc"][idx].astype(np.int32))
            s = _state(env)
            _assert_state_equal(s, dict(rows=f["o_rows"][idx], lines=f["o_lines"][idx], moves=f["o_moves"][idx],
                                        state=f["o_state"][idx]), f"L={L} M={M}")
            checked += len(idx)
            env.terminate()
    assert checked == 4096


# ------------------------------------------------------------------------------------------------- vs oracle
def _dense_boards(rng, n):
    """Tall, dense boards with nearly-full rows so that clears, multi-clears and top-outs all occur."""
    height = rng.integers(0, 21, n)
    cells = rng.random((n, 20, 10)) < rng.uniform(0.3, 0.95, (n, 1, 1))
    near = rng.random((n, 20)) < 0.5
    holes = rng.integers(0, 10, (n, 20))
    full = np.ones((n, 20, 10), bool)
    full[np.arange(n)[:, None], np.arange(20)[None, :], holes] = False
    cells = np.where(near[:, :, None], full, cells)
    cells &= (np.arange(20)[None, :, None] >= (20 - height)[:, None, None])
    return (cells.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(-1).astype(np.uint16)


@pytest.mark.parametrize("params,want", [((0.1, 0.0, -0.3), 0.0), ((0.7, 0.0, 0.1), 2.1999998)])
def test_reward_is_two_roundings_not_a_fused_multiply_add(T, oracle, params, want):
    """reward = per_line * rows (+ win) (+ lose) with TWO roundings, as on the CPU.  A move that clears three rows and
    loses at the move limit: 0.1f * 3 - 0.3f is exactly 0 in two roundings and -7.45e-9 as one fused multiply-add (and
    0.7f * 3 + 0.1f is 2.1999998 against 2.2) -- the compiler fuses the pair unless told not to, and random boards almost
    never make this move.  Every entry point that computes a reward: move, step, step + observe, fused rollout."""
    import torch
    L, M, n = 10, 1, 130
    rows = np.zeros((n, 20), np.uint16)
    rows[:, 17:20] = 0x3FE                             # three rows with the cell at x = 0 missing
    pieces = np.zeros((n, M + 1), np.uint8)            # I first: upright at x = 0 it clears all three
    want32 = np.float32(want)
    assert np.float32(np.float32(params[0]) * np.float32(3)) + np.float32(params[2]) == want32
    cpu = oracle.Env(n, L, M)
    cpu.set_pool(rows, pieces)
    cpu.set_options(auto_reset=False, assign_mode=1, per_line=params[0], win=params[1], lose=params[2])
    cpu.reset()
    r_c, d_c = cpu.step(np.full(n, 10, np.uint8))
    assert np.all(r_c == want32) and np.all(d_c == 1) and np.all(cpu.get_state()["lines"] == 3)
    act = torch.full((n,), 10, dtype=torch.uint8, device="cuda")
    for how in ("move", "step", "step_observe", "rollout"):
        gpu = T.BatchedTetris(L, M, n, assign="sequential", reward=params, config_pool=(rows, pieces))
        gpu.reset()
        if how == "move":
            r, d, cleared = gpu.move(np.ones(n, np.uint8), np.zeros(n, np.uint8))
            assert np.all(_np(cleared) == 3)
        elif how == "step":
            _, r, d, _ = gpu.step(act, observe=False)
        elif how == "step_observe":
            _, r, d, _ = gpu.step(act)
        else:
            rsum, fin, rs, ds = gpu.rollout(act.unsqueeze(0), per_step=True)
            assert np.array_equal(_np(rsum), _np(rs[0]))
            r, d = rs[0], ds[0]
        got = _np(r)
        assert got.dtype == np.float32 and np.all(got == want32), (how, got[:4], want32)
        assert np.all(_np(d).astype(np.uint8) == 1)
        gpu.terminate()


@pytest.mark.parametrize("L,M,n", [(3, 12, 8192), (10, 40, 65536), (250, 254, 4096), (1, 1, 1024)])
def test_move_matches_oracle_on_dense_random_boards(T, oracle, L, M, n):
    rng = np.random.default_rng(L * 1000 + M)
    rows = _dense_boards(rng, n)
    pieces = rng.integers(0, 7, (n, M + 1)).astype(np.uint8)
    # reward parameters whose products are not exact in float32 (0.1 * 3): the device must round twice like the CPU
    gpu = T.BatchedTetris(L, M, n, assign="sequential", reward=(0.1, 0.7, -0.3), config_pool=(rows, pieces))
    cpu = oracle.Env(n, L, M)
    cpu.set_pool(rows, pieces)
    cpu.set_options(auto_reset=False, assign_mode=1, per_line=0.1, win=0.7, lose=-0.3)
    gpu.reset(); cpu.reset()
    _assert_state_equal(_state(gpu), cpu.get_state(), "after reset")
    steps = min(M, 48)
    for t in range(steps):
        rot = rng.integers(0, 9, n).astype(np.uint8)        # > 3: exercises rotations % len
        loc = rng.integers(0, 11, n).astype(np.uint8)       # 10: exercises the right clamp
        r_g, d_g, c_g = gpu.move(rot, loc)
        r_c, d_c, c_c = cpu.move(rot, loc)
        assert np.array_equal(_np(r_g), r_c) and np.array_equal(_np(d_g).astype(np.uint8), d_c) and np.array_equal(_np(c_g), c_c), t
        _assert_state_equal(_state(gpu), cpu.get_state(), f"step {t}")
    gpu.terminate()


@pytest.mark.parametrize("dtype", ["uint8", "int32", "int64"])
def test_action_dtypes(T, oracle, dtype):
    import torch
    L, M, n = 5, 20, 4096
    gpu = T.BatchedTetris(L, M, n, seed=11, assign="sequential")
    rows, pieces = gpu.synthetic_configs(n)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, 11)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(assign_mode=1)
    cpu.reset()
    for t in range(6):
        a = gpu.synthetic_actions(t)
        _, r_g, d_g, _ = gpu.step(a.to(getattr(torch, dtype)), observe=False)
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(r_g), r_c) and np.array_equal(_np(d_g).astype(np.uint8), d_c)
    _assert_state_equal(_state(gpu), cpu.get_state(), dtype)
    gpu.terminate()


@pytest.mark.parametrize("assign", ["hash", "sequential"])
def test_auto_reset_rollout_matches_oracle(T, oracle, assign):
    """BASELINE config 2 shape: 65,536 boards, L=5 M=20, auto-reset from a pool, statistics."""
    L, M, n, pool, seed = 5, 20, 65536, 4096, 5
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, assign=assign, reward=(1.0, 5.0, -1.0))
    rows, pieces = gpu.synthetic_configs(pool)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0 if assign == "hash" else 1, per_line=1.0, win=5.0, lose=-1.0)
    cpu.reset()
    _assert_state_equal(_state(gpu), cpu.get_state(), "reset")
    for t in range(3 * M):
        a = gpu.synthetic_actions(t)
        _, r_g, d_g, _ = gpu.step(a, observe=False)
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(r_g), r_c) and np.array_equal(_np(d_g).astype(np.uint8), d_c), t
        if t % 7 == 0 or t == 3 * M - 1:
            _assert_state_equal(_state(gpu), cpu.get_state(), f"step {t}")
    assert gpu.stats() == cpu.stats()
    assert gpu.stats()["episodes"] > n
    # masked reset
    mask = (np.arange(n) % 5 == 0).astype(np.uint8)
    gpu.reset(mask); cpu.reset(mask)
    _assert_state_equal(_state(gpu), cpu.get_state(), "masked reset")
    gpu.terminate()


@pytest.mark.parametrize("bpl", [1, 2, 4])
def test_many_short_episodes(T, oracle, bpl):
    """L=1, M=2: every episode lasts at most two moves, so 700 steps are > 300 episodes per board, each assigned by
    its birth step; also runs each boards-per-lane variant of the step kernel and a ragged batch size."""
    L, M, n, pool, seed = 1, 2, 5000, 777, 3
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, assign="hash")
    gpu.set_tuning(bpl)
    rows, pieces = gpu.synthetic_configs(pool)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    for t in range(700):
        a = gpu.synthetic_actions(t)
        _, r_g, d_g, _ = gpu.step(a, observe=False)
        r_c, d_c = cpu.step(_np(a))
        if t % 50 == 0 or t == 699:
            assert np.array_equal(_np(r_g), r_c) and np.array_equal(_np(d_g).astype(np.uint8), d_c), t
            _assert_state_equal(_state(gpu), cpu.get_state(), f"step {t}")
    assert gpu.stats() == cpu.stats() and gpu.stats()["episodes"] > 300 * n
    gpu.terminate()


@pytest.mark.parametrize("M", [7, 9, 10, 11, 19, 20, 21, 40, 49, 50, 51, 129, 130, 131, 254])   # 49|50, 129|130: record stride 64|128|256
def test_piece_window_refills_at_every_word_boundary(T, oracle, M):
    """Boards that survive all M moves (empty start, O pieces side by side never top out before M for small M;
    otherwise whatever happens) with random piece lists: checks cur/nxt against the oracle on every step."""
    L, n = 250, 2048
    rng = np.random.default_rng(M)
    rows = np.zeros((n, 20), np.uint16)
    pieces = rng.integers(0, 7, (n, M + 1)).astype(np.uint8)
    gpu = T.BatchedTetris(L, M, n, assign="sequential", config_pool=(rows, pieces))
    cpu = oracle.Env(n, L, M)
    cpu.set_pool(rows, pieces)
    cpu.set_options(assign_mode=1)
    gpu.reset(); cpu.reset()
    for t in range(M):
        # spread the pieces out so that many boards live long: column from the step index
        rot = np.zeros(n, np.uint8)
        loc = ((np.arange(n) + 3 * t) % 10).astype(np.uint8)
        gpu.move(rot, loc); cpu.move(rot, loc)
        _assert_state_equal(_state(gpu), cpu.get_state(), f"M={M} step {t}")
    assert (cpu.get_state()["moves"] >= min(M, 10)).mean() > 0.2     # the refill path was really exercised
    gpu.terminate()


@pytest.mark.parametrize("auto", [True, False])
def test_fused_rollout_equals_k_single_steps_and_the_oracle(T, oracle, auto):
    """tpl_rollout (K moves per launch, board held in registers) == K x tpl_step == the oracle."""
    import torch
    L, M, n, pool, seed, K = 5, 20, 10007, 999, 13, 57
    envs = [T.BatchedTetris(L, M, n, seed=seed, auto_reset=auto, reward=(1.0, 3.0, -0.5)) for _ in range(2)]
    rows, pieces = envs[0].synthetic_configs(pool)
    for e in envs:
        e.load_configs(rows, pieces)
        e.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=auto, assign_mode=0, per_line=1.0, win=3.0, lose=-0.5)
    cpu.reset()
    actions = torch.stack([envs[0].synthetic_actions(t) for t in range(K)])
    # two rollout launches (K1 + K2) to also cover resuming from a stored state
    K1 = 20
    rsum1, fin1, rs1, ds1 = envs[0].rollout(actions[:K1], per_step=True)
    rsum2, fin2, rs2, ds2 = envs[0].rollout(actions[K1:], per_step=True)
    rs, ds = torch.cat([rs1, rs2]), torch.cat([ds1, ds2])
    acc1 = np.zeros(n, np.float32); acc2 = np.zeros(n, np.float32); fins = np.zeros(n, np.int64)
    for t in range(K):
        _, r_b, d_b, _ = envs[1].step(actions[t], observe=False)
        before = cpu.get_state()["state"]
        r_c, d_c = cpu.step(_np(actions[t]))
        assert np.array_equal(_np(rs[t]), r_c) and np.array_equal(_np(ds[t]).astype(np.uint8), d_c), t
        assert np.array_equal(_np(r_b), r_c) and np.array_equal(_np(d_b).astype(np.uint8), d_c), t
        if t < K1:
            acc1 += r_c
        else:
            acc2 += r_c
        fins += (d_c == 1) & (before == 0)
    want = cpu.get_state()
    _assert_state_equal(_state(envs[0]), want, "rollout")
    _assert_state_equal(_state(envs[1]), want, "steps")
    assert np.array_equal(_np(rsum1), acc1) and np.array_equal(_np(rsum2), acc2)
    assert np.array_equal(_np(fin1).astype(np.int64) + _np(fin2).astype(np.int64), fins)
    assert envs[0].stats() == cpu.stats() == envs[1].stats()
    for e in envs:
        e.terminate()


def _argmax_first(vals):
    best, idx = -float("inf"), 0
    for k, v in enumerate(vals):
        if v > best:
            best, idx = v, k
    return idx


def test_compat_tetris_pool_walk_freeze_and_forward_supplier(T):
    """The single-board `Tetris`: (a) after any number of moves and resets, `pieces` (host list of the entry the wrapper
    believes it is on) agrees with what the device shows as current / next piece and with the board of that entry --
    i.e. the wrapper follows the device's sequential assignment by birth step; (b) a finished game is frozen and
    move() on it raises (the reference keeps mutating: documented divergence); (c) with warm_reset=True the pool also
    holds the winnable games of the reference's second producer, the forward generator + solver over seeds 0..99,
    with one random piece in front (game/tetris.py:19-20, 205-211)."""
    L, M = 5, 20
    game = T.Tetris(L, M, warm_reset=True, seed=11, pool_size=8)
    rows_host = np.asarray(game._env._pool[0].cpu().numpy()).view(np.uint16)
    fw = T.forward_generate(L, M, np.arange(100))
    n_fw = int(fw["winnable"].sum())
    assert n_fw > 0 and game._pieces_host.shape[0] == 8 + n_fw
    assert np.array_equal(rows_host[8:], fw["rows"][fw["winnable"]])
    assert np.array_equal(game._pieces_host[8:, 1:], fw["sequence"][fw["winnable"]])
    rng = np.random.default_rng(0)
    seen = set()
    for episode in range(12):
        k = game._config()
        seen.add(k)
        board0 = ((rows_host[k][:, None] >> np.arange(10)) & 1).astype(bool)
        assert np.array_equal(game.board, board0) and game.pieces == game._pieces_host[k].tolist()
        for _ in range(int(rng.integers(1, 9))):
            if game.state is not None:
                break
            game.move(int(rng.integers(0, 4)), int(rng.integers(0, 10)))
            st = game.get_state()
            assert (st[1], st[2]) == tuple(game.pieces[:2])
        if game.state is not None:
            frozen = game.board.copy()
            with pytest.raises(RuntimeError, match="finished game"):
                game.move(0, 0)
            assert np.array_equal(game.board, frozen)
        game.reset()
        assert game.state is None and game.moves_used == 0 and game.lines_cleared == 0
    assert len(seen) > 3
    # reset() twice with no move in between deals two different games, as the reference's queue does
    # (game/tetris.py:445-447); the device's step clock follows the wrapper's count
    dealt = []
    for _ in range(5):
        game.reset()
        dealt.append(game._config())
        assert game.pieces == game._pieces_host[dealt[-1]].tolist()
    assert len(set(dealt)) == 5 and game._env.step_clock() == game._steps
    game.terminate()


@pytest.mark.parametrize("auto,n,L,M,K", [(True, 10007, 5, 20, 57), (False, 4133, 3, 12, 30), (True, 65536, 5, 20, 43),
                                          (True, 1 << 20, 10, 40, 10), (True, 777, 4, 9, 1), (True, 777, 4, 9, 4)])
def test_compact_trajectory_decodes_to_the_per_step_outputs_and_the_oracle(T, oracle, auto, n, L, M, K):
    """tpl_rollout_trajectory writes ONE byte per board-step (rows cleared | how the move ended | reset | frozen) as a dword
    per lane every fourth step; tpl_decode_trajectory must give back, bit for bit, the reward f32 / done u8 that tpl_rollout
    writes for the same steps (a reward with two roundings: per_line 0.1, lose -0.7) and that the oracle computes; the boards,
    counters and statistics after the launch are tpl_rollout's; the bytes themselves say what the oracle saw happen."""
    import torch
    seed, reward = 23, (0.1, 1.5, -0.7)
    envs = [T.BatchedTetris(L, M, n, seed=seed, auto_reset=auto, reward=reward) for _ in range(2)]
    rows, pieces = envs[0].synthetic_configs(min(n, 5000))
    for e in envs:
        e.load_configs(rows, pieces)
        e.reset()
    actions = torch.stack([envs[0].synthetic_actions(t) for t in range(K)])
    _, _, rs, ds = envs[0].rollout(actions, per_step=True)
    traj = envs[1].rollout_trajectory(actions)
    assert traj.shape == ((K + 3) // 4, n) and traj.dtype == torch.int32
    rs2, ds2 = envs[1].decode_trajectory(traj, K)
    assert torch.equal(rs.view(torch.int32), rs2.view(torch.int32)) and torch.equal(ds, ds2)
    _assert_state_equal(_state(envs[1]), _state(envs[0]), "after the launch")
    assert envs[0].stats() == envs[1].stats()
    codes = _np(traj).view(np.uint8).reshape((K + 3) // 4, n, 4).transpose(0, 2, 1).reshape(-1, n)   # [step, board]
    assert not codes[K:].any()                                  # bytes past the last step are zero
    codes = codes[:K]
    ended, cleared, was_reset, frozen = (codes >> 3) & 3, codes & 7, (codes >> 5) & 1, (codes >> 6) & 1
    assert cleared.max() <= 4 and not (codes >> 7).any()
    assert np.array_equal((ended != 0) | (frozen != 0), _np(ds))
    assert np.array_equal(was_reset.astype(bool), (ended != 0) if auto else np.zeros_like(ended, bool))
    if auto:
        assert not frozen.any()
    if n <= 70000:                                              # the oracle, step by step: what happened, not just what it was worth
        cpu = oracle.Env(n, L, M, 0, seed)
        cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
        cpu.set_options(auto_reset=auto, assign_mode=0, per_line=reward[0], win=reward[1], lose=reward[2])
        cpu.reset()
        for t in range(K):
            before = cpu.get_state()
            r_c, d_c = cpu.step(_np(actions[t]))
            assert np.array_equal(_np(rs2[t]), r_c) and np.array_equal(_np(ds2[t]).astype(np.uint8), d_c), t
            assert np.array_equal(frozen[t].astype(bool), before["state"] != 0), t
            if not auto:
                after = cpu.get_state()
                live = before["state"] == 0
                assert np.array_equal(cleared[t][live], (after["lines"] - before["lines"])[live]), t
                won, lost = after["state"] == 1, after["state"] == 2
                assert np.array_equal(ended[t][live] == 1, won[live]) and np.array_equal(ended[t][live] >= 2, lost[live]), t
    for e in envs:
        e.terminate()


def test_compact_trajectory_of_the_device_drawn_policy(T):
    """tpl_rollout_random_trajectory == tpl_rollout_random: the same actions are played and recorded, the trajectory decodes
    to the same per-step rewards and dones, a second launch continues the first (step0)."""
    import torch
    L, M, n, K = 6, 25, 9001, 41
    envs = [T.BatchedTetris(L, M, n, seed=19, auto_reset=True, reward=(1.0, 2.0, -1.0)) for _ in range(2)]
    rows, pieces = envs[0].synthetic_configs(2000)
    for e in envs:
        e.load_configs(rows, pieces)
        e.reset()
    for step0, steps in ((0, 17), (17, K - 17)):
        _, _, acts, rs, ds = envs[0].rollout_random(steps, seed=5, step0=step0, record=True)
        acts2 = torch.empty((steps, n), dtype=torch.uint8, device=envs[1].device)
        traj = envs[1].rollout_random_trajectory(steps, seed=5, step0=step0, actions_out=acts2)
        rs2, ds2 = envs[1].decode_trajectory(traj, steps)
        assert torch.equal(acts, acts2) and torch.equal(rs, rs2) and torch.equal(ds, ds2)
    _assert_state_equal(_state(envs[1]), _state(envs[0]))
    assert envs[0].stats() == envs[1].stats()
    with pytest.raises(ValueError):
        envs[0].decode_trajectory(traj[:, :-1].contiguous(), steps)
    for e in envs:
        e.terminate()


@pytest.mark.parametrize("auto", [True, False])
def test_rollout_random_equals_explore_then_step_and_the_oracle(T, oracle, auto):
    """tpl_rollout_random (the uniform random policy drawn on the device, K steps per launch) == K x (tpl_explore_actions at
    epsilon 1, tpl_step) on a twin handle; the recorded actions replayed through the oracle give the same rewards, dones,
    boards and statistics; two launches continue one another (step0); the draws are uniform."""
    import torch
    L, M, n, seed, K = 6, 25, 9001, 19, 41
    envs = []
    for _ in range(2):
        env = T.BatchedTetris(L, M, n, seed=seed, global_offset=12345, auto_reset=auto, reward=(1.0, 2.0, -0.5))
        rows, pieces = env.synthetic_configs(777)
        env.load_configs(rows, pieces)
        env.reset()
        envs.append(env)
    fused, ref = envs
    cpu = oracle.Env(n, L, M, 12345, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=auto, assign_mode=0, per_line=1.0, win=2.0, lose=-0.5)
    cpu.reset()
    # the draws come in pairs of steps sharing one hash word: the first launch begins on an ODD step and the second one
    # too (in the middle of a pair), so both ways into the loop are taken
    cut, step0 = 18, 1001
    r1 = fused.rollout_random(cut, seed=5, step0=step0, record=True)
    r2 = fused.rollout_random(K - cut, seed=5, step0=step0 + cut, record=True)
    acts, rs, ds = (torch.cat([a, b]) for a, b in zip(r1[2:], r2[2:]))
    rsum = np.zeros(n, np.float32)
    for t in range(K):
        a = ref.explore_actions(torch.zeros(n, dtype=torch.uint8, device=ref.device), 1.0, seed=5, step=step0 + t)
        assert torch.equal(acts[t], a), t
        # ... and the oracle's own restatement of the draw
        assert np.array_equal(_np(a), oracle.explore_actions(np.zeros(n, np.uint8), 1.0, 5, step0 + t, global_offset=12345)), t
        _, r, d, _ = ref.step(a, observe=False)
        assert torch.equal(rs[t], r) and torch.equal(ds[t], d), t
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(r), r_c) and np.array_equal(_np(d).astype(np.uint8), d_c), t
        if t >= cut:
            rsum += r_c
    assert np.array_equal(_np(r2[0]), rsum)
    _assert_state_equal(_state(fused), cpu.get_state(), "rollout_random")
    assert fused.stats() == ref.stats() == cpu.stats() and fused.step_clock() == K
    counts = np.bincount(_np(acts).ravel(), minlength=40).astype(np.float64)
    assert counts.shape == (40,) and abs(counts / counts.sum() - 1 / 40).max() < 0.002
    for e in envs:
        e.terminate()


def test_decode_actions(T):
    import torch
    n = 5000
    env = T.BatchedTetris(5, 20, n)
    g = torch.Generator(device="cuda").manual_seed(0)
    logits = torch.randn((n, 14), device="cuda", generator=g)
    logits[::7, 1] = logits[::7, 0]                       # ties -> lowest index
    logits[::5, 9] = logits[::5, 6] = 9.0
    logits[3, 2] = float("nan")                           # NaN never wins
    for t in (logits, logits.to(torch.bfloat16)):
        host = t.float().cpu().tolist()
        want = torch.tensor([_argmax_first(r[:4]) * 10 + _argmax_first(r[4:]) for r in host])
        assert torch.equal(env.decode_actions(t.contiguous()).cpu().long(), want)
    env.terminate()


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_actor_loop_graph_equals_eager_and_the_oracle(T, oracle, dtype):
    """BASELINE config 5 in small: obs -> MLP -> action -> step on the device, graph-captured; the recorded
    actions replayed through the oracle give the same rewards, dones and boards."""
    import torch
    L, M, n, seed, steps = 10, 40, 4096, 17, 45
    torch.manual_seed(0)
    model = T.PolicyMLP()
    with torch.no_grad():
        for prm in model.parameters():                     # default init gives a near-constant argmax; spread it
            prm.normal_(0.0, 0.35)
        model.layer1.weight[:, 214:] = 0.0                 # let cells and pieces, not the two big counters, decide
    actors = []
    for use_graph in (True, False):
        env = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
        rows, pieces = env.synthetic_configs(512)
        env.load_configs(rows, pieces)
        env.reset()
        actors.append(T.Actor(env, model, dtype=getattr(torch, dtype), use_graph=use_graph))
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    distinct = set()
    for t in range(steps):
        for a in actors:
            a.step()
        g, e = actors
        assert torch.equal(g.action, e.action) and torch.equal(g.reward, e.reward) and torch.equal(g.done, e.done), t
        r_c, d_c = cpu.step(_np(g.action))
        assert np.array_equal(_np(g.reward), r_c) and np.array_equal(_np(g.done), d_c), t
        distinct.update(np.unique(_np(g.action)).tolist())
    assert len(distinct) > 8                               # the policy really depends on the observation
    _assert_state_equal(_state(actors[0].env), cpu.get_state(), "actor")
    assert actors[0].env.stats() == cpu.stats()
    for a in actors:
        a.env.terminate()


def test_observation_matches_oracle(T, oracle):
    import torch
    L, M, n = 10, 40, 1000          # not a multiple of 64: exercises the ragged tail
    gpu = T.BatchedTetris(L, M, n, seed=2, assign="sequential")
    rows, pieces = gpu.synthetic_configs(n)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, 2)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(assign_mode=1)
    cpu.reset()
    for t in range(12):
        a = gpu.synthetic_actions(t)
        obs, _, _, _ = gpu.step(a)
        cpu.step(_np(a))
        want = cpu.expand_obs()
        assert np.array_equal(_np(obs), want), t
        bf = gpu.observe(torch.bfloat16)
        assert np.array_equal(_np(bf.float()), want), t     # every value is exactly representable in bf16
    board, cur, nxt, l_rem, m_rem, state = gpu.get_state()
    s = cpu.get_state()
    assert np.array_equal(_np(board).reshape(n, 200), want[:, :200].astype(bool))
    assert np.array_equal(_np(l_rem), L - s["lines"].astype(np.int32)) and np.array_equal(_np(m_rem), M - s["moves"].astype(np.int32))
    gpu.terminate()


@pytest.mark.parametrize("n", [1, 7, 64, 65, 129, 4097])
def test_observation_on_dense_boards_frozen_wins_and_odd_sizes(T, oracle, n):
    """The 16-byte-store observation kernel: boards with every cell pattern (dense random rows), boards frozen after a
    win whose last clear overshot L (lines left below zero), batch sizes around the 64-board span of a wave, f32
    and bf16, and an output that is not 16-byte aligned (element-wise path)."""
    import torch
    L, M = 2, 30
    rng = np.random.default_rng(100 + n)
    rows = rng.integers(0, 1 << 10, (n, 20)).astype(np.uint16)
    rows[:, :6] = 0                                    # room to play
    rows[rows == 0x3FF] = 0x3FE
    if n >= 7:                                         # a double clear from lines=1 on a board that needs L=2: overshoot
        rows[3] = 0
        rows[3, 17] = 0x3FE                            # one cell missing at x=0: an upright I at x=0 clears rows 17..19
        rows[3, 18] = 0x3FE
        rows[3, 19] = 0x3FE
    pieces = rng.integers(0, 7, (n, M + 1)).astype(np.uint8)
    pieces[:, 0] = 0                                   # I first
    gpu = T.BatchedTetris(L, M, n, assign="sequential", config_pool=(rows, pieces))
    cpu = oracle.Env(n, L, M)
    cpu.set_pool(rows, pieces)
    cpu.set_options(assign_mode=1)
    gpu.reset(); cpu.reset()
    spare = torch.empty(n * 217 + 8, dtype=torch.float32, device=gpu.device)
    for t in range(10):
        rot = np.full(n, 1 if t == 0 else t % 4, np.uint8)
        loc = np.full(n, 0 if t == 0 else (3 * t) % 10, np.uint8)
        gpu.move(rot, loc); cpu.move(rot, loc)
        want = cpu.expand_obs()
        assert np.array_equal(_np(gpu.observe(torch.float32)), want), t
        assert np.array_equal(_np(gpu.observe(torch.bfloat16).float()), want), t
        unaligned = spare[1: 1 + n * 217].view(n, 217)             # 4 bytes off a 16-byte boundary
        assert np.array_equal(_np(gpu.observe(out=unaligned)), want), t
        board = gpu.get_state()[0]                                 # Tetris.board, bool [n, 20, 10] (tpl_get_board)
        assert board.dtype == torch.bool and np.array_equal(_np(board).reshape(n, 200), want[:, :200].astype(bool)), t
    if n >= 7:
        assert want[3, 214] < 0 and want[3, 216] == 1              # the overshoot really happened and is frozen
    gpu.terminate()


@pytest.mark.parametrize("L,M,n,bf16,auto,steps", [(5, 20, 65536, False, True, 45), (10, 40, 1 << 20, True, True, 14),
                                                  (4, 12, 1000, False, True, 40), (4, 12, 63, True, True, 40),
                                                  (3, 9, 2049, True, False, 12), (2, 30, 129, False, False, 12)])
def test_step_observe_is_step_then_observe_and_the_oracle(T, oracle, L, M, n, bf16, auto, steps):
    """tpl_step_observe -- the move and the [N,217] observation in one launch -- against the oracle's step followed by
    its expand_obs (rewards and dones at every step, the observation at every step of the small cases and at a few steps
    of the large ones), and against tpl_step + tpl_expand_obs on a twin handle.  BASELINE's configs[1] and configs[2]
    sizes, ragged sizes around a wave's and a block's span, with and without auto-reset (without it: frozen boards,
    among them a win whose last clear overshot L -- lines left below zero)."""
    import torch
    dtype = torch.bfloat16 if bf16 else torch.float32
    seed = 31 + n % 97
    pool_n = min(n, 4096)
    rows, pieces = oracle.synth_boards(seed, 0, pool_n, L), oracle.synth_pieces(seed, 0, pool_n, M)
    if not auto and pool_n >= 7:                       # the overshoot of test_observation_on_dense_boards...: I upright at x = 0
        rows[3] = 0
        rows[3, 17:20] = 0x3FE
        pieces[3, 0] = 0
    make = lambda: T.BatchedTetris(L, M, n, seed=seed, auto_reset=auto, assign="sequential", config_pool=(rows, pieces))
    gpu, twin = make(), make()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows, pieces)
    cpu.set_options(auto_reset=auto, assign_mode=1)
    gpu.reset(); twin.reset(); cpu.reset()
    obs = torch.empty((n, 217), dtype=dtype, device=gpu.device)
    reward = torch.empty(n, dtype=torch.float32, device=gpu.device)
    done = torch.empty(n, dtype=torch.uint8, device=gpu.device)
    r2, d2 = torch.empty_like(reward), torch.empty_like(done)
    look = set(range(steps)) if n <= 4096 else {0, 1, steps // 2, steps - 1}
    for t in range(steps):
        a = gpu.synthetic_actions(t)
        if t == 0 and not auto:
            a[:] = 10                                  # rot 1, loc 0: the upright I into the three-row well of board 3
        gpu.step_observe_into(a, reward, done, obs)
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(reward), r_c) and np.array_equal(_np(done), d_c), t
        twin.step_into(a, r2, d2)
        if t in look:
            want = cpu.expand_obs()
            got = _np(obs.float())
            assert np.array_equal(got, want), (t, np.argwhere(got != want)[:5].tolist())
            assert torch.equal(twin.observe(dtype), obs), t
    _assert_state_equal(_state(gpu), cpu.get_state(), "step_observe")
    assert gpu.stats() == cpu.stats() and gpu.stats() == twin.stats()
    if not auto and pool_n >= 7 and L < 3:
        final = _np(obs.float())
        assert final[3, 214] < 0 and final[3, 216] == 1               # three rows cleared where two were asked for; frozen
    # an observation buffer that is not 16-byte aligned is refused (tpl_step + tpl_expand_obs serve it)
    spare = torch.empty(n * 217 + 8, dtype=dtype, device=gpu.device)
    with pytest.raises(T.TplError, match="16-byte aligned"):
        gpu.step_observe_into(a, reward, done, spare[1: 1 + n * 217].view(n, 217))
    gpu.terminate(); twin.terminate()


def test_sharding_is_independent_of_the_number_of_gpus(T):
    """Two handles with global offsets reproduce one handle over the whole batch (the multi-GPU partition)."""
    import torch
    L, M, n, seed = 5, 20, 16384, 9
    whole = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
    rows, pieces = whole.synthetic_configs(2048)
    parts = [T.BatchedTetris(L, M, n // 2, seed=seed, auto_reset=True, global_offset=k * (n // 2)) for k in range(2)]
    for e in [whole] + parts:
        e.load_configs(rows, pieces)
        e.reset()
    for t in range(2 * M):
        whole.step(whole.synthetic_actions(t), observe=False)
        for e in parts:
            e.step(e.synthetic_actions(t), observe=False)
    w = _state(whole)
    p = [_state(e) for e in parts]
    for k in w:
        assert np.array_equal(w[k], np.concatenate([q[k] for q in p]))
    total = sum(torch.stack([e.stats_tensor() for e in parts]))
    assert total.tolist() == whole.stats_tensor().tolist()
    for e in [whole] + parts:
        e.terminate()


def test_full_size_run_properties_and_oracle_equality(T, oracle):
    """BASELINE's roofline configuration (1,048,576 boards, L=10, M=40): size-independent properties on every
    step, and equality with the oracle over the whole batch at the end."""
    L, M, n, seed = 10, 40, 1 << 20, 0
    gpu = T.BatchedTetris(L, M, n, seed=seed, assign="sequential")
    rows, pieces = gpu.synthetic_configs(n)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    for t in range(M):
        a = gpu.synthetic_actions(t)
        _, r, d, _ = gpu.step(a, observe=False)
        if t % 8 == 0 or t == M - 1:
            s = _state(gpu)
            assert np.all(s["moves"] <= M) and np.all(s["state"] <= 2)
            assert np.all(s["rows"] < 1024)                                    # only 10 columns ever set
            assert np.all((s["state"] != 0) == _np(d))
    # conservation on the final state: every successful move adds 4 cells, every cleared line removes 10
    s = _state(gpu)
    start_cells = _popcount(_np(rows).view(np.uint16))
    assert np.array_equal(_popcount(s["rows"]), start_cells + 4 * s["moves"].astype(np.int64) - 10 * s["lines"].astype(np.int64))
    # whole-batch equality with the oracle
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(assign_mode=1)
    cpu.reset()
    for t in range(M):
        cpu.step(oracle.synth_actions(seed, 0, n, t))
    _assert_state_equal(s, cpu.get_state(), "1M boards")
    # idempotence: every board is finished, one more step changes nothing
    gpu.step(gpu.synthetic_actions(M), observe=False)
    _assert_state_equal(_state(gpu), s, "frozen")
    gpu.terminate()


def test_benched_configuration_in_lockstep_with_the_oracle(T, oracle):
    """Exactly what bench.py times (BASELINE configs[2]): 1,048,576 boards, one pool entry per board, auto-reset, hashed
    assignment, tpl_step through step_into() with uint8 actions -- 60 lockstep steps against the oracle: rewards and
    dones of every step, the whole state at the end, statistics.  (Reference semantics: game/tetris.py:354-449.)"""
    import os
    import torch
    L, M, n, seed = 10, 40, 1 << 20, 0
    steps = int(os.environ.get("TPL_BENCH_PARITY_STEPS", "60"))     # a one-off longer run: TPL_BENCH_PARITY_STEPS=1500
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, assign="hash")
    rows, pieces = gpu.synthetic_configs(n)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    reward = torch.empty(n, dtype=torch.float32, device=gpu.device)
    done = torch.empty(n, dtype=torch.uint8, device=gpu.device)
    for t in range(steps):
        a = gpu.synthetic_actions(t)
        gpu.step_into(a, reward, done)
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(reward), r_c) and np.array_equal(_np(done), d_c), t
    _assert_state_equal(_state(gpu), cpu.get_state(), "benched configuration")
    st = gpu.stats()
    assert st == cpu.stats() and st["episodes"] > 4 * n and gpu.step_clock() == cpu.clock == steps
    gpu.terminate()


def test_configs3_shard_geometry_eight_shards_of_131072_equal_the_one_gpu_job(T, oracle):
    """BASELINE configs[3] in its exact geometry, on one GPU: EIGHT handles of 131,072 boards at global offsets k x 131,072
    (`sharding.strong_shard(k, 8, 2^20)`: what the eight ranks of `bench.py --gpus 8` build), every one over the same
    2^20-entry pool, stepped 60 times through step_into() with their own slice of the synthetic actions.  Per step the
    eight shards' rewards and dones, concatenated, are the oracle's over the ONE 2^20-board job; at the end so are the state,
    the summed statistics and the mean episodic return the ranks' all-reduce would give (the sum of the eight [return sum,
    episodes] pairs).  Reference: boards are independent objects (game/tetris.py:354-449; one board per Tetris instance)."""
    import os
    import torch
    L, M, total, ranks, seed = 10, 40, 1 << 20, 8, 0
    steps = int(os.environ.get("TPL_BENCH_PARITY_STEPS", "60"))
    shards = [T.sharding.strong_shard(k, ranks, total) for k in range(ranks)]
    assert [(s.boards, s.global_offset) for s in shards] == [(131072, k * 131072) for k in range(ranks)]
    envs = [T.BatchedTetris(L, M, s.boards, seed=seed, global_offset=s.global_offset, auto_reset=True, assign="hash") for s in shards]
    rows, pieces = envs[0].synthetic_configs(total, first=0)          # the same pool on every rank, as bench.py loads it
    for e in envs:
        e.load_configs(rows, pieces)
        e.reset()
    cpu = oracle.Env(total, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    dev = envs[0].device
    reward = [torch.empty(s.boards, dtype=torch.float32, device=dev) for s in shards]
    done = [torch.empty(s.boards, dtype=torch.uint8, device=dev) for s in shards]
    for t in range(steps):
        acts = [e.synthetic_actions(t) for e in envs]                # keyed by the GLOBAL board index
        for e, a, r, d in zip(envs, acts, reward, done):
            e.step_into(a, r, d)
        r_c, d_c = cpu.step(_np(torch.cat(acts)))
        assert np.array_equal(_np(torch.cat(reward)), r_c) and np.array_equal(_np(torch.cat(done)), d_c), t
    got = [_state(e) for e in envs]
    _assert_state_equal({k: np.concatenate([g[k] for g in got]) for k in got[0]}, cpu.get_state(), "eight shards")
    summed = {k: sum(e.stats()[k] for e in envs) for k in envs[0].stats()}
    assert summed == cpu.stats() and summed["episodes"] > 4 * total
    assert all(e.step_clock() == steps for e in envs)
    # what the job's one collective computes: the SUM over ranks of [return sum, episodes]
    acc = sum(T.sharding.return_sum(e.stats_tensor(), e.reward_params) for e in envs)
    one = T.sharding.return_sum(torch.tensor([cpu.stats()[k] for k in ("episodes", "lines", "wins", "topouts")], dtype=torch.int64),
                                envs[0].reward_params)
    assert torch.equal(acc.cpu(), one)
    for e in envs:
        e.terminate()


def test_config1_shape_in_lockstep_with_the_oracle(T, oracle):
    """BASELINE configs[1] as bench.py's config1_run drives it: 65,536 boards, L=5, M=20, pool = boards, auto-reset, hash,
    fused rollouts of 50 steps and single steps interleaved."""
    import torch
    L, M, n, seed = 5, 20, 65536, 0
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, assign="hash")
    rows, pieces = gpu.synthetic_configs(n)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    actions = torch.stack([gpu.synthetic_actions(t) for t in range(130)])
    rsum, fin, rs, ds = gpu.rollout(actions[:50], per_step=True)
    for t in range(50):
        r_c, d_c = cpu.step(_np(actions[t]))
        assert np.array_equal(_np(rs[t]), r_c) and np.array_equal(_np(ds[t]).astype(np.uint8), d_c), t
    for t in range(50, 80):
        _, r, d, _ = gpu.step(actions[t], observe=False)
        r_c, d_c = cpu.step(_np(actions[t]))
        assert np.array_equal(_np(r), r_c) and np.array_equal(_np(d).astype(np.uint8), d_c), t
    gpu.rollout_into(actions[80:], 50)
    for t in range(80, 130):
        cpu.step(_np(actions[t]))
    _assert_state_equal(_state(gpu), cpu.get_state(), "config 1")
    assert gpu.stats() == cpu.stats()
    gpu.terminate()


def _popcount(rows):
    r = rows.astype(np.uint16)
    return np.unpackbits(r.view(np.uint8), axis=-1).reshape(r.shape[0], -1).sum(1).astype(np.int64)


@pytest.mark.parametrize("assign", ["hash", "sequential"])
def test_global_offsets_beyond_32_bits(T, oracle, assign):
    """A shard far into a (hypothetical) huge global batch: board indices above 2^32 key the generator and the
    configuration assignment exactly as on the CPU."""
    L, M, n, pool, seed, off = 5, 20, 3000, 1237, 41, (1 << 33) + 12345
    gpu = T.BatchedTetris(L, M, n, seed=seed, global_offset=off, auto_reset=True, assign=assign)
    rows, pieces = gpu.synthetic_configs(pool, first=off)
    assert np.array_equal(_np(rows).view(np.uint16), oracle.synth_boards(seed, off, pool, L))
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, off, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0 if assign == "hash" else 1)
    cpu.reset()
    for t in range(50):
        a = gpu.synthetic_actions(t)
        assert np.array_equal(_np(a), oracle.synth_actions(seed, off, n, t))
        _, r, d, _ = gpu.step(a, observe=False)
        r_c, d_c = cpu.step(_np(a))
        assert np.array_equal(_np(r), r_c) and np.array_equal(_np(d).astype(np.uint8), d_c), t
    _assert_state_equal(_state(gpu), cpu.get_state(), "offset")
    assert gpu.stats() == cpu.stats()
    gpu.terminate()


@pytest.mark.parametrize("L,M", [(2, 7), (4, 33), (10, 40)])
def test_long_soak_against_the_oracle(T, oracle, L, M):
    """Two thousand lockstep steps with auto-reset from carved (solvable) configurations mixed with synthetic
    ones, alternating single steps and fused rollouts; compared with the oracle at the end and on the way."""
    import torch
    import os
    n, seed, steps = 16384, 1000 + L, int(os.environ.get("TPL_SOAK_STEPS", "2000"))    # a one-off longer run: TPL_SOAK_STEPS=40000
    gen = min(L, 16)
    c_rows, c_pieces = T.generate_configs(gen, M, 1500, seed=seed) if M >= 2 * gen else (np.zeros((0, 20), np.uint16), np.zeros((0, M + 1), np.uint8))
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, reward=(0.5, 4.0, -0.25))
    s_rows, s_pieces = gpu.synthetic_configs(1500)
    rows = np.concatenate([c_rows, _np(s_rows).view(np.uint16)])
    pieces = np.concatenate([c_pieces, _np(s_pieces)])
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(rows, pieces)
    cpu.set_options(auto_reset=True, assign_mode=0, per_line=0.5, win=4.0, lose=-0.25)
    cpu.reset()
    t = 0
    while t < steps:
        if (t // 100) % 2 == 0:                                  # a block of single steps
            for _ in range(100):
                a = gpu.synthetic_actions(t)
                _, r, d, _ = gpu.step(a, observe=False)
                r_c, d_c = cpu.step(_np(a))
                t += 1
            assert np.array_equal(_np(r), r_c) and np.array_equal(_np(d).astype(np.uint8), d_c), t
        else:                                                    # a fused rollout of 100 steps
            acts = torch.stack([gpu.synthetic_actions(t + k) for k in range(100)])
            rsum, fin = gpu.rollout(acts)
            acc = np.zeros(n, np.float32)
            for k in range(100):
                r_c, d_c = cpu.step(_np(acts[k]))
                acc += r_c
            t += 100
            assert np.array_equal(_np(rsum), acc), t
        _assert_state_equal(_state(gpu), cpu.get_state(), f"L={L} M={M} step {t}")
    assert gpu.stats() == cpu.stats() and gpu.stats()["wins"] > 0 or L > 4
    gpu.terminate()


def test_step_is_graph_capturable_and_replays_exactly(T, oracle):
    """The ABI promises no host synchronisation inside tpl_step: capture 8 steps into one HIP graph, replay it."""
    import torch
    L, M, n, seed = 5, 20, 20000, 31
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
    rows, pieces = gpu.synthetic_configs(2048)
    gpu.load_configs(rows, pieces)
    gpu.reset()
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
    cpu.set_options(auto_reset=True, assign_mode=0)
    cpu.reset()
    actions = torch.stack([gpu.synthetic_actions(t) for t in range(8)])
    rewards = torch.zeros((8, n), dtype=torch.float32, device=gpu.device)
    dones = torch.zeros((8, n), dtype=torch.uint8, device=gpu.device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        saved = gpu.snapshot()
        gpu.step_into(actions[0], rewards[0], dones[0])         # warm-up outside the capture
        gpu.restore(saved)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for t in range(8):
                gpu.step_into(actions[t], rewards[t], dones[t])
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(3):                                         # the same 8 actions again each replay
        graph.replay()
        torch.cuda.synchronize()
        for t in range(8):
            r_c, d_c = cpu.step(_np(actions[t]))
            assert np.array_equal(_np(rewards[t]), r_c) and np.array_equal(_np(dones[t]), d_c), (rep, t)
    _assert_state_equal(_state(gpu), cpu.get_state(), "graph")
    gpu.terminate()


def test_capture_steps_replays_and_recaptures_after_a_pool_swap(T, oracle):
    """BatchedTetris.capture_steps: K steps in one HIP graph, replayed; after load_configs() the callable captures again
    (the old graph would read the old pool buffer).  Against the oracle throughout."""
    import torch
    L, M, n, seed, K = 5, 20, 7000, 41, 6
    gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True)
    cpu = oracle.Env(n, L, M, 0, seed)
    cpu.set_options(auto_reset=True, assign_mode=0)
    pool_a = (oracle.synth_boards(1, 0, 500, L), oracle.synth_pieces(1, 0, 500, M))
    pool_b = (oracle.synth_boards(2, 0, 300, L), oracle.synth_pieces(2, 0, 300, M))
    gpu.load_configs(*pool_a); cpu.set_pool(*pool_a)
    gpu.reset(); cpu.reset()
    actions = torch.empty((K, n), dtype=torch.uint8, device=gpu.device)
    rewards = torch.empty((K, n), dtype=torch.float32, device=gpu.device)
    dones = torch.empty((K, n), dtype=torch.uint8, device=gpu.device)
    replay = gpu.capture_steps(actions, rewards, dones)
    t = 0
    for block in range(12):
        if block == 5:
            gpu.load_configs(*pool_b); cpu.set_pool(*pool_b)
        if block == 10:                                            # 30 replayed steps later (M + 1 = 21): the guard has been told
            assert gpu.pool_info()["steps_until_swap"] == 0
            gpu.load_configs(*pool_a); cpu.set_pool(*pool_a)
        for k in range(K):
            gpu.synthetic_actions(t + k, out=actions[k])
        replay()
        for k in range(K):
            r_c, d_c = cpu.step(_np(actions[k]))
            assert np.array_equal(_np(rewards[k]), r_c) and np.array_equal(_np(dones[k]), d_c), (block, k)
        t += K
    _assert_state_equal(_state(gpu), cpu.get_state(), "capture_steps")
    assert gpu.stats() == cpu.stats() and gpu.step_clock() == cpu.clock == t
    gpu.terminate()


@pytest.mark.parametrize("n", [1, 63, 1000, 2049, 5000])
def test_ragged_batches_match_oracle_in_every_geometry(T, oracle, n):
    """Batch sizes that do not fill the last block: the step kernel reads and writes back the padding boards behind
    the batch (created finished, never advanced, never counted).  Every geometry, auto-reset on, statistics equal."""
    L, M, seed = 4, 12, 11
    for bpl in (1, 2, 4):
        for threads in (64, 128, 256, 512):
            gpu = T.BatchedTetris(L, M, n, seed=seed, auto_reset=True, reward=(1.0, 2.0, -3.0))
            gpu.set_tuning(bpl, threads)
            rows, pieces = gpu.synthetic_configs(97)
            gpu.load_configs(rows, pieces)
            gpu.reset()
            cpu = oracle.Env(n, L, M, 0, seed)
            cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
            cpu.set_options(auto_reset=True, assign_mode=0, per_line=1.0, win=2.0, lose=-3.0)
            cpu.reset()
            for t in range(2 * M + 3):
                a = gpu.synthetic_actions(t)
                _, r_g, d_g, _ = gpu.step(a, observe=False)
                r_c, d_c = cpu.step(_np(a))
                assert np.array_equal(_np(r_g), r_c) and np.array_equal(_np(d_g).astype(np.uint8), d_c), (bpl, threads, t)
            _assert_state_equal(_state(gpu), cpu.get_state(), f"n={n} bpl={bpl} threads={threads}")
            assert gpu.stats() == cpu.stats()
            gpu.terminate()


def test_randomised_sweep_of_sizes_rules_and_geometries(T, oracle):
    """Thirty random draws of (L, M, batch, pool, assignment, auto-reset, reward, geometry, action dtype), each stepped
    past several episode ends and compared with the oracle: rewards, dones, final state, statistics."""
    import torch
    rng = np.random.default_rng(2024)
    for case in range(30):
        L = int(rng.integers(1, 13))
        M = int(rng.choice([1, 2, 5, 7, 8, 9, 16, 23, 40, 71, 72, 100]))
        n = int(rng.choice([1, 17, 64, 100, 513, 3000]))
        pool = int(rng.integers(1, 200))
        assign = ["hash", "sequential"][int(rng.integers(0, 2))]
        auto = bool(rng.integers(0, 2))
        reward = tuple(float(x) for x in rng.choice([0.0, 0.1, 1.0, -0.3, 2.5], 3))
        bpl, threads = int(rng.choice([1, 2, 4])), int(rng.choice([64, 128, 256, 512]))
        dtype = [np.uint8, np.int32, np.int64][int(rng.integers(0, 3))]
        seed, offset = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 20))
        tag = f"case {case}: L={L} M={M} n={n} pool={pool} {assign} auto={auto} reward={reward} bpl={bpl}x{threads} {dtype.__name__}"
        gpu = T.BatchedTetris(L, M, n, seed=seed, global_offset=offset, auto_reset=auto, assign=assign, reward=reward)
        gpu.set_tuning(bpl, threads)
        rows, pieces = gpu.synthetic_configs(pool, seed=seed + 1)
        gpu.load_configs(rows, pieces)
        gpu.reset()
        cpu = oracle.Env(n, L, M, offset, seed)
        cpu.set_pool(_np(rows).view(np.uint16), _np(pieces))
        cpu.set_options(auto_reset=auto, assign_mode=0 if assign == "hash" else 1, per_line=reward[0], win=reward[1], lose=reward[2])
        cpu.reset()
        twin = T.BatchedTetris(L, M, n, seed=seed, global_offset=offset, auto_reset=auto, assign=assign, reward=reward)
        twin.load_configs(rows, pieces)                            # the same run through the fused K-step kernel
        twin.reset()
        steps = min(3 * M + 5, 60)
        acts = rng.integers(0, 40, (steps, n)).astype(np.uint8)
        rewards = np.zeros((steps, n), np.float32)
        dones = np.zeros((steps, n), np.uint8)
        for t in range(steps):
            _, r_g, d_g, _ = gpu.step(torch.from_numpy(acts[t].astype(dtype)), observe=False)
            rewards[t], dones[t] = cpu.step(acts[t])
            assert np.array_equal(_np(r_g), rewards[t]) and np.array_equal(_np(d_g).astype(np.uint8), dones[t]), (tag, t)
        _assert_state_equal(_state(gpu), cpu.get_state(), tag)
        assert gpu.stats() == cpu.stats(), tag
        assert np.array_equal(_np(gpu.observe()), cpu.expand_obs()), tag
        cut = int(rng.integers(1, steps)) if steps > 1 else 1
        dev_acts = torch.from_numpy(acts).to(twin.device)
        parts = [twin.rollout(dev_acts[:cut], per_step=True)] + ([twin.rollout(dev_acts[cut:], per_step=True)] if cut < steps else [])
        assert np.array_equal(np.concatenate([_np(q[2]) for q in parts]), rewards), tag
        assert np.array_equal(np.concatenate([_np(q[3]).astype(np.uint8) for q in parts]), dones), tag
        _assert_state_equal(_state(twin), cpu.get_state(), tag + " (rollout)")
        assert twin.stats() == cpu.stats(), tag
        gpu.terminate(); twin.terminate()


def test_ragged_sizes_and_errors(T):
    import torch
    for n in (1, 63, 65, 257):
        env = T.BatchedTetris(4, 9, n, assign="sequential")
        rows, pieces = env.synthetic_configs(7)
        env.load_configs(rows, pieces)
        env.reset()
        env.step(env.synthetic_actions(0))
        s = env.packed_state()
        assert s["moves"].shape[0] == n
        # the reference's public attributes, batched: each is one field of packed_state()
        assert torch.equal(env.lines_cleared, s["lines"]) and torch.equal(env.moves_used, s["moves"]) and torch.equal(env.state, s["state"])
        env.terminate()
    env = T.BatchedTetris(4, 9, 8)
    with pytest.raises(T.TplError):
        env.reset()                       # no configurations loaded yet
    with pytest.raises(ValueError):
        env.step(np.zeros(5, np.uint8))   # wrong batch size
    env.terminate()
