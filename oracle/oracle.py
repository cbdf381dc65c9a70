"""ctypes loader for the CPU oracle (oracle/tetris_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package never does (tests/test_boundary.py greps for it).  Every wrapper maps 1:1 onto a C function whose
header comment cites the reference lines it restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtetris_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("tetris_oracle.c", "tetris_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


class _Game(C.Structure):
    _fields_ = [("rows", C.c_uint16 * 20), ("lines_cleared", C.c_int32), ("moves_used", C.c_int32),
                ("state", C.c_int32), ("cursor", C.c_int32)]


class _Shape(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("mask", C.c_uint8 * 4), ("revtopo", C.c_uint8 * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
        L.to_num_rotations.restype = i32
        L.to_num_rotations.argtypes = [i32]
        L.to_get_tetromino.argtypes = [i32, i32, C.POINTER(_Shape)]
        L.to_move.restype = i32
        L.to_move.argtypes = [C.POINTER(_Game), vp, i32, i32, i32, i32]
        L.to_env_create.restype = vp
        L.to_env_create.argtypes = [i64, i32, i32, i64, u64]
        L.to_env_destroy.argtypes = [vp]
        L.to_env_set_pool.argtypes = [vp, vp, vp, i64]
        L.to_env_set_options.argtypes = [vp, i32, i32, f32, f32, f32]
        L.to_env_assign.restype = i64
        L.to_env_assign.argtypes = [vp, i64, u64]
        L.to_env_clock.restype = u64
        L.to_env_clock.argtypes = [vp]
        L.to_env_birth.restype = u64
        L.to_env_birth.argtypes = [vp, i64]
        L.to_env_reset.argtypes = [vp, vp]
        L.to_env_move.argtypes = [vp, vp, vp, vp, vp, vp]
        L.to_env_step.argtypes = [vp, vp, vp, vp]
        L.to_env_get_state.argtypes = [vp] * 8
        L.to_env_expand_obs.argtypes = [vp, vp]
        L.to_env_get_stats.argtypes = [vp, vp]
        L.to_explore_actions.argtypes = [vp, i64, i64, u64, C.c_uint32, C.c_uint32]
        L.to_explore_actions.restype = None
        L.to_rng.restype = u64
        L.to_rng.argtypes = [u64, u64, u64, u64]
        L.to_synth_boards.argtypes = [u64, i64, i64, i32, vp]
        L.to_synth_pieces.argtypes = [u64, i64, i64, i32, vp]
        L.to_synth_actions.argtypes = [u64, i64, i64, u64, vp]
        L.to_carve.restype = i32
        L.to_carve.argtypes = [vp, i32, i32, i32, i32]
        L.to_generate_config_tape.restype = i64
        L.to_generate_config_tape.argtypes = [i32, i32, vp, i64, C.POINTER(i64), vp, vp, vp, C.POINTER(i32)]
        L.to_generate_config_seeded.restype = i64
        L.to_generate_config_seeded.argtypes = [i32, i32, u64, u64, i64, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
        L.to_carve_attempt_limit.restype = i64
        L.to_carve_attempt_limit.argtypes = [i32, i64, i32]
        L.to_board_hash.restype = u64
        L.to_board_hash.argtypes = [vp]
        L.to_bench_run.restype = i64
        L.to_bench_run.argtypes = [u64, i64, i32, i32, i64, i32, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def get_tetromino(piece: int, rotations: int):
    s = _Shape()
    lib().to_get_tetromino(piece, rotations, C.byref(s))
    return s.h, s.w, list(s.mask)[: s.h], list(s.revtopo)[: s.w]


def num_rotations(piece: int) -> int:
    return lib().to_num_rotations(piece)


class Game:
    """One board with the reference's attribute names (game/tetris.py:143-151,186-187)."""

    def __init__(self, L, M, rows=None, pieces=None, lines_cleared=0, moves_used=0, state=0):
        self.L, self.M = L, M
        self._g = _Game()
        if rows is not None:
            for r in range(20):
                self._g.rows[r] = int(rows[r])
        self._g.lines_cleared, self._g.moves_used, self._g.state, self._g.cursor = lines_cleared, moves_used, state, 0
        self._pieces = np.ascontiguousarray(pieces if pieces is not None else [], dtype=np.uint8)

    def move(self, rotations: int, location: int) -> int:
        return lib().to_move(C.byref(self._g), _p(self._pieces), self.L, self.M, rotations, location)

    rows = property(lambda self: np.array(list(self._g.rows), dtype=np.uint16))
    lines_cleared = property(lambda self: self._g.lines_cleared)
    moves_used = property(lambda self: self._g.moves_used)
    state = property(lambda self: self._g.state)
    cursor = property(lambda self: self._g.cursor)
    pieces = property(lambda self: self._pieces[self._g.cursor:].tolist())


class Env:
    """Batched oracle environment with the same rules as the HIP library (freeze / auto-reset / reward)."""

    def __init__(self, n, L, M, global_offset=0, seed=0):
        self.n, self.L, self.M = n, L, M
        self._h = lib().to_env_create(n, L, M, global_offset, seed)
        self._pool = None

    def close(self):
        if self._h:
            lib().to_env_destroy(self._h)
            self._h = None

    __del__ = close

    def set_pool(self, rows, pieces):
        rows = np.ascontiguousarray(rows, dtype=np.uint16).reshape(-1, 20)
        pieces = np.ascontiguousarray(pieces, dtype=np.uint8).reshape(rows.shape[0], self.M + 1)
        self._pools = getattr(self, "_pools", []) + [(rows, pieces)]   # borrowed by the C side: keep alive
        lib().to_env_set_pool(self._h, _p(rows), _p(pieces), rows.shape[0])

    def set_options(self, auto_reset=False, assign_mode=0, per_line=1.0, win=0.0, lose=0.0):
        lib().to_env_set_options(self._h, int(auto_reset), int(assign_mode), per_line, win, lose)

    def assign(self, board, birth):
        """Pool entry (current slot) of the episode of `board` that begins at step `birth`."""
        return lib().to_env_assign(self._h, board, birth)

    @property
    def clock(self):
        return lib().to_env_clock(self._h)

    def birth(self, board):
        return lib().to_env_birth(self._h, board)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().to_env_reset(self._h, _p(m))

    def move(self, rot, loc):
        rot = np.ascontiguousarray(rot, dtype=np.uint8)
        loc = np.ascontiguousarray(loc, dtype=np.uint8)
        reward = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        cleared = np.empty(self.n, np.uint8)
        lib().to_env_move(self._h, _p(rot), _p(loc), _p(reward), _p(done), _p(cleared))
        return reward, done, cleared

    def step(self, action):
        action = np.ascontiguousarray(action, dtype=np.uint8)
        reward = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        lib().to_env_step(self._h, _p(action), _p(reward), _p(done))
        return reward, done

    def get_state(self):
        n = self.n
        out = dict(rows=np.empty((n, 20), np.uint16), cur=np.empty(n, np.uint8), nxt=np.empty(n, np.uint8),
                   lines=np.empty(n, np.uint8), moves=np.empty(n, np.uint8), state=np.empty(n, np.uint8),
                   pieces_left=np.empty(n, np.uint8))
        lib().to_env_get_state(self._h, *[_p(out[k]) for k in
                                          ("rows", "cur", "nxt", "lines", "moves", "state", "pieces_left")])
        return out

    def expand_obs(self):
        out = np.empty((self.n, 217), np.float32)
        lib().to_env_expand_obs(self._h, _p(out))
        return out

    def stats(self):
        raw = (C.c_uint64 * 4)()
        lib().to_env_get_stats(self._h, raw)
        return dict(episodes=raw[0], lines=raw[1], wins=raw[2], topouts=raw[3])


def rng(seed, stream, index, counter):
    return lib().to_rng(seed, stream, index, counter)


def synth_boards(seed, first, count, L):
    rows = np.empty((count, 20), np.uint16)
    lib().to_synth_boards(seed, first, count, L, _p(rows))
    return rows


def synth_pieces(seed, first, count, M):
    p = np.empty((count, M + 1), np.uint8)
    lib().to_synth_pieces(seed, first, count, M, _p(p))
    return p


def synth_actions(seed, first, count, step):
    a = np.empty(count, np.uint8)
    lib().to_synth_actions(seed, first, count, step, _p(a))
    return a


def explore_actions(action, epsilon, seed, step, global_offset=0):
    """tpl_explore_actions on a copy of `action` (uint8): replaced by the exploration draw with probability epsilon."""
    a = np.ascontiguousarray(action, dtype=np.uint8).copy()
    eps_q24 = int(np.float32(epsilon) * np.float32(16777216.0))
    lib().to_explore_actions(_p(a), len(a), global_offset, seed, step, eps_q24)
    return a


def carve(rows, piece, rotations, location, allow_partial):
    """Tetris.carve on a copy of rows; returns (ok, rows_after)."""
    r = np.ascontiguousarray(rows, dtype=np.uint16).copy()
    ok = lib().to_carve(_p(r), piece, rotations, location, int(allow_partial))
    return bool(ok), r


def _gen_outputs(M):
    return np.zeros(20, np.uint16), np.zeros(M + 1, np.uint8), np.zeros((M, 2), np.uint8), C.c_int32(0)


def generate_config_tape(L, M, tape):
    """The carving generator driven by a tape of (lo, hi, value) decisions recorded from the reference."""
    tape = np.ascontiguousarray(tape, dtype=np.int32).reshape(-1, 3)
    rows, pieces, sol, n = _gen_outputs(M)
    used = C.c_int64(0)
    it = lib().to_generate_config_tape(L, M, _p(tape), tape.shape[0], C.byref(used), _p(rows), _p(pieces), _p(sol), C.byref(n))
    return it, used.value, rows, pieces, sol[: n.value]


def generate_config_seeded(L, M, seed, index, cutoff=0, with_attempt=False):
    """The build's seeded generator under its restart rule: (iterations or -1, rows, pieces, solution[, winning attempt])."""
    rows, pieces, sol, n = _gen_outputs(M)
    attempt = C.c_int32(0)
    it = lib().to_generate_config_seeded(L, M, seed, index, cutoff, _p(rows), _p(pieces), _p(sol), C.byref(n), C.byref(attempt))
    return (it, rows, pieces, sol[: n.value], attempt.value) if with_attempt else (it, rows, pieces, sol[: n.value])


def carve_attempt_limit(L, cutoff, attempt):
    return lib().to_carve_attempt_limit(L, cutoff, attempt)


def board_hash(rows):
    rows = np.ascontiguousarray(rows, dtype=np.uint16)
    return lib().to_board_hash(_p(rows))


def bench_run(seed, count, L, M, steps, threads=1):
    sec = C.c_double(0.0)
    done = lib().to_bench_run(seed, count, L, M, steps, threads, C.byref(sec))
    return done, sec.value
