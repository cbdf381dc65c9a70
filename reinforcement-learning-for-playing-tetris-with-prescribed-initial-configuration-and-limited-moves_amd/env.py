"""Host-side mirror of the reference's `Tetris` interface, batched over the C ABI.

`BatchedTetris` keeps the reference's surface -- reset() / move(rotations, location) / get_state() /
terminate() and the attributes L, M, lines_cleared, moves_used, state (game/tetris.py:140-151,354,435-451 of
the upstream repo) -- for N boards that live on one MI355X, plus the step(action) / reward / done form that
BASELINE.json's north_star asks for.  `Tetris` is the single-board form with the reference's exact attribute
types, used by tests that read like the reference's own.

All compute happens in csrc/tetris_piclim.hip; this module only owns torch buffers and passes raw pointers
and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check

OBS_DIM = 217
NUM_ACTIONS = 40
RUNNING, WON, LOST = 0, 1, 2

_INT_CODES = {torch.uint8: _lib.TPL_U8, torch.int32: _lib.TPL_I32, torch.int64: _lib.TPL_I64}
_OBS_CODES = {torch.float32: _lib.TPL_F32, torch.bfloat16: _lib.TPL_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Snapshot:
    """What BatchedTetris.snapshot() returns: a copy of the resident state together with what it refers to -- the pool
    generation and assignment mode it was taken under and the pool-swap guard's value at that moment.  The three travel
    WITH the copy (clone(), to()), so restore() can always tell a foreign snapshot from its own."""

    __slots__ = ("data", "pool_generation", "assign", "hold")

    def __init__(self, data: torch.Tensor, pool_generation: int, assign: str, hold: int):
        self.data, self.pool_generation, self.assign, self.hold = data, int(pool_generation), assign, int(hold)

    def clone(self) -> "Snapshot":
        return Snapshot(self.data.clone(), self.pool_generation, self.assign, self.hold)

    def to(self, *args, **kwargs) -> "Snapshot":
        return Snapshot(self.data.to(*args, **kwargs), self.pool_generation, self.assign, self.hold)


class BatchedTetris:
    """N independent Tetris-piclim boards on one GPU, stepped in lockstep.

    Parameters mirror `Tetris(L, M, ...)` (game/tetris.py:141); the warm-reset worker processes of the
    reference (:189-214) are replaced by a device-resident pool of prescribed configurations
    (`load_configs`), from which resets draw.

    Which configuration an episode starts from is a function of (seed, global board index, the step at which the
    episode begins): assign="hash" hashes the three, assign="sequential" takes entry (global index + step) mod pool
    size.  Steps are counted on the device since the last full reset(), 64 bits wide, so a board's sequence of
    configurations never repeats and does not depend on how many GPUs share the batch.  The pool can be replaced
    while boards run (`load_configs` again): boards that are mid-episode finish on the pool they started from.
    """

    def __init__(self, L: int, M: int, num_envs: int, device="cuda:0", seed: int = 0, global_offset: int = 0,
                 auto_reset: bool = False, assign: str = "hash", reward=(1.0, 0.0, 0.0), config_pool=None):
        self.L, self.M, self.num_envs = int(L), int(M), int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("BatchedTetris runs on a GPU only (device must be cuda:N); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP library cannot run and no fallback exists")
        self._lib = _lib.lib()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._index = idx
        self._shape1 = torch.Size((self.num_envs,))
        self.seed, self.global_offset = int(seed), int(global_offset)
        nbytes = self._lib.tpl_workspace_bytes(self.num_envs, self.M)
        self._workspace = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        check(self._lib.tpl_create(C.byref(h), self.num_envs, self.L, self.M, idx, self.global_offset, self.seed,
                                   _ptr(self._workspace), nbytes))
        self._h = h
        self._pool_mems = [None, None]       # the two pool buffers of the handle (device memory it borrows)
        self._pool = None
        self.n_configs = 0
        self.pool_generation = 0             # bumped by every load_configs(): graphs captured before it are stale
        self.auto_reset = bool(auto_reset)
        self.assign = assign
        self.reward_params = tuple(float(x) for x in reward)
        self._apply_options()
        if config_pool is not None:
            self.load_configs(*config_pool)

    # ------------------------------------------------------------------------------------------ plumbing
    def _stream(self):
        # the current stream of this device as a raw hipStream_t (an int; ctypes passes it as void*)
        return torch._C._cuda_getCurrentRawStream(self._index)

    def _apply_options(self):
        mode = {"hash": _lib.TPL_ASSIGN_HASH, "sequential": _lib.TPL_ASSIGN_SEQUENTIAL}[self.assign]
        check(self._lib.tpl_set_options(self._h, int(self.auto_reset), mode, *self.reward_params))

    def set_tuning(self, boards_per_lane: int = 2, block_threads: int = 256) -> None:
        """Step-kernel geometry (boards per lane 1/2/4, threads per block 64..512); results do not depend on it."""
        check(self._lib.tpl_set_tuning(self._h, int(boards_per_lane), int(block_threads)))

    def set_options(self, auto_reset=None, assign=None, reward=None):
        """Change the options.  A new `assign` mode takes effect with the next full reset(), which the library insists
        on before it advances boards again (running boards find their piece lists through the assignment)."""
        if auto_reset is not None:
            self.auto_reset = bool(auto_reset)
        if assign is not None:
            self.assign = assign
        if reward is not None:
            self.reward_params = tuple(float(x) for x in reward)
        self._apply_options()

    def _dev(self, x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)

    def _int_arg(self, x):
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.ascontiguousarray(x))
        if x.dtype not in _INT_CODES:
            x = x.to(torch.int64)
        x = x.to(self.device).contiguous()
        if x.numel() != self.num_envs:
            raise ValueError(f"expected {self.num_envs} values, got {x.numel()}")
        return x, _INT_CODES[x.dtype]

    # ------------------------------------------------------------------------------------------ configs
    def load_configs(self, rows, pieces, validate: bool = True) -> None:
        """Upload a pool of prescribed (board, pieces) configurations (the supply behind reset()).

        rows: [n_cfg, 20] uint16 (bit x = column x) or [n_cfg, 20, 10] bool; pieces: [n_cfg, M+1] uint8.
        On a live environment this REPLACES the supply without disturbing running boards: the pool goes into the
        handle's other buffer and becomes current; boards that are mid-episode finish on the one they started from.
        Raises (TPL_ERR_STATE) if that other buffer may still be in use -- see pool_info().  The work is enqueued on
        the current stream (PoolRefresher shows the side-stream form).  validate=False skips the piece-id check, which
        reads device data back and so makes the host wait for the stream: for batches the caller vouches for (the
        device generator cannot emit an invalid id)."""
        if not isinstance(rows, torch.Tensor):
            rows = np.asarray(rows)
            if rows.ndim == 3:
                rows = (rows.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(-1).astype(np.uint16)
            rows = torch.from_numpy(np.ascontiguousarray(rows.astype(np.uint16)).view(np.int16))
        elif rows.dim() == 3:
            rows = (rows.to(torch.int32) << torch.arange(10, device=rows.device, dtype=torch.int32)).sum(-1).to(torch.int16)
        rows = rows.to(self.device).contiguous()
        if rows.dtype not in (torch.int16, torch.uint16):
            rows = rows.to(torch.int16)
        pieces = self._dev(pieces, torch.uint8)
        n_cfg = rows.shape[0]
        if rows.shape != (n_cfg, 20) or pieces.shape != (n_cfg, self.M + 1):
            raise ValueError(f"rows must be [n,20] and pieces [n,{self.M + 1}]; got {tuple(rows.shape)} {tuple(pieces.shape)}")
        if n_cfg == 0:
            raise ValueError("the pool needs at least one configuration")
        if validate and int(pieces.max()) > 6:          # tetrominos[piece] of the reference raises IndexError here
            raise ValueError("piece ids must be in 0..6 (I L J T S Z O)")
        nbytes = self._lib.tpl_pool_bytes(n_cfg, self.M)
        pool_mem = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_load_configs(self._h, _ptr(rows), _ptr(pieces), n_cfg, _ptr(pool_mem), nbytes, self._stream()))
        # the handle now reads this buffer as its current slot; what that slot held before is out of use (the library
        # refused the call otherwise) but launches that read it may still be enqueued on this stream
        slot = self.pool_info()["current_slot"]
        old = self._pool_mems[slot]
        self._pool_mems[slot], self._pool, self.n_configs = pool_mem, (rows, pieces), n_cfg
        self.pool_generation += 1
        if old is not None:
            old.record_stream(torch.cuda.current_stream(self.device))

    def pool_info(self) -> dict:
        """Which of the handle's two pool buffers is current, their sizes, and how many more steps must be enqueued
        before load_configs() may replace the other one (0: it may)."""
        slot, cur, other, wait = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.tpl_pool_info(self._h, C.byref(slot), C.byref(cur), C.byref(other), C.byref(wait)))
        return dict(current_slot=slot.value, n_configs=cur.value, n_configs_other=other.value, steps_until_swap=wait.value)

    def _clocks(self) -> torch.Tensor:
        """The step clocks (one int64 per group of 32 boards) as a view of the workspace."""
        ptr, count = C.c_void_p(), C.c_int64()
        check(self._lib.tpl_clock_ptr(self._h, C.byref(ptr), C.byref(count)))
        off = ptr.value - self._workspace.data_ptr()
        return self._workspace[off: off + 8 * count.value].view(torch.int64)

    def note_steps(self, steps: int) -> None:
        """Tell the pool-swap guard that `steps` steps were enqueued by replaying a captured HIP graph (it counts the
        steps that pass through the C API, which a replay does not)."""
        check(self._lib.tpl_note_steps(self._h, int(steps)))

    def step_clock(self) -> int:
        """Steps since the last full reset(), as the device counts them (host sync)."""
        clocks = self._clocks()
        first = int(clocks[0])
        if not bool((clocks == first).all()):
            raise _lib.TplError("the step clocks of the board groups disagree")
        return first

    def carved_configs(self, count: int, seed: Optional[int] = None, first: int = 0, with_solutions: bool = False, *,
                       cutoff: int = 0, waves: int = 0, return_status: bool = False):
        """Carved (solvable) configurations generated ON THE DEVICE (a persistent kernel whose lanes take configurations
        from a queue; `waves` = how many 64-lane waves share it, 0 = automatic) -- the same configurations
        generate_configs(L, M, count, seed, first) builds on the host.  Returns device tensors (rows int16 [count, 20],
        pieces uint8 [count, M+1]) and, with_solutions, (solution uint8 [count, M, 2], solution_len int32 [count]).
        `cutoff` (keyword only) overrides the restart rule's base cut-off (0 = by L: csrc/tpl_device.h `carve_cutoff`); raises
        if every attempt of some configuration ran into its cut-off -- unless `return_status`, which appends the int32 status
        tensor (0 finished, 1 capped: all-zero outputs) and leaves the decision to the caller."""
        seed = self.seed if seed is None else seed
        d = self.device
        rows = torch.empty((count, 20), dtype=torch.int16, device=d)
        pieces = torch.empty((count, self.M + 1), dtype=torch.uint8, device=d)
        sol = torch.zeros((count, self.M, 2), dtype=torch.uint8, device=d) if with_solutions else None
        sol_len = torch.zeros(count, dtype=torch.int32, device=d) if with_solutions else None
        status = torch.empty(count, dtype=torch.int32, device=d)
        nbytes = self._lib.tpl_generate_configs_device_work_bytes(self.M, count)
        work = torch.empty(nbytes, dtype=torch.uint8, device=d)
        check(self._lib.tpl_generate_configs_device_waves(self.L, self.M, seed, first, count, int(cutoff), int(waves), _ptr(rows),
                                                          _ptr(pieces), _ptr(sol), _ptr(sol_len), _ptr(status), _ptr(work), nbytes,
                                                          self._stream()))
        out = (rows, pieces, sol, sol_len) if with_solutions else (rows, pieces)
        if return_status:
            return out + (status,)
        if bool(status.any()):
            raise _lib.TplError(f"{int(status.sum())} of {count} configuration(s) did not finish: every attempt of the restart rule ran "
                                f"into its cut-off (L={self.L} M={self.M} cutoff={int(cutoff) or 'by L'}; a larger `cutoff` searches on)")
        return out

    def forward_configs(self, seeds, initial_height_max: int = 4, max_attempts: int = 1000):
        """The reference's forward generator + solver (game/tetris_algo_main/) ON THE DEVICE, one game per lane, for the
        given integer seeds: the same games, seed for seed, as `forward_generate()` builds on the host and as the reference's
        TetrisGameGenerator(seed, goal=L, tetrominoes=M, initial_height_max) + TetrisSolver(...).solve().  Returns a dict of
        device tensors over ALL seeds: rows int16 [n, 20], sequence uint8 [n, M] (piece ids of Tetris.move), winnable bool
        [n], failed_attempts int32 [n], solution uint8 [n, M, 2] (rotations, location), solver_stack uint8 [n, M, 3],
        solution_len int32 [n]."""
        seeds = torch.as_tensor(np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64)).view(np.int64)).to(self.device)
        n, d, M = int(seeds.numel()), self.device, self.M
        out = dict(rows=torch.empty((n, 20), dtype=torch.int16, device=d), sequence=torch.empty((n, M), dtype=torch.uint8, device=d),
                   winnable=torch.empty(n, dtype=torch.uint8, device=d), failed_attempts=torch.empty(n, dtype=torch.int32, device=d),
                   solution=torch.empty((n, M, 2), dtype=torch.uint8, device=d),
                   solver_stack=torch.empty((n, M, 3), dtype=torch.uint8, device=d),
                   solution_len=torch.empty(n, dtype=torch.int32, device=d))
        nbytes = self._lib.tpl_forward_generate_device_work_bytes(M, n)
        work = torch.empty(nbytes, dtype=torch.uint8, device=d)
        check(self._lib.tpl_forward_generate_device(self.L, M, int(initial_height_max), int(max_attempts), _ptr(seeds), n,
                                                    _ptr(out["rows"]), _ptr(out["sequence"]), _ptr(out["winnable"]),
                                                    _ptr(out["failed_attempts"]), _ptr(out["solution"]), _ptr(out["solver_stack"]),
                                                    _ptr(out["solution_len"]), _ptr(work), nbytes, self._stream()))
        work.record_stream(torch.cuda.current_stream(d))
        out["winnable"] = out["winnable"].view(torch.bool)
        return out

    def synthetic_configs(self, count: int, seed: Optional[int] = None, first: int = 0):
        """The synthetic boards / piece lists of SURVEY 8(d), generated on the device."""
        seed = self.seed if seed is None else seed
        rows = torch.empty((count, 20), dtype=torch.int16, device=self.device)
        pieces = torch.empty((count, self.M + 1), dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_synth_configs(self._h, seed, first, count, _ptr(rows), _ptr(pieces), self._stream()))
        return rows, pieces

    def synthetic_actions(self, step: int, seed: Optional[int] = None, out: Optional[torch.Tensor] = None):
        """Uniform synthetic actions for lockstep step `step`, keyed by global board index."""
        seed = self.seed if seed is None else seed
        if out is None:
            out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_synth_actions(self._h, seed, self.global_offset, self.num_envs, step, _ptr(out), self._stream()))
        return out

    # ------------------------------------------------------------------------------------------ reference surface
    def reset(self, mask=None) -> None:
        """Tetris.reset() (game/tetris.py:438-443) for all boards, or for the boards where mask is true."""
        m = None if mask is None else self._dev(mask, torch.uint8)
        check(self._lib.tpl_reset(self._h, _ptr(m), self._stream()))

    def move(self, rotations, locations) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Tetris.move(rotations, location) (game/tetris.py:354) on every board.

        Returns (reward f32[N], done bool[N], rows_cleared u8[N])."""
        rot, code = self._int_arg(rotations)
        loc, code2 = self._int_arg(locations)
        if code2 != code:
            rot, loc = rot.to(torch.int64), loc.to(torch.int64)
            code = _lib.TPL_I64
        reward = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
        done = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        cleared = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_move(self._h, _ptr(rot), _ptr(loc), code, _ptr(reward), _ptr(done), _ptr(cleared), self._stream()))
        return reward, done.view(torch.bool), cleared

    def _own(self, t: torch.Tensor, dtype, what: str, shape=None) -> torch.Tensor:
        """A caller-owned buffer the library will read or write through a raw pointer: it must be exactly what the
        kernel assumes (an out-of-range access on the device can take the whole GPU down)."""
        shape = (self.num_envs,) if shape is None else shape
        ok = dtype if isinstance(dtype, (tuple, list, dict)) else (dtype,)
        if (not isinstance(t, torch.Tensor) or t.dtype not in ok or tuple(t.shape) != tuple(shape) or t.device != self.device
                or not t.is_contiguous()):
            raise ValueError(f"{what} must be a contiguous {'/'.join(str(d) for d in ok)} tensor of shape {tuple(shape)} on {self.device}")
        return t

    def step_into(self, action: torch.Tensor, reward: torch.Tensor, done: torch.Tensor) -> None:
        """step() writing into caller-owned buffers (uint8/int32/int64 action, f32 reward, uint8 done)."""
        code, dev, shape = _INT_CODES.get(action.dtype), self.device, self._shape1
        # the hot call of a host-driven loop: the same checks as _own(), written flat (about a microsecond for all three)
        if not (code is not None and reward.dtype is torch.float32 and done.dtype is torch.uint8
                and action.shape == shape and reward.shape == shape and done.shape == shape
                and action.device == dev and reward.device == dev and done.device == dev
                and action.is_contiguous() and reward.is_contiguous() and done.is_contiguous()):
            self._own(action, _INT_CODES, "action")
            self._own(reward, torch.float32, "reward")
            self._own(done, torch.uint8, "done")
        rc = self._lib.tpl_step(self._h, action.data_ptr(), code, reward.data_ptr(), done.data_ptr(), self._stream())
        if rc:
            check(rc)

    def capture_steps(self, actions: torch.Tensor, rewards: torch.Tensor, dones: torch.Tensor):
        """K consecutive step() calls captured into ONE HIP graph: actions uint8/int32/int64 [K, N], rewards f32 [K, N],
        dones uint8 [K, N], all caller-owned and reused by every replay.  Returns a callable: each call replays the K
        steps (the boards move on; refill the action rows in between).  For batches small enough that the host's few
        microseconds per step_into() call are the limit (65,536 boards: 6.3 -> 5.0 us per step); at 2^20 boards it
        changes nothing.  The graph carries the pool pointers of the moment of capture: the callable captures again by
        itself after a load_configs().  `actions` may also be a sequence of K row tensors (rows that are not neighbours in
        memory); `replay.prepare()` captures ahead of the first replay (a capture costs a snapshot, a restore and K enqueues:
        not something to leave inside a timed region)."""
        K = len(actions)
        for t in range(K):
            self._own(actions[t], _INT_CODES, "actions[t]")
            self._own(rewards[t], torch.float32, "rewards[t]")
            self._own(dones[t], torch.uint8, "dones[t]")
        state = {"graph": None, "pool": -1}

        def capture():
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                saved = self.snapshot()
                self.step_into(actions[0], rewards[0], dones[0])          # load the kernel outside the capture
                self.restore(saved)
                torch.cuda.synchronize(self.device)
                with torch.cuda.graph(graph, stream=side):
                    for t in range(K):
                        self.step_into(actions[t], rewards[t], dones[t])
            torch.cuda.current_stream(self.device).wait_stream(side)
            self.restore(saved)                                           # capture does not execute; the swap guard saw K + 1 calls
            state["graph"], state["pool"] = graph, self.pool_generation

        def replay():
            if state["pool"] != self.pool_generation:
                capture()
            state["graph"].replay()
            # the library counts steps as they pass through its API (the pool-swap guard); a replay does not: tell it
            check(self._lib.tpl_note_steps(self._h, K))

        def prepare():
            if state["pool"] != self.pool_generation:
                capture()

        replay.prepare = prepare
        replay.steps = K
        return replay

    def step(self, action, observe: bool = True, obs_dtype=torch.float32):
        """step(action) with action = rot*10 + loc.  Returns (obs [N,217] or None, reward, done, info).  With the
        observation it is ONE kernel launch (tpl_step_observe): the move and the [N,217] rows of the boards as they stand
        after it."""
        act, code = self._int_arg(action)
        reward = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
        done = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        obs = None
        if observe:
            obs = torch.empty((self.num_envs, OBS_DIM), dtype=obs_dtype, device=self.device)
            check(self._lib.tpl_step_observe(self._h, _ptr(act), code, _ptr(reward), _ptr(done), _ptr(obs), _OBS_CODES[obs_dtype],
                                             self._stream()))
        else:
            check(self._lib.tpl_step(self._h, _ptr(act), code, _ptr(reward), _ptr(done), self._stream()))
        return obs, reward, done.view(torch.bool), {}

    def step_observe_into(self, action: torch.Tensor, reward: torch.Tensor, done: torch.Tensor, obs: torch.Tensor) -> None:
        """step() with the observation, writing into caller-owned buffers (as step_into; obs f32 / bf16 [N, 217])."""
        self._own(action, _INT_CODES, "action")
        self._own(reward, torch.float32, "reward")
        self._own(done, torch.uint8, "done")
        self._own(obs, _OBS_CODES, "obs", (self.num_envs, OBS_DIM))
        check(self._lib.tpl_step_observe(self._h, action.data_ptr(), _INT_CODES[action.dtype], reward.data_ptr(), done.data_ptr(),
                                         obs.data_ptr(), _OBS_CODES[obs.dtype], self._stream()))

    def rollout(self, actions: torch.Tensor, per_step: bool = False):
        """K consecutive steps in one kernel launch: actions uint8 [K, N] on the device.  Equivalent to K calls
        of step(); the board stays in registers between moves.  Returns (reward_sum f32[N], finished int32[N])
        and, if per_step, also (reward f32[K,N], done bool[K,N])."""
        if actions.dtype != torch.uint8 or actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise ValueError(f"actions must be uint8 [K, {self.num_envs}]")
        actions = actions.to(self.device)
        if actions.stride(1) != 1:
            actions = actions.contiguous()
        K = actions.shape[0]
        rsum = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
        fin = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        rs = ds = None
        if per_step:
            rs = torch.empty((K, self.num_envs), dtype=torch.float32, device=self.device)
            ds = torch.empty((K, self.num_envs), dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_rollout(self._h, _ptr(actions), actions.stride(0), K, _ptr(rs), _ptr(ds), _ptr(rsum), _ptr(fin),
                                    self._stream()))
        return (rsum, fin, rs, ds.view(torch.bool)) if per_step else (rsum, fin)

    def rollout_trajectory(self, actions: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rollout() recording the COMPACT TRAJECTORY: one byte per board-step -- rows cleared (bits 0-2), how the move ended
        (bits 3-4: 0 goes on, 1 won, 2 lost at the move limit, 3 topped out), re-initialised from the pool (bit 5), frozen
        (bit 6) -- as int32 words [ceil(K / 4), N], byte j of word w = step 4 w + j.  decode_trajectory() gives the
        (reward, done) of rollout(per_step=True), bit for bit."""
        if actions.dtype != torch.uint8 or actions.dim() != 2 or actions.shape[1] != self.num_envs or actions.stride(1) != 1 \
                or actions.device != self.device:
            raise ValueError(f"actions must be uint8 [K, {self.num_envs}] on {self.device} with unit inner stride")
        K = actions.shape[0]
        if out is None:
            out = torch.empty(((K + 3) // 4, self.num_envs), dtype=torch.int32, device=self.device)
        self._own(out, torch.int32, "out", ((K + 3) // 4, self.num_envs))
        check(self._lib.tpl_rollout_trajectory(self._h, _ptr(actions), actions.stride(0), K, _ptr(out), None, self._stream()))
        return out

    def rollout_random_trajectory(self, steps: int, seed: int = 0, step0: int = 0, out: Optional[torch.Tensor] = None,
                                  actions_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rollout_random() recording the compact trajectory (and, into `actions_out` uint8 [steps, N], what was played)."""
        if steps < 1:
            raise ValueError("steps must be positive")
        if out is None:
            out = torch.empty(((steps + 3) // 4, self.num_envs), dtype=torch.int32, device=self.device)
        self._own(out, torch.int32, "out", ((steps + 3) // 4, self.num_envs))
        if actions_out is not None:
            self._own(actions_out, torch.uint8, "actions_out", (steps, self.num_envs))
        check(self._lib.tpl_rollout_random_trajectory(self._h, int(seed), int(step0), int(steps), _ptr(actions_out), _ptr(out), None,
                                                      self._stream()))
        return out

    def decode_trajectory(self, trajectory: torch.Tensor, steps: int):
        """(reward f32 [steps, N], done bool [steps, N]) of a compact trajectory, under this environment's reward parameters."""
        self._own(trajectory, torch.int32, "trajectory", ((steps + 3) // 4, self.num_envs))
        rs = torch.empty((steps, self.num_envs), dtype=torch.float32, device=self.device)
        ds = torch.empty((steps, self.num_envs), dtype=torch.uint8, device=self.device)
        check(self._lib.tpl_decode_trajectory(self._h, _ptr(trajectory), int(steps), _ptr(rs), _ptr(ds), self._stream()))
        return rs, ds.view(torch.bool)

    def rollout_into(self, actions: torch.Tensor, K: int) -> None:
        """rollout() without outputs (statistics only): the throughput form used by bench.py."""
        if (actions.dtype != torch.uint8 or actions.dim() != 2 or actions.shape[1] != self.num_envs or actions.stride(1) != 1
                or actions.device != self.device or not 1 <= K <= actions.shape[0]):
            raise ValueError(f"actions must be uint8 [>= K, {self.num_envs}] on {self.device} with unit inner stride")
        check(self._lib.tpl_rollout(self._h, _ptr(actions), actions.stride(0), K, None, None, None, None, self._stream()))

    def rollout_random(self, steps: int, seed: int = 0, step0: int = 0, record: bool = False):
        """`steps` consecutive steps in one launch under the UNIFORM RANDOM POLICY drawn on the device (the loop of
        game/performance_test.py:13-17): step k plays the action explore_actions(epsilon=1, seed, step0 + k) would draw.
        No actions are staged.  Returns (reward_sum f32[N], finished int32[N]) and, if record, also
        (actions uint8[K,N], reward f32[K,N], done bool[K,N])."""
        if steps < 1:
            raise ValueError("steps must be positive")
        n, d = self.num_envs, self.device
        rsum = torch.empty(n, dtype=torch.float32, device=d)
        fin = torch.empty(n, dtype=torch.int32, device=d)
        acts = rs = ds = None
        if record:
            acts = torch.empty((steps, n), dtype=torch.uint8, device=d)
            rs = torch.empty((steps, n), dtype=torch.float32, device=d)
            ds = torch.empty((steps, n), dtype=torch.uint8, device=d)
        check(self._lib.tpl_rollout_random(self._h, int(seed), int(step0), int(steps), _ptr(acts), _ptr(rs), _ptr(ds), _ptr(rsum),
                                           _ptr(fin), self._stream()))
        return (rsum, fin, acts, rs, ds.view(torch.bool)) if record else (rsum, fin)

    def observe(self, dtype=torch.float32, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[N, 217] observation for Model(217, 14) (model/train.py:26)."""
        if out is None:
            out = torch.empty((self.num_envs, OBS_DIM), dtype=dtype, device=self.device)
        self._own(out, _OBS_CODES, "out", (self.num_envs, OBS_DIM))
        check(self._lib.tpl_expand_obs(self._h, _ptr(out), _OBS_CODES[out.dtype], self._stream()))
        return out

    def decode_actions(self, logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[N, 14] policy outputs (f32 or bf16) -> uint8 actions: argmax of 4 rotation logits and 10 location logits."""
        self._own(logits, _OBS_CODES, "logits", (self.num_envs, 14))
        if out is None:
            out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        self._own(out, torch.uint8, "out")
        check(self._lib.tpl_decode_actions(self._h, _ptr(logits), _OBS_CODES[logits.dtype], _ptr(out), self._stream()))
        return out

    def policy_act(self, image: torch.Tensor, out: Optional[torch.Tensor] = None, logits: Optional[torch.Tensor] = None):
        """Fused observation -> Model(217, 14) -> action on the matrix cores.  `image` is the device copy of
        pack_policy(...): the bf16 image (weights and hidden activations rounded to bf16) or the float32 one
        (pack_policy(..., f32=True): the reference's arithmetic width, about 1/8 of the rate) -- told apart by size.
        Returns the uint8 actions; fills `logits` ([N, 14] float32) when given."""
        if out is None:
            out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        self._own(out, torch.uint8, "out")
        kind = self._image(image, either=True, split=True)
        if logits is not None:
            self._own(logits, torch.float32, "logits", (self.num_envs, 14))
        act = {"split": self._lib.tpl_policy_act_split, True: self._lib.tpl_policy_act_f32, False: self._lib.tpl_policy_act}[kind]
        check(act(self._h, _ptr(image), _ptr(out), _ptr(logits), self._stream()))
        return out

    def _image(self, image: torch.Tensor, either: bool = False, split: bool = False):
        """Checks a policy image; returns False for the bf16 kind, True for the float32 kind (accepted only where `either`),
        "split" for the three-piece kind (only where `split`)."""
        if split and isinstance(image, torch.Tensor) and image.numel() == self._lib.tpl_policy_image_bytes_split():
            self._own(image, torch.uint8, "image (the device copy of pack_policy(..., f32='split'))", (image.numel(),))
            return "split"
        if either and isinstance(image, torch.Tensor) and image.numel() == self._lib.tpl_policy_image_bytes_f32():
            self._own(image, torch.uint8, "image (the device copy of pack_policy(..., f32=True))", (image.numel(),))
            return True
        self._own(image, torch.uint8, "image (the device copy of pack_policy(...))", (self._lib.tpl_policy_image_bytes(),))
        return False

    def explore_actions(self, action: torch.Tensor, epsilon: float, seed: int = 0, step: int = 0) -> torch.Tensor:
        """Epsilon-greedy in place: action[i] becomes uniform in [0, 40) with probability epsilon."""
        self._own(action, torch.uint8, "action")
        check(self._lib.tpl_explore_actions(self._h, _ptr(action), float(epsilon), int(seed), int(step), self._stream()))
        return action

    def actor_rollout(self, image: torch.Tensor, steps: int, epsilon: float = 0.0, seed: int = 0, step0: int = 0,
                      record: bool = True, record_states: bool = False) -> dict:
        """`steps` iterations of policy -> epsilon-greedy -> step in one kernel launch (weights in LDS, boards in
        registers).  `image`: the bf16 image, the float32 one (pack_policy(..., f32=True): the reference's arithmetic
        width, the weights stream through LDS once per step) or the split one (f32="split") -- told apart by size.  Returns the trajectory:
        actions/rewards/dones [steps, N] and, if asked, the 32-byte state of every board before each step as two int32
        [steps, N, 4] tensors."""
        n, d = self.num_envs, self.device
        kind = self._image(image, either=True, split=True)
        if steps < 1:
            raise ValueError("steps must be positive")
        out = {}
        if record:
            out["actions"] = torch.empty((steps, n), dtype=torch.uint8, device=d)
            out["rewards"] = torch.empty((steps, n), dtype=torch.float32, device=d)
            out["dones"] = torch.empty((steps, n), dtype=torch.uint8, device=d)
        if record_states:
            out["states_a"] = torch.empty((steps, n, 4), dtype=torch.int32, device=d)
            out["states_b"] = torch.empty((steps, n, 4), dtype=torch.int32, device=d)
        launch = {"split": self._lib.tpl_actor_rollout_split, True: self._lib.tpl_actor_rollout_f32, False: self._lib.tpl_actor_rollout}[kind]
        check(launch(self._h, _ptr(image), int(steps), float(epsilon), int(seed), int(step0),
                     _ptr(out.get("actions")), _ptr(out.get("rewards")), _ptr(out.get("dones")),
                     _ptr(out.get("states_a")), _ptr(out.get("states_b")), self._stream()))
        if record:
            out["dones"] = out["dones"].view(torch.bool)
        return out

    def raw_planes(self):
        """Copies of the two resident state planes, int32 [N, 4] each (layout in DESIGN.md section 2)."""
        n = self.num_envs
        pa, pb = C.c_void_p(), C.c_void_p()
        check(self._lib.tpl_state_ptrs(self._h, C.byref(pa), C.byref(pb)))
        base = self._workspace.data_ptr()
        off_a, off_b = pa.value - base, pb.value - base
        a = self._workspace[off_a: off_a + n * 16].view(torch.int32).view(n, 4).clone()
        b = self._workspace[off_b: off_b + n * 16].view(torch.int32).view(n, 4).clone()
        return a, b

    def write_raw_planes(self, a: torch.Tensor, b: torch.Tensor) -> None:
        """Put two int32 [N, 4] planes (as raw_planes() returns them) into the resident state: mid-game positions that
        the public surface cannot create -- e.g. lines_cleared / moves_used of a game in progress (the reference lets a
        caller assign them, game/tetris.py:149-150).  The caller answers for their consistency."""
        n = self.num_envs
        pa, pb = C.c_void_p(), C.c_void_p()
        check(self._lib.tpl_state_ptrs(self._h, C.byref(pa), C.byref(pb)))
        base = self._workspace.data_ptr()
        for ptr, src in ((pa, a), (pb, b)):
            src = src.to(device=self.device, dtype=torch.int32).contiguous()
            if tuple(src.shape) != (n, 4):
                raise ValueError(f"a plane must be int32 [{n}, 4]")
            off = ptr.value - base
            self._workspace[off: off + n * 16].view(torch.int32).view(n, 4).copy_(src)

    def expand_states(self, states_a: torch.Tensor, states_b: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """[K, 217] observations of K recorded 32-byte states (int32 [K, 4] pairs as actor_rollout(record_states=True)
        returns them, e.g. a replay-buffer minibatch)."""
        a = states_a.to(self.device).reshape(-1, 4).contiguous()
        b = states_b.to(self.device).reshape(-1, 4).contiguous()
        if a.shape != b.shape or a.dtype != torch.int32 or b.dtype != torch.int32:
            raise ValueError("states_a / states_b must be int32 [..., 4] tensors of equal shape")
        out = torch.empty((a.shape[0], OBS_DIM), dtype=dtype, device=self.device)
        check(self._lib.tpl_expand_states(self._h, _ptr(a), _ptr(b), a.shape[0], _ptr(out), _OBS_CODES[dtype], self._stream()))
        return out

    def packed_state(self) -> dict:
        """Everything get_state() and the public attributes expose, in the interchange layout (device tensors)."""
        n, d = self.num_envs, self.device
        out = dict(rows=torch.empty((n, 20), dtype=torch.int16, device=d))
        for k in ("cur", "nxt", "lines", "moves", "state", "pieces_left"):
            out[k] = torch.empty(n, dtype=torch.uint8, device=d)
        check(self._lib.tpl_get_state(self._h, _ptr(out["rows"]), _ptr(out["cur"]), _ptr(out["nxt"]), _ptr(out["lines"]),
                                      _ptr(out["moves"]), _ptr(out["state"]), _ptr(out["pieces_left"]), self._stream()))
        return out

    def get_state(self):
        """Batched Tetris.get_state() (game/tetris.py:435-436):
        (board bool[N,20,10], pieces[0], pieces[1], L - lines_cleared, M - moves_used, state)."""
        n, d = self.num_envs, self.device
        cells = torch.empty((n, 20, 10), dtype=torch.uint8, device=d)
        check(self._lib.tpl_get_board(self._h, _ptr(cells), self._stream()))
        s = {k: torch.empty(n, dtype=torch.uint8, device=d) for k in ("cur", "nxt", "lines", "moves", "state")}
        check(self._lib.tpl_get_state(self._h, None, _ptr(s["cur"]), _ptr(s["nxt"]), _ptr(s["lines"]), _ptr(s["moves"]),
                                      _ptr(s["state"]), None, self._stream()))
        return (cells.view(torch.bool), s["cur"], s["nxt"], self.L - s["lines"].to(torch.int32),
                self.M - s["moves"].to(torch.int32), s["state"])

    _FIELDS = ("rows", "cur", "nxt", "lines", "moves", "state", "pieces_left")

    def _field(self, name: str) -> torch.Tensor:
        """One array of packed_state() (the kernel skips the outputs it is not asked for)."""
        out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        ptrs = [_ptr(out) if k == name else None for k in self._FIELDS]
        check(self._lib.tpl_get_state(self._h, *ptrs, self._stream()))
        return out

    lines_cleared = property(lambda self: self._field("lines"))
    moves_used = property(lambda self: self._field("moves"))
    state = property(lambda self: self._field("state"))

    def snapshot(self) -> Snapshot:
        """Copy of the whole resident state (boards, counters, piece windows, step clocks, statistics).  The boards
        refer to the pool that is loaded now: the Snapshot remembers which."""
        return Snapshot(self._workspace.clone(), self.pool_generation, self.assign, self.pool_info()["steps_until_swap"])

    def restore(self, saved: Snapshot) -> None:
        """Put a snapshot() back.  Refused if the pool or the assignment mode has changed since it was taken: the
        running boards of the snapshot would continue with the piece lists of other configurations.  A bare tensor is
        refused too: it cannot say what it was taken under."""
        if not isinstance(saved, Snapshot):
            raise TypeError("restore() takes the Snapshot that snapshot() returned (a bare tensor does not say which "
                            "configuration pool its boards refer to)")
        if (saved.pool_generation, saved.assign) != (self.pool_generation, self.assign):
            raise _lib.TplError("this snapshot was taken under another configuration pool or assignment mode")
        if saved.data.shape != self._workspace.shape:
            raise _lib.TplError("this snapshot belongs to an environment of another size")
        self._workspace.copy_(saved.data)
        # the boards are as old again as they were then: so is the guard on the other pool buffer
        check(self._lib.tpl_pool_set_hold(self._h, saved.hold))

    def stats(self) -> dict:
        """Episodes finished since the last full reset: counts for the episodic-return mean (host sync)."""
        out = torch.empty(4, dtype=torch.int64, device=self.device)
        check(self._lib.tpl_get_stats(self._h, _ptr(out), self._stream()))
        return dict(zip(("episodes", "lines", "wins", "topouts"), out.tolist()))

    def stats_tensor(self) -> torch.Tensor:
        """The four counters as a device int64 tensor (no host sync) -- what the RCCL all-reduce sums."""
        out = torch.empty(4, dtype=torch.int64, device=self.device)
        check(self._lib.tpl_get_stats(self._h, _ptr(out), self._stream()))
        return out

    def return_sum(self, stats: torch.Tensor) -> torch.Tensor:
        """[sum of episodic returns, episodes] (f64) from a stats tensor, under the current reward parameters."""
        from .sharding import return_sum
        return return_sum(stats, self.reward_params)

    def terminate(self) -> None:
        """Tetris.terminate() (game/tetris.py:451-470): releases the handle."""
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            check(self._lib.tpl_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.terminate()
        except Exception:
            pass


class Tetris:
    """One board with the reference's own constructor, surface and attribute types (game/tetris.py:140-214,
    354-449): `Tetris(L, M, warm_reset=True, render=False, framerate=30, debug=False)`.

    `board` is a 20x10 numpy bool array, `pieces` a Python list, `state` None / True / False, and with debug=True
    `solution` is the list of (rotations, location) that wins the current configuration (:155-156).  Like the
    reference's constructor it carves its own prescribed configurations -- with the native generator, a pool of
    `pool_size` at a time instead of two worker processes; `configs=(rows, pieces)` supplies them instead.
    reset() moves on to another configuration of the pool (entry `steps so far` mod pool size: the sequential
    assignment of the batched environment, a step being a move() or a reset() that follows no move -- so that, as with
    the reference's queue (:445-447), resetting twice in a row gives two different games).  render=True (the pygame
    window, :158-182) is out of scope and raises.

    One divergence from the reference BY DEFAULT: a FINISHED game is frozen.  The reference's move() keeps popping pieces
    and changing board / moves_used / lines_cleared after `state` has been set (game/tetris.py:354-422 never looks at it), and
    its reset() leaves lines_cleared / moves_used / state as they were (:438-449); here move() on a game whose state is not
    None raises RuntimeError, so that a ported caller fails loudly instead of reading a board that no longer changes, and
    reset() starts the counters from zero.

    `reference_quirks=True` gives the reference's behaviour instead, both parts (pinned by tests/golden/afterlife.npz, recorded
    from the reference): a finished game goes on -- `state` is overwritten only where the reference assigns it, so a won game
    can turn lost and a lost one won; move() raises IndexError once the M + 1 pieces are used up (`pieces.pop(0)`, :356) -- and
    the three scalars survive reset().  The board mechanics of every move (clamp, drop, top-out, lock, full-row test,
    compaction) run on the device as ever; what the device cannot hold is a finished game's bookkeeping -- its packed state
    derives the piece cursor from moves_used and stops at M -- so in this mode lines_cleared / moves_used / state live in the
    object, every move is handed to the device as the first move of a game whose next two pieces are the current ones, and
    the reference's six lines of terminal tests (:372-374, 389-394, 415-422) are applied to what came back."""

    _STATE = {RUNNING: None, WON: True, LOST: False}

    def __init__(self, L: int, M: int, warm_reset: bool = True, render: bool = False, framerate: int = 30,
                 debug: bool = False, configs=None, device="cuda:0", seed: Optional[int] = None, pool_size: int = 64,
                 reference_quirks: bool = False):
        if render:
            raise NotImplementedError("the pygame window of the reference is not part of this build")
        del framerate
        self.L, self.M, self.debug = L, M, debug
        self._solutions = None
        if configs is None:
            seed = int.from_bytes(__import__("os").urandom(4), "little") if seed is None else seed
            rows, pieces, sol, sol_len = _lib.generate_configs(L, M, pool_size, seed=seed, with_solutions=True)
            self._solutions = [[(int(r), int(c)) for r, c in sol[k, : sol_len[k]]] for k in range(pool_size)]
            if warm_reset and not debug:
                # the reference's warm-reset queue has a SECOND producer (game/tetris.py:205-211, 482-488): the forward
                # generator + solver over seeds 0..99 (tetris_algo_main/main.py:39-40), whose winnable games enter the
                # same queue through translate() -- one random piece put in FRONT of the sequence (:19-20).  Same here:
                # the winnable ones join the pool.  (With debug=True the pool stays carved-only: only carved
                # configurations have a recorded `solution`.)
                fw = _lib.forward_generate(L, M, np.arange(100))
                win = fw["winnable"]
                if win.any():
                    lead = np.random.default_rng(seed).integers(0, 7, (int(win.sum()), 1)).astype(np.uint8)
                    rows = np.concatenate([rows, fw["rows"][win]])
                    pieces = np.concatenate([pieces, np.concatenate([lead, fw["sequence"][win]], axis=1)])
                    self._solutions = None
        else:
            rows, pieces = configs
        self._pieces_host = np.asarray(pieces, dtype=np.uint8).reshape(-1, M + 1)
        self._env = BatchedTetris(L, M, 1, device=device, assign="sequential", config_pool=(rows, self._pieces_host))
        self._steps = 0                    # step launches since construction = the device's step clock
        self._birth = 0                    # the step at which the current episode began
        self._finished = False
        self._board = None
        self._cache = None                 # packed_state() of the current position (one export per move, not per read)
        self._quirks = bool(reference_quirks)
        self._lines = self._moves = self._consumed = 0                     # reference_quirks: the scalars the object keeps
        self._state = None
        self._env.reset()

    def _config(self) -> int:
        return self._birth % self._pieces_host.shape[0]                    # sequential assignment, board index 0

    @property
    def solution(self) -> list:
        if not self.debug or self._solutions is None:
            raise AttributeError("solution is recorded only with debug=True and self-generated configurations")
        return self._solutions[self._config()]

    def _move_as_the_reference(self, rotations: int, location: int) -> None:
        """move() under reference_quirks: game/tetris.py:354-422 with `state` never read."""
        pieces = self._pieces_host[self._config()]
        if self._consumed >= len(pieces):
            raise IndexError("pop from empty list")                       # :356 self.pieces.pop(0)
        # the device plays this move as the FIRST move of a game: counters zero, running, window = [current, next, none...]
        cur = int(pieces[self._consumed])
        nxt = int(pieces[self._consumed + 1]) if self._consumed + 1 < len(pieces) else 7
        window = cur | (nxt << 3) | (((1 << 30) - 1) << 6)               # twelve 3-bit ids, 7 = none
        a, b = (x.cpu().numpy().view(np.uint32).copy() for x in self._env.raw_planes())
        a[0, 1] &= 0x0FFFFFFF                                              # moves_used, low nibble
        a[0, 3] &= 0x0FFFFFFF                                              #             high nibble
        b[0, 1] &= 0xCFFFFFFF                                              # state = running (the pool-slot bit stays)
        b[0, 2] = (b[0, 2] & 0x000FFFFF) | np.uint32(((window >> 32) & 15) << 28)          # lines_cleared = 0, window[35:32]
        b[0, 3] = np.uint32(window & 0xFFFFFFFF)
        self._env.write_raw_planes(torch.from_numpy(a.view(np.int32)), torch.from_numpy(b.view(np.int32)))
        _, _, cleared = self._env.move(torch.tensor([rotations], dtype=torch.int64), torch.tensor([location], dtype=torch.int64))
        self._steps += 1
        self._consumed += 1
        self._board = self._cache = None
        placed = int(self._s()["moves"][0]) == 1                          # a top-out consumes the piece and counts no move (:372-374)
        if not placed:
            self._state = False
            return
        self._moves += 1                                                  # :379
        rows_cleared = int(cleared.item())
        if rows_cleared == 0:                                             # :389-394
            if self._moves >= self.M:
                self._state = False
            return
        self._lines += rows_cleared                                       # :409
        if self._lines >= self.L:                                         # :415-417 (tested before the move limit)
            self._state = True
        elif self._moves >= self.M:                                       # :420-422
            self._state = False

    def move(self, rotations: int, location: int) -> None:
        if self._quirks:
            return self._move_as_the_reference(rotations, location)
        if self._finished:
            raise RuntimeError("move() on a finished game: this build freezes a game once state is set (the reference "
                               "keeps mutating it); call reset() first")
        _, done, _ = self._env.move(torch.tensor([rotations], dtype=torch.int64), torch.tensor([location], dtype=torch.int64))
        self._finished = bool(done.item())
        self._steps += 1
        self._board = self._cache = None

    def reset(self) -> None:
        if self._steps == self._birth:
            # no move since this game was dealt: let one step pass unplayed (the board is about to be replaced, so its
            # birth step need not survive), or the masked reset below would deal the same configuration again
            self._env._clocks().add_(1)
            self._steps += 1
        self._birth = self._steps
        self._env.reset(mask=[1])
        self._finished = False
        self._consumed = 0                 # (reference_quirks: lines / moves / state are NOT touched, as in :438-449)
        self._board = self._cache = None

    def _s(self):
        if self._cache is None:
            self._cache = {k: v.cpu().numpy() for k, v in self._env.packed_state().items()}
        return self._cache

    @property
    def board(self) -> np.ndarray:
        """The 20x10 bool array, as in the reference a LIVE object: the same array is returned until the next move()
        or reset(), carve() works on it, and writes to it are seen by carve() (not by move(): the board the moves
        are played on lives on the device)."""
        if self._board is None:
            rows = self._s()["rows"][0].view(np.uint16)
            self._board = ((rows[:, None] >> np.arange(10)) & 1).astype(bool)
        return self._board

    def carve(self, piece: int, rotations: int, location: int, allow_partial: bool) -> bool:
        """Tetris.carve (game/tetris.py:286-352): removes the piece from where move() would have put it, on `board`."""
        board = self.board
        rows = (board.astype(np.uint16) << np.arange(10, dtype=np.uint16)).sum(1).astype(np.uint16)
        ok, after = _lib.carve(rows, piece, rotations, location, allow_partial)
        if ok:
            board[:] = ((after[:, None] >> np.arange(10)) & 1).astype(bool)
        return ok

    @property
    def pieces(self) -> list:
        if self._quirks:
            return self._pieces_host[self._config()][self._consumed:].tolist()
        left = int(self._s()["pieces_left"][0])
        return self._pieces_host[self._config()][self.M + 1 - left:].tolist()

    lines_cleared = property(lambda self: self._lines if self._quirks else int(self._s()["lines"][0]))
    moves_used = property(lambda self: self._moves if self._quirks else int(self._s()["moves"][0]))
    state = property(lambda self: self._state if self._quirks else self._STATE[int(self._s()["state"][0])])

    def get_state(self):
        if self._quirks:
            pieces = self.pieces                                           # (IndexError with fewer than two left, as :436)
            return (self.board, pieces[0], pieces[1], self.L - self._lines, self.M - self._moves, self._state)
        s = self._s()
        return (self.board, int(s["cur"][0]), int(s["nxt"][0]), self.L - int(s["lines"][0]),
                self.M - int(s["moves"][0]), self._STATE[int(s["state"][0])])

    def terminate(self) -> None:
        _ = self.board                     # the reference's attributes stay readable after terminate()
        self._env.terminate()
