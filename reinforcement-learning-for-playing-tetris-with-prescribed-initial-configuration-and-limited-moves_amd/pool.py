"""On-disk format of a pool of prescribed configurations (the reference keeps them only in a process queue,
game/tetris.py:195,473-488 of the upstream repo).

One .npz file: L, M, rows uint16 [n, 20] (bit x = column x), pieces uint8 [n, M+1] (ids I0 L1 J2 T3 S4 Z5 O6) and,
optionally, the carved solution uint8 [n, M, 2] (rotations, location) with its length int32 [n] -- the triple
(board, pieces, solution) of the reference's debug mode (game/tetris.py:155-156, 259-260)."""
from __future__ import annotations

import numpy as np


# CU-masked / low-priority streams made through tpl_stream_create, one per (device, cu_count, low_priority), kept for the
# life of the process like torch's own stream pool: torch's allocators remember the streams a block was used on and record
# events on them when the block is freed (a pinned host buffer does, long after a refresher is gone), so a stream that
# torch has seen must never be destroyed under it.
_SIDE_STREAMS = {}


def side_stream(env, reserved_cus: int = 0, low_priority: bool = False):
    """A torch stream on `env`'s device whose work runs on `reserved_cus` compute units only (or at the lowest priority)."""
    import ctypes as C
    import torch
    from ._lib import check
    key = (env._index, int(reserved_cus), bool(low_priority))
    if key not in _SIDE_STREAMS:
        raw = C.c_void_p()
        check(env._lib.tpl_stream_create(env._index, int(reserved_cus), int(bool(low_priority)), C.byref(raw)))
        _SIDE_STREAMS[key] = torch.cuda.ExternalStream(raw.value, device=env.device)
    return _SIDE_STREAMS[key]


_CONCURRENT = {}          # (device index, stepping stream) -> a torch stream whose kernels were SEEN to run beside it
_UNTESTED = {}            # device index -> the one side stream handed out when the probe could not tell (never a new one per call)


def concurrent_stream(env):
    """A torch stream for the generator, tested to run CONCURRENTLY with the current (stepping) stream.

    HIP deals its streams out over a few hardware queues (four by default), the stepping stream's among them, and two
    streams on one queue do not overlap: a generator kernel submitted between two steps makes the next step wait until it
    has finished -- 30 ms for a batch of 65,536 configurations, which is what every fourth stream of torch's pool did to
    the step loop (profiles/r03_live_supply/hardware_queue_aliasing.log).  Nothing in the API says which queue a stream
    is on, so the candidates are tried: a generator launch of some ten milliseconds goes onto the candidate, a tiny launch
    onto the stepping stream behind it, and the candidate is good if the tiny launch is done while the generator is still
    running.  The first good one is kept for the life of the process.  The probe synchronises the device: it must not be
    called while the stepping stream is being captured into a graph (PoolRefresher does not)."""
    import ctypes as C
    import torch
    from ._lib import check
    d, main = env.device, torch.cuda.current_stream(env.device)
    key = (env._index, main.cuda_stream)                   # tested against THIS stepping stream
    if key in _CONCURRENT:
        return _CONCURRENT[key]
    # The probe launch does not depend on the environment: 64 configurations at L = 12, M = 40 on ONE wave -- searches of
    # 4,600 trips at the median, the slowest of 64 some four times that: ten to thirty milliseconds whatever the seed -- and its
    # output is discarded.
    L, M, count = 12, 40, 64
    rows = torch.empty((count, 20), dtype=torch.int16, device=d)
    pieces = torch.empty((count, M + 1), dtype=torch.uint8, device=d)
    status = torch.empty(count, dtype=torch.int32, device=d)
    nbytes = env._lib.tpl_generate_configs_device_work_bytes(M, count)
    work = torch.empty(nbytes, dtype=torch.uint8, device=d)
    scratch = torch.empty(4, dtype=torch.int64, device=d)
    torch.cuda.synchronize(d)
    chosen, conclusive = None, 0
    for attempt in range(12):
        cand = torch.cuda.Stream(d)
        busy, quick = torch.cuda.Event(), torch.cuda.Event()
        check(env._lib.tpl_generate_configs_device_waves(L, M, 0x5EED, attempt * count, count, 0, 1, C.c_void_p(rows.data_ptr()),
                                                         C.c_void_p(pieces.data_ptr()), None, None, C.c_void_p(status.data_ptr()),
                                                         C.c_void_p(work.data_ptr()), nbytes, cand.cuda_stream))
        busy.record(cand)
        pending = not busy.query()                # still running when the quick launch goes out?  else the trial says nothing
        check(env._lib.tpl_get_stats(env._h, C.c_void_p(scratch.data_ptr()), main.cuda_stream))
        quick.record(main)
        quick.synchronize()
        overlapped = not busy.query()             # the stepping stream's launch is done and the generator's is not
        cand.synchronize()
        if not pending:
            continue                              # inconclusive: the probe had ended before the comparison began
        conclusive += 1
        chosen = cand
        if overlapped:
            break
    else:
        import warnings
        if conclusive:
            warnings.warn("no stream was seen to run beside the stepping stream: the supply generator will serialise with the "
                          "steps (a stall of one generator launch per pool swap)")
        else:
            warnings.warn("the stream probe was inconclusive (its generator launch ended before it could be compared): "
                          "using an untested side stream")
            if env._index not in _UNTESTED:
                _UNTESTED[env._index] = torch.cuda.Stream(d)
            return _UNTESTED[env._index]          # not cached as tested: the next refresher tries again
    _CONCURRENT[key] = chosen
    return chosen


class ForwardGames:
    """The winnable games of the reference's forward generator + solver for a set of seeds (its own: 0..99,
    tetris_algo_main/main.py:39-40), produced on the device once and kept there, ready to be blended into pools of carved
    configurations -- the reference's queue has both producers (game/tetris.py:195-211, 482-488).

    `translate(batch)` gives them the way the reference's `translate()` does (game/tetris.py:19-20): ONE random piece put in
    FRONT of the sequence, so that `pieces` has M + 1 entries -- which also means the game no longer starts with the piece its
    solver planned for: a translated game is not winnable by construction, in the reference either.  `lead=False` pads at the
    END instead (the recorded solution still wins; the last piece is never played)."""

    def __init__(self, env, seeds=range(100), initial_height_max: int = 4, max_attempts: int = 1000):
        import torch
        out = env.forward_configs(list(seeds), initial_height_max, max_attempts)
        keep = out["winnable"]                                 # one host sync, at construction
        self.device, self.M = env.device, env.M
        self.tried = int(keep.numel())
        self.rows = out["rows"][keep].contiguous()
        self.sequence = out["sequence"][keep].contiguous()
        self.solution = out["solution"][keep].contiguous()
        self.solution_len = out["solution_len"][keep].contiguous()
        self.count = int(self.rows.shape[0])
        self._torch = torch
        # game indices for translate()'s draw, made HERE and waited for (construction has a host sync anyway): translate() runs on
        # whatever side stream a refresher uses, and must not depend on a tensor another stream may still be writing
        self._index = torch.arange(self.count, dtype=torch.int64, device=self.device)
        torch.cuda.synchronize(self.device)

    def translate(self, seed: int = 0, batch: int = 0, lead: bool = True):
        """(rows int16 [count, 20], pieces uint8 [count, M + 1]) on the device, enqueued on the current stream.  NO host wait:
        the extra piece of game g in batch b is drawn ON THE DEVICE, a splitmix64 hash of (seed, b, g) in torch's wrapping int64
        arithmetic (through round 5 it was drawn by numpy and copied up from pageable memory -- a copy that holds the calling
        thread until the stream has reached it, i.e. a stall of the step loop's host thread at every swap)."""
        torch = self._torch
        pieces = torch.cat([self._draw(seed, batch), self.sequence] if lead else [self.sequence, self._draw(seed, batch)], dim=1)
        return self.rows, pieces

    def _draw(self, seed: int, batch: int):
        """uint8 [count, 1] in 0..6, a function of (seed, batch, game index) alone (the same on any stream, any device)."""
        torch = self._torch
        if self._index is None:
            self._index = torch.arange(self.count, dtype=torch.int64, device=self.device)

        def signed(v):                                          # a 64-bit pattern as the int64 torch computes in
            v &= (1 << 64) - 1
            return v - (1 << 64) if v >> 63 else v

        def shr(x, s):                                          # logical shift right of an int64 tensor
            return (x >> s) & ((1 << (64 - s)) - 1)
        key = signed((int(seed) * 0xD1342543DE82EF95 + int(batch) * 0xA0761D6478BD642F + 0x2545F4914F6CDD1D))
        x = self._index * signed(0x9E3779B97F4A7C15) + key
        x = (x ^ shr(x, 30)) * signed(0xBF58476D1CE4E5B9)
        x = (x ^ shr(x, 27)) * signed(0x94D049BB133111EB)
        x = x ^ shr(x, 31)
        return (shr(x, 33) % 7).to(torch.uint8).unsqueeze(1)


def blend(carved, forward_games, seed: int = 0, batch: int = 0, lead: bool = True):
    """One pool from both of the reference's producers: `carved` = (rows, pieces) device tensors of the carving generator,
    followed by the winnable forward games as `translate()` hands them over.  (Which entry an episode draws is the
    environment's assignment; the reference's queue interleaves by arrival.)"""
    import torch
    rows, pieces = carved
    if forward_games is None or forward_games.count == 0:
        return rows, pieces
    f_rows, f_pieces = forward_games.translate(seed, batch, lead)
    return torch.cat([rows, f_rows]), torch.cat([pieces, f_pieces])


class PoolRefresher:
    """Keeps a running BatchedTetris supplied with FRESH prescribed configurations -- the analogue of the reference's
    two producer processes feeding the reset queue while games are played (game/tetris.py:195-211, 473-488).

    Carved (solvable) configurations are generated on the device (`tpl_generate_configs_device`, one configuration per
    lane) on a SIDE stream while the environment steps on its own stream; when a batch is finished and the handle's
    other pool buffer is out of use, poll() packs the batch into that buffer (still on the side stream), makes the
    stepping stream wait for exactly that, and the buffer becomes current: episodes that begin afterwards draw from
    it, boards that are mid-episode finish on the old one.  Batches never repeat: batch b is configurations
    [b * count, (b + 1) * count) of the generator's counter-based stream.

        feeder = PoolRefresher(env, count=1 << 20)
        while training:
            env.step_into(...)
            feeder.poll()            # cheap: an event query, no wait on any stream; swaps at most once per M + 1 steps
    """

    # What a footprint costs a 2^20-board step loop and what it supplies, measured (profiles/r05_live_supply/
    # footprint_grid_waves_x_batch.log: pool-sized batches; L = 10, M = 40): waves -> (step time with the generator beside it /
    # alone, fresh configurations a second).  The reference hands every reset a FRESH game (reset() blocks on queue.get(),
    # game/tetris.py:445-447); here a pool is re-dealt until the next batch replaces it, resets / fresh = the reuse factor.
    FOOTPRINTS = ((64, 1.04, 0.8e6), (256, 1.07, 5.8e6), (512, 1.095, 10.3e6), (768, 1.126, 14.1e6), (1024, 1.155, 16.7e6))

    @classmethod
    def waves_for(cls, target_slowdown: float) -> int:
        """The largest measured footprint whose cost stays within `target_slowdown` (at least the smallest)."""
        fits = [w for w, cost, _ in cls.FOOTPRINTS if cost <= target_slowdown]
        return max(fits) if fits else cls.FOOTPRINTS[0][0]

    def __init__(self, env, count: int = 0, seed: int = 0, first: int = 0, waves: int = 0, reserved_cus: int = 0,
                 low_priority: bool = False, cutoff: int = 0, strict: bool = False, max_capped_batches: int = 3,
                 forward_seeds=None, forward_lead: bool = True, target_slowdown: float = 1.13):
        """count: configurations per batch = the size of the pool the batch becomes (0 = as many as the environment has
        boards: with fewer, the generator idles between swaps -- a swap has to wait M + 1 steps for the boards still on the
        other buffer -- and the pool is re-dealt more often for the same cost).
        waves: how many persistent 64-lane waves share the generator's queue: its footprint beside the stepping environment.
        0 = chosen by `target_slowdown` from the measured table above (the default 1.13 picks 768 waves: 1.13 x the step
        time alone for 14 M fresh configurations a second, a pool reuse factor of about 300 at 2^20 boards under random play;
        through round 4 the default was count / 256 waves on batches of 65,536: 1.02-1.10 x, 2 M/s, a factor of 2,300).
        What a footprint costs and supplies is in bench.py's `live_supply_run`
        (`by_generator_footprint`; profiles/NOTES.md has the history).  (The step kernel raises its waves'
        issue priority above the generator's; without that any generator wave on a SIMD cost the whole launch 18-29 %.)
        reserved_cus > 0 runs the generator on a CU-masked stream of that many compute units (`tpl_stream_create`),
        low_priority on a lowest-priority stream: both measured 3x SLOWER steps than a plain side stream -- kept as
        options because the review of round 2 asked for the comparison, not because they help.
        An (L, M, cutoff) under which NONE of the generator's pilot configurations finishes is refused HERE (TplError from the first start()).  A batch
        in which some configuration ran into all of its cut-offs is dropped: `strict` raises at once; otherwise a warning names
        (L, M, cutoff), and after `max_capped_batches` such batches IN A ROW the refresher stops (`stopped`; poll() returns
        False from then on) instead of spending the generator's worst case beside the training loop for ever."""
        import torch
        self.env, self.count, self.seed, self.next_first = env, int(count) or int(env.num_envs), int(seed), int(first)
        self.cutoff = int(cutoff)     # the restart rule's iteration cut-off (0 = by L), as generate_configs / carved_configs take it
        self.target_slowdown = float(target_slowdown)
        self.waves = int(waves) or self.waves_for(self.target_slowdown)   # (the launch never has more lanes than configurations)
        self.strict, self.max_capped_batches = bool(strict), int(max_capped_batches)
        self._masked = bool(reserved_cus or low_priority)
        self.side = side_stream(env, reserved_cus, low_priority) if self._masked else concurrent_stream(env)
        self._stepping = torch.cuda.current_stream(env.device).cuda_stream      # the stream `side` was tested against
        self._retest = False          # stepping has moved to another stream: test a side stream against it at the next start()
        self._ready = None            # event recorded behind the batch being generated
        self._bad_host = None         # pinned: the batch's count of configurations that could not be carved
        self._batch = None
        self.swaps = 0
        self.capped_batches = 0       # batches dropped because a configuration's attempts all ran into their cut-off
        self._capped_in_a_row = 0
        self.stopped = False
        self.forward = ForwardGames(env, forward_seeds) if forward_seeds is not None else None
        self.forward_lead, self._batches = bool(forward_lead), 0
        self._stats_at_swap = None    # the environment's episode counters when the current pool came in (device tensor)
        self._pool_entries = 0
        self.start()

    def start(self) -> None:
        """Begin generating the next batch on the side stream (returns at once)."""
        import ctypes as C
        import torch
        from ._lib import check
        env, d, n = self.env, self.env.device, self.count
        if self._retest and not torch.cuda.is_current_stream_capturing():
            # stepping has moved to another stream since the side stream was tested: HIP may have put the two on one hardware
            # queue (a 30-ms stall per swap).  Tested here, between batches (the probe synchronises the device: never while the
            # stepping stream is being captured); cached per stepping stream, so a loop that alternates between two pays once each.
            self.side = concurrent_stream(env)
            self._stepping = torch.cuda.current_stream(d).cuda_stream
            self._retest = False
        with torch.cuda.stream(self.side):
            rows = torch.empty((n, 20), dtype=torch.int16, device=d)
            pieces = torch.empty((n, env.M + 1), dtype=torch.uint8, device=d)
            status = torch.empty(n, dtype=torch.int32, device=d)
            nbytes = env._lib.tpl_generate_configs_device_work_bytes(env.M, n)
            work = torch.empty(nbytes, dtype=torch.uint8, device=d)
            check(env._lib.tpl_generate_configs_device_waves(env.L, env.M, self.seed, self.next_first, n, self.cutoff, self.waves,
                                                             C.c_void_p(rows.data_ptr()), C.c_void_p(pieces.data_ptr()), None, None,
                                                             C.c_void_p(status.data_ptr()), C.c_void_p(work.data_ptr()), nbytes,
                                                             self.side.cuda_stream))
            # how many configurations hit the iteration cap: into pinned host memory, still on the side stream and ahead
            # of the event, so that poll() reads a host value (no .item(), no wait on any stream)
            if self._bad_host is None:
                self._bad_host = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._bad_host.copy_(status.sum().reshape(1), non_blocking=True)
            self._ready = torch.cuda.Event()
            self._ready.record(self.side)
        self._batch = (rows, pieces, status, work, self.next_first, self.side)
        self.next_first += n

    def poll(self) -> bool:
        """Swap the finished batch in if it is ready and the handle can take it; True when a swap happened."""
        import torch
        if self.stopped or self._ready is None or not self._ready.query() or self.env.pool_info()["steps_until_swap"] > 0:
            return False
        rows, pieces, _, _, first, made_on = self._batch
        bad = int(self._bad_host[0])                               # host memory, written ahead of the event that has fired
        if bad:
            # every attempt of some configuration ran into its cut-off although a pilot configuration of this (L, M, cutoff)
            # finished: the cut-off is marginal for this (L, M).  The batch is dropped, the pool stays as it is.
            import warnings
            self.capped_batches += 1
            self._capped_in_a_row += 1
            what = (f"{bad} configuration(s) of the batch at {first} ran into every cut-off of the restart rule "
                    f"(L={self.env.L}, M={self.env.M}, cutoff={self.cutoff or 'by L'}): batch dropped, the pool is not refreshed")
            if self.strict:
                self._ready = self._batch = None
                raise RuntimeError(what)
            if self._capped_in_a_row >= self.max_capped_batches:
                self.stopped = True
                self._ready = self._batch = None
                warnings.warn(what + f"; {self._capped_in_a_row} batches in a row: the refresher has STOPPED (pass a larger cutoff)")
                return False
            if self.capped_batches == 1:
                warnings.warn(what)
            self.start()
            return False
        self._capped_in_a_row = 0
        main = torch.cuda.current_stream(self.env.device)
        if main.cuda_stream != self._stepping and not self._masked:
            self._retest = True                                    # the NEXT batch goes to a stream tested against `main` (start())
        # the finished batch is packed on the stream that made it: its tensors belong to that stream's allocations
        made_on.wait_stream(main)                                  # launches that still read the buffer being replaced
        with torch.cuda.stream(made_on):
            if self.forward is not None:
                made_on.wait_stream(main)                          # the forward games were made on the stepping stream
                rows, pieces = blend((rows, pieces), self.forward, self.seed, self._batches, self.forward_lead)
            self._batches += 1
            self.env.load_configs(rows, pieces, validate=False)    # the generators cannot emit an invalid piece id
        main.wait_stream(made_on)                                  # the next step sees the packed pool
        for mem in self.env._pool_mems:                            # allocated on the side stream, read on the stepping one
            if mem is not None:
                mem.record_stream(main)
        self.last_batch = (rows, pieces, first)
        self.swaps += 1
        self._stats_at_swap = self.env.stats_tensor()              # enqueued behind the swap: no host wait
        self._pool_entries = int(rows.shape[0])
        self.start()
        return True

    def reuse_factor(self) -> float:
        """How many times over the CURRENT pool has been dealt: episodes finished since it was swapped in / its entries (host
        sync: one small copy).  The reference deals every game once (reset() blocks on queue.get(), game/tetris.py:445-447);
        here a pool is re-dealt until the next batch replaces it -- this is the number to watch, hold_reuse() the way to bound
        it.  0.0 before the first swap."""
        if self._stats_at_swap is None or not self._pool_entries:
            return 0.0
        finished = int((self.env.stats_tensor()[0] - self._stats_at_swap[0]).item())
        return finished / float(self._pool_entries)

    def hold_reuse(self, limit: float = 1.0) -> bool:
        """Bound the reuse of a pool: if the current one has been dealt `limit` times over, WAIT on the host for the batch being
        generated and swap it in (True when a swap happened; False when the limit is not reached, or the handle cannot take a
        pool yet -- fewer than M + 1 steps since the last swap: step on and call again).  With limit = 1 an episode starts, in
        expectation, from a configuration no other episode of this pool has had: the reference's supply semantics, at the
        generator's rate instead of the step kernel's (bench.py: `live_supply_run.reuse_held_at_1`).  Costs a host
        synchronisation per call: every few steps is enough."""
        if self.stopped or self._ready is None or self.reuse_factor() < limit:
            return False
        self._ready.synchronize()
        return self.poll()

    def close(self) -> None:
        """Waits for the batch in flight and drops it (the side stream itself lives as long as the process)."""
        if self._batch is not None:
            self._batch[5].synchronize()
        self.side.synchronize()
        self._ready = self._batch = None


def save_pool(path: str, L: int, M: int, rows, pieces, solution=None, solution_len=None) -> None:
    rows = np.ascontiguousarray(rows, dtype=np.uint16)
    pieces = np.ascontiguousarray(pieces, dtype=np.uint8)
    if rows.ndim != 2 or rows.shape[1] != 20 or pieces.shape != (rows.shape[0], M + 1):
        raise ValueError(f"rows must be [n, 20] and pieces [n, {M + 1}]")
    if pieces.max(initial=0) > 6 or rows.max(initial=0) > 0x3FF:
        raise ValueError("piece ids must be 0..6 and rows must use 10 columns")
    extra = {}
    if solution is not None:
        extra = dict(solution=np.ascontiguousarray(solution, dtype=np.uint8),
                     solution_len=np.ascontiguousarray(solution_len, dtype=np.int32))
    np.savez_compressed(path, L=np.int32(L), M=np.int32(M), rows=rows, pieces=pieces, **extra)


def load_pool(path: str) -> dict:
    f = np.load(path, allow_pickle=False)
    out = dict(L=int(f["L"]), M=int(f["M"]), rows=f["rows"], pieces=f["pieces"])
    if "solution" in f.files:
        out["solution"], out["solution_len"] = f["solution"], f["solution_len"]
    if out["pieces"].shape != (out["rows"].shape[0], out["M"] + 1):
        raise ValueError("corrupt pool file: pieces do not match rows / M")
    return out
