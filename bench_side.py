"""bench_side.py -- everything in bench.py's record that is NOT the headline: the rooflines' yardsticks, the CPU baseline, the
guard the side figures run under (`SideFigures`) and the side figures themselves (`measure_*`).  None of these enters `value`.
bench.py holds the arguments, the timed region and the line; it imports this module AFTER the timed region's imports, and the
tests import `SideFigures` through either module.

Side figures (keys of the detail record): `fused_rollout` (tpl_rollout, K steps per launch), `scaling_model` + `shard_run`
(N = 1: the shards of a 2-, 4- and 8-GPU run of the same job, measured on this one GPU), `weak_scaling_job` (N > 1: 2^20 boards on
EVERY rank), `out_of_cache` (nothing fits the Infinity Cache), `carved_pool_run`, `live_supply_run`, `config1_run` (BASELINE
configs[1]), `config_supply` (the generators), `actor_loop` (BASELINE configs[4])."""
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

ALGO_BYTES_PER_BOARD_STEP = 96          # 46 B read + 49 B write, rounded
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16 (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3            # f32-input MFMA = the f32 vector rate (MI355X_MICROARCH.md)
MODEL_FLOPS_PER_BOARD = 2.0 * (217 * 128 + 3 * 128 * 128 + 128 * 14)     # Model(217, 14), model/model.py:9-20: 157,440
SIMDS, PEAK_CLOCK_GHZ = 1024, 2.4       # 256 CUs x 4 SIMDs; peak engine clock (MI355X_MICROARCH.md)
HW_CYCLES_PER_WAVE64_VALU = 2.0         # a SIMD-32 issues a wave64 vector instruction over 2 cycles (MI355X_MICROARCH.md, "Wave scheduling")
HBM_ACHIEVABLE_GBS = 6300.0             # what a streaming kernel reaches of the 8 TB/s (MI355X_MICROARCH.md: "~6.3 TB/s achievable")
PROFILED_L_M = (10, 40)                 # the forms of profiles/valu_issue.json were counted at the bench's own L, M
_VALU_ISSUE = None


def library_digest():
    """Digest of the sources lib/libtetris_piclim.so was built from (written beside it by the build)."""
    try:
        return open(os.path.join(ROOT, "lib", "libtetris_piclim.so.sha256")).read().strip()
    except OSError:
        pass
    try:                                  # no stamp beside the library: the digest of the sources as they stand
        import tetris_piclim as T
        return T._lib._source_digest()
    except Exception:                     # noqa: BLE001
        return None


def valu_roofline(form, units_per_s, L=None, M=None, steps_per_launch=None):
    """The roofline of a kernel that HBM does not bound: VECTOR-INSTRUCTION ISSUE, priced twice.

    `frac` -- the fair price: a SIMD issues one wave64 vector instruction per c cycles at best, c = the kernel's static
    instruction mix priced with the issue costs MEASURED on gfx950 (tools/valu_mix.py, tools/valu_rates.hip: 2.5 cycles for a
    handful of plain two-operand opcodes, 4 for the rest -- v_cndmask / v_lshl_or / v_bfe / v_and_or / v_perm / the 64-bit
    shifts, which make up the move; c = 3.3 for both kernels), `peak` = 1024 SIMDs x 2.4 GHz / c.
    `frac_hw` -- the hardware's own price: the guide's 2 cycles per wave64 instruction on a SIMD-32 (what several waves
    together may reach with the cheapest opcodes), `peak_hw` = 1024 x 2.4 / 2 = 1,229 G wave-instructions/s.
    `frac_of_lane_slots` -- `frac_hw` x the lanes active per instruction / 64: of the lane-slots the hardware offers, the share
    that did work (divergence inside a wave included).  Always frac_of_lane_slots <= frac_hw <= frac.

    `achieved` = vector instructions per unit of work (SQ_INSTS_VALU of the committed counter pass of this same form,
    profiles/valu_issue.json <- tools/update_valu_issue.py) x the units per second measured HERE; the per-unit count is a
    property of the code path (1.6800 per board-step at 2^20 boards, 1.6802 at 131,072), NOT of the grid size, but it does
    depend on L, M and on the steps per launch (a launch's fixed part is spread over them): a run at another (L, M, steps per
    launch) than the profiled one gets no price (`not_comparable`), whatever its speed (round-5 advisor finding)."""
    global _VALU_ISSUE
    if _VALU_ISSUE is None:
        try:
            _VALU_ISSUE = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))["forms"]
        except (OSError, KeyError, ValueError):
            _VALU_ISSUE = {}
    f = _VALU_ISSUE.get(form)
    if not f or not units_per_s:
        return None
    profiled_steps = f["units_per_launch"] // f["grid"] if f["unit"] == "board-step" else None
    differs = [f"{k} = {got} (profiled: {want})" for k, got, want in (("L", L, PROFILED_L_M[0]), ("M", M, PROFILED_L_M[1]),
                                                                       ("steps per launch", steps_per_launch, profiled_steps))
               if got is not None and want is not None and got != want]
    if differs:
        return {"bound": "valu-issue", "frac": None, "not_comparable": "the committed instruction count is for another form: " + ", ".join(differs)}
    c = f["cycles_per_valu_instruction"]
    achieved = f["valu_per_unit"] * units_per_s / 1e9
    peak, peak_hw = SIMDS * PEAK_CLOCK_GHZ / c, SIMDS * PEAK_CLOCK_GHZ / HW_CYCLES_PER_WAVE64_VALU
    lanes = f.get("lanes_active_per_valu_instruction")
    stale = f["stamp"].get("source_digest") != library_digest()
    profiled = {"frac": f["frac"], "duration_ns": f["duration_ns_median"]}
    if f["duration_ns_median"] >= 150000 and f["clock_GHz_held"] <= PEAK_CLOCK_GHZ:
        # the clock derived as GRBM_GUI_ACTIVE / 8 / duration: only for launches long enough for the ratio to mean something
        profiled.update(frac_at_the_clock_held=f["frac_at_clock_held"], clock_GHz_held=f["clock_GHz_held"])
    return {"bound": "valu-issue", "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s", "frac": achieved / peak,
            "peak_hw": peak_hw, "frac_hw": achieved / peak_hw,
            "frac_of_lane_slots": (achieved / peak_hw * lanes / 64.0) if lanes else None,
            "valu_instructions_per_" + f["unit"].replace("-", "_"): f["valu_per_unit"], "cycles_per_valu_instruction": c,
            "cycles_per_valu_instruction_hw": HW_CYCLES_PER_WAVE64_VALU, "lanes_active_per_valu_instruction": lanes,
            "in_the_profiled_run": profiled, "count_stale": stale,
            "source": f"NOT counted in this run: {f['source']} (commit {f['stamp'].get('git_head')}, source digest "
                      f"{str(f['stamp'].get('source_digest'))[:12]}), {f['kernel']} at grid {f['grid']}"}


def numpy_port_leg(L, M, seed, cores, seconds=2.0):
    """the NumPy per-board restatement of the reference's move (oracle/numpy_port.py: the reference's
    own operation sequence, so its rate on a core is the reference's rate on that core), one process per host core.
    Started BEFORE this process touches the GPU (a process that has initialised HIP must not exec another)."""
    cmd = [sys.executable, "-m", "oracle.numpy_port", str(seed), "256", str(L), str(M), str(seconds)]
    procs = [subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True) for _ in range(cores)]
    rates = []
    for p in procs:
        out, _ = p.communicate(timeout=120)
        if p.returncode == 0:
            rates.append(json.loads(out.strip().splitlines()[-1])["moves_per_s"])
    if len(rates) != cores:
        return None
    return {"value": sum(rates), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "per_core": sum(rates) / cores,
            "sample": f"{cores} processes x {seconds:.0f} s of move-and-reset over 256 synthetic configurations, L={L} M={M}"}


def cpu_model():
    """Model name of the host CPU (SURVEY 8d-ii asks for it beside the core count)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_limits():
    """What bounds the host threads of the CPU baseline: the machine's CPUs, this process's affinity mask, the cgroup's CPU
    quota, and TPL_CPU_BUDGET when set -- `cores_used` = _lib.cpu_budget(), the smallest of them, and `limited_by` names it."""
    import tetris_piclim as T
    machine = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else machine
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        pass
    used = T._lib.cpu_budget()
    override = os.environ.get("TPL_CPU_BUDGET")
    if override:
        limited_by = f"TPL_CPU_BUDGET={override}"
    elif quota is not None and used < affinity:
        limited_by = f"cgroup CPU quota (cpu.max = {quota:g} CPUs; the affinity mask allows {affinity})"
    elif affinity < machine:
        limited_by = f"affinity mask ({affinity} of the machine's {machine} CPUs)"
    else:
        limited_by = "nothing: every CPU of the machine"
    return {"cores_available": machine, "cores_in_affinity_mask": affinity, "cgroup_cpu_quota": quota, "cores_used": used,
            "limited_by": limited_by}


def cpu_baseline(L, M, seed, numpy_leg, seconds=2.0):
    """The oracle's loop (game/performance_test.py:13-17 shape: move, reset when finished) on the host cores: a bounded
    sample of about `seconds` of wall time on all the cores the container may use (`limited_by` says what bounds them)."""
    from oracle import oracle as O
    limits = cpu_limits()
    cores = limits["cores_used"]
    boards, steps = 262144, 40
    O.bench_run(seed, 4096, L, M, 8, cores)                      # warm the thread pool / page in
    done, sec = O.bench_run(seed, boards, L, M, steps, cores)    # calibration: a few hundredths of a second
    steps = int(max(40, min(20000, seconds * (done / sec) / boards)))
    done, sec = O.bench_run(seed, boards, L, M, steps, cores)
    out = dict({"value": done / sec, "unit": "env-steps/s", "cores": cores}, **limits, cpu_model=cpu_model(), kind="port",
               sample=f"{boards} boards x {steps} lockstep steps, L={L} M={M}, auto-reset, {cores} threads, {sec:.1f}s",
               numpy_port=numpy_leg)
    if numpy_leg and "per_core" in numpy_leg:
        out["c_port_over_numpy_port_per_core"] = (done / sec / cores) / numpy_leg["per_core"]
    try:
        # the Python reference itself cannot travel to this box; its rate beside both restatements was measured in
        # the build container (tests/golden/time_reference.py), one core each, same workload
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_timing.json")))
        out["reference_in_build_container"] = {k: ref[k] for k in (
            "reference_moves_per_s", "numpy_port_moves_per_s", "c_port_env_steps_per_s", "c_port_over_reference",
            "numpy_port_over_reference")}
    except (OSError, KeyError, ValueError):
        pass
    return out


def timed(torch, dev, fn, reps):
    """Average milliseconds of fn() over `reps` calls, by HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


WATCHDOG_STATUS = 3


class SideFigures:
    """Runs the side figures of the line so that none of them can take the headline with it (round-4 review: an exception in a
    side figure lost the line, and at N > 1 left the other ranks in a collective until the process group timed out).

    * `run(name, fn)` calls fn() under a guard: an exception becomes `{"error": "<type>: <message>"}` under that key.
    * At world > 1 a figure runs on every rank or on none, and every collective of a figure happens OUTSIDE fn: before it the
      ranks agree whether to start (one of them may be out of time), after it whether all of them came through -- both over
      `ctl`, a gloo group of host tensors that a sick GPU cannot take down.  Only then are the ranks' numbers combined
      (`max_over_ranks`), so no rank ever waits in a collective for one that raised.
    * `budget_s`: figures that would start after that many seconds of side figures are skipped (`{"skipped": ...}`).
    * `watchdog(seconds, emit)`: if the side figures are still running after that long, `emit()` (rank 0: print the line with
      what there is) is called from a timer thread and the process exits with status 3 (WATCHDOG_STATUS): the measured
      headline is on stdout, and the hang is visible to whoever started the run (round-5 advisor finding: status 0 made a
      kernel hung on the GPU look like a clean run; find its cause from `side_figures.abandoned`, which names the figure).
    `inject` names figures made to fail on purpose ("name" or "name@rank", comma separated; TPL_BENCH_INJECT_FAILURE): the
    tests' way to see all of the above happen."""

    def __init__(self, world=1, rank=0, dist=None, ctl=None, budget_s=None, inject="", clock=time.perf_counter, gather=None,
                 distributed=None):
        self.world, self.rank, self.dist, self.ctl = world, rank, dist, ctl
        self.distributed = world > 1 if distributed is None else bool(distributed)   # (a forced one-rank group counts)
        self._gather_fn = gather      # main() hands in the device-tensor gather of the job's own group when no gloo group could be made
        self.budget_s, self.clock, self.t0 = budget_s, clock, clock()
        self.inject = [x.strip() for x in (inject or "").split(",") if x.strip()]
        self.log = []                 # (name, seconds, outcome) in the order run
        self.running = None
        self._timer = None

    # -- host-side agreement between the ranks (gloo)
    def _gather(self, value):
        if self._gather_fn is not None:
            return self._gather_fn(float(value))
        if not self.distributed:
            return [float(value)]
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        got = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(got, t, group=self.ctl)
        return [float(g.item()) for g in got]

    def max_over_ranks(self, value):
        return max(self._gather(value))

    def _injected(self, name):
        return any(x == name or x == f"{name}@{self.rank}" for x in self.inject)

    def run(self, name, fn):
        over = self.budget_s is not None and self.clock() - self.t0 > self.budget_s
        if any(self._gather(1.0 if over else 0.0)):
            self.log.append((name, 0.0, "skipped"))
            return {"skipped": "the side figures had used their budget (--side-budget"
                               + (f" = {self.budget_s:.0f} s" if self.budget_s is not None else "") + ") on some rank when this one's turn came"}
        self.running = name
        t0 = self.clock()
        result = error = None
        try:
            if self._injected(name):
                raise RuntimeError(f"failure injected into '{name}' (TPL_BENCH_INJECT_FAILURE)")
            result = fn()
        except Exception as e:        # noqa: BLE001 -- whatever it is, the headline survives it
            error = f"{type(e).__name__}: {e}"[:400]
        failed = [r for r, f in enumerate(self._gather(0.0 if error is None else 1.0)) if f]
        self.running = None
        self.log.append((name, self.clock() - t0, "ok" if not failed else "failed"))
        if failed:
            out = {"error": error or f"rank(s) {failed} failed; this rank's own measurement was dropped with theirs"}
            if self.distributed:
                out["failed_ranks"] = failed
            return out
        return result

    @staticmethod
    def ok(result):
        return isinstance(result, dict) and "error" not in result and "skipped" not in result

    def summary(self):
        return {"seconds": {n: round(s, 3) for n, s, _ in self.log}, "failed": [n for n, _, o in self.log if o == "failed"],
                "skipped": [n for n, _, o in self.log if o == "skipped"], "total_seconds": round(self.clock() - self.t0, 3)}

    def watchdog(self, seconds, emit):
        import threading

        def fire():
            try:
                emit(self.running)
            finally:
                sys.stdout.flush()
                os._exit(WATCHDOG_STATUS)
        self._timer = threading.Timer(seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def releases_envs(fn):
    """The measure_* functions that build environments of their own register them through `keep(...)`: whatever happens
    inside -- the function's guard in SideFigures.run turns an exception into an {"error": ...} entry -- their handles and
    device memory are released before the next figure starts."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        made = []

        def keep(env):
            made.append(env)
            return env
        try:
            return fn(*args, keep=keep, **kwargs)
        finally:
            for env in made:
                try:
                    env.terminate()
                except Exception:      # noqa: BLE001
                    pass
    return wrapper


def measure_fused_rollout(torch, T, env, actions, first, K, chunk, compact=False):
    """tpl_rollout over the same pre-staged actions, `chunk` steps per launch, writing the same
    per-step reward/done outputs as the step loop -- or, `compact`, tpl_rollout_trajectory: one byte per board-step."""
    n, dev = env.num_envs, env.device
    rs = torch.empty((chunk, n), dtype=torch.float32, device=dev)
    ds = torch.empty((chunk, n), dtype=torch.uint8, device=dev)
    traj = torch.empty(((chunk + 3) // 4, n), dtype=torch.int32, device=dev)
    launches = K // chunk

    def run():
        for c in range(launches):
            a = actions[first + c * chunk: first + (c + 1) * chunk]
            if compact:
                T._lib.check(env._lib.tpl_rollout_trajectory(env._h, ctypes.c_void_p(a.data_ptr()), a.stride(0), chunk,
                                                             ctypes.c_void_p(traj.data_ptr()), None, env._stream()))
                continue
            T._lib.check(env._lib.tpl_rollout(env._h, ctypes.c_void_p(a.data_ptr()), a.stride(0), chunk,
                                              ctypes.c_void_p(rs.data_ptr()), ctypes.c_void_p(ds.data_ptr()), None, None,
                                              env._stream()))
    run()
    torch.cuda.synchronize(dev)
    return timed(torch, dev, run, 1) / (launches * chunk)


def action_rows(actions):
    """The rows of an [S, n] action tensor as S views made ONCE.  `actions[t]` inside a loop builds a new view object per step
    (0.8 us of Python): nothing at 2^20 boards, a seventh of the period at a shard's 131,072, where the host's calls per second
    are the limit (`timing.host_call_us`; profiles/NOTES.md, round-4 host issue-rate table) -- and no part of what is being measured."""
    return list(actions.unbind(0))


def measure_weak_job(torch, T, dev, rank, world, L, M, per_gpu, seed, K):
    """Side figure for N > 1 (the headline is BASELINE configs[3], fixed total work): `per_gpu` boards on EVERY rank,
    i.e. a job that grows with the node.  Same pool on every rank, everything keyed by the global board index.  Returns this
    rank's ms per step; no collective in here (SideFigures.run), the ranks start together within the agreement that precedes it."""
    shard = T.sharding.weak_shard(rank, world, per_gpu)
    env = T.BatchedTetris(L, M, shard.boards, device=dev, seed=seed, global_offset=shard.global_offset,
                          auto_reset=True, assign="hash")
    try:
        rows, pieces = env.synthetic_configs(per_gpu, first=0)
        env.load_configs(rows, pieces)
        del rows, pieces
        env.reset()
        S = max(1, min(K, 200))
        actions = torch.empty((S, shard.boards), dtype=torch.uint8, device=dev)
        for t in range(S):
            env.synthetic_actions(t, out=actions[t])
        reward = torch.empty(shard.boards, dtype=torch.float32, device=dev)
        done = torch.empty(shard.boards, dtype=torch.uint8, device=dev)
        rows_of = action_rows(actions)
        for t in range(20):
            env.step_into(rows_of[t % S], reward, done)
        torch.cuda.synchronize(dev)
        step = iter(range(S))
        ms = timed(torch, dev, lambda: env.step_into(rows_of[next(step)], reward, done), S)
    finally:
        env.terminate()
    return {"ms": ms, "steps": S, "global_boards": shard.global_boards}


def weak_job_line(ms, steps, per_gpu, global_boards):
    gbs = ALGO_BYTES_PER_BOARD_STEP * per_gpu / (ms * 1e-3) / 1e9
    return {"scaling": "weak", "boards_per_gpu": per_gpu, "global_boards": global_boards, "unit": "env-steps/s",
            "value": float(global_boards) / (ms * 1e-3), "ms_per_step": ms, "steps": steps,
            "per_gpu_roofline_frac": gbs / HBM_PEAK_GBS,
            "note": "NOT the BASELINE workload for N > 1 (that is 1,048,576 boards in total): a job N times as large"}


@releases_envs
def measure_shard_run(torch, T, dev, L, M, seed, total, ranks, chunk, keep=None):
    """What ONE rank of a `ranks`-GPU run of BASELINE configs[3] does, measured on this one GPU: rank 0's shard of `total`
    boards (131,072 at 8 ranks), over the whole `total`-entry pool, in the three forms the library offers: one tpl_step launch
    per step issued eagerly, `chunk` such steps as one replayed HIP graph (what `bench.py --gpus N` times below 2^19 boards per
    GPU), and `chunk` steps per launch (tpl_rollout, same per-step outputs).  `frac` prices the per-launch forms against HBM
    at the canonical 96 B per board-step ON THE SHARD's boards; `host_call_us` is what one step_into() costs the host thread."""
    shard = T.sharding.strong_shard(0, ranks, total)
    n = shard.boards
    env = keep(T.BatchedTetris(L, M, n, device=dev, seed=seed, global_offset=shard.global_offset, auto_reset=True, assign="hash"))
    rows, pieces = env.synthetic_configs(total, first=0)
    env.load_configs(rows, pieces)
    del rows, pieces
    env.reset()
    S = 8 * chunk
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    rows_of = action_rows(actions)
    for t in range(50):
        env.step_into(rows_of[t % S], reward, done)
    torch.cuda.synchronize(dev)

    def leg(ms, **more):
        gbs = ALGO_BYTES_PER_BOARD_STEP * n / (ms * 1e-3) / 1e9
        return dict({"us_per_step": ms * 1e3, "value_per_gpu": float(n) / (ms * 1e-3), "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                     f"value_x{ranks}_if_every_rank_matches": float(n) * ranks / (ms * 1e-3)}, **more)
    step = iter(range(S))
    ms_step = timed(torch, dev, lambda: env.step_into(rows_of[next(step)], reward, done), S)
    # the host's share: the same calls with the queue never waited for (a few hundred launches fit the queue)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for t in range(200):
        env.step_into(rows_of[t], reward, done)
    host_us = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize(dev)
    rs = torch.empty((chunk, n), dtype=torch.float32, device=dev)
    ds = torch.empty((chunk, n), dtype=torch.uint8, device=dev)
    replay = env.capture_steps(actions[:chunk], rs, ds)
    replay()
    torch.cuda.synchronize(dev)
    ms_graph = timed(torch, dev, replay, 8) / chunk
    ms_fused = measure_fused_rollout(torch, T, env, actions, 0, S, chunk)
    env.terminate()
    # the multi-step kernel is not an HBM kernel (a board's 64 B cross the memory once per `chunk` steps): priced by vector issue
    fused_leg = {"us_per_step": ms_fused * 1e3, "value_per_gpu": float(n) / (ms_fused * 1e-3),
                 f"value_x{ranks}_if_every_rank_matches": float(n) * ranks / (ms_fused * 1e-3), "steps_per_launch": chunk,
                 "outputs": "per-step reward f32 + done u8 written",
                 "roofline": valu_roofline("rollout_shard_131072" if n == 131072 else "rollout_f32_u8_50", float(n) / (ms_fused * 1e-3),
                                           L, M, chunk)}
    return {"workload": f"rank 0's shard of {total} boards over {ranks} GPUs = {n} boards, {total}-entry pool, L={L} M={M}",
            "boards": n, "global_boards": total, "ranks": ranks, "unit": "env-steps/s",
            "tpl_step": leg(ms_step, launches_per_step=1, host_call_us=host_us),
            "capture_steps": leg(ms_graph, steps_per_graph=chunk),
            "tpl_rollout": fused_leg,
            "note": "measured on ONE GPU; an N-GPU run's headline is global_boards / (the slowest rank's period)"}


GRAPH_BELOW_BOARDS = 1 << 19            # bench.py --gpus N replays captured graphs when a rank holds fewer boards than this


def launch_mode_for(world, boards_per_gpu, asked="auto"):
    """How the timed region issues its K steps: "eager" (one tpl_step call per step) or "graph" (the same step_kernel launches,
    captured <= 50 at a time and replayed).  auto: graphs for N > 1 below 2^19 boards per GPU, where the host's 5 us per call
    are as long as the device's period (profiles/NOTES.md, round-4 host issue-rate table); always eager at N = 1, so that the
    N = 1 line of a scaling run is the plain bench line."""
    if asked in ("eager", "graph"):
        return asked
    return "graph" if world > 1 and boards_per_gpu < GRAPH_BELOW_BOARDS else "eager"


def scaling_model(shard_runs, total, value_x1, us_x1, fused_value_x1, chunk):
    """The scaling ceiling as numbers, before hardware gives them: from the shards of a 2-, 4- and 8-GPU run measured on ONE GPU
    (`measure_shard_run`), what `bench.py --gpus N` of the SAME job (fixed total work) can read at best -- every rank as fast as
    this GPU, nothing lost between them (there is no data-path collective) -- in the launch mode it would use at that size,
    and the same for the multi-step kernel.  efficiency = value_xN / (N x value_x1).  The weak-scaling counterpart (2^20 boards on
    EVERY rank) has the one-GPU period at any N: ranks share nothing but the host."""
    per_launch, fused = {}, {}
    for ranks, run in sorted(shard_runs.items()):
        if not SideFigures.ok(run):
            continue
        mode = launch_mode_for(ranks, run["boards"])
        us = run["tpl_step" if mode == "eager" else "capture_steps"]["us_per_step"]
        value = float(total) / (us * 1e-6)
        per_launch[f"x{ranks}"] = {"boards_per_gpu": run["boards"], "launch_mode": mode, "us_per_step": us, "value": value,
                                   "efficiency": value / (ranks * value_x1), "host_call_us": run["tpl_step"]["host_call_us"],
                                   "eager_us_per_step": run["tpl_step"]["us_per_step"], "graph_us_per_step": run["capture_steps"]["us_per_step"]}
        us_f = run["tpl_rollout"]["us_per_step"]
        fused[f"x{ranks}"] = {"us_per_step": us_f, "value": float(total) / (us_f * 1e-6),
                              "efficiency": (float(total) / (us_f * 1e-6)) / (ranks * fused_value_x1) if fused_value_x1 else None}
    return {"job": f"{total} boards IN TOTAL (BASELINE configs[3]), sharded; every period measured on THIS one GPU at the shard's size",
            "x1": {"us_per_step": us_x1, "value": value_x1, "fused_value": fused_value_x1},
            "per_launch": per_launch, f"fused_{chunk}_steps_per_launch": fused,
            "weak": {"boards_per_gpu": total, "us_per_step": us_x1, "value_x8": 8 * value_x1, "efficiency": 1.0,
                     "note": "2^20 boards on EVERY rank (a job N times as large): the one-GPU period at any N, by construction"},
            "reading": "strong scaling of a 15-us step is latency-bound: a shard's launch cannot be shorter than one wave's chain of "
                       "dependent memory trips (~3.7 us) plus the distance between dependent launches (~1.5 us)"}


@releases_envs
def measure_out_of_cache(torch, T, dev, L, M, seed, boards=1 << 23, pool=1 << 21, steps=100, keep=None):
    """The step loop where nothing fits the 256 MiB Infinity Cache: 2^23 boards (256 MiB of state) over a 2^21-entry
    pool (another 256 MiB).  Reported per 2^20 boards so that it reads beside the main line."""
    env = keep(T.BatchedTetris(L, M, boards, device=dev, seed=seed, auto_reset=True, assign="hash"))
    rows, pieces = env.synthetic_configs(pool)
    env.load_configs(rows, pieces)
    del rows, pieces
    env.reset()
    S = 8
    actions = torch.empty((S, boards), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(boards, dtype=torch.float32, device=dev)
    done = torch.empty(boards, dtype=torch.uint8, device=dev)
    for t in range(10):
        env.step_into(actions[t % S], reward, done)
    torch.cuda.synchronize(dev)
    step = iter(range(steps))
    ms = timed(torch, dev, lambda: env.step_into(actions[next(step) % S], reward, done), steps)
    env.terminate()
    gbs = ALGO_BYTES_PER_BOARD_STEP * boards / (ms * 1e-3) / 1e9
    return {"boards": boards, "pool_entries": pool, "resident_bytes": boards * 32 + pool * 128,
            "kernel_ms": ms, "us_per_2^20_boards": ms * 1e3 / (boards / float(1 << 20)),
            "value": float(boards) / (ms * 1e-3), "unit": "env-steps/s", "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
            "note": "state (32 B/board) + pool (128 B/entry) = 512 MiB, twice the Infinity Cache: every launch streams from HBM"}


@releases_envs
def measure_config_supply(torch, T, dev, L, M, seed, keep=None):
    """rates of the prescribed-configuration suppliers (side figures).  Carving on the device
    (a persistent kernel: lanes take configurations from a queue, and once it is dry run further attempts of their wave's
    stragglers under the restart rule) and on the host cores produce the same configurations; the forward generator +
    solver is host code.  Device rates by batch size (a launch lasts as long as its slowest wave) and at the reference's
    own test configuration L = 15, M = 40 (game/main.py:33,50)."""
    import numpy as np
    env = keep(T.BatchedTetris(L, M, 64, device=dev, seed=seed))

    def device_rate(e, count, reps=2):
        e.carved_configs(count)                                  # load the kernel, and let torch's allocator keep the buffers
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for r in range(reps):
            rows, _ = e.carved_configs(count, first=(r + 1) * count)     # returns after the status check (host sync)
        return count * reps / (time.perf_counter() - t0), rows
    count = 1 << 20                                              # a pool's worth
    rate_big, _ = device_rate(env, count)
    rate_small, _ = device_rate(env, 1 << 18)
    rate_huge, _ = device_rate(env, 1 << 22, reps=1)             # the end of a launch amortised over sixteen configurations a lane
    rows = env.carved_configs(1 << 14, first=0)[0]
    host_count = 1 << 14
    t0 = time.perf_counter()
    hrows, _ = T.generate_configs(L, M, host_count, seed=seed)
    dt_host = time.perf_counter() - t0
    same = bool(np.array_equal(rows.cpu().numpy().view(np.uint16), hrows))
    env.terminate()
    ref_env = keep(T.BatchedTetris(15, 40, 64, device=dev, seed=seed))
    rate_ref, _ = device_rate(ref_env, 1 << 18, reps=1)
    ref_env.terminate()
    games = 4000
    t0 = time.perf_counter()
    fw = T.forward_generate(5, 20, np.arange(games))
    dt_fw = time.perf_counter() - t0
    # the same generator + solver as a HIP kernel (one game per lane), seed for seed the host's games
    fenv = keep(T.BatchedTetris(5, 20, 64, device=dev, seed=seed))
    games_dev = 1 << 16                                          # a launch of one wave per SIMD: a wave lasts as long as its slowest game
    fenv.forward_configs(np.arange(games))                       # load the kernel
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dfw = fenv.forward_configs(np.arange(games_dev))
    torch.cuda.synchronize(dev)
    dt_dfw = time.perf_counter() - t0
    same_fw = bool(np.array_equal(dfw["winnable"][:games].cpu().numpy(), fw["winnable"]) and
                   np.array_equal(dfw["rows"][:games].cpu().numpy().view(np.uint16), fw["rows"]) and
                   np.array_equal(dfw["failed_attempts"][:games].cpu().numpy(), fw["failed_attempts"]))
    del dfw
    fenv.terminate()
    return {"unit": "configurations/s", "L": L, "M": M,
            "carve_device": {"value": rate_big, "count": count, "roofline": valu_roofline("carve_1048576", rate_big, L, M),
                             "roofline_batch_of_262144": valu_roofline("carve_262144", rate_small, L, M),
                             "batch_of_262144": rate_small,
                             "rate_ratio_2^20_over_2^18": rate_big / rate_small, "batch_of_4194304": rate_huge,
                             "L15_M40_batch_of_262144": rate_ref},
            "carve_host": {"value": host_count / dt_host, "count": host_count, "threads": T._lib.cpu_budget(),
                           "equal_to_device_output": same},
            "forward_generator_solver_host": {"value": games / dt_fw, "games": games, "L": 5, "M": 20,
                                              "winnable_fraction": float(fw["winnable"].mean())},
            "forward_generator_solver_device": {"value": games_dev / dt_dfw, "games": games_dev, "L": 5, "M": 20, "equal_to_host_output": same_fw,
                                                "note": "one game per lane, a CPython-compatible MT19937 per lane: a serial, divergent "
                                                        "search (the reference feeds it a hundred seeds per batch; the carving generator "
                                                        "is the supply)"}}


def measure_live_supply(torch, T, env, actions, reward, done, seed, count=0, min_swaps=3, min_steps=4000, max_steps=40000,
                        step_fn=None, hold=None, **where):
    """The replenished supply under load (game/tetris.py:195-211, 473-488: producers feed the reset queue while games
    run): PoolRefresher carves `count` configurations at a time on a side stream while the main stream steps, and each
    finished batch becomes the current pool (boards in mid-episode finish on the buffer they started from).  Steady state:
    the timed region starts at the FIRST swap and runs until `min_swaps` more batches have been swapped in (and at least
    `min_steps` steps); the supply rate is those batches over the wall time between the first and the last swap.  `where` = PoolRefresher's waves / reserved_cus / low_priority.
    count = 0: PoolRefresher's default, pool-sized batches (one configuration per board).  `step_fn(t)` replaces the random-action
    step (the policy-driven loop of measure_actor_loop).  `hold` = a reuse limit: instead of poll() the loop calls
    PoolRefresher.hold_reuse(hold) every 32 steps, which WAITS for the generator once the pool has been dealt that many times
    over -- the step rate then is what the reference's fresh-game-per-episode supply costs."""
    n, dev = env.num_envs, env.device
    S = actions.shape[0] if actions is not None else 1
    if step_fn is None:
        def step_fn(t):
            env.step_into(actions[t % S], reward, done)
    rows, pieces = T.generate_configs(env.L, env.M, 4096, seed=seed)          # something carved to start from
    if env.n_configs:
        env.reset()                                                           # no board left on the buffer about to be replaced
    env.load_configs(rows, pieces)
    env.reset()
    for t in range(50):
        step_fn(t)
    torch.cuda.synchronize(dev)
    alone = iter(range(50, 10 ** 9))
    ms_alone = timed(torch, dev, lambda: step_fn(next(alone)), 500 if actions is not None else 100)
    feeder = T.PoolRefresher(env, count, seed=seed, first=4096, **where)
    count = feeder.count
    try:
        # the supplier's first batch is its start-up (the generator's code is loaded, its work memory and the batch's tensors are
        # allocated -- a hipMalloc is a device synchronisation): stepped through untimed, the timed region begins at the first swap
        lead = 0
        while lead < max_steps and not feeder.poll():
            for t in range(32):
                step_fn(lead + t)
            lead += 32
        episodes0 = env.stats()["episodes"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        swap_times = [time.perf_counter()]
        e0.record()
        steps = 0
        while steps < max_steps and (steps < min_steps or len(swap_times) < min_swaps + 1):
            for t in range(32):
                step_fn(lead + steps + t)
            steps += 32
            if feeder.hold_reuse(hold) if hold is not None else feeder.poll():
                swap_times.append(time.perf_counter())       # the host loop runs a bounded queue ahead of the GPU: wall time tracks it
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        resets_per_s = (env.stats()["episodes"] - episodes0) / (ms * 1e-3 * steps)
        waves = feeder.waves
    finally:
        feeder.close()                                        # whatever happened, no generator keeps running beside the next figure
    fresh_per_s = (len(swap_times) - 1) * count / (swap_times[-1] - swap_times[0]) if len(swap_times) >= 2 else None
    return {"unit": "env-steps/s", "value": float(n) / (ms * 1e-3), "ms_per_step": ms, "ms_per_step_without_refresher": ms_alone,
            "slowdown": ms / ms_alone, "configurations_per_batch": count, "generator_waves": waves, "pool_swaps": len(swap_times) - 1, "steps": steps,
            "steps_before_the_first_swap_untimed": lead,
            "configurations_supplied_per_s": fresh_per_s, "resets_per_s": resets_per_s,
            # the reference's reset() blocks on queue.get() (game/tetris.py:445-447): every episode a fresh game, factor 1
            "pool_reuse_factor": (resets_per_s / fresh_per_s) if fresh_per_s else None}


def measure_carved_pool(torch, T, env, actions, reward, done, W, K, pool, seed):
    """SURVEY 8(d) "realism run": the same step loop on a pool of CARVED (solvable) configurations."""
    n, dev, S = env.num_envs, env.device, actions.shape[0]
    rows, pieces = T.generate_configs(env.L, env.M, pool, seed=seed)
    env.load_configs(rows, pieces)
    env.reset()
    for t in range(W):
        env.step_into(actions[t % S], reward, done)
    torch.cuda.synchronize(dev)
    kc = min(K, 500)
    step = iter(range(W, W + kc))
    ms = timed(torch, dev, lambda: env.step_into(actions[next(step) % S], reward, done), kc)
    st = env.stats()
    return {"value": float(n) / (ms * 1e-3), "unit": "env-steps/s", "ms_per_step": ms, "pool": pool,
            "mean_moves_per_episode": (W + kc) * float(n) / max(st["episodes"], 1),
            "win_rate": st["wins"] / max(st["episodes"], 1)}


@releases_envs
def measure_actor_loop(torch, T, dev, L, M, boards, seed, keep=None):
    """BASELINE configs[4]: boards driven by the policy MLP, obs -> action -> step on the device, three ways."""
    env = keep(T.BatchedTetris(L, M, boards, device=dev, seed=seed, auto_reset=True, assign="hash"))
    rows, pieces = env.synthetic_configs(boards)
    env.load_configs(rows, pieces)
    env.reset()
    out = {"boards": boards, "unit": "env-steps/s", "policy": "MLP 217-128-128-128-128-14, greedy, random init",
           "arithmetic": {"value": "float32 accuracy on the bf16 pipe (split_megakernel)", "fused_mfma_kernel": "bf16 operands, f32 accumulation",
                          "fused_f32_kernel": "float32 operands and accumulation: the reference's nn.Linear width (model/model.py:9-20)",
                          "f32_megakernel": "float32 operands and accumulation (v_mfma_f32_16x16x4_f32), T steps per launch",
                          "split_megakernel": "as fused_split_kernel, T steps per launch",
                          "fused_split_kernel": "float32 accuracy on the bf16 pipe: every weight and activation as three bf16 pieces, "
                                                "six v_mfma_f32_16x16x32_bf16 per product, float32 accumulation (within the float32 "
                                                "kernel's tolerance of a float64 evaluation; not bit-identical to a float32 FMA chain)",
                          "megakernel": "bf16 operands, f32 accumulation, T steps per launch",
                          "torch_linear_layers": "torch bf16 Linear layers (hipBLASLt)",
                          "torch_linear_layers_f32": "torch float32 Linear layers (hipBLASLt)"}}
    for name, use_fused, dtype in (("fused_mfma_kernel", True, torch.bfloat16), ("fused_f32_kernel", True, torch.float32),
                                   ("fused_split_kernel", True, torch.float32),
                                   ("torch_linear_layers", False, torch.bfloat16), ("torch_linear_layers_f32", False, torch.float32)):
        torch.manual_seed(0)
        # two launches per iteration when fused: a graph replay costs more than it saves there
        actor = T.Actor(env, T.PolicyMLP(), dtype=dtype, use_graph=not use_fused, fused=use_fused, split=name == "fused_split_kernel")
        actor.run(20)
        torch.cuda.synchronize(dev)
        ms = timed(torch, dev, actor.step, 300 if dtype is torch.bfloat16 else 60)
        out[name] = {"value": boards / (ms * 1e-3), "ms_per_step": ms}
    # T iterations per launch (tpl_actor_rollout): weights stay in LDS, boards in registers; trajectory written
    torch.manual_seed(0)
    image = T.actor.policy_image(T.PolicyMLP(), dev)
    iters = 50
    env.actor_rollout(image, iters)
    torch.cuda.synchronize(dev)
    ms = timed(torch, dev, lambda: env.actor_rollout(image, iters), 6) / iters
    out["megakernel"] = {"value": boards / (ms * 1e-3), "ms_per_step": ms, "steps_per_launch": iters,
                         "outputs": "per-step action u8 + reward f32 + done u8 written",
                         # the model's FLOPs over the WHOLE iteration (policy, exploration draw, move, trajectory stores): what
                         # the matrix pipe delivers when the per-launch costs (weights into LDS, launch gap) are paid once per T
                         "model_flops_over_whole_step_frac_of_bf16_peak": MODEL_FLOPS_PER_BOARD * boards / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
    # the same loop at the reference's arithmetic width: float32 operands and accumulation, T iterations per launch
    image32m = T.actor.policy_image(T.PolicyMLP(), dev, f32=True)
    iters32 = 10
    env.actor_rollout(image32m, iters32)
    torch.cuda.synchronize(dev)
    ms = timed(torch, dev, lambda: env.actor_rollout(image32m, iters32), 3) / iters32
    out["f32_megakernel"] = {"value": boards / (ms * 1e-3), "ms_per_step": ms, "steps_per_launch": iters32,
                             "outputs": "per-step action u8 + reward f32 + done u8 written"}
    image_sm = T.actor.policy_image(T.PolicyMLP(), dev, f32="split")
    env.actor_rollout(image_sm, iters32)
    torch.cuda.synchronize(dev)
    ms = timed(torch, dev, lambda: env.actor_rollout(image_sm, iters32), 3) / iters32
    out["split_megakernel"] = {"value": boards / (ms * 1e-3), "ms_per_step": ms, "steps_per_launch": iters32,
                               "outputs": "per-step action u8 + reward f32 + done u8 written"}
    # BASELINE configs[4] is "driven by model/model.py policy": a float32 nn.Linear stack (model/model.py:9-20).  The figure of
    # this block is therefore the fastest form at FLOAT32 ACCURACY -- the split megakernel (every product from three bf16 pieces
    # per operand, float32 accumulation; within 2e-5 (1 + max|ref|) of a float64 evaluation, the float32 kernel's own tolerance:
    # tests/test_policy_kernel.py) -- and the bf16 megakernel a named side key with ITS tolerance
    out["value"] = out["split_megakernel"]["value"]
    out["value_is"] = "split_megakernel: float32-accuracy policy (|logit - float64 ref| <= 2e-5 (1 + max|ref|)), T steps per launch"
    out["bf16_megakernel"] = dict(out["megakernel"], tolerance="|logit - float64 ref| <= 2e-2 (1 + max|ref|): bf16 operands, "
                                  "f32 accumulation -- NOT the reference's arithmetic width; actions agree with the float64 "
                                  "policy wherever its margin exceeds twice that")
    # the policy kernel alone against the dense bf16 MFMA peak: USEFUL FLOPs per board -- 2 x (217 x 128 + 3 x 128 x 128 +
    # 128 x 14) = 157,440, the model's own (the kernel issues 159,744: K padded to 224, the head as a 16-row tile)
    # -- over its own duration
    act = torch.empty(boards, dtype=torch.uint8, device=dev)
    for _ in range(5):
        env.policy_act(image, out=act)
    ms = timed(torch, dev, lambda: env.policy_act(image, out=act), 100)
    tflops = MODEL_FLOPS_PER_BOARD * boards / (ms * 1e-3) / 1e12
    out["policy_kernel"] = {"ms": ms, "flops_per_board": MODEL_FLOPS_PER_BOARD,
                            "roofline": {"bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": tflops / MFMA_BF16_PEAK_TFLOPS}}
    image32 = T.actor.policy_image(T.PolicyMLP(), dev, f32=True)
    for _ in range(3):
        env.policy_act(image32, out=act)
    ms = timed(torch, dev, lambda: env.policy_act(image32, out=act), 20)
    tflops = MODEL_FLOPS_PER_BOARD * boards / (ms * 1e-3) / 1e12
    out["policy_kernel_f32"] = {"ms": ms, "roofline": {"bound": "mfma", "achieved": tflops, "peak": MFMA_F32_PEAK_TFLOPS,
                                                       "unit": "TFLOP/s", "frac": tflops / MFMA_F32_PEAK_TFLOPS}}
    image_split = T.actor.policy_image(T.PolicyMLP(), dev, f32="split")
    for _ in range(3):
        env.policy_act(image_split, out=act)
    ms_s = timed(torch, dev, lambda: env.policy_act(image_split, out=act), 20)
    # six bf16 MFMAs per product in the hidden layers and the head, three in layer 1: the FLOPs it ISSUES against the bf16 peak
    issued = 2.0 * (3 * 224 * 128 + 6 * 3 * 128 * 128 + 6 * 128 * 16) * boards / (ms_s * 1e-3) / 1e12
    out["policy_kernel_split"] = {"ms": ms_s, "speedup_over_policy_kernel_f32": ms / ms_s,
                                  "roofline": {"bound": "mfma", "achieved": issued, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": issued / MFMA_BF16_PEAK_TFLOPS,
                                               "note": "bf16 FLOPs issued (3-6 per float32-grade product), not model FLOPs"}}
    # the replenished supply under the POLICY-driven loop (round-4 review: episodes last 2-3 times as long under a policy as
    # under random play, so the reuse factor a LEARNER sees is this one): the bf16 policy kernel + tpl_step per step, with
    # PoolRefresher's defaults beside them (pool-sized batches, the footprint `target_slowdown` picks)
    torch.manual_seed(0)
    actor = T.Actor(env, T.PolicyMLP(), dtype=torch.bfloat16, use_graph=False, fused=True)
    live = measure_live_supply(torch, T, env, None, None, None, seed, min_swaps=2, min_steps=1000, max_steps=12000,
                               step_fn=lambda t: actor.step())
    keep = ("value", "ms_per_step", "ms_per_step_without_refresher", "slowdown", "configurations_per_batch", "generator_waves",
            "pool_swaps", "steps", "configurations_supplied_per_s", "resets_per_s", "pool_reuse_factor")
    out["live_supply"] = dict({k: live[k] for k in keep},
                              loop="bf16 policy kernel + tpl_step per step (two launches), greedy, random-init weights; carved pool",
                              note="the reference's reset() hands every episode a fresh game (game/tetris.py:445-447): factor 1")
    # ... and at the reference's arithmetic width: the split policy kernel (float32 accuracy) + tpl_step per step -- a slower loop
    # finishes fewer episodes a second beside the same generator, so its pool is re-dealt the fewest times
    torch.manual_seed(0)
    actor_s = T.Actor(env, T.PolicyMLP(), dtype=torch.float32, use_graph=False, fused=True, split=True)
    live_s = measure_live_supply(torch, T, env, None, None, None, seed, min_swaps=2, min_steps=300, max_steps=4000,
                                 step_fn=lambda t: actor_s.step())
    out["live_supply"]["float32_accuracy_loop"] = dict({k: live_s[k] for k in keep},
                                                       loop="split policy kernel + tpl_step per step (two launches)")
    env.terminate()
    return out


@releases_envs
def measure_config1(torch, T, dev, seed, chunk, keep=None):
    """BASELINE configs[1]: 65,536 boards, random prescribed initial configurations, L=5, M=20, one GPU.  Side figure
    with its own roofline: a launch this small is bound by the dispatch period of dependent launches, not by HBM."""
    n, L, M, K = 65536, 5, 20, 400
    env = keep(T.BatchedTetris(L, M, n, device=dev, seed=seed, auto_reset=True, assign="hash"))
    rows, pieces = env.synthetic_configs(n)
    env.load_configs(rows, pieces)
    env.reset()
    actions = torch.empty((K, n), dtype=torch.uint8, device=dev)
    for t in range(K):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    rows_of = action_rows(actions)
    for t in range(50):
        env.step_into(rows_of[t], reward, done)
    torch.cuda.synchronize(dev)
    step = iter(range(K))
    ms = timed(torch, dev, lambda: env.step_into(rows_of[next(step)], reward, done), K)
    out = {"workload": f"{n} boards, random initial configs, L={L} M={M}, auto-reset, uniform actions", "unit": "env-steps/s",
           "value": float(n) / (ms * 1e-3), "ms_per_step": ms,
           "roofline": {"bound": "hbm", "achieved": ALGO_BYTES_PER_BOARD_STEP * n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ALGO_BYTES_PER_BOARD_STEP * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "kernel": "step_kernel<action, auto_reset>", "kernel_ms": ms}}
    # the same steps as one replayed HIP graph of 50: at this size the host's few microseconds per call are the limit
    G = 50
    rs = torch.empty((G, n), dtype=torch.float32, device=dev)
    ds = torch.empty((G, n), dtype=torch.uint8, device=dev)
    replay = env.capture_steps(actions[:G], rs, ds)
    replay()
    torch.cuda.synchronize(dev)
    ms_g = timed(torch, dev, replay, 8) / G
    out["graph_replay"] = {"value": float(n) / (ms_g * 1e-3), "ms_per_step": ms_g, "steps_per_graph": G}
    if chunk > 0:
        ms_f = measure_fused_rollout(torch, T, env, actions, 0, K // chunk * chunk, chunk)
        out["fused_rollout"] = {"value": float(n) / (ms_f * 1e-3), "ms_per_step": ms_f, "steps_per_launch": chunk}
    # north_star's step(action) -> (obs, reward, done) for a host-driven loop: the move and the [n,217] float32 observation
    # as ONE launch (tpl_step_observe) against tpl_step followed by tpl_expand_obs
    obs = torch.empty((n, 217), dtype=torch.float32, device=dev)
    step = iter(range(2 * K))

    def two_launches():
        env.step_into(rows_of[next(step) % K], reward, done)
        env.observe(out=obs)
    for t in range(20):
        env.step_observe_into(rows_of[t], reward, done, obs)
    torch.cuda.synchronize(dev)
    ms_two = timed(torch, dev, two_launches, K)
    ms_one = timed(torch, dev, lambda: env.step_observe_into(rows_of[next(step) % K], reward, done, obs), K)
    bytes_moved = (32 + 32 + 1 + 4 + 1 + 217 * 4) * n            # state in and out, action, reward, done, observation
    out["obs_step"] = {"unit": "env-steps/s", "observation": "float32 [n, 217] written every step",
                       "step_then_observe": {"value": float(n) / (ms_two * 1e-3), "ms_per_step": ms_two, "launches": 2},
                       "step_observe": {"value": float(n) / (ms_one * 1e-3), "ms_per_step": ms_one, "launches": 1,
                                        "achieved_GBs": bytes_moved / (ms_one * 1e-3) / 1e9},
                       "speedup": ms_two / ms_one}
    env.terminate()
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# The order the side figures run in, behind bench.py's timed region.  `c` is bench.py's context (a SimpleNamespace): torch, T,
# dev, args, side, rank, world, total, n, L, M, K, W, S, env, actions, reward, done, value (the headline) and the two dicts the
# line is assembled from, `more` (figures on the main boards) and `figures` (figures that own their boards).

def sustained_pass(c):
    """The headline's loop again, `--sustained` launches with an event every 50, right after the timed region; the host's time
    per step_into() call rides along (the calls never wait for the GPU: the queue is deeper than a 50-launch group)."""
    torch, env = c.torch, c.env
    groups = c.args.sustained // 50
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    rows_of = action_rows(c.actions)
    host = 0.0
    evs[0].record()
    for g in range(groups):
        t0 = time.perf_counter()
        for t in range(50):
            env.step_into(rows_of[(g * 50 + t) % c.S], c.reward, c.done)
        host += time.perf_counter() - t0
        evs[g + 1].record()
    torch.cuda.synchronize(c.dev)
    per = [evs[g].elapsed_time(evs[g + 1]) / 50 for g in range(groups)]
    return {"launches": groups * 50, "kernel_ms_mean_this_rank": evs[0].elapsed_time(evs[-1]) / (groups * 50),
            "kernel_ms_median_of_50s": statistics.median(per), "kernel_ms_min_of_50s": min(per), "kernel_ms_max_of_50s": max(per),
            "host_us_per_step_into_call": host / (groups * 50) * 1e6}


def fused_figures(c):
    """This rank's milliseconds per step in each form of the fused rollout (no collective in here)."""
    torch, T, env, chunk = c.torch, c.T, c.env, c.args.chunk
    if c.S < max(2 * chunk, 200):                                # the fused form wants whole chunks of distinct steps
        c.S = max(2 * chunk, 200)
        c.actions = torch.empty((c.S, c.n), dtype=torch.uint8, device=c.dev)
        for t in range(c.S):
            env.synthetic_actions(t, out=c.actions[t])
    whole = c.S // chunk * chunk
    ms = {"f32_u8": measure_fused_rollout(torch, T, env, c.actions, 0, whole, chunk),
          # the same steps recorded as the compact trajectory (one byte per board-step, decoded on the learner's side)
          "compact": measure_fused_rollout(torch, T, env, c.actions, 0, whole, chunk, compact=True),
          # ... and with 200 steps per launch (a launch's fixed part -- the 64 B per board of state in and out, the launch
          # gap -- is some 20 us: a quarter of a 50-step launch's step time, a fifteenth of a 200-step one's)
          "compact_200": measure_fused_rollout(torch, T, env, c.actions, 0, 200, 200, compact=True),
          "f32_u8_200": measure_fused_rollout(torch, T, env, c.actions, 0, 200, 200)}
    # the same kernel under the uniform random policy drawn on the device (no actions staged, no per-step outputs)
    env.rollout_random(100, seed=c.args.seed)
    torch.cuda.synchronize(c.dev)
    ms["device_random"] = timed(torch, c.dev, lambda: env.rollout_random(100, seed=c.args.seed), 4) / 100
    return ms


def fused_line(c, ms):
    total, n, chunk, L, M = c.total, c.n, c.args.chunk, c.L, c.M
    return {"value": float(total) / (ms["f32_u8"] * 1e-3), "unit": "env-steps/s", "steps_per_launch": chunk,
            "ms_per_step": ms["f32_u8"], "outputs": "per-step reward f32 + done u8 written", "kernel": "rollout_kernel<auto_reset>",
            # priced on rank 0's boards over the slowest rank's time, like the headline's roofline
            "roofline": valu_roofline("rollout_f32_u8_50", float(n) / (ms["f32_u8"] * 1e-3), L, M, chunk),
            "compact_trajectory": {"value": float(total) / (ms["compact"] * 1e-3), "ms_per_step": ms["compact"],
                                   "roofline": valu_roofline("rollout_compact_50", float(n) / (ms["compact"] * 1e-3), L, M, chunk),
                                   "steps_per_launch": chunk,
                                   "outputs": "one byte per board-step (rows cleared, how the move ended, reset, frozen), "
                                              "a dword per board every fourth step; tpl_decode_trajectory -> reward f32, done u8"},
            "at_200_steps_per_launch": {"compact_trajectory": float(total) / (ms["compact_200"] * 1e-3),
                                        "reward_f32_and_done_u8": float(total) / (ms["f32_u8_200"] * 1e-3)},
            "device_random_policy": {"value": float(total) / (ms["device_random"] * 1e-3), "ms_per_step": ms["device_random"],
                                     "roofline": valu_roofline("rollout_random_100", float(n) / (ms["device_random"] * 1e-3), L, M, 100),
                                     "steps_per_launch": 100, "outputs": "reward sums and episode counts only"}}


def live_figures(c):
    torch, T, env = c.torch, c.T, c.env
    live = measure_live_supply(torch, T, env, c.actions, c.reward, c.done, c.args.seed)
    # the same run by the generator's footprint: how many persistent waves share its queue, and confined to 32 compute units by
    # a CU-masked stream (which turns out to be the expensive way: profiles/r03_live_supply)
    live["generator"] = ("PoolRefresher defaults: a plain side stream, pool-sized batches (one configuration per board), "
                         "the footprint its target_slowdown = 1.13 picks from the measured table")
    keep = ("ms_per_step", "slowdown", "pool_swaps", "steps", "configurations_per_batch", "generator_waves",
            "configurations_supplied_per_s", "pool_reuse_factor")
    # ... and what the reference's supply semantics cost: the pool's reuse HELD at 1.  The loop waits for the generator, which gets
    # the whole chip (4096 waves); batches of FOUR configurations per board, because a swap has to wait M + 1 = 41 steps and a
    # board-sized pool is dealt 2.5 times over in those (6 % of the boards finish at every step under random play)
    held = measure_live_supply(torch, T, env, c.actions, c.reward, c.done, c.args.seed, count=4 * c.n, waves=4096, hold=1.0,
                               min_swaps=3, min_steps=64, max_steps=20000)
    live["reuse_held_at_1"] = {k: held[k] for k in ("value", "ms_per_step", "pool_swaps", "steps", "configurations_per_batch",
                                                    "configurations_supplied_per_s", "resets_per_s", "pool_reuse_factor")}
    live["reuse_held_at_1"]["note"] = ("PoolRefresher.hold_reuse(1.0): every pool dealt about once, as the reference's queue deals every game once "
                                       "(game/tetris.py:445-447) -- the step loop then runs at the generator's rate")
    live["by_generator_footprint"] = [
        dict(generator=name, **{k: v for k, v in measure_live_supply(torch, T, env, c.actions, c.reward, c.done, c.args.seed, **kw).items() if k in keep})
        for name, kw in (("256 waves, batches of 65,536 (the default through round 4)", dict(waves=256, count=65536)),
                         ("1024 waves, pool-sized batches", dict(waves=1024)), ("64 waves, batches of 65,536", dict(waves=64, count=65536)))]
    return live


def own_board_figures(c):
    """The side figures that build boards of their own (never part of `value`), each under SideFigures' guard."""
    torch, T, dev, args, side, L, M, total = c.torch, c.T, c.dev, c.args, c.side, c.L, c.M, c.total
    if c.world == 1:
        if args.shard_ranks > 1 and args.chunk > 0:
            # the scaling ceiling first: it is the figure the first 8-GPU run will be read against
            runs = {}
            for ranks in sorted({2, 4, args.shard_ranks}):
                runs[ranks] = side.run(f"shard_run_x{ranks}", lambda r=ranks: measure_shard_run(torch, T, dev, L, M, args.seed, total, r, args.chunk))
            c.figures["shard_run"] = runs[args.shard_ranks]
            fused = c.more["fused_rollout"]
            c.figures["scaling_model"] = scaling_model(runs, total, c.value, c.ms_per_step * 1e3,
                                                       fused["value"] if side.ok(fused) else None, args.chunk)
        if args.actor_boards > 0:
            c.figures["actor_loop"] = side.run("actor_loop", lambda: measure_actor_loop(torch, T, dev, L, M, args.actor_boards, args.seed))
        if args.carved_pool > 0:
            c.figures["config_supply"] = side.run("config_supply", lambda: measure_config_supply(torch, T, dev, L, M, args.seed))
        if not args.no_config1:
            c.figures["config1_run"] = side.run("config1_run", lambda: measure_config1(torch, T, dev, args.seed, args.chunk))
    elif not args.no_weak_job:
        got = side.run("weak_scaling_job", lambda: measure_weak_job(torch, T, dev, c.rank, c.world, L, M, total, args.seed, c.K))
        if side.ok(got):          # every rank came through: only now are their numbers combined
            got = weak_job_line(side.max_over_ranks(got["ms"]), got["steps"], total, got["global_boards"])
        c.figures["weak_scaling_job"] = got


def after_the_timed_region(c, numpy_leg):
    """Every side figure, in order: the figures on the main boards (sustained pass, fused rollout, carved pool, live supply), the
    out-of-cache run, the figures that own their boards, the C leg of the CPU baseline.  The timed region is the first GPU work
    of the process (behind the matrix kernels of the actor loop the same twenty steps read 0.2 us a step slower)."""
    args, side, more = c.args, c.side, c.more
    if args.sustained >= 100:
        sustained = side.run("sustained", lambda: sustained_pass(c))
        if side.ok(sustained):
            sustained["kernel_ms_mean"] = side.max_over_ranks(sustained.pop("kernel_ms_mean_this_rank"))
            sustained["host_us_per_step_into_call"] = side.max_over_ranks(sustained["host_us_per_step_into_call"])
            sustained["value"] = float(c.total) / (sustained["kernel_ms_mean"] * 1e-3)
            sustained["frac"] = ALGO_BYTES_PER_BOARD_STEP * c.n / (sustained["kernel_ms_median_of_50s"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        more["sustained"] = sustained
    if args.chunk > 0:
        fused = side.run("fused_rollout", lambda: fused_figures(c))
        if side.ok(fused):
            fused = fused_line(c, {k: side.max_over_ranks(v) for k, v in sorted(fused.items())})
        more["fused_rollout"] = fused
    if args.carved_pool > 0 and c.world == 1:
        more["carved_pool_run"] = side.run("carved_pool_run", lambda: measure_carved_pool(c.torch, c.T, c.env, c.actions, c.reward, c.done,
                                                                                           c.W, c.K, args.carved_pool, args.seed))
        more["live_supply_run"] = side.run("live_supply_run", lambda: live_figures(c))
    try:
        c.env.terminate()
    except Exception:                 # noqa: BLE001 -- a side figure may have left the handle in a state it cannot be destroyed from
        pass
    c.actions = None
    if c.world == 1 and not args.no_out_of_cache:
        more["out_of_cache"] = side.run("out_of_cache", lambda: measure_out_of_cache(c.torch, c.T, c.dev, c.L, c.M, args.seed))
    own_board_figures(c)
    if c.rank == 0 and not args.no_cpu_baseline:
        # rank 0's host cores, after every collective of the job (the other ranks are on their way out)
        try:
            more["cpu_baseline"] = cpu_baseline(c.L, c.M, args.seed, numpy_leg)
        except Exception as e:        # noqa: BLE001
            more["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:400]}
