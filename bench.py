#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Tetris-piclim board step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one lockstep pass of the hot path (tpl_step: one Tetris.move per board, auto-reset of finished boards from the device
pool) over one batch of synthetic actions.  Workload = BASELINE.json configs[2] at N = 1 and configs[3] at N > 1: 1,048,576 boards
IN TOTAL, L=10, M=40, synthetic boards / 7-bag piece lists / uniform actions, all resident in HBM before the timed
region.  N > 1 shards that one batch by global board index (1,048,576 / N boards per GPU: fixed total work, "scaling": "strong"),
one process per GPU, the same pool on every rank, no data-path collective; one RCCL all-reduce of the episodic-return counters
closes the timed region.  The N-GPU job is the one-GPU job sharded: same episodes, same mean return.
`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (a child process, spawned before
anything touches the GPU) and exits with the child's status.

Launch mode (`config.launch_mode`): every step is ONE step_kernel launch over the rank's boards in either mode.  "eager" = one
tpl_step() call per step (N = 1 always: the N = 1 point of a scaling run is the plain bench line).  "graph" = the same launches
captured <= 50 at a time and replayed ("persistent stream, no host sync per step, HIP graphs for the rollout loop"),
chosen for N > 1 when a rank holds fewer than 2^19 boards: there the host's ~5 us per tpl_step() call are as long as the device's
period, and a host-bound rank would set the node's figure.  `timing.host_call_us` is the measured cost of a tpl_step() call,
`timing.host_issue_us_per_step` what the timed loop's issue cost the host per step in the mode used.

Timing: `value` and `ms_per_step` come from a HIP-event pair on the launch stream around EXACTLY K steps (max over
ranks), the region bracketed by barrier + synchronize on both sides; the all-reduce closes the region and is reported apart
(`timing.collective_ms`).  The last of the W warm-up steps is enqueued after the synchronize, directly ahead of the first timed
launch ("excluding one warm-up"): a launch into a queue that has run dry pays the GPU's wake-up, 20-160 us, which is
not a property of a step (`timing.launch_after_synchronize_ms`).

Output.  stdout carries ONE JSON line, the contract's: the headline keys first, `roofline` (the step kernel against HBM at the
canonical 96 B/board-step of SURVEY 8(d); `frac` = this run, `frac_hbm_resident` = the same kernel where nothing fits the 256 MiB
Infinity Cache), `cpu_baseline` (the C oracle on this box's host cores; `limited_by` says what bounds them), `scaling_model` (N = 1:
the periods of a 2-, 4- and 8-GPU run's shards measured on this GPU, and the value and efficiency they imply) and one number per
side figure.  The full record -- every side figure with its own roofline, bench_side.py -- goes to `--detail` (default
gpurun_out/bench_detail_n<N>.json) and, as one line prefixed BENCH_DETAIL, to stderr; the headline alone goes there once already
right behind the timed region (BENCH_HEADLINE), before any side figure runs.  Side figures never enter `value`; each
runs under a guard (bench_side.SideFigures): an exception costs its own key, figures past --side-budget are skipped, and if one
hangs past --side-timeout the line is printed with what there is and the process exits with status 3.  A failure INSIDE the timed
region is not guarded: the run ends non-zero with no line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_side import (ALGO_BYTES_PER_BOARD_STEP, HBM_ACHIEVABLE_GBS, HBM_PEAK_GBS, SideFigures, WATCHDOG_STATUS,  # noqa: E402,F401
                        action_rows, launch_mode_for, library_digest, numpy_port_leg, valu_roofline)
import bench_side as B  # noqa: E402

GRAPH_STEPS = 50                        # steps per captured graph in "graph" mode (the last one holds the remainder)


def graph_plan(W, K, S, g_max=GRAPH_STEPS):
    """The K timed steps of "graph" mode as chunks of <= g_max steps: [[action row of each step], ...].  Step t (W <= t < W + K)
    plays action row t % S -- exactly the row the eager loop hands to tpl_step() for that step."""
    plan, t = [], W
    while t < W + K:
        g = min(g_max, K, W + K - t)
        plan.append([(t + i) % S for i in range(g)])
        t += g
    return plan


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a CHILD process -- this process has not
    imported torch or touched the GPU, and it never execs -- and hand back the child's exit status."""
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20,
                    help="boards of the job, IN TOTAL: sharded by global board index over the --gpus ranks (BASELINE configs[3])")
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--M", type=int, default=40)
    ap.add_argument("--pool", type=int, default=0, help="pool entries, the same pool on every rank (default: one per board of the job)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--launch-mode", choices=("auto", "eager", "graph"), default="auto",
                    help="auto: replayed graphs of <= 50 steps for N > 1 below 2^19 boards per GPU, eager tpl_step() calls otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--actor-boards", type=int, default=262144, help="boards of the config-5 actor-loop side measurement (0 = skip)")
    ap.add_argument("--carved-pool", type=int, default=65536, help="size of the carved pool of the realism run (0 = skip)")
    ap.add_argument("--chunk", type=int, default=50, help="steps per launch of the fused-rollout side measurement (0 = skip)")
    ap.add_argument("--sustained", type=int, default=2000, help="launches of the sustained pass after the timed region (0 = skip)")
    ap.add_argument("--no-config1", action="store_true", help="skip the BASELINE configs[1] side line")
    ap.add_argument("--shard-ranks", type=int, default=8,
                    help="N = 1 only: measure the shards of a 2-, 4- and this-many-GPU run on this GPU (`scaling_model`, `shard_run`; 0 = skip)")
    ap.add_argument("--no-weak-job", action="store_true", help="N > 1: skip the weak-scaling side figure (--boards per GPU)")
    ap.add_argument("--no-out-of-cache", action="store_true", help="skip the 2^23-board side run (N = 1 only)")
    ap.add_argument("--no-side-figures", action="store_true", help="the headline, its roofline and the CPU baseline only")
    ap.add_argument("--side-budget", type=float, default=120.0,
                    help="seconds of side figures after which the remaining ones are skipped (the headline is never skipped)")
    ap.add_argument("--side-timeout", type=float, default=300.0,
                    help="seconds after which a side figure that is STILL running is abandoned: the line is printed, exit status 3")
    ap.add_argument("--detail", default=None, help="where the full record goes (default gpurun_out/bench_detail_n<N>.json; '-' = nowhere)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus and --steps must be positive, --warmup non-negative")
    if args.no_side_figures:
        args.actor_boards = args.carved_pool = args.chunk = args.sustained = args.shard_ranks = 0
        args.no_config1 = args.no_weak_job = args.no_out_of_cache = True
    return args


def traffic_of(n):
    """HBM bytes per launch by the PMC counters, scaled to `n` boards.  The counters cannot be read from inside this process
    (rocprofv3 collects them in passes of their own), so the figure comes from the committed profile of this same command
    (profiles/traffic.json <- tools/profile_step.sh + tools/update_traffic.py) and says so; `stale` = that profile was taken on
    other kernel sources than the library loaded now."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        scale = n / float(tj.get("boards_per_launch_measured", 1 << 20))      # the kernel's traffic is linear in the board count
        dec = tj.get("decomposed_estimate_bytes")
        return {"traffic": tj["hbm_bytes_per_launch"] * scale, "traffic_decomposed": dec * scale if dec else None,
                "traffic_stale": tj.get("source_digest") != library_digest(),
                "traffic_source": f"NOT measured in this run: rocprofv3 --pmc passes of this command, {tj.get('source')} (commit "
                                  f"{tj.get('commit')}, source digest {str(tj.get('source_digest'))[:12]}), scaled to {n} boards"}
    except Exception:                 # noqa: BLE001
        return {"traffic": None, "traffic_decomposed": None, "traffic_stale": None, "traffic_source": None}


def compact(d):
    """The ONE line of stdout from the full record `d`: the contract's keys, scalars only inside `config` / `roofline` /
    `cpu_baseline` (a reader that truncates nested objects and long strings keeps every number), one number per side figure."""
    def pick(src, keys, always=()):
        return {k: src.get(k) for k in keys if k in src or k in always} if isinstance(src, dict) else src
    ok = SideFigures.ok
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data")}
    out["config"] = pick(d["config"], ("workload", "launch_mode", "global_boards", "boards_per_gpu", "L", "M", "parallelism"))
    if d["config"].get("graph_capture_failed"):
        out["config"]["graph_capture_failed"] = str(d["config"]["graph_capture_failed"])[:128]
    out["roofline"] = pick(d["roofline"], ("bound", "achieved", "peak", "unit", "frac", "frac_hbm_resident", "hbm_resident_working_set_bytes",
                                           "traffic", "traffic_stale", "frac_traffic", "kernel", "kernel_ms", "kernel_ms_median",
                                           "frac_median", "boards_per_launch", "algorithmic_bytes_per_launch"),
                           always=("frac_hbm_resident", "traffic", "traffic_stale"))
    if "cpu_baseline" in d:
        out["cpu_baseline"] = pick(d["cpu_baseline"], ("value", "unit", "cores", "cores_available", "cores_used", "limited_by", "cpu_model",
                                                       "kind", "sample", "error"))
    out["timing"] = pick(d["timing"], ("per_rank_ms_per_step", "host_call_us", "host_issue_us_per_step", "collective_ms", "wall_ms_per_step",
                                       "launch_after_synchronize_ms"))
    sm = d.get("scaling_model")
    if ok(sm):
        # twelve numbers: period, implied value and efficiency of a 2-, 4- and 8-GPU run of this job at one launch per step, the 8-GPU
        # figure with `chunk` steps per launch, the weak-scaling counterpart (boards per GPU and launch modes: the detail record)
        fused = sm[next(k for k in sm if k.startswith("fused_"))]
        top = max(sm["per_launch"], key=lambda k: int(k[1:]), default=None)
        out["scaling_model"] = {"basis": "shards of this 2^20-board job measured on THIS one GPU; value = boards / period, efficiency = value_xN / (N x value_x1)",
                                "per_launch": {k: pick(v, ("us_per_step", "value", "efficiency")) for k, v in sm["per_launch"].items()},
                                f"fused_{top}": pick(fused.get(top, {}), ("value", "efficiency")),
                                "weak_x8_value": sm["weak"]["value_x8"]}
    side = {}
    fr = d.get("fused_rollout")
    if ok(fr):
        side["fused_rollout"] = dict({"value": fr["value"], "steps_per_launch": fr["steps_per_launch"]},
                                     **pick(fr["roofline"] or {}, ("bound", "frac", "frac_hw", "frac_of_lane_slots")))
    for key, sub in (("shard_run", "tpl_step"), ("shard_run", "capture_steps"), ("shard_run", "tpl_rollout")):
        if ok(d.get(key)):
            side.setdefault(key, {"boards": d[key]["boards"]})[sub + "_us_per_step"] = d[key][sub]["us_per_step"]
    for key in ("weak_scaling_job", "carved_pool_run", "config1_run", "actor_loop"):
        if ok(d.get(key)):
            side[key] = d[key].get("value")
    if ok(d.get("live_supply_run")):
        side["live_supply_run"] = pick(d["live_supply_run"], ("slowdown", "configurations_supplied_per_s", "pool_reuse_factor"))
        held = d["live_supply_run"].get("reuse_held_at_1")
        if isinstance(held, dict):
            side["live_supply_run"].update(value_with_reuse_held_at_1=held.get("value"), reuse_factor_when_held=held.get("pool_reuse_factor"))
    if ok(d.get("config_supply")):
        cd = d["config_supply"]["carve_device"]
        side["config_supply_carve_device"] = dict({"value": cd["value"]}, **pick(cd["roofline"] or {}, ("frac", "frac_hw", "frac_of_lane_slots")))
    out["side"] = side
    out["side_figures"] = pick(d["side_figures"], ("failed", "skipped", "total_seconds", "abandoned"))
    for k in ("mean_episodic_return", "episodes", "ranks_seen", "backend", "detail"):
        out[k] = d.get(k)
    return out


def only_the_line_on_stdout():
    """From here on file descriptor 1 is stderr's: whatever a library prints on stdout while this process runs -- gloo's
    "[Gloo] Rank 0 is connected to 1 peer ranks" at every group creation, a runtime's warnings -- lands on stderr, and the ONE
    JSON line goes out through the returned writer on the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def write(text):
        data = (text + "\n").encode()
        while data:
            data = data[os.write(real, data):]
    return write


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    write_line = only_the_line_on_stdout()
    world, rank, local = (int(os.environ.get(k, "0" if k != "WORLD_SIZE" else "1")) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                 f"torch.distributed.run with --nproc-per-node equal to --gpus")

    # the NumPy leg of the CPU baseline runs in child processes, so it goes first: nothing here has touched the GPU yet
    # (rank 0 of any world size: the other ranks wait for it in init_process_group)
    numpy_leg = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            import tetris_piclim as T0
            numpy_leg = numpy_port_leg(args.L, args.M, args.seed, T0._lib.cpu_budget())
        except Exception as e:        # noqa: BLE001 -- a reported baseline, not the measurement
            numpy_leg = {"error": f"{type(e).__name__}: {e}"[:400]}

    import torch
    import torch.distributed as dist
    import tetris_piclim as T

    # TPL_BENCH_BACKEND=gloo and TPL_BENCH_ONE_GPU=1 exist only to rehearse the multi-rank path on a one-GPU box;
    # TPL_BENCH_FORCE_DIST=1 sends a ONE-rank run through the process group as an N-rank run (real RCCL on a one-GPU box)
    backend = os.environ.get("TPL_BENCH_BACKEND", "nccl")
    if os.environ.get("TPL_BENCH_ONE_GPU") == "1":
        local = 0
    ctl = ctl_note = None
    dist_on = world > 1 or os.environ.get("TPL_BENCH_FORCE_DIST") == "1"
    if dist_on:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # the ranks' agreements around the side figures travel over a gloo group of HOST tensors (SideFigures); if that group
        # cannot be made, they fall back to the job's own group and device tensors -- the headline must not depend on it
        try:
            ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=max(600.0, 2 * args.side_timeout)))
            dist.all_reduce(torch.zeros(1, dtype=torch.float64), group=ctl)
            ctl_note = "gloo group of host tensors"
        except Exception as e:        # noqa: BLE001
            ctl, ctl_note = None, f"the job's own group, device tensors (no gloo control group: {type(e).__name__}: {e})"[:300]
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    def barrier():
        if dist_on:
            dist.barrier()

    def every_rank(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if not dist_on:
            return [float(x)]
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return [float(g.item()) for g in got]

    side = SideFigures(world, rank, dist, ctl, budget_s=args.side_budget, inject=os.environ.get("TPL_BENCH_INJECT_FAILURE", ""),
                       gather=every_rank if (dist_on and ctl is None) else None, distributed=dist_on)

    # BASELINE configs[2] (N = 1) / configs[3] (N > 1): ONE batch of `total` boards, sharded by global board index; the same
    # pool on every rank (entry e = synthetic configuration e), actions and assignment keyed by the GLOBAL board index: the
    # N-GPU job is the one-GPU job sharded -- the same episodes and the same mean return for every N (tests/test_multi_rank_gpu.py)
    total, L, M, K, W = args.boards, args.L, args.M, args.steps, args.warmup
    shard = T.sharding.strong_shard(rank, world, total)
    n, pool = shard.boards, args.pool or total
    mode = launch_mode_for(world, T.sharding.strong_shard(0, world, total).boards, args.launch_mode)   # by the largest shard: the same on every rank
    env = T.BatchedTetris(L, M, n, device=dev, seed=args.seed, global_offset=shard.global_offset, auto_reset=True, assign="hash")
    rows, pieces = env.synthetic_configs(pool, first=0)
    env.load_configs(rows, pieces)
    del rows, pieces
    env.reset()
    # synthetic actions for every step, staged in HBM before timing (at most 4096 distinct steps; a longer run cycles).  No more
    # rows than the W + K steps need: every megabyte written here pushes boards and pool out of the Infinity Cache
    S = min(max(W + K, 1), 4096)
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    rows_of = action_rows(actions)              # the row views, made once: at a shard's size the host's call rate is the period

    # ---- the timed region: W untimed + exactly K timed steps, barrier + synchronize on both sides.  NOTHING in here is
    # guarded: a failure ends the run with a non-zero status -- an unmeasured headline must not look measured.
    for t in range(max(W - 1, 0)):
        env.step_into(rows_of[t % S], reward, done)
    replays, capture_failed = [], None
    if mode == "graph":
        # the K timed steps as captured graphs of <= GRAPH_STEPS step_kernel launches, step t on action row t as in eager mode;
        # captured HERE (a capture runs nothing: snapshot, K enqueues into the graph, restore), replayed inside the region.
        # Should a capture fail (eight RCCL ranks capturing at once is the one thing no one-GPU box can rehearse) the rank says so
        # and EVERY rank falls back to eager calls: a slower, host-dependent headline instead of none.
        saved = env.snapshot()
        try:
            if os.environ.get("TPL_BENCH_INJECT_FAILURE") == "graph_capture":
                raise RuntimeError("failure injected into the graph capture (TPL_BENCH_INJECT_FAILURE)")
            g_max = min(GRAPH_STEPS, K)
            rs = torch.empty((g_max, n), dtype=torch.float32, device=dev)
            ds = torch.empty((g_max, n), dtype=torch.uint8, device=dev)
            for chunk in graph_plan(W, K, S):
                replays.append(env.capture_steps([rows_of[r] for r in chunk], rs[:len(chunk)], ds[:len(chunk)]))
                replays[-1].prepare()
        except Exception as e:            # noqa: BLE001
            capture_failed = f"{type(e).__name__}: {e}"[:300]
            torch.cuda.synchronize(dev)
            env.restore(saved)
        del saved
        if any(every_rank(1.0 if capture_failed else 0.0)):
            capture_failed = capture_failed or "the capture failed on another rank"
            mode, replays = "eager", []
    T.sharding.mean_episodic_return(env.stats_tensor(), env.reward_params)   # load the reduction kernels / RCCL rings
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    ev_w, ev_a, ev_c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    ev_w.record()                                                 # same stream as the kernel launches
    if W >= 1:
        env.step_into(rows_of[(W - 1) % S], reward, done)        # warm-up step W of W: takes the idle queue's wake-up
    ev_a.record()
    h0 = time.perf_counter()
    if mode == "graph":
        for replay in replays:
            replay()
    else:
        for t in range(W, W + K):
            env.step_into(rows_of[t % S], reward, done)
    h1 = time.perf_counter()
    ev_c.record()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    # the one collective of the job: RCCL all-reduce (sum) of [return sum, episodes] over xGMI
    mean_return, episodes = T.sharding.mean_episodic_return(env.stats_tensor(), env.reward_params)
    t2 = time.perf_counter()
    barrier()
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    host_issue_us = (h1 - h0) / K * 1e6
    host_call_us = host_issue_us
    if mode == "graph":                                           # what a tpl_step() call costs the host, measured apart
        torch.cuda.synchronize(dev)
        h0 = time.perf_counter()
        for t in range(64):
            env.step_into(rows_of[t % S], reward, done)
        host_call_us = (time.perf_counter() - h0) / 64 * 1e6
        torch.cuda.synchronize(dev)
    per_rank_ms = every_rank(ev_a.elapsed_time(ev_c) / K)
    steady_ms = max(per_rank_ms)                                  # max over ranks
    wake_ms = max(every_rank(ev_w.elapsed_time(ev_a))) if W >= 1 else None
    wall_ms, collective_ms = max(every_rank((t3 - t0) * 1e3)), max(every_rank((t2 - t1) * 1e3))
    host_call_us, host_issue_us = max(every_rank(host_call_us)), max(every_rank(host_issue_us))
    value = float(total) / (steady_ms * 1e-3)

    per_rank_roofline = []                                        # each rank's kernel priced on the boards of ITS shard
    for r, ms_r in enumerate(per_rank_ms):
        nb = T.sharding.strong_shard(r, world, total).boards
        gbs = ALGO_BYTES_PER_BOARD_STEP * nb / (ms_r * 1e-3) / 1e9
        per_rank_roofline.append({"rank": r, "boards": nb, "kernel_ms": ms_r, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS})

    # ---- the headline is complete from here on; everything below adds side keys to the record and none of it can lose it
    c = types.SimpleNamespace(torch=torch, T=T, dev=dev, args=args, side=side, rank=rank, world=world, total=total, n=n, L=L, M=M, K=K,
                              W=W, S=S, env=env, actions=actions, reward=reward, done=done, value=value, ms_per_step=steady_ms,
                              more={"sustained": None, "fused_rollout": None, "carved_pool_run": None, "live_supply_run": None,
                                    "out_of_cache": None, "cpu_baseline": None},
                              figures={"actor_loop": None, "config_supply": None, "config1_run": None, "weak_scaling_job": None,
                                       "shard_run": None, "scaling_model": None})
    detail_path = args.detail or os.path.join(ROOT, "gpurun_out", f"bench_detail_n{world}.json")

    def record(abandoned=None):
        """The full record, from what has been measured so far."""
        more, figures = c.more, c.figures
        achieved = ALGO_BYTES_PER_BOARD_STEP * n / (steady_ms * 1e-3) / 1e9
        tr = traffic_of(n)
        sustained = more["sustained"] if side.ok(more["sustained"]) else {}
        ooc = more["out_of_cache"] if side.ok(more["out_of_cache"]) else {}
        what = f"{total} boards " + (f"sharded over {world} GPUs" if world > 1 else "on 1 GPU")
        out = {
            "metric": "env-steps/sec (whole node) at 1M parallel 20x10 boards", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": steady_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{what}, L={L} M={M}, random configs + actions, auto-reset, 1 launch/step ({mode}), "
                                   f"BASELINE configs[{3 if world > 1 else 2}]",
                       "workload_detail": f"{total} boards in total" + (f", batch-sharded over {world} GPUs ({n} on rank 0)" if world > 1 else " on 1 GPU")
                                          + f", random initial configs, L={L} M={M}, uniform random actions, auto-reset from a {pool}-entry "
                                          "device pool (the same pool on every rank), one tpl_step launch per step",
                       "launch_mode": mode, "graph_capture_failed": capture_failed,
                       "launch_mode_is": ("one tpl_step() call per step" if mode == "eager" else
                                                               f"the same step_kernel launches, captured <= {GRAPH_STEPS} at a time and replayed "
                                                               f"({len(replays)} graph(s) for the {K} timed steps)"),
                       "global_boards": total, "boards_per_gpu": n, "L": L, "M": M, "parallelism": f"batch-shard x{world}"},
            "timing": {"clock": "HIP events on the launch stream around the K steps, max over ranks",
                       "warmup_placement": f"{max(W - 1, 0)} warm-up step(s) before the synchronize, {min(W, 1)} after it directly "
                                           "ahead of the first timed launch (it takes the wake-up of the idle queue)",
                       "per_rank_ms_per_step": per_rank_ms, "wall_ms_per_step": wall_ms / K, "collective_ms": collective_ms,
                       "launch_after_synchronize_ms": wake_ms, "host_call_us": host_call_us, "host_issue_us_per_step": host_issue_us,
                       "host_call_is": "what one tpl_step() call costs the host thread (max over ranks)"
                                       + ("" if mode == "eager" else ", measured over 64 eager calls behind the region"),
                       "host_bound": bool(host_issue_us > 0.9 * steady_ms * 1e3)},
            "ranks_seen": dist.get_world_size() if dist_on else 1, "backend": dist.get_backend() if dist_on else None,
            "per_rank_roofline": per_rank_roofline,
            "roofline": dict({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                              # the same kernel where NOTHING fits the 256 MiB Infinity Cache (2^23 boards + a 2^21-entry pool = 512 MiB):
                              # `frac` above is helped by the cache (32 MB of state + 26 MB of action rows sit inside it)
                              "frac_hbm_resident": ooc.get("frac"), "hbm_resident_working_set_bytes": ooc.get("resident_bytes"),
                              # what the memory system DELIVERED (counter bytes / period) against the ~6.3 TB/s a streaming kernel reaches
                              "traffic_rate_over_achievable_hbm": (tr["traffic"] / (steady_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS) if tr["traffic"] else None},
                             **tr,
                             frac_traffic=(tr["traffic"] / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr["traffic"] else None,
                             frac_traffic_decomposed=(tr["traffic_decomposed"] / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr["traffic_decomposed"] else None,
                             out_of_cache=more["out_of_cache"], kernel="step_kernel<action, auto_reset>", kernel_ms=steady_ms, boards_per_launch=n,
                             priced_on="rank 0's shard of the job over the slowest rank's launch period",
                             kernel_ms_source="launch period over the K timed steps (HIP events on the launch stream)",
                             kernel_ms_median=sustained.get("kernel_ms_median_of_50s"), frac_median=sustained.get("frac"),
                             sustained=more["sustained"], algorithmic_bytes_per_launch=ALGO_BYTES_PER_BOARD_STEP * n,
                             node={"achieved": ALGO_BYTES_PER_BOARD_STEP * total / (steady_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * world,
                                   "unit": "GB/s", "frac": ALGO_BYTES_PER_BOARD_STEP * total / (steady_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)}),
            # (the record ends with what a reader of its tail wants first: the fused form, the shards, the scaling model)
            "carved_pool_run": more["carved_pool_run"], "live_supply_run": more["live_supply_run"], "config1_run": figures["config1_run"],
            "config_supply": figures["config_supply"], "actor_loop": figures["actor_loop"], "weak_scaling_job": figures["weak_scaling_job"],
            "fused_rollout": more["fused_rollout"], "shard_run": figures["shard_run"], "scaling_model": figures["scaling_model"],
            "mean_episodic_return": mean_return if episodes else None, "episodes": episodes,
            "side_figures": dict(side.summary(), agreements_over=ctl_note,
                                 guard="each side figure runs under a guard (an exception becomes {\"error\": ...} under its key; at N > 1 the "
                                       "ranks agree over a gloo group before and after each); the timed region is not guarded"),
            "detail": None if detail_path == "-" else os.path.relpath(detail_path, ROOT),
        }
        if abandoned is not None:
            out["side_figures"]["abandoned"] = (f"'{abandoned}' was still running {args.side_timeout:.0f} s into the side figures "
                                                f"(--side-timeout): line printed by the watchdog, exit status {WATCHDOG_STATUS}")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = more["cpu_baseline"]
        return out

    def emit(abandoned=None):
        if rank != 0:
            return
        full = record(abandoned)
        if detail_path != "-":
            try:
                os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
                with open(detail_path, "w") as f:
                    json.dump(full, f, indent=1)
            except OSError as e:
                full["detail"] = f"not written: {e}"
        print("BENCH_DETAIL " + json.dumps(full), file=sys.stderr, flush=True)
        write_line(json.dumps(compact(full)))

    if rank == 0:
        # the headline on record BEFORE any side figure runs (stderr and the detail file; stdout keeps its one line for the end):
        # should a side figure take the process down instead of raising, the measurement is not lost with it
        early = record()
        print("BENCH_HEADLINE " + json.dumps({k: early[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config")}),
              file=sys.stderr, flush=True)
        if detail_path != "-":
            try:
                os.makedirs(os.path.dirname(detail_path) or ".", exist_ok=True)
                with open(detail_path, "w") as f:
                    json.dump(early, f, indent=1)
            except OSError:
                pass
    side.watchdog(args.side_timeout, emit)
    B.after_the_timed_region(c, numpy_leg)
    side.disarm()
    emit()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
