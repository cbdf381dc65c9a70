#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Tetris-piclim board step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one lockstep pass of the hot path (tpl_step: one Tetris.move per board, auto-reset of finished
boards from the device pool) over one batch of synthetic actions.  Workload = BASELINE.json configs[2]:
1,048,576 boards per GPU, L=10, M=40, synthetic boards / 7-bag piece lists / uniform actions,
all resident in HBM before the timed region.  N > 1 shards boards by global index, one process per GPU, no
data-path collective; one RCCL all-reduce of the episodic-return counters closes the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the step kernel against HBM with the canonical
96 B/board-step of SURVEY 8(d); `cpu_baseline` is the CPU oracle (a scalar C port of the reference's move)
timed on this box's host cores over a bounded sample of the same workload.  Side figures that never enter
`value`: `fused_rollout` (tpl_rollout), `carved_pool_run` (the step loop on carved configurations),
`actor_loop` (BASELINE configs[4]: 262,144 boards driven by the 217-128-128-128-128-14 policy).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_BOARD_STEP = 96          # 46 B read + 49 B write, rounded
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0          # dense bf16 (MI355X_MICROARCH.md)


def cpu_baseline(L, M, seed):
    """The oracle's loop (game/performance_test.py:13-17 shape: move, reset when finished) on the host cores."""
    from oracle import oracle as O
    import tetris_piclim as T
    cores = T._lib.cpu_budget()                                  # affinity mask capped by the cgroup CPU quota
    boards, steps = 262144, 40
    O.bench_run(seed, 4096, L, M, 8, cores)                      # warm the thread pool / page in
    done, sec = O.bench_run(seed, boards, L, M, steps, cores)
    # bounded sample: grow the step count until the timed part is a few seconds of wall time on all cores
    for _ in range(3):
        if sec >= 3.0 or steps >= 20000:
            break
        steps = int(max(40, min(20000, 5.0 * (done / sec) / boards)))
        done, sec = O.bench_run(seed, boards, L, M, steps, cores)
    return {"value": done / sec, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{boards} boards x {steps} lockstep steps, L={L} M={M}, auto-reset, {cores} threads, {sec:.1f}s",
            "note": "the C port of the step; the Python reference itself (it cannot travel to this box) ran the same "
                    "workload at 61-68 k moves/s per core in the build container, the port at 12-14 M: 175-235x per core "
                    "(tests/golden/time_reference.py)"}


def timed(torch, dev, fn, reps):
    """Average milliseconds of fn() over `reps` calls, by HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps


def measure_fused_rollout(torch, T, env, actions, first, K, chunk):
    """tpl_rollout over the same pre-staged actions, `chunk` steps per launch, writing the same
    per-step reward/done outputs as the step loop."""
    n, dev = env.num_envs, env.device
    rs = torch.empty((chunk, n), dtype=torch.float32, device=dev)
    ds = torch.empty((chunk, n), dtype=torch.uint8, device=dev)
    launches = K // chunk

    def run():
        for c in range(launches):
            a = actions[first + c * chunk: first + (c + 1) * chunk]
            T._lib.check(env._lib.tpl_rollout(env._h, ctypes.c_void_p(a.data_ptr()), a.stride(0), chunk,
                                              ctypes.c_void_p(rs.data_ptr()), ctypes.c_void_p(ds.data_ptr()), None, None,
                                              env._stream()))
    run()
    torch.cuda.synchronize(dev)
    return timed(torch, dev, run, 1) / (launches * chunk)


def measure_strong_scaling(torch, T, dev, rank, world, L, M, total, seed, K, chunk, barrier, max_over_ranks):
    """BASELINE configs[3]: ONE batch of `total` boards sharded by global board index over the ranks (fixed total
    work).  Side figure only; `value` stays the weak-scaling job.  Reported both ways: one launch per step (where the
    per-launch dispatch gap dominates a shard of 131,072 boards) and `chunk` steps per launch."""
    shard = T.sharding.strong_shard(rank, world, total)
    env = T.BatchedTetris(L, M, shard.boards, device=dev, seed=seed, global_offset=shard.global_offset,
                          auto_reset=True, assign="hash")
    rows, pieces = env.synthetic_configs(shard.boards, first=shard.global_offset)
    env.load_configs(rows, pieces)
    env.reset()
    S = max(chunk, min(K, 500) // chunk * chunk)
    actions = torch.empty((S, shard.boards), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(shard.boards, dtype=torch.float32, device=dev)
    done = torch.empty(shard.boards, dtype=torch.uint8, device=dev)
    for t in range(20):
        env.step_into(actions[t % S], reward, done)
    torch.cuda.synchronize(dev)
    barrier()
    step = iter(range(S))
    ms_step = max_over_ranks(timed(torch, dev, lambda: env.step_into(actions[next(step)], reward, done), S))
    barrier()
    ms_fused = max_over_ranks(measure_fused_rollout(torch, T, env, actions, 0, S, chunk))
    env.terminate()
    return {"global_boards": total, "boards_per_gpu": shard.boards, "unit": "env-steps/s",
            "one_launch_per_step": {"value": float(total) / (ms_step * 1e-3), "ms_per_step": ms_step},
            "fused_rollout": {"value": float(total) / (ms_fused * 1e-3), "ms_per_step": ms_fused, "steps_per_launch": chunk}}


def measure_config_supply(torch, T, dev, L, M, seed):
    """rates of the prescribed-configuration suppliers (side figures).  Carving on the device
    (one configuration per lane) and on the host cores produce the same configurations; the forward generator +
    solver is host code."""
    import numpy as np
    env = T.BatchedTetris(L, M, 64, device=dev, seed=seed)
    count = 1 << 18
    env.carved_configs(4096)                                     # load the kernel
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    rows, _ = env.carved_configs(count)                           # returns after the status check (host sync)
    dt_dev = time.perf_counter() - t0
    host_count = 1 << 14
    t0 = time.perf_counter()
    hrows, _ = T.generate_configs(L, M, host_count, seed=seed)
    dt_host = time.perf_counter() - t0
    same = bool(np.array_equal(rows[:host_count].cpu().numpy().view(np.uint16), hrows))
    env.terminate()
    games = 4000
    t0 = time.perf_counter()
    fw = T.forward_generate(5, 20, np.arange(games))
    dt_fw = time.perf_counter() - t0
    return {"unit": "configurations/s", "L": L, "M": M,
            "carve_device": {"value": count / dt_dev, "count": count},
            "carve_host": {"value": host_count / dt_host, "count": host_count, "threads": T._lib.cpu_budget(),
                           "equal_to_device_output": same},
            "forward_generator_solver_host": {"value": games / dt_fw, "games": games, "L": 5, "M": 20,
                                              "winnable_fraction": float(fw["winnable"].mean())}}


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


def measure_actor_loop(torch, T, dev, L, M, boards, seed):
    """BASELINE configs[4]: boards driven by the policy MLP, obs -> action -> step on the device, three ways."""
    env = T.BatchedTetris(L, M, boards, device=dev, seed=seed, auto_reset=True, assign="hash")
    rows, pieces = env.synthetic_configs(boards)
    env.load_configs(rows, pieces)
    env.reset()
    out = {"boards": boards, "unit": "env-steps/s",
           "policy": "MLP 217-128-128-128-128-14, bf16 operands, greedy, random init"}
    for name, use_fused in (("fused_mfma_kernel", True), ("torch_linear_layers", False)):
        torch.manual_seed(0)
        # two launches per iteration when fused: a graph replay costs more than it saves there
        actor = T.Actor(env, T.PolicyMLP(), dtype=torch.bfloat16, use_graph=not use_fused, fused=use_fused)
        actor.run(20)
        torch.cuda.synchronize(dev)
        ms = timed(torch, dev, actor.step, 300)
        out[name] = {"value": boards / (ms * 1e-3), "ms_per_step": ms}
    # T iterations per launch (tpl_actor_rollout): weights stay in LDS, boards in registers; trajectory written
    torch.manual_seed(0)
    image = T.actor.policy_image(T.PolicyMLP(), dev)
    iters = 50
    env.actor_rollout(image, iters)
    torch.cuda.synchronize(dev)
    ms = timed(torch, dev, lambda: env.actor_rollout(image, iters), 6) / iters
    out["megakernel"] = {"value": boards / (ms * 1e-3), "ms_per_step": ms, "steps_per_launch": iters,
                         "outputs": "per-step action u8 + reward f32 + done u8 written"}
    out["value"] = out["megakernel"]["value"]
    # the policy kernel alone against the dense bf16 MFMA peak: FLOPs it issues per board (K padded to 224, the
    # 14-row head run as one 16-row tile) over its own duration
    act = torch.empty(boards, dtype=torch.uint8, device=dev)
    for _ in range(5):
        env.policy_act(image, out=act)
    ms = timed(torch, dev, lambda: env.policy_act(image, out=act), 100)
    tflops = 2.0 * (224 * 128 + 3 * 128 * 128 + 128 * 16) * boards / (ms * 1e-3) / 1e12
    out["policy_kernel"] = {"ms": ms, "roofline": {"bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_PEAK_TFLOPS,
                                                   "unit": "TFLOP/s", "frac": tflops / MFMA_BF16_PEAK_TFLOPS}}
    env.terminate()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--L", type=int, default=10)
    ap.add_argument("--M", type=int, default=40)
    ap.add_argument("--pool", type=int, default=0, help="pool size per GPU (default: one config per board)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--actor-boards", type=int, default=262144, help="boards of the config-5 actor-loop side measurement (0 = skip)")
    ap.add_argument("--carved-pool", type=int, default=65536, help="size of the carved pool of the realism run (0 = skip)")
    ap.add_argument("--chunk", type=int, default=50, help="steps per launch of the fused-rollout side measurement (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tetris_piclim as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TPL_BENCH_BACKEND=gloo and TPL_BENCH_ONE_GPU=1 exist only to rehearse the multi-rank path on a one-GPU box
    backend = os.environ.get("TPL_BENCH_BACKEND", "nccl")
    if os.environ.get("TPL_BENCH_ONE_GPU") == "1":
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n, L, M, K, W = args.boards, args.L, args.M, args.steps, args.warmup
    pool = args.pool or n
    shard = T.sharding.weak_shard(rank, world, n)               # batch-index sharding: contiguous blocks
    env = T.BatchedTetris(L, M, n, device=dev, seed=args.seed, global_offset=shard.global_offset, auto_reset=True,
                          assign="hash")
    rows, pieces = env.synthetic_configs(pool, first=shard.global_offset)
    env.load_configs(rows, pieces)
    del rows, pieces
    env.reset()
    # synthetic actions for every step, staged in HBM before timing (at most 4096 distinct steps = 4 GiB at 2^20
    # boards; a longer run cycles through them)
    S = min(W + K, 4096)
    actions = torch.empty((S, n), dtype=torch.uint8, device=dev)
    for t in range(S):
        env.synthetic_actions(t, out=actions[t])
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)

    # ---- the timed region: W untimed + exactly K timed steps, barrier + synchronize on both sides
    for t in range(W):
        env.step_into(actions[t % S], reward, done)
    T.sharding.mean_episodic_return(env.stats_tensor(), env.reward_params)   # load the reduction kernels / RCCL rings
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for t in range(W, W + K):
        env.step_into(actions[t % S], reward, done)
    ev1.record()                                                  # same stream as the kernel launches
    # the one collective of the job: RCCL all-reduce (sum) of [return sum, episodes] over xGMI
    mean_return, episodes = T.sharding.mean_episodic_return(env.stats_tensor(), env.reward_params)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = max_over_ranks(time.perf_counter() - t0)
    kernel_ms = ev0.elapsed_time(ev1) / K                         # average launch-to-launch duration of the step kernel

    # ---- side figures (not part of `value`)
    fused = None
    if args.chunk > 0 and K >= args.chunk and W + K <= S:
        barrier()
        ms = max_over_ranks(measure_fused_rollout(torch, T, env, actions, W, K, args.chunk))
        fused = {"value": float(n) * world / (ms * 1e-3), "unit": "env-steps/s", "steps_per_launch": args.chunk,
                 "ms_per_step": ms, "outputs": "per-step reward f32 + done u8 written", "kernel": "rollout_kernel<auto_reset>"}
    carved = None
    if args.carved_pool > 0 and world == 1:
        carved = measure_carved_pool(torch, T, env, actions, reward, done, W, K, args.carved_pool, args.seed)
    env.terminate()
    del actions
    strong = None
    if world > 1 and args.chunk > 0:
        strong = measure_strong_scaling(torch, T, dev, rank, world, L, M, n, args.seed, K, args.chunk, barrier, max_over_ranks)
    actor = None
    if args.actor_boards > 0 and world == 1:
        actor = measure_actor_loop(torch, T, dev, L, M, args.actor_boards, args.seed)

    supply = None
    if world == 1 and args.carved_pool > 0:
        supply = measure_config_supply(torch, T, dev, L, M, args.seed)

    if rank == 0:
        achieved = ALGO_BYTES_PER_BOARD_STEP * n / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                # measured at 1,048,576 boards per launch; the kernel's traffic is linear in the board count
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch") * (n / float(1 << 20))
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node) at 1M parallel 20x10 boards",
            "value": float(n) * world * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{n} boards per GPU x {world} GPU, random initial configs, L={L} M={M}, "
                                   f"uniform random actions, auto-reset from a {pool}-entry device pool "
                                   "(BASELINE configs[2])",
                       "boards_per_gpu": n, "L": L, "M": M, "parallelism": f"batch-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "step_kernel<action, auto_reset>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_BOARD_STEP * n},
            "fused_rollout": fused,
            "strong_scaling": strong,
            "carved_pool_run": carved,
            "config_supply": supply,
            "actor_loop": actor,
            "mean_episodic_return": mean_return if episodes else None,
            "episodes": episodes,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, M, args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
