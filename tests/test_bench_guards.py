"""bench.py's side figures cannot take the headline with them (round-4 review, "make the N = 8 run unable to come back
empty"): `bench.SideFigures` on the CPU -- one process, then eight gloo ranks (the world size of BASELINE configs[3]) with a
failure on ONE rank: every rank must come out of the figure with an error entry, none may be left waiting in a collective,
and the figures after it still run.  The GPU side of the same property is tests/test_multi_rank_gpu.py."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_a_failing_side_figure_becomes_an_error_entry_and_the_next_one_runs():
    bench = _bench()
    side = bench.SideFigures(inject="carved_pool_run, live_supply_run@3")
    assert side.run("fused_rollout", lambda: {"ms": 1.5}) == {"ms": 1.5}
    got = side.run("carved_pool_run", lambda: {"never": "reached"})                     # injected by name
    assert "injected" in got["error"] and got["error"].startswith("RuntimeError") and "failed_ranks" not in got
    assert side.run("live_supply_run", lambda: {"rank": 0}) == {"rank": 0}              # injected for rank 3 only: not this rank
    boom = side.run("out_of_cache", lambda: (_ for _ in ()).throw(MemoryError("HIP out of memory. Tried to allocate 512 MiB")))
    assert boom["error"].startswith("MemoryError: HIP out of memory")
    assert side.ok({"ms": 1.0}) and not side.ok(boom) and not side.ok(None) and not side.ok({"skipped": "..."})
    summary = side.summary()
    assert summary["failed"] == ["carved_pool_run", "out_of_cache"] and summary["skipped"] == []
    assert list(summary["seconds"]) == ["fused_rollout", "carved_pool_run", "live_supply_run", "out_of_cache"]


def test_figures_past_the_budget_are_skipped_not_started():
    bench = _bench()
    now = [0.0]
    side = bench.SideFigures(budget_s=10.0, clock=lambda: now[0])
    assert side.run("first", lambda: now.__setitem__(0, 11.0) or {"ok": 1}) == {"ok": 1}   # takes 11 s: allowed to finish
    started = []
    got = side.run("second", lambda: started.append(1))
    assert "skipped" in got and not started and side.summary()["skipped"] == ["second"]


def test_agreements_fall_back_to_a_gather_of_the_jobs_own_group():
    """When no gloo control group can be made, main() hands SideFigures the device-tensor all-gather of the job's own process
    group; here a stand-in that reports a second rank as failed (and as over budget)."""
    bench = _bench()
    other = iter([0.0, 1.0, 1.0])                          # the other rank: within budget, then failed; then out of budget
    side = bench.SideFigures(world=2, rank=0, gather=lambda v: [v, next(other)], distributed=True)
    got = side.run("fused_rollout", lambda: {"ms": 1.0})
    assert got["failed_ranks"] == [1] and "rank(s) [1] failed" in got["error"]
    assert "skipped" in side.run("carved_pool_run", lambda: {"ms": 1.0})
    side = bench.SideFigures(world=2, rank=0, gather=lambda v: [v, 0.0], distributed=True)
    assert side.run("fused_rollout", lambda: {"ms": 1.0}) == {"ms": 1.0} and side.max_over_ranks(3.0) == 3.0


def _rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    ctl = dist.new_group(backend="gloo")
    side = bench.SideFigures(world, rank, dist, ctl, budget_s=600.0, inject="weak_scaling_job@5")
    record = {}
    # (1) every rank comes through: only then are the numbers combined
    got = side.run("sustained", lambda: {"ms": 1.0 + rank})
    record["sustained"] = side.max_over_ranks(got["ms"]) if side.ok(got) else got
    # (2) rank 5 fails inside the figure (injected), rank 2 fails with an error of its own: nobody hangs, everybody knows
    def fused():
        if rank == 2:
            raise ValueError("rank 2's own trouble")
        time.sleep(0.05 * rank)                        # the ranks leave the figure at different times
        return {"ms": 2.0}
    record["fused_rollout"] = side.run("fused_rollout", fused)
    record["weak_scaling_job"] = side.run("weak_scaling_job", lambda: {"ms": 3.0})
    # (3) the figure after the failures runs on all ranks again
    got = side.run("after", lambda: {"ms": 10.0 - rank})
    record["after"] = side.max_over_ranks(got["ms"]) if side.ok(got) else got
    record["summary"] = side.summary()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(record, f)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_eight_ranks_agree_about_a_figure_that_failed_on_one_of_them(tmp_path):
    world = 8
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    records = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    for r, rec in enumerate(records):
        assert rec["sustained"] == 8.0 and rec["after"] == 10.0                          # max over ranks, on every rank
        assert rec["fused_rollout"]["failed_ranks"] == [2] and rec["weak_scaling_job"]["failed_ranks"] == [5]
        assert ("rank 2's own trouble" in rec["fused_rollout"]["error"]) == (r == 2)     # the failing rank keeps its own message
        assert ("injected" in rec["weak_scaling_job"]["error"]) == (r == 5)
        assert rec["summary"]["failed"] == ["fused_rollout", "weak_scaling_job"]


def test_watchdog_prints_the_line_and_ends_the_process_with_a_status_that_shows_the_hang(tmp_path):
    """A side figure that never returns: after --side-timeout the line goes out with what there is and the process exits with
    WATCHDOG_STATUS (3), not 0 -- the measured headline is on stdout, and whoever started the run sees that something hung
    (round-5 advisor finding: status 0 reported a kernel hung on the GPU as a clean run)."""
    script = tmp_path / "hang.py"
    script.write_text(
        "import sys, time, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "side = bench.SideFigures()\n"
        "side.watchdog(0.5, lambda running: print(json.dumps({'value': 42.0, 'abandoned': running}), flush=True))\n"
        "side.run('live_supply_run', lambda: time.sleep(60))\n"
        "print('not reached')\n")
    t0 = time.time()
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=50)
    assert res.returncode == _bench().WATCHDOG_STATUS == 3 and time.time() - t0 < 30
    assert json.loads(res.stdout.strip().splitlines()[-1]) == {"value": 42.0, "abandoned": "live_supply_run"}
    assert "not reached" not in res.stdout


def test_issue_bound_rooflines_are_priced_twice_and_in_order():
    """Round-5 review: the `valu-issue` rooflines carry the builder's price (c = 3.3 cycles per instruction, from the measured
    issue costs of the kernel's mix) AND the hardware's (2 cycles per wave64 instruction), and the share of lane-slots that did
    work: frac_of_lane_slots <= frac_hw <= frac < 1 for every committed form; a run at another (L, M, steps per launch) than
    the profiled one gets no price at all (round-5 advisor finding)."""
    bench = _bench()
    forms = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))["forms"]
    assert set(forms) >= {"rollout_f32_u8_50", "rollout_compact_50", "rollout_random_100", "rollout_shard_131072", "carve_1048576"}
    for name, f in forms.items():
        units_per_s = f["units_per_launch"] / (f["duration_ns_median"] * 1e-9)           # the profiled run's own rate
        r = bench.valu_roofline(name, units_per_s, 10, 40, f["units_per_launch"] // f["grid"] if f["unit"] == "board-step" else None)
        assert 0 < r["frac_of_lane_slots"] <= r["frac_hw"] <= r["frac"] < 1, (name, r)
        assert r["frac"] == pytest.approx(f["frac"], rel=1e-6)
        assert r["frac_hw"] == pytest.approx(r["achieved"] / (1024 * 2.4 / 2.0), rel=1e-9) and r["peak_hw"] == pytest.approx(1228.8)
        assert r["frac_of_lane_slots"] == pytest.approx(r["frac_hw"] * f["lanes_active_per_valu_instruction"] / 64, rel=1e-9)
        # the counter-derived clock only where the launch is long enough for it to mean something, and never above the peak clock
        held = r["in_the_profiled_run"].get("clock_GHz_held")
        assert held is None or (held <= 2.4 and f["duration_ns_median"] >= 150000), name
    for other in (dict(L=5, M=20, steps_per_launch=50), dict(L=10, M=40, steps_per_launch=200)):
        r = bench.valu_roofline("rollout_f32_u8_50", 2.4e11, **other)
        assert r["frac"] is None and "not_comparable" in r and "achieved" not in r
    assert bench.valu_roofline("carve_1048576", 9e7, 15, 40)["frac"] is None


def test_scaling_model_is_global_boards_over_the_shard_period_in_the_mode_the_bench_would_use():
    bench = _bench()
    import bench_side
    total = 1 << 20

    def run(ranks, eager_us, graph_us, fused_us):
        n = total // ranks
        return {"boards": n, "ranks": ranks, "tpl_step": {"us_per_step": eager_us, "host_call_us": 5.0},
                "capture_steps": {"us_per_step": graph_us}, "tpl_rollout": {"us_per_step": fused_us}}
    runs = {2: run(2, 9.0, 9.2, 2.3), 4: run(4, 6.5, 6.0, 1.4), 8: run(8, 5.6, 5.1, 1.04)}
    sm = bench_side.scaling_model(runs, total, value_x1=total / 15.68e-6, us_x1=15.68, fused_value_x1=2.4e11, chunk=50)
    pl = sm["per_launch"]
    assert pl["x2"]["launch_mode"] == "eager" and pl["x4"]["launch_mode"] == pl["x8"]["launch_mode"] == "graph"       # 2^19 boards per GPU: eager
    assert pl["x2"]["us_per_step"] == 9.0 and pl["x4"]["us_per_step"] == 6.0 and pl["x8"]["us_per_step"] == 5.1
    assert pl["x8"]["value"] == pytest.approx(total / 5.1e-6) and pl["x8"]["boards_per_gpu"] == 131072
    assert pl["x8"]["efficiency"] == pytest.approx(15.68 / 5.1 / 8)
    assert sm["fused_50_steps_per_launch"]["x8"]["value"] == pytest.approx(total / 1.04e-6)
    assert sm["weak"]["value_x8"] == pytest.approx(8 * total / 15.68e-6) and sm["weak"]["efficiency"] == 1.0
    # a shard whose measurement failed is left out, the others stay
    runs[4] = {"error": "RuntimeError: ..."}
    assert set(bench_side.scaling_model(runs, total, 6.7e10, 15.68, None, 50)["per_launch"]) == {"x2", "x8"}
    assert bench.launch_mode_for(1, 1 << 17) == "eager" and bench.launch_mode_for(8, 1 << 17) == "graph"
    assert bench.launch_mode_for(2, 1 << 19) == "eager" and bench.launch_mode_for(8, 1 << 17, "eager") == "eager"


def test_compact_line_keeps_the_contract_keys_and_fits_a_truncating_reader():
    """stdout's ONE line is made from the full record by bench.compact(): the contract's keys, `roofline` / `config` /
    `cpu_baseline` with scalars only and no string past 128 characters (the driver's record keeps that much), the traffic's
    staleness flag, the HBM-resident fraction beside the cache-assisted one, and well under the 9 KB of stdout a driver keeps."""
    bench = _bench()
    tr = bench.traffic_of(1 << 20)
    assert tr["traffic"] > 5e7 and tr["traffic_stale"] in (True, False)
    full = {"metric": "env-steps/sec (whole node) at 1M parallel 20x10 boards", "value": 6.7e10, "unit": "env-steps/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 0.01568, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "w" * 120, "workload_detail": "x" * 400, "launch_mode": "eager", "launch_mode_is": "...", "global_boards": 1 << 20,
                       "boards_per_gpu": 1 << 20, "L": 10, "M": 40, "parallelism": "batch-shard x1"},
            "timing": {"clock": "...", "per_rank_ms_per_step": [0.01568], "host_call_us": 5.0, "host_issue_us_per_step": 5.0, "collective_ms": 0.1,
                       "wall_ms_per_step": 0.02, "launch_after_synchronize_ms": 0.06, "note": "y" * 300},
            "roofline": dict({"bound": "hbm", "achieved": 6400.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.80, "frac_hbm_resident": 0.77,
                              "hbm_resident_working_set_bytes": 1 << 29, "kernel": "step_kernel<action, auto_reset>",
                              "kernel_ms": 0.01568, "kernel_ms_median": 0.0152, "frac_median": 0.83, "boards_per_launch": 1 << 20,
                              "algorithmic_bytes_per_launch": 96 << 20, "sustained": {"launches": 2000}, "out_of_cache": {"frac": 0.77}}, **tr),
            "cpu_baseline": {"value": 4e8, "unit": "env-steps/s", "cores": 16, "cores_available": 128, "cores_used": 16,
                             "limited_by": "cgroup CPU quota (cpu.max = 16 CPUs; the affinity mask allows 128)", "cpu_model": "AMD EPYC 9575F",
                             "kind": "port", "sample": "...", "numpy_port": {"value": 2.4e6}},
            "scaling_model": {"job": "...", "x1": {}, "per_launch": {"x8": {"boards_per_gpu": 131072, "launch_mode": "graph", "us_per_step": 5.1,
                                                                             "value": 2.05e11, "efficiency": 0.38, "host_call_us": 5.0}},
                              "fused_50_steps_per_launch": {"x8": {"us_per_step": 1.04, "value": 1.0e12, "efficiency": 0.52}},
                              "weak": {"value_x8": 5.3e11, "efficiency": 1.0, "note": "..."}, "reading": "z" * 300},
            "fused_rollout": {"value": 2.4e11, "steps_per_launch": 50, "roofline": {"bound": "valu-issue", "frac": 0.54, "frac_hw": 0.33,
                                                                                      "frac_of_lane_slots": 0.22, "source": "s" * 300}},
            "shard_run": {"boards": 131072, "tpl_step": {"us_per_step": 5.6}, "capture_steps": {"us_per_step": 5.1}, "tpl_rollout": {"us_per_step": 1.04}},
            "actor_loop": {"value": 1.67e9, "policy_kernel": {"ms": 0.03}}, "config_supply": {"carve_device": {"value": 8.9e7, "roofline": None}},
            "carved_pool_run": {"error": "RuntimeError: ..."}, "live_supply_run": None, "config1_run": {"value": 1.2e10}, "weak_scaling_job": None,
            "side_figures": {"seconds": {"a": 1.0}, "failed": ["carved_pool_run"], "skipped": [], "total_seconds": 40.0, "guard": "g" * 300},
            "mean_episodic_return": -0.9, "episodes": 123, "ranks_seen": 1, "backend": None, "detail": "gpurun_out/bench_detail_n1.json"}
    line = bench.compact(full)
    text = json.dumps(line)
    assert len(text) < 5000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert list(line)[:12] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                               "dtype", "data"]
    for block in ("config", "roofline", "cpu_baseline"):
        for k, v in line[block].items():
            assert not isinstance(v, (dict, list)), (block, k)
            assert not isinstance(v, str) or len(v) <= 128, (block, k)
    assert line["config"]["launch_mode"] == "eager"
    assert line["roofline"]["frac"] == 0.80 and line["roofline"]["frac_hbm_resident"] == 0.77 and line["roofline"]["traffic_stale"] in (True, False)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample", "cores_available", "cores_used", "limited_by"} <= set(line["cpu_baseline"])
    assert line["timing"]["host_call_us"] == 5.0 and "error" not in line["cpu_baseline"]

    def fractions(node, path=""):                                   # no fraction anywhere in the line exceeds 1
        for k, v in (node.items() if isinstance(node, dict) else []):
            if isinstance(v, dict):
                yield from fractions(v, path + k + ".")
            elif isinstance(v, (int, float)) and not isinstance(v, bool) and (k.startswith("frac") or k == "efficiency"):
                yield path + k, v
    assert all(0 <= v <= 1 for _, v in fractions(line)), list(fractions(line))
    sm = line["scaling_model"]
    assert sm["per_launch"]["x8"] == {"us_per_step": 5.1, "value": 2.05e11, "efficiency": 0.38}
    assert sm["fused_x8"] == {"value": 1.0e12, "efficiency": 0.52} and sm["weak_x8_value"] == 5.3e11

    def numbers(node):
        return sum(numbers(v) for v in node.values()) if isinstance(node, dict) else int(isinstance(node, (int, float)) and not isinstance(node, bool))
    full["scaling_model"]["per_launch"].update({k: dict(full["scaling_model"]["per_launch"]["x8"]) for k in ("x2", "x4")})
    full["scaling_model"]["fused_50_steps_per_launch"].update({k: dict(full["scaling_model"]["fused_50_steps_per_launch"]["x8"]) for k in ("x2", "x4")})
    assert numbers(bench.compact(full)["scaling_model"]) == 12               # the review's bound: one key, at most twelve numbers
    assert line["side"]["fused_rollout"]["frac_hw"] == 0.33 and line["side"]["actor_loop"] == 1.67e9 and "carved_pool_run" not in line["side"]
    assert line["side"]["shard_run"] == {"boards": 131072, "tpl_step_us_per_step": 5.6, "capture_steps_us_per_step": 5.1, "tpl_rollout_us_per_step": 1.04}
    assert line["side_figures"]["failed"] == ["carved_pool_run"]


def test_cpu_limits_name_what_bounds_the_baseline_threads(monkeypatch):
    import bench_side
    lim = bench_side.cpu_limits()
    assert lim["cores_available"] == os.cpu_count() and 1 <= lim["cores_used"] <= lim["cores_in_affinity_mask"] <= lim["cores_available"]
    assert isinstance(lim["limited_by"], str) and lim["limited_by"]
    monkeypatch.setenv("TPL_CPU_BUDGET", "3")
    lim = bench_side.cpu_limits()
    assert lim["cores_used"] == 3 and lim["limited_by"] == "TPL_CPU_BUDGET=3"


def test_committed_counter_profiles_were_taken_on_these_kernel_sources():
    """`roofline.traffic` and the per-unit instruction counts of the `valu-issue` rooflines come from committed rocprofv3 counter
    passes (PMC counters cannot be read from inside the bench process): they must be OF THIS LIBRARY -- the digest of the kernel
    sources stamped into profiles/traffic.json and into every form of profiles/valu_issue.json is the digest of csrc/ + include/ as
    they stand (a kernel change without a new `tools/profile_step.sh` / `tools/profile_valu.sh` run fails here; at run time the
    line's `roofline.traffic_stale` / `count_stale` say the same about the loaded library)."""
    import tetris_piclim as T
    digest = T._lib._source_digest()
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert traffic["source_digest"] == digest, "profiles/traffic.json was measured on other kernel sources: run tools/profile_step.sh + tools/update_traffic.py"
    for name, f in json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))["forms"].items():
        assert f["stamp"]["source_digest"] == digest and f["stamp"]["library_built_from_these_sources"], name


def test_graph_mode_plays_every_step_on_the_action_row_the_eager_loop_would():
    """bench.graph_plan: the K timed steps as chunks of at most 50 captured launches; step t on action row t % S, whatever K, W and
    the number of staged rows S are (the driver's K = 20 is one graph; 70 = 50 + 20; past 4096 staged rows the rows wrap)."""
    bench = _bench()
    assert bench.graph_plan(5, 20, 25) == [list(range(5, 25))]
    assert [len(c) for c in bench.graph_plan(5, 70, 75)] == [50, 20]
    assert [len(c) for c in bench.graph_plan(0, 1, 1)] == [1] and bench.graph_plan(0, 1, 1) == [[0]]
    for W, K, S in ((5, 20, 25), (50, 2000, 2050), (50, 5000, 4096), (0, 7, 7), (3, 149, 100), (10, 50, 60)):
        plan = bench.graph_plan(W, K, S)
        flat = [r for chunk in plan for r in chunk]
        assert flat == [t % S for t in range(W, W + K)] and all(1 <= len(c) <= 50 for c in plan)
        assert all(len(c) == 50 for c in plan[:-1])
