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


def test_watchdog_prints_the_line_and_ends_the_process_with_status_zero(tmp_path):
    """A side figure that never returns: after --side-timeout the line goes out with what there is and the process exits 0."""
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
    assert res.returncode == 0 and time.time() - t0 < 30
    assert json.loads(res.stdout.strip().splitlines()[-1]) == {"value": 42.0, "abandoned": "live_supply_run"}
    assert "not reached" not in res.stdout
