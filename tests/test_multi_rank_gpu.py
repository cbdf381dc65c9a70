"""The N > 1 path with the HIP library doing the stepping: two ranks (gloo process group, both on the test box's one
GPU) shard a batch by global board index, step their shards through the C ABI, and all-reduce the return counters
through the product's sharding module.  The result must equal ONE process stepping the whole batch on the GPU, and the
CPU oracle.  Also: `python bench.py --gpus 2` starts its two ranks itself."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "dist"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["strong", "weak"])
def test_two_hip_ranks_equal_one_process_and_the_oracle(tmp_path, oracle, mode):
    import hip_shard_worker as W
    import tetris_piclim as T
    out = str(tmp_path / "dist.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist", "hip_shard_worker.py"), mode, out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    got = np.load(out)
    assert int(got["ranks"]) == 2 and str(got["lib"]) == T.LIB_PATH
    n_total = W.BOARDS if mode == "strong" else (W.BOARDS // 2) * 2
    stats, state = W.run_shard(T.sharding.strong_shard(0, 1, n_total))
    mean, episodes = T.sharding.mean_episodic_return(stats, W.REWARD)
    assert int(got["episodes"]) == episodes and episodes > n_total and float(got["mean"]) == mean
    for k in ("rows", "lines", "moves", "state", "cur", "nxt", "reward_sum"):
        assert np.array_equal(got[k], state[k]), k
    # and the oracle, one process over the whole batch
    cpu = oracle.Env(n_total, W.L, W.M, 0, W.SEED)
    cpu.set_pool(oracle.synth_boards(W.SEED, 0, W.POOL, W.L), oracle.synth_pieces(W.SEED, 0, W.POOL, W.M))
    cpu.set_options(auto_reset=True, assign_mode=0, per_line=W.REWARD[0], win=W.REWARD[1], lose=W.REWARD[2])
    cpu.reset()
    rsum = np.zeros(n_total, np.float64)
    for t in range(W.STEPS):
        r, _ = cpu.step(oracle.synth_actions(W.SEED, 0, n_total, t))
        rsum += r
    want = cpu.get_state()
    for k in ("rows", "lines", "moves", "state", "cur", "nxt"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["reward_sum"], rsum) and cpu.stats()["episodes"] == episodes


def _bench(gpus, extra=(), boards=65536, inject=None, base=("--sustained", "100", "--actor-boards", "0", "--carved-pool", "0",
                                                             "--no-config1", "--no-out-of-cache"), env_more=None, rehearsal=True, steps=20):
    """One run of bench.py; returns the FULL record (--detail) after checking what stdout carries: exactly ONE JSON line, the
    compact form of that record, with the contract's keys."""
    import tempfile
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TPL_BENCH_BACKEND", "TPL_BENCH_ONE_GPU", "TPL_BENCH_FORCE_DIST"):
        env.pop(k, None)
    if rehearsal:                                 # several ranks on this box's one GPU, over gloo
        env.update(TPL_BENCH_ONE_GPU="1", TPL_BENCH_BACKEND="gloo")
    if inject:
        env["TPL_BENCH_INJECT_FAILURE"] = inject
    env.update(env_more or {})
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", "5", "--boards", str(boards),
               "--detail", detail, *base, *extra]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        full = json.load(open(detail))
    lines = res.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), lines   # ONE line on stdout and nothing else (gloo's chatter goes to stderr)
    line = json.loads(lines[0])
    assert len(lines[0]) < 6000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["workload"] == full["config"]["workload"] and len(line["config"]["workload"]) <= 128
    assert line["config"]["launch_mode"] == full["config"]["launch_mode"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "frac_hbm_resident"} <= set(line["roofline"])
    assert line["timing"]["host_call_us"] == full["timing"]["host_call_us"] > 0
    assert line["episodes"] == full["episodes"] and line["mean_episodic_return"] == full["mean_episodic_return"]
    assert [l for l in res.stderr.splitlines() if l.startswith("BENCH_DETAIL {")]
    if full.get("scaling_model"):                 # the N = 1 line carries the scaling ceiling, the same numbers as the record
        for k, leg in full["scaling_model"]["per_launch"].items():
            assert line["scaling_model"]["per_launch"][k] == {f: leg[f] for f in ("us_per_step", "value", "efficiency")}
    else:
        assert "scaling_model" not in line
    return full


@pytest.mark.gpu
def test_bench_gpus_2_headline_is_the_one_gpu_job_sharded():
    """`python bench.py --gpus 2` outside torchrun (the form the driver uses for N = 1): the parent starts two ranks.
    On a one-GPU box they share cuda:0 and use gloo (TPL_BENCH_ONE_GPU / TPL_BENCH_BACKEND exist for this rehearsal).
    The HEADLINE of an N > 1 run is BASELINE configs[3] -- `--boards` boards IN TOTAL, sharded -- so it must be the
    one-GPU job sharded: the same episodes and the same mean episodic return as `--gpus 1` of the same command, with
    "scaling": "strong".  The job that grows with the node is a side key."""
    out = _bench(2)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo"      # what dist.get_backend() says
    assert out["steps"] == 20 and out["warmup"] == 5 and len(out["timing"]["per_rank_ms_per_step"]) == 2
    assert out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["global_boards"] == 65536 and cfg["boards_per_gpu"] == 32768 and "65536 boards sharded over 2 GPUs" in cfg["workload"]
    assert "configs[3]" in cfg["workload"] and "65536 boards in total" in cfg["workload_detail"]
    # below 2^19 boards per GPU an N > 1 run replays captured graphs (one step_kernel launch per step all the same): the
    # host's cost per step of the timed loop is then a fraction of a tpl_step() call's
    assert cfg["launch_mode"] == "graph" and "(graph)" in cfg["workload"] and "1 graph(s) for the 20 timed steps" in cfg["launch_mode_is"]
    assert 0 < out["timing"]["host_issue_us_per_step"] < 4 * out["timing"]["host_call_us"]      # (2 us against 5 on a quiet box; no tight bound on a clock)
    assert [(r["rank"], r["boards"]) for r in out["per_rank_roofline"]] == [(0, 32768), (1, 32768)]
    assert all(r["frac"] > 0 for r in out["per_rank_roofline"])
    # value = GLOBAL boards x steps / time (never boards-per-GPU x ranks of a larger job)
    assert out["value"] == pytest.approx(65536 / (out["ms_per_step"] * 1e-3), rel=1e-9)
    assert out["roofline"]["boards_per_launch"] == 32768
    assert out["roofline"]["frac"] == pytest.approx(96 * 32768 / (out["ms_per_step"] * 1e-3) / 8e12, rel=1e-6)
    assert out["fused_rollout"]["value"] == pytest.approx(65536 / (out["fused_rollout"]["ms_per_step"] * 1e-3), rel=1e-9)
    weak = out["weak_scaling_job"]
    assert weak["scaling"] == "weak" and weak["boards_per_gpu"] == 65536 and weak["global_boards"] == 131072
    assert out["shard_run"] is None
    # the line of a multi-rank run carries the CPU baseline too (rank 0's host cores)
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["cpu_model"]
    one = _bench(1, ["--no-cpu-baseline", "--shard-ranks", "2"])
    assert one["n_gpus"] == 1 and one["backend"] is None and one["scaling"] == "strong"
    assert one["config"]["launch_mode"] == "eager" and "(eager)" in one["config"]["workload"]      # N = 1 is always the plain line
    assert one["timing"]["host_issue_us_per_step"] == one["timing"]["host_call_us"]
    assert one["config"]["global_boards"] == 65536 and one["config"]["boards_per_gpu"] == 65536
    assert "configs[2]" in one["config"]["workload"] and one["weak_scaling_job"] is None
    assert out["episodes"] > 65536
    for k in ("episodes", "mean_episodic_return"):
        assert out[k] == one[k], k
    # shard_run of the one-GPU line = what one rank of the two-rank run does: the same shard, all three forms
    sr = one["shard_run"]
    assert sr["boards"] == 32768 and sr["global_boards"] == 65536 and sr["ranks"] == 2
    for form in ("tpl_step", "capture_steps"):
        assert sr[form]["us_per_step"] > 0 and 0 < sr[form]["frac"] < 1
        assert sr[form]["frac"] == pytest.approx(96 * 32768 / (sr[form]["us_per_step"] * 1e-6) / 8e12, rel=1e-6)
    # the multi-step kernel is priced by vector-instruction issue, not by the 96 B a step it does not move
    fused_leg = sr["tpl_rollout"]
    assert fused_leg["us_per_step"] < sr["tpl_step"]["us_per_step"] and "frac" not in fused_leg
    for r in (fused_leg["roofline"], one["fused_rollout"]["roofline"]):
        assert r["bound"] == "valu-issue" and 0 < r["frac_of_lane_slots"] <= r["frac_hw"] <= r["frac"] < 1
    # the scaling model of the one-GPU line: the 2- and 4-rank shards of THIS job measured on this GPU, value = global boards / period
    sm = one["scaling_model"]
    assert set(sm["per_launch"]) == {"x2", "x4"} and sm["per_launch"]["x2"]["boards_per_gpu"] == 32768
    assert sm["per_launch"]["x2"]["launch_mode"] == "graph"
    assert sm["per_launch"]["x2"]["us_per_step"] == sr["capture_steps"]["us_per_step"]
    for k, leg in sm["per_launch"].items():
        assert leg["value"] == pytest.approx(65536 / (leg["us_per_step"] * 1e-6), rel=1e-9)
        assert leg["efficiency"] == pytest.approx(leg["value"] / (int(k[1:]) * one["value"]), rel=1e-9)
    # the eager form of the same two-rank run: the same job again (same episodes, same mean return)
    eager = _bench(2, ["--no-cpu-baseline", "--launch-mode", "eager", "--no-weak-job"])
    assert eager["config"]["launch_mode"] == "eager"
    for k in ("episodes", "mean_episodic_return"):
        assert eager[k] == one[k], k


@pytest.mark.gpu
def test_bench_graph_mode_with_a_remainder_graph_is_the_eager_job():
    """70 timed steps in graph mode = one graph of 50 step_kernel launches and one of 20, step t on action row t: the same
    episodes and mean return as 70 eager tpl_step() calls -- on two ranks (where graph mode is the default below 2^19 boards per
    GPU) and forced on one."""
    quiet = ("--no-side-figures", "--no-cpu-baseline")
    eager = _bench(1, boards=65536, base=quiet, steps=70)
    assert eager["config"]["launch_mode"] == "eager" and eager["steps"] == 70
    for gpus, extra in ((2, ()), (1, ("--launch-mode", "graph"))):
        out = _bench(gpus, extra, boards=65536, base=quiet, steps=70)
        assert out["config"]["launch_mode"] == "graph" and "2 graph(s) for the 70 timed steps" in out["config"]["launch_mode_is"]
        assert out["steps"] == 70 and out["value"] == pytest.approx(65536 / (out["ms_per_step"] * 1e-3), rel=1e-9)
        for k in ("episodes", "mean_episodic_return"):
            assert out[k] == eager[k], (gpus, k)


@pytest.mark.gpu
def test_bench_falls_back_to_eager_calls_when_a_graph_capture_fails():
    """Eight RCCL ranks capturing graphs at once is the one thing no one-GPU box can rehearse: should a capture fail on any rank,
    every rank drops to eager calls (the ranks agree before the timed region) and the line says why -- a slower headline instead of
    none.  Injected here on both of two ranks; the job is still the one-GPU job."""
    out = _bench(2, ("--no-cpu-baseline", "--no-weak-job"), inject="graph_capture")
    assert out["config"]["launch_mode"] == "eager" and "injected into the graph capture" in out["config"]["graph_capture_failed"]
    assert "(eager)" in out["config"]["workload"] and out["value"] > 0
    plain = _bench(1, ["--no-cpu-baseline", "--shard-ranks", "0"])
    assert plain["config"]["graph_capture_failed"] is None
    for k in ("episodes", "mean_episodic_return"):
        assert out[k] == plain[k], k


@pytest.mark.gpu
def test_bench_sharded_job_of_1048576_boards_on_four_ranks_is_the_one_gpu_job():
    """BASELINE configs[3]'s job -- 1,048,576 boards IN TOTAL, sharded by global board index, `--steps 20 --warmup 5`, side
    figures off -- through bench.py itself with FOUR ranks sharing this box's one GPU over gloo (a GPU box of this pool lets
    at most six processes use its card at once, so the eight-rank form cannot be rehearsed here: its shard geometry is
    tests/test_gpu_parity.py::test_configs3_shard_geometry_..., its control flow with eight ranks tests/test_bench_guards.py).
    The line must be the one-GPU line's job: the same episodes and the same mean episodic return."""
    out = _bench(4, boards=1 << 20, base=("--no-side-figures", "--no-cpu-baseline"))
    assert out["n_gpus"] == 4 and out["ranks_seen"] == 4 and out["scaling"] == "strong"
    assert [(r["rank"], r["boards"]) for r in out["per_rank_roofline"]] == [(k, 262144) for k in range(4)]
    assert all(0 < r["frac"] < 1 for r in out["per_rank_roofline"])
    assert out["config"]["global_boards"] == 1 << 20 and out["config"]["boards_per_gpu"] == 262144
    assert out["value"] == pytest.approx((1 << 20) / (out["ms_per_step"] * 1e-3), rel=1e-9)
    assert out["config"]["launch_mode"] == "graph" and out["timing"]["host_call_us"] > 0       # 262,144 boards per GPU: below 2^19
    assert out["fused_rollout"] is None and out["weak_scaling_job"] is None and "cpu_baseline" not in out
    assert out["side_figures"]["failed"] == [] and out["side_figures"]["seconds"] == {}
    one = _bench(1, boards=1 << 20, base=("--sustained", "0", "--actor-boards", "0", "--carved-pool", "0", "--no-config1", "--no-out-of-cache",
                                          "--no-cpu-baseline"))
    assert one["episodes"] > 1 << 20
    for k in ("episodes", "mean_episodic_return"):
        assert out[k] == one[k], k
    # ... and that one-GPU line states the ceiling of the eight-GPU run before hardware does: value_x8 = 2^20 / period(131,072 boards)
    x8 = one["scaling_model"]["per_launch"]["x8"]
    assert x8["boards_per_gpu"] == 131072 and x8["launch_mode"] == "graph" and x8["us_per_step"] == one["shard_run"]["capture_steps"]["us_per_step"]
    assert x8["value"] == pytest.approx((1 << 20) / (x8["us_per_step"] * 1e-6), rel=1e-9) and 0.1 < x8["efficiency"] < 1
    assert one["scaling_model"]["per_launch"]["x2"]["launch_mode"] == "eager" and set(one["scaling_model"]["per_launch"]) == {"x2", "x4", "x8"}
    # the replayed graph against eager calls: 9 % faster on a box whose host needs 5.5 us per call, 3 % slower on one that needs 4.7
    # (profiles/NOTES.md, round 6) -- never far behind, and independent of the host
    assert one["shard_run"]["capture_steps"]["us_per_step"] <= one["shard_run"]["tpl_step"]["us_per_step"] * 1.25


@pytest.mark.gpu
def test_bench_headline_survives_failing_side_figures_on_one_and_on_two_ranks():
    """A side figure that raises (injected: TPL_BENCH_INJECT_FAILURE) costs its own key, nothing else: the run ends with status 0
    and ONE line whose headline, roofline and remaining side figures are there.  With two ranks the failure is on rank 1 ONLY:
    rank 0 must not be left in a collective (the run would end at the process group's timeout instead), and the figure after it
    (the weak job) still runs on both."""
    one = _bench(1, inject="fused_rollout", extra=("--no-cpu-baseline", "--shard-ranks", "2"))
    assert one["value"] > 0 and 0 < one["roofline"]["frac"] < 1 and one["episodes"] > 65536
    assert "injected" in one["fused_rollout"]["error"] and one["side_figures"]["failed"] == ["fused_rollout"]
    assert one["roofline"]["sustained"]["launches"] == 100 and one["shard_run"]["tpl_step"]["us_per_step"] > 0
    two = _bench(2, inject="fused_rollout@1", extra=("--no-cpu-baseline",))
    assert two["ranks_seen"] == 2 and two["value"] > 0 and two["episodes"] == one["episodes"]
    assert two["fused_rollout"]["failed_ranks"] == [1] and "rank(s) [1] failed" in two["fused_rollout"]["error"]
    assert two["weak_scaling_job"]["value"] > 0 and two["side_figures"]["failed"] == ["fused_rollout"]


@pytest.mark.gpu
def test_bench_one_rank_through_rccl_and_the_gloo_control_group():
    """The multi-rank plumbing on REAL RCCL, as far as a one-GPU box allows: TPL_BENCH_FORCE_DIST=1 makes a one-rank run initialise
    the "nccl" process group (bound to its device), create the gloo control group beside it, and send every collective of the N > 1
    path -- barriers, the all-reduce of the return counters, the all-gather of the ranks' times, SideFigures' agreements -- through
    them.  (Two RCCL ranks cannot share one GPU, so this is the only way the nccl + gloo combination runs before an 8-GPU node
    does.)  The line must be the plain one-GPU line's job."""
    out = _bench(1, ["--no-cpu-baseline", "--shard-ranks", "0"], rehearsal=False,
                 env_more=dict(TPL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert out["backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["side_figures"]["agreements_over"] == "gloo group of host tensors" and out["side_figures"]["failed"] == []
    assert out["fused_rollout"]["value"] > 0 and out["roofline"]["sustained"]["launches"] == 100
    plain = _bench(1, ["--no-cpu-baseline", "--shard-ranks", "0"])
    for k in ("episodes", "mean_episodic_return"):
        assert out[k] == plain[k], k
    # ... and the graph launch mode of an N > 1 run UNDER a live RCCL process group (its watchdog thread polls events while the
    # step launches are being captured): capture and replay must go through, and give the same job
    graph = _bench(1, ["--no-cpu-baseline", "--shard-ranks", "0", "--launch-mode", "graph"], rehearsal=False,
                   env_more=dict(TPL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert graph["backend"] == "nccl" and graph["config"]["launch_mode"] == "graph" and graph["side_figures"]["failed"] == []
    for k in ("episodes", "mean_episodic_return"):
        assert graph[k] == plain[k], k


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
