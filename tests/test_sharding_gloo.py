"""The N > 1 path on CPU: two gloo ranks shard a batch by global board index, step their shards (the CPU
oracle stands in for the GPU step, which cannot run here), and all-reduce the episodic-return counters through
the product's sharding module.  The result must equal one process over the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

L, M, BOARDS, POOL, STEPS, SEED = 5, 20, 4096, 512, 50, 21
REWARD = (1.0, 5.0, -1.0)


def _run_shard(O, shard):
    rows, pieces = O.synth_boards(SEED, 0, POOL, L), O.synth_pieces(SEED, 0, POOL, M)
    env = O.Env(shard.boards, L, M, shard.global_offset, SEED)
    env.set_pool(rows, pieces)
    env.set_options(auto_reset=True, assign_mode=0, per_line=REWARD[0], win=REWARD[1], lose=REWARD[2])
    env.reset()
    for t in range(STEPS):
        env.step(O.synth_actions(SEED, shard.global_offset, shard.boards, t))
    st = env.stats()
    state = env.get_state()
    return torch.tensor([st["episodes"], st["lines"], st["wins"], st["topouts"]], dtype=torch.int64), state


def _worker(rank, world, port, mode, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tetris_piclim as T
    from oracle import oracle as O
    sh = T.sharding.strong_shard(rank, world, BOARDS) if mode == "strong" else T.sharding.weak_shard(rank, world, BOARDS // world)
    stats, state = _run_shard(O, sh)
    mean, episodes = T.sharding.mean_episodic_return(stats, REWARD)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.global_offset, state["rows"], state["lines"], state["moves"]))
    if rank == 0:
        gathered.sort(key=lambda g: g[0])
        np.savez(out, mean=mean, episodes=episodes, rows=np.concatenate([g[1] for g in gathered]),
                 lines=np.concatenate([g[2] for g in gathered]), moves=np.concatenate([g[3] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode,world", [("weak", 2), ("strong", 2), ("strong", 3)])
def test_two_rank_gloo_matches_single_process(tmp_path, oracle, mode, world):
    import tetris_piclim as T
    out = str(tmp_path / "dist.npz")
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    got = np.load(out)
    n_total = BOARDS if mode == "strong" else (BOARDS // world) * world
    stats, state = _run_shard(oracle, T.sharding.strong_shard(0, 1, n_total))
    mean, episodes = T.sharding.mean_episodic_return(stats, REWARD)
    assert int(got["episodes"]) == episodes and episodes > n_total
    assert float(got["mean"]) == mean
    assert np.array_equal(got["rows"], state["rows"]) and np.array_equal(got["lines"], state["lines"])
    assert np.array_equal(got["moves"], state["moves"])


def test_shard_arithmetic():
    import tetris_piclim as T
    for world in (1, 2, 3, 8):
        parts = [T.sharding.strong_shard(r, world, 1000) for r in range(world)]
        assert sum(p.boards for p in parts) == 1000
        assert all(parts[r].global_offset == sum(p.boards for p in parts[:r]) for r in range(world))
    w = T.sharding.weak_shard(3, 8, 1 << 20)
    assert (w.global_offset, w.global_boards) == (3 << 20, 8 << 20)
