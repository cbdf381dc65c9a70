"""One rank of the two-rank HIP rehearsal (tests/test_multi_rank_gpu.py): started by torch.distributed.run, steps its
shard of the batch through the C ABI on cuda:0 (the ranks share the one GPU of the test box; the process group is
gloo) and all-reduces the episodic-return counters through the product's sharding module."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import tetris_piclim as T  # noqa: E402

L, M, BOARDS, POOL, STEPS, SEED = 5, 20, 4096 + 37, 512, 60, 21
REWARD = (1.0, 5.0, -1.0)


def run_shard(shard, device="cuda:0"):
    env = T.BatchedTetris(L, M, shard.boards, device=device, seed=SEED, global_offset=shard.global_offset, auto_reset=True,
                          assign="hash", reward=REWARD)
    rows, pieces = env.synthetic_configs(POOL)
    env.load_configs(rows, pieces)
    env.reset()
    reward = torch.empty(shard.boards, dtype=torch.float32, device=device)
    done = torch.empty(shard.boards, dtype=torch.uint8, device=device)
    rsum = torch.zeros(shard.boards, dtype=torch.float64, device=device)
    for t in range(STEPS):
        if t % 3 == 2:                                       # the fused form on some steps: same results by contract
            r, _ = env.rollout(env.synthetic_actions(t).unsqueeze(0))
            rsum += r.double()
        else:
            env.step_into(env.synthetic_actions(t), reward, done)
            rsum += reward.double()
    stats = env.stats_tensor().cpu()
    state = {k: v.cpu().numpy() for k, v in env.packed_state().items()}
    state["rows"] = state["rows"].view(np.uint16)
    state["reward_sum"] = rsum.cpu().numpy()
    env.terminate()
    return stats, state


def main():
    mode, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sh = T.sharding.strong_shard(rank, world, BOARDS) if mode == "strong" else T.sharding.weak_shard(rank, world, BOARDS // world)
    stats, state = run_shard(sh)
    mean, episodes = T.sharding.mean_episodic_return(stats, REWARD)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.global_offset, state))
    if rank == 0:
        gathered.sort(key=lambda g: g[0])
        keys = ("rows", "lines", "moves", "state", "cur", "nxt", "reward_sum")
        np.savez(out, mean=mean, episodes=episodes, ranks=world, lib=T.LIB_PATH,
                 **{k: np.concatenate([g[1][k] for g in gathered]) for k in keys})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
