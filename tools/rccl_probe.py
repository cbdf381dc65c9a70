#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: a process group of one rank over the "nccl" backend and the job's one collective
(mean_episodic_return's all-reduce)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import tetris_piclim as T

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
env = T.BatchedTetris(5, 20, 4096, auto_reset=True, reward=(1.0, 5.0, -1.0))
rows, pieces = env.synthetic_configs(512)
env.load_configs(rows, pieces)
env.reset()
for t in range(60):
    env.step(env.synthetic_actions(t), observe=False)
mean, episodes = T.sharding.mean_episodic_return(env.stats_tensor(), env.reward_params)
dist.barrier()
print(f"rccl ok: mean episodic return {mean:.4f} over {episodes} episodes")
dist.destroy_process_group()
