"""Multi-GPU layout of the batched environment: one process per GPU, boards sharded by global batch index.

Boards are independent (nothing in game/tetris.py:354-449 of the upstream repo couples two boards), so a rank
owns a contiguous block of global board indices and the data path needs NO collective.  Everything random
(synthetic configurations, synthetic actions, configuration assignment) is keyed by the GLOBAL board index, so
results do not depend on how many GPUs share the batch.  The one exchange is the episodic-return mean: a sum
of two scalars per rank, all-reduced over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    boards: int              # boards owned by this rank
    global_offset: int       # global index of this rank's board 0
    global_boards: int


def weak_shard(rank: int, world: int, boards_per_rank: int) -> Shard:
    """Fixed work per GPU: rank r owns [r*B, (r+1)*B)."""
    return Shard(rank, world, boards_per_rank, rank * boards_per_rank, world * boards_per_rank)


def strong_shard(rank: int, world: int, global_boards: int) -> Shard:
    """Fixed total work: contiguous blocks, the first `global_boards % world` ranks hold one board more."""
    base, extra = divmod(global_boards, world)
    boards = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return Shard(rank, world, boards, offset, global_boards)


def return_sum(stats: torch.Tensor, reward_params) -> torch.Tensor:
    """[sum of episodic returns, finished episodes] (f64) from the four exact counters
    {episodes, lines at finish, wins, top-outs}: return = per_line*lines + win*wins + lose*(episodes - wins)."""
    per_line, win, lose = reward_params
    s = stats.to(torch.float64)
    return torch.stack([per_line * s[1] + win * s[2] + lose * (s[0] - s[2]), s[0]])


def mean_episodic_return(stats: torch.Tensor, reward_params, group=None):
    """All-reduce (sum) of [return sum, episodes] across ranks -> (mean return, episodes).  The only collective
    of the whole job; 16 bytes per rank."""
    acc = return_sum(stats, reward_params)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    total, episodes = acc.tolist()
    return (total / episodes if episodes else float("nan")), int(episodes)
