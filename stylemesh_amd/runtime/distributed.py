"""Multi-GPU protocol of the view-sharded path (SURVEY.md section 8 e): one process per GPU, views of the scene
shard over ranks, ONE exchange per step - a sum all-reduce of the flat texture-gradient arena (RCCL over xGMI on
the GPU box: ``torch.distributed`` backend "nccl"; "gloo" in the CPU tests) - then the identical fused update on
every rank with ``grad_scale = 1 / world_size`` (DDP-mean semantics).

Nothing here touches the kernels; it is the host logic around them, testable without a GPU.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_views(indices, rank: int, world_size: int):
    """Rank r takes views r, r + R, r + 2R, ... of the scene's train list (keeps the per-rank consecutive-repeat
    structure of RepeatingSampler, so the per-view constant cache still hits)."""
    return list(indices)[rank::world_size]


def steps_per_epoch(n_views: int, index_repeat: int, world_size: int) -> int:
    """All ranks must take the same number of steps (every step contains a collective): ranks with one view
    fewer repeat their last view."""
    per_rank = -(-n_views // world_size)
    return per_rank * index_repeat


def rank_schedule(indices, rank, world_size, index_repeat):
    """View index per step for this rank, padded so that every rank has ``steps_per_epoch`` entries."""
    mine = shard_views(indices, rank, world_size)
    per_rank = -(-len(list(indices)) // world_size)
    if not mine:
        raise ValueError("more ranks than views")
    while len(mine) < per_rank:
        mine.append(mine[-1])
    return [v for v in mine for _ in range(index_repeat)]


def make_grad_reducer(dist_module, world_size: int):
    """In-place SUM all-reduce of the gradient arena (None for a single rank)."""
    if world_size <= 1:
        return None

    def reduce(flat_grad):
        dist_module.all_reduce(flat_grad, op=dist_module.ReduceOp.SUM)
    return reduce
