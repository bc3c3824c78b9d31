"""Sequence sharding across ranks: sequences are independent serial chains, so every
rank owns a disjoint set and no data-path collective exists.  torch.distributed is used only for
the barrier and the MAX-over-ranks timing the bench contract requires ("nccl" = RCCL on the GPU
box, "gloo" in the CPU tests)."""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def sequences_of_rank(n_sequences, rank, world):
    """sequence s runs on rank s mod world (config 5: KITTI 00-07, one per GPU)"""
    return [s for s in range(n_sequences) if s % world == rank]


def seed_of_sequence(config_index, sequence):
    from .synthetic import seed_for
    return seed_for(config_index, sequence)


def max_over_ranks(value, device=None):
    """elapsed time of the slowest rank (identity when not distributed)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
