"""Sequence sharding across ranks: sequences are independent serial chains, so every
rank owns a disjoint set and no data-path collective exists.  torch.distributed is used only for
the barrier and the MAX-over-ranks timing the bench contract requires ("nccl" = RCCL on the GPU
box, "gloo" in the CPU tests and when several ranks share one GPU).  bench.py drives its N > 1 runs
through these helpers; tests/test_sharding_gloo.py runs the same code on two gloo ranks."""
import os


def rank_world():
    """(rank, world_size, local_rank) from the launcher's environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None, device_index=None):
    """rendezvous on 127.0.0.1 (the container hostname may not resolve); backend None = nccl when a device is given"""
    import torch
    import torch.distributed as dist
    rank, world, _ = rank_world()
    if world <= 1 or dist.is_initialized():
        return rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if device_index is not None else "gloo"
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(device_index)))
    else:
        dist.init_process_group(backend=backend)
    return rank, world


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def sequences_of_rank(n_sequences, rank, world):
    """sequence s runs on rank s mod world (config 5: KITTI 00-07, one per GPU)"""
    return [s for s in range(n_sequences) if s % world == rank]


def balanced_sequences_of_rank(lengths, rank, world):
    """longest-processing-time assignment for sequences of unequal length (KITTI 00 has 4541 frames, 04 has 271): the
    sequences, longest first, go to the rank with the least work so far.  Returns the indices owned by `rank`."""
    load = [0] * world
    owner = {}
    for s in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):
        r = min(range(world), key=lambda i: (load[i], i))
        owner[s] = r
        load[r] += lengths[s]
    return sorted(s for s, r in owner.items() if r == rank)


def seed_of_sequence(config_index, sequence):
    from .synthetic import seed_for
    return seed_for(config_index, sequence)


def _reduce(value, op, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device=None):
    """elapsed time of the slowest rank (identity when not distributed)"""
    import torch.distributed as dist
    return _reduce(value, dist.ReduceOp.MAX, device)


def sum_over_ranks(value, device=None):
    import torch.distributed as dist
    return _reduce(value, dist.ReduceOp.SUM, device)


def gather_over_ranks(value, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank (a one-element list when not distributed)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    world = dist.get_world_size()
    t = torch.zeros((world,), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def backend_world_size():
    """the world size the initialised backend reports (1 when not distributed)"""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def aggregate_throughput(units_this_rank, elapsed_this_rank, device=None):
    """whole-job throughput of the bench contract: units of ALL ranks / time of the SLOWEST rank -> (units/s, seconds)"""
    total = sum_over_ranks(units_this_rank, device)
    slowest = max_over_ranks(elapsed_this_rank, device)
    return total / slowest, slowest


KITTI_SEQUENCE_FRAMES = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101)  # sequences 00-07 (BASELINE.json config 5)


def run_sequences_over_ranks(lengths, scale, run_sequence, reduce_device=None, min_frames=4):
    """BASELINE.json config 5 as written: whole sequences (serial chains) are assigned to ranks longest-first, every rank runs
    its own one after the other, and the job's rate is all tracked frames / the slowest rank's time.
    run_sequence(sequence_index, n_frames) -> (tracked_frames, seconds) runs ONE sequence (on this rank's device) for n_frames =
    round(scale * length).  Returns (frames/s over all ranks, seconds of the slowest rank, [(sequence, frames, seconds)] of this rank)."""
    rank, world, _ = rank_world()
    mine = balanced_sequences_of_rank(list(lengths), rank, world)
    tracked, elapsed, parts = 0, 0.0, []
    barrier()
    for s in mine:
        n = max(int(round(lengths[s] * scale)), min_frames)
        got, seconds = run_sequence(s, n)
        tracked += got
        elapsed += seconds
        parts.append((s, n, seconds))
    barrier()
    fps, slowest = aggregate_throughput(tracked, max(elapsed, 1e-9), reduce_device)
    return fps, slowest, parts
