"""N > 1 path on CPU: world_size-2 gloo processes shard independent sequences, run them (with the
CPU oracle standing in for the device, as the checker), and aggregate like bench.py does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from srrg2_proslam_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sequences, out_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    assert sharding.init_distributed("gloo") == (rank, world)  # the rendezvous bench.py uses
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from helpers import kitti_frame, oracle_stereo_params
    from oracle import binding as ob
    mine = sharding.sequences_of_rank(n_sequences, rank, world)
    frames = 0
    digest = []
    for s in mine:
        cfg, fr = kitti_frame(sharding.seed_of_sequence(1, s), 300)
        corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(ob, cfg["stereo_matcher"]))
        digest.append((s, int(len(corr)), int(corr["moving_idx"].sum())))
        frames += 1
    sharding.barrier()
    elapsed = 0.5 + 0.25 * rank  # deterministic stand-in for a measured time
    fps, t_max = sharding.aggregate_throughput(frames, elapsed)  # bench.py's own aggregation
    total = sharding.sum_over_ranks(frames)
    assert fps == total / t_max
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array([t_max, total] + [v for d in digest for v in d], dtype=np.float64))
    sharding.barrier()
    sharding.shutdown()


def test_two_ranks_shard_sequences_without_overlap(tmp_path):
    world, n_sequences = 2, 8  # config 5: KITTI 00-07 sharded across ranks
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_sequences, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % k)) for k in range(world)]
    # MAX over ranks of the time, SUM of the frames
    assert r[0][0] == r[1][0] == 0.75 and r[0][1] == r[1][1] == n_sequences
    seqs = [set(int(x) for x in rr[2::3]) for rr in r]
    assert seqs[0] == {0, 2, 4, 6} and seqs[1] == {1, 3, 5, 7}
    # every sequence produced work, and the results are the single-process results
    from helpers import kitti_frame, oracle_stereo_params
    from oracle import binding as ob
    for rr in r:
        for s, n, chk in zip(rr[2::3], rr[3::3], rr[4::3]):
            cfg, fr = kitti_frame(sharding.seed_of_sequence(1, int(s)), 300)
            corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(ob, cfg["stereo_matcher"]))
            assert len(corr) == int(n) and int(corr["moving_idx"].sum()) == int(chk)


def test_sharding_is_a_partition():
    for world in (1, 2, 3, 4, 8):
        parts = [sharding.sequences_of_rank(11, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(11))
    assert sharding.max_over_ranks(1.5) == 1.5 and sharding.sum_over_ranks(3) == 3.0
    assert sharding.aggregate_throughput(10, 2.0) == (5.0, 2.0)


def test_uneven_sequences_are_balanced():
    """config 5: KITTI 00-07 have 4541, 1101, 4661, 801, 271, 2761, 1101, 1101 frames; one per GPU leaves seven GPUs idle while
    00 / 02 finish.  With fewer ranks than sequences the longest-first assignment evens the load out."""
    lengths = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101]
    for world in (1, 2, 4, 8):
        parts = [sharding.balanced_sequences_of_rank(lengths, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(8))
        loads = [sum(lengths[s] for s in part) for part in parts]
        assert max(loads) <= max(max(lengths), -(-sum(lengths) // world) + max(lengths) // 2)
    two = [sum(lengths[s] for s in sharding.balanced_sequences_of_rank(lengths, r, 2)) for r in range(2)]
    assert abs(two[0] - two[1]) <= 300
