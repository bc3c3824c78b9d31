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


def _loop_worker(rank, world, port, out_dir):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    sharding.init_distributed("gloo")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_tracking as bt
    from srrg2_proslam_amd import configs
    cfg = configs.get("kitti")

    def run_sequence(s, n_frames):
        # the closed loop of tools/bench_tracking.py for one short sequence, on the CPU checker instead of the device
        gt = bt.kitti00_poses(n_frames)
        gt = np.linalg.inv(gt[0]) @ gt
        seq = bt.make_sequences(cfg, 1, gt, 300, sharding.seed_of_sequence(1, s))[0]
        poses, sizes, flags, seconds = bt.oracle_chain(cfg, seq, bt.split_schedule(gt), 2048, n_frames + 1, 1.0)
        assert len(poses) == n_frames and all(f >= 0 for f in flags)
        return n_frames - 2, 0.1 * (n_frames - 2) * (1 + rank)  # deterministic stand-in for the measured time

    fps, slowest, parts = sharding.run_sequences_over_ranks(sharding.KITTI_SEQUENCE_FRAMES, 0.001, run_sequence, min_frames=3)
    np.save(os.path.join(out_dir, "loop%d.npy" % rank), np.array([fps, slowest] + [v for p in parts for v in p[:2]], dtype=np.float64))
    sharding.shutdown()


def test_closed_loop_entry_on_two_ranks(tmp_path):
    """bench.py --mode closed-loop under torch.distributed.run (BASELINE.json config 5): the same driver, sharding.run_sequences_over_ranks,
    on two gloo ranks with the CPU checker's loop standing in for the device"""
    world, port = 2, _free_port()
    mp.spawn(_loop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), "loop%d.npy" % k)) for k in range(world)]
    seqs = [[int(x) for x in rr[2::2]] for rr in r]
    assert sorted(seqs[0] + seqs[1]) == list(range(8)) and not set(seqs[0]) & set(seqs[1])
    assert seqs[0] == sharding.balanced_sequences_of_rank(list(sharding.KITTI_SEQUENCE_FRAMES), 0, 2)
    frames = [[int(x) for x in rr[3::2]] for rr in r]
    tracked = [sum(n - 2 for n in f) for f in frames]
    slowest = max(0.1 * tracked[0] * 1, 0.1 * tracked[1] * 2)
    for rr in r:  # every rank holds the job's figures: all frames / the slowest rank's time
        assert abs(rr[1] - slowest) < 1e-9 and abs(rr[0] - sum(tracked) / slowest) < 1e-6


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


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher: the parent starts two ranks of itself (gloo rendezvous on 127.0.0.1), rank 0 prints
    ONE line with n_gpus 2, the world size the backend reported and one rate per rank.  --dry-run replaces the device step by a sleep
    (no GPU here); the spawn, rendezvous, barrier and SUM / MAX code is the one the GPU run takes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--batch", "32", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["backend_world_size"] == 2 and len(line["ranks"]["fps_per_rank"]) == 2
    assert line["value"] == 0.0 and line["data"].startswith("dry-run")  # never mistaken for a measurement


def test_bench_refuses_more_ranks_than_devices():
    """no silent 1-GPU run: --gpus 2 with fewer than two visible devices (none here) exits non-zero before any rank starts"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PRS_BENCH_SHARE_GPU")}
    env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "device(s) visible" in (out.stderr + out.stdout)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
