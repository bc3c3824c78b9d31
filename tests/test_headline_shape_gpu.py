"""The SHIPPED instantiations at the shapes bench.py times, against the CPU checker (needs an MI355X).

The other GPU tests stop at 800 fixed / 900 moving points.  Here: the headline shape (KITTI, 2000 keypoints per image, 2000
local-map points, max_fixed 896: stereo_match5_kernel<2> with its persistent grid wrapped, align_kernel<512, true, circle> with
several workgroups per CU, gn_kernel<8, stereo, false, 3, 5>) and the EuRoC / TUM bench shapes (1000 / 1000, max_fixed 512 / 1024),
batches of 1024+ frames tiled from 37 distinct ones, matcher epilogue feeding the aligner in place -- every distinct frame is
compared with the checker (correspondences bit-exact incl. order, pose bit-exact AND within the 1e-4 of BASELINE.json), and every
replica in the batch with its source frame (bit-exact: a frame's result must not depend on which workgroup / CU slot ran it).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4
UNIQUE = 37  # a prime: every distinct frame visits many CU slots / XCDs


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _run(cfg_name, keypoints, moving, max_fixed, batch, seed, steps=2, unique=UNIQUE, corr_of=None):
    import torch
    import bench
    from srrg2_proslam_amd import configs
    cfg = configs.get(cfg_name)
    w = bench.FrameWorkload(cfg, 0, batch, keypoints, moving, max_fixed, unique, seed)
    stream = torch.cuda.Stream(device=w.dev)
    with torch.cuda.stream(stream):
        w.ctx.use_torch_stream()
        for _ in range(steps):  # the second step runs on warm caches and on the buffers the first one left behind
            w.step()
        torch.cuda.synchronize()
        snap = w.snapshot()
        ok, _ = w.check(snap)
        X_all = w.aframes.X.cpu().numpy()
        n_corr_all = w.aframes.n_corr.cpu().numpy()
        corr_all = ({b: w.aframes.corr_of(b) for b in corr_of} if corr_of is not None else
                    ([w.aframes.corr_of(b) for b in range(batch)] if batch <= 2048 else None))
        n_match_all = w.sframes.n_matches.cpu().numpy() if w.stereo else None
    _, _, poses = bench.cpu_baseline(cfg, w.uniq, len(w.uniq))
    out = {"snap": snap, "ok": ok, "X": X_all, "n_corr": n_corr_all, "corr": corr_all, "n_match": n_match_all, "poses": poses, "uniq": w.uniq,
           "cfg": cfg}
    w.close()
    del w
    torch.cuda.empty_cache()
    return out


def _check(r, batch):
    poses = r["poses"]
    assert len(poses) == UNIQUE
    n_big = 0
    for u, (Xr, c) in enumerate(poses):
        gc = r["corr"][u]
        assert len(gc) == len(c) and np.array_equal(gc["fixed_idx"], c["fixed_idx"]) and np.array_equal(gc["moving_idx"], c["moving_idx"]), "frame %d" % u
        assert np.array_equal(gc["response"].view(np.uint32), c["response"].view(np.uint32)), "frame %d" % u
        Xg = r["X"][u].reshape(4, 4)
        Xr = np.asarray(Xr, np.float32).reshape(4, 4)
        assert np.linalg.norm(Xg.astype(np.float64) - Xr) / np.linalg.norm(Xr) <= POSE_TOL, "frame %d" % u
        assert np.array_equal(_bits(Xg), _bits(Xr)), "frame %d: pose not bit-identical" % u
        n_big += len(c) > 100
    assert n_big >= UNIQUE // 2  # the frames really carry headline-sized correspondence sets
    # every replica equals its source frame, whichever workgroup ran it
    for b in range(UNIQUE, batch):
        u = b % UNIQUE
        assert r["n_corr"][b] == r["n_corr"][u], "replica %d of frame %d" % (b, u)
        assert np.array_equal(_bits(r["X"][b]), _bits(r["X"][u])), "replica %d of frame %d" % (b, u)
        gb, gu = r["corr"][b], r["corr"][u]
        assert np.array_equal(gb["fixed_idx"], gu["fixed_idx"]) and np.array_equal(gb["moving_idx"], gu["moving_idx"]), "replica %d" % b
    if r["n_match"] is not None:
        for b in range(UNIQUE, batch):
            assert r["n_match"][b] == r["n_match"][b % UNIQUE]


def test_kitti_headline_shape_matcher_epilogue_into_aligner():
    """2000 keypoints, 2000 local-map points, max_fixed 896, 1184 frames (4.6 per CU: the matcher's persistent grid wraps, the search
    kernel keeps three workgroups per CU, the GN kernel ten frames)"""
    batch = 32 * UNIQUE
    r = _run("kitti", 2000, 2000, 896, batch, 90210)
    assert r["ok"] > 0.9
    assert 380 <= r["snap"]["n_corr"] <= 560 and 600 <= r["snap"]["n_match"] <= 800  # the headline's ~459 correspondences / ~704 matches
    _check(r, batch)


@pytest.mark.parametrize("cfg_name,max_fixed,seed", [("euroc", 512, 4242), ("tum", 1024, 777)])
def test_euroc_and_tum_bench_shapes(cfg_name, max_fixed, seed):
    """1000 / 1000 at the bench's max_fixed: gn_kernel<4, stereo, false> (EuRoC) and the generic gn_kernel<8, 0, true> with
    inlier-only runs and kept inlier classes (TUM, 524 correspondences: rows beyond the parked ones are streamed)"""
    batch = 28 * UNIQUE
    r = _run(cfg_name, 1000, 1000, max_fixed, batch, seed)
    assert r["ok"] > 0.9
    _check(r, batch)


def test_full_bench_batch_every_distinct_frame_and_every_replica():
    """bench.py's own allocation: B = 55296 frames per step tiled from 251 distinct ones (its --unique default), two steps.  Every
    distinct frame against the CPU checker (correspondences bit-exact incl. order, pose bit-exact and <= 1e-4); pose, correspondence
    count and stereo match count of ALL 55296 replicas against their source frame; the correspondence vectors of a strided sample."""
    import bench
    batch, unique = 55296, 251
    sample = sorted(set(range(unique)) | set(range(unique, batch, 397)))
    from srrg2_proslam_amd import synthetic as syn
    r = _run("kitti", 2000, 2000, 896, batch, syn.seed_for(1, 0), unique=unique, corr_of=sample)  # (the seed of bench.py's rank 0)
    assert r["ok"] > 0.9
    poses = r["poses"]
    assert len(poses) == unique
    for u, (Xr, c) in enumerate(poses):
        gc = r["corr"][u]
        assert len(gc) == len(c) and np.array_equal(gc["fixed_idx"], c["fixed_idx"]) and np.array_equal(gc["moving_idx"], c["moving_idx"]), "frame %d" % u
        assert np.array_equal(gc["response"].view(np.uint32), c["response"].view(np.uint32)), "frame %d" % u
        Xg = r["X"][u].reshape(4, 4)
        Xr = np.asarray(Xr, np.float32).reshape(4, 4)
        assert np.linalg.norm(Xg.astype(np.float64) - Xr) / np.linalg.norm(Xr) <= POSE_TOL, "frame %d" % u
        assert np.array_equal(_bits(Xg), _bits(Xr)), "frame %d: pose not bit-identical" % u
    src = np.arange(batch) % unique
    X = _bits(r["X"]).reshape(batch, -1)
    bad = np.nonzero((X != X[src]).any(axis=1) | (r["n_corr"] != r["n_corr"][src]) | (r["n_match"] != r["n_match"][src]))[0]
    assert bad.size == 0, "replicas that differ from their source frame: %s" % bad[:10]
    for b in sample:
        gb, gu = r["corr"][b], r["corr"][b % unique]
        assert np.array_equal(gb["fixed_idx"], gu["fixed_idx"]) and np.array_equal(gb["moving_idx"], gu["moving_idx"]), "replica %d" % b
