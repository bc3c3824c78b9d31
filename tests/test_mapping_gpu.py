"""GPU parity of the landmark estimators + projective mergers through the C-ABI:
after every merged frame ALL map arrays (local coordinates, world state, covariance, descriptor, counters,
inlier flags, measurement history, pose table) equal the oracle's bit for bit (float32 bit patterns; the
EKF's double arithmetic lands in the float covariance / state)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import binding as ob
from oracle import binding_mapping as om
from srrg2_proslam_amd import _lib, configs, ops
from tests import helpers as hp
from tests.test_oracle_mapping import merger_params as oracle_merger_params, stereo_scene

pytestmark = pytest.mark.gpu


def _copy_struct(src, dst_type):
    dst = dst_type()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(dst_type))
    return dst


def _gpu_params(po):
    assert C.sizeof(_lib.MergerParams) == C.sizeof(om.MergerParams)
    return _copy_struct(po, _lib.MergerParams)


def _upload_map(maps, b, m, poses):
    n, dev = m.n_points, maps.coords.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    maps.coords[b, :n], maps.desc[b, :n], maps.state[b, :n] = t(m.coords[:n]), t(m.desc[:n]), t(m.state[:n])
    maps.covariance[b, :n] = t(m.covariance[:n])
    maps.n_opt[b, :n], maps.inlier[b, :n] = t(m.n_opt[:n].view(np.int32)), t(m.inlier[:n])
    maps.n_meas[b, :n] = t(m.n_meas[:n].view(np.int32))
    if m.max_measurements > 0:
        maps.meas[b, :n] = t(m.meas[:n].view(np.int32).reshape(n, m.meas.shape[1], 7))
    maps.poses[b, : len(poses)] = t(poses.view(np.float32).reshape(len(poses), 24))
    maps.n_points[b] = n


def _upload_frame(maps, b, fixed, desc, corr, Tw, Ts, frame):
    dev = maps.coords.device
    n = len(fixed)
    z4 = np.zeros((n, 4), np.float32)
    z4[:, : fixed.shape[1]] = fixed
    maps.measurement[b, :n] = torch.from_numpy(z4).to(dev)
    maps.measurement_desc[b, :n] = torch.from_numpy(np.ascontiguousarray(desc)).to(dev)
    maps.n_measured[b] = n
    raw = np.zeros((len(corr), 3), np.int32)
    raw[:, 0], raw[:, 1], raw[:, 2] = corr["fixed_idx"], corr["moving_idx"], corr["response"].view(np.int32)
    if len(corr):
        maps.corr[b, : len(corr)] = torch.from_numpy(raw).to(dev)
    maps.n_corr[b] = len(corr)
    maps.measurement_in_world[b] = torch.from_numpy(np.ascontiguousarray(Tw, dtype=np.float32)).to(dev)
    maps.measurement_in_scene[b] = torch.from_numpy(np.ascontiguousarray(Ts, dtype=np.float32)).to(dev)
    maps.frame[b] = frame


def _assert_map_equal(maps, b, m, poses, n_frames):
    n = m.n_points
    assert int(maps.n_points[b].item()) == n
    def bits(a):
        # float32 bit patterns; every NaN counts as the same value (a degenerate landmark is NaN on both sides,
        # sign / payload of the NaN are not part of the contract)
        a = np.ascontiguousarray(a, dtype=np.float32)
        return np.where(np.isnan(a), np.uint32(0x7FC00000), a.view(np.uint32))
    assert np.array_equal(bits(maps.coords[b, :n, :3].cpu().numpy()), bits(m.coords[:n, :3])), "coords"
    assert np.array_equal(bits(maps.state[b, :n, :3].cpu().numpy()), bits(m.state[:n, :3])), "state"
    assert np.array_equal(bits(maps.covariance[b, :n].cpu().numpy()), bits(m.covariance[:n])), "covariance"
    assert np.array_equal(maps.desc[b, :n].cpu().numpy(), m.desc[:n]), "desc"
    assert np.array_equal(maps.n_opt[b, :n].cpu().numpy().view(np.uint32), m.n_opt[:n]), "n_opt"
    assert np.array_equal(maps.inlier[b, :n].cpu().numpy(), m.inlier[:n]), "inlier"
    assert np.array_equal(maps.n_meas[b, :n].cpu().numpy().view(np.uint32), m.n_meas[:n]), "n_meas"
    if m.max_measurements > 0:
        g = maps.meas[b, :n].cpu().numpy()
        o = m.meas[:n].view(np.int32).reshape(n, m.meas.shape[1], 7)
        for i in range(n):
            k = int(m.n_meas[i])
            assert np.array_equal(g[i, :k], o[i, :k]), ("history", i)
    gp = maps.poses[b, :n_frames].cpu().numpy()
    assert np.array_equal(bits(gp), bits(poses[:n_frames].view(np.float32).reshape(n_frames, 24))), "pose table"


def _frame_pose(rng, k):
    T = hp.rot("y", 0.02 * k) @ hp.rot("x", -0.01 * k)
    T[:3, 3] = (0.05 * k, -0.02 * k, 0.6 * k)
    return T.astype(np.float32)


CASES = [("weighted_mean", om.MERGER_STEREO_TRIANGULATION, 0), ("smoother", om.MERGER_STEREO_TRIANGULATION, 8), ("stereo_ekf", om.MERGER_STEREO_EKF, 0),
         ("depth_ekf", om.MERGER_DEPTH_EKF, 0)]


def _estimator(kind, K, bpx):
    if kind == "weighted_mean":
        return om.estimator_params(om.EST_WEIGHTED_MEAN, 4, K, max_dist2=25.0)
    if kind == "smoother":
        return om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=100.0, chi2_delta=1e-6)  # kitti.conf:503-517
    if kind == "depth_ekf":
        return om.estimator_params(om.EST_EKF, 3, K, max_dist2=1.0)  # MergerProjectiveDepthEKF + LandmarkEstimatorProjectiveEKF3D3D (icl.conf:506)
    return om.estimator_params(om.EST_EKF, 4, K, baseline_px=bpx, max_dist2=25.0, max_cov_norm2=0.25)  # kitti.conf:1-17


@pytest.mark.parametrize("kind,variant,max_meas", CASES)
@pytest.mark.parametrize("binning", [0, 1])
def test_sequence_of_frames_merges_identically(oracle, hip_ctx, kind, variant, max_meas, binning):
    """frame 0 seeds the map (no correspondences), frames 1..4 merge with descriptor-derived correspondences"""
    rng = np.random.default_rng(31)
    depth = variant == om.MERGER_DEPTH_EKF  # RGB-D: measurements are (u, v, depth) of the ICL camera
    cfg = configs.get("icl" if depth else "kitti")
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    po = oracle_merger_params(cfg, variant, _estimator(kind, K, (configs.baseline_pixels(cfg), 0.0)), enable_binning=binning,
                              target_merges=10 ** 6 if binning else 50)
    pg = _gpu_params(po)
    cap, n_frames = 2500, 6
    m = om.Map(cap, max_meas)
    poses = om.pose_table(n_frames)
    maps = ops.MapBatch(0, 1, cap, max_meas, n_frames, 1024, 1024)
    _upload_map(maps, 0, m, poses)
    prev = None
    for k in range(5):
        if depth:
            fr = hp.syn.rgbd_frame(np.random.default_rng(40 + k), cfg, 500)
            fixed, desc = fr["fixed"][:, :3].astype(np.float32), fr["desc_fixed"].copy()
        else:
            _, fixed, desc, xyz = stereo_scene(40 + k, n_kp=500)
        # the same world seen again: reuse part of the previous frame's descriptors so that correspondences exist
        if prev is not None:
            reuse = min(len(fixed), len(prev[1])) // 2
            desc[:reuse] = prev[1][:reuse]
        Tw = _frame_pose(rng, k)
        Ts = Tw.copy()  # one local map whose frame is the world
        corr = np.zeros(0, ob.CORR_DTYPE)
        if k > 0:
            # scene points are matched by descriptor equality (order = ascending scene index: unique indices)
            pairs = []
            lut = {bytes(d): i for i, d in enumerate(desc)}
            for s in range(m.n_points):
                i = lut.get(bytes(m.desc[s]))
                if i is not None:
                    pairs.append((s, i))
            pairs = pairs[: maps.corr_stride]
            corr = np.zeros(len(pairs), ob.CORR_DTYPE)
            corr["fixed_idx"] = [p[0] for p in pairs]
            corr["moving_idx"] = [p[1] for p in pairs]
            corr["response"] = rng.integers(0, 70, len(pairs)).astype(np.float32)  # some exceed maximum_distance_appearance
            assert len(corr) > 50
        rc, res = om.merge(po, Tw, Ts, poses, k, m, fixed, desc, corr)
        assert rc == 0
        _upload_frame(maps, 0, fixed, desc, corr, Tw, Ts, k)
        ops.merge_batch(hip_ctx, pg, maps)
        hip_ctx.synchronize()
        got = maps.result[0].cpu().numpy()
        assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags), (k, got, res.n_merged, res.n_added, res.flags)
        _assert_map_equal(maps, 0, m, poses, k + 1)
        prev = (fixed, desc)
    assert m.n_points > 300
    if kind not in ("stereo_ekf", "depth_ekf"):
        assert m.n_opt[: m.n_points].max() >= 2


def test_depth_ekf_merger_and_index_map(oracle, hip_ctx):
    """MergerProjectiveDepthEKF (icl.conf:506) with measurements (u, v, d) and a clipped -> scene index map"""
    rng = np.random.default_rng(5)
    cfg = configs.get("icl")
    cam = cfg["camera"]
    K = (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    est = om.estimator_params(om.EST_EKF, 3, K, max_dist2=1.0)
    po = oracle_merger_params(cfg, om.MERGER_DEPTH_EKF, est, row_bins=10, col_bins=30, target_merges=10 ** 6)
    pg = _gpu_params(po)
    fr0, fr1 = hp.syn.rgbd_frame(rng, cfg, 400), hp.syn.rgbd_frame(rng, cfg, 400)
    cap = 1200
    m = om.Map(cap, 0)
    poses = om.pose_table(3)
    maps = ops.MapBatch(0, 1, cap, 0, 3, 512, 512)
    _upload_map(maps, 0, m, poses)
    I4 = np.eye(4, dtype=np.float32)
    T1 = I4.copy()
    T1[:3, 3] = (0.01, 0.0, 0.02)
    for k, (fr, T) in enumerate(((fr0, I4), (fr1, T1))):
        fixed, desc = fr["fixed"][:, :3].astype(np.float32), fr["desc_fixed"]
        corr = np.zeros(0, ob.CORR_DTYPE)
        imap = None
        if k == 1:
            # correspondences index a clipped view of the scene: clipped i -> scene perm[i]
            perm = rng.permutation(m.n_points)[:200].astype(np.int32)
            imap = np.zeros(cap, np.int32)
            imap[: len(perm)] = perm
            corr = np.zeros(len(perm), ob.CORR_DTYPE)
            corr["fixed_idx"] = np.arange(len(perm))
            corr["moving_idx"] = rng.permutation(len(fixed))[: len(perm)]
            corr["response"] = rng.integers(0, 45, len(perm)).astype(np.float32)
            maps.scene_index_map = torch.from_numpy(imap.reshape(1, cap)).cuda()
        rc, res = om.merge(po, T, T, poses, k, m, fixed, desc, corr, imap)
        assert rc == 0
        # the device side receives the vector in the aligner's orientation (fixed -> measurement, moving -> scene)
        swapped = corr.copy()
        swapped["fixed_idx"], swapped["moving_idx"] = corr["moving_idx"], corr["fixed_idx"]
        maps.corr_from_aligner = 1
        _upload_frame(maps, 0, fixed, desc, swapped, T, T, k)
        ops.merge_batch(hip_ctx, pg, maps)
        hip_ctx.synchronize()
        got = maps.result[0].cpu().numpy()
        assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags)
        _assert_map_equal(maps, 0, m, poses, k + 1)
    assert m.n_points > 100


def test_batched_maps_and_loud_errors(oracle, hip_ctx):
    cfg = configs.get("kitti")
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    est = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=100.0)
    po = oracle_merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est, target_merges=10 ** 6)
    pg = _gpu_params(po)
    B, cap = 5, 900
    maps = ops.MapBatch(0, B, cap, 2, 4, 700, 700)
    I4 = np.eye(4, dtype=np.float32)
    ref = []
    for b in range(B):
        _, fixed, desc, xyz = stereo_scene(60 + b, n_kp=300 + 60 * b)
        m = om.Map(cap if b != 3 else 40, 2)  # map 3 is too small: PRS_ERR_SCENE_FULL
        poses = om.pose_table(4)
        rc, res = om.merge(po, I4, I4, poses, 0, m, fixed, desc, np.zeros(0, ob.CORR_DTYPE))
        ref.append((m, poses, fixed, desc, rc, res))
        _upload_frame(maps, b, fixed, desc, np.zeros(0, ob.CORR_DTYPE), I4, I4, 0)
    # map 3: emulate the small capacity by pre-filling the point count
    maps.n_points[3] = cap - 40
    ops.merge_batch(hip_ctx, pg, maps)
    hip_ctx.synchronize()
    for b, (m, poses, fixed, desc, rc, res) in enumerate(ref):
        got = maps.result[b].cpu().numpy()
        if b == 3:
            assert rc == om.ERR_SCENE_FULL and int(got[2]) == -8
            continue
        assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags)
        _assert_map_equal(maps, b, m, poses, 1)
    # second frame: identity correspondences -> a second measurement per landmark; third frame overflows the history (2)
    for step, expect in ((1, 0), (2, -7)):
        for b, (m, poses, fixed, desc, rc, res) in enumerate(ref):
            if b == 3:
                maps.n_corr[b] = 0
                maps.n_measured[b] = 0
                continue
            n = min(m.n_points, 100)
            corr = np.zeros(n, ob.CORR_DTYPE)
            corr["fixed_idx"] = corr["moving_idx"] = np.arange(n)
            # measurement i of the first frame produced landmark i only without invalid triangulations: use the
            # landmark's own first measurement as the new observation
            z = np.zeros((n, 4), np.float32)
            z[:, :3] = m.meas[:n, 0]["point_in_image"]
            z[:, 3] = z[:, 1]
            d = m.desc[:n].copy()
            rc2, res2 = om.merge(po, I4, I4, poses, step, m, z, d, corr)
            _upload_frame(maps, b, z, d, corr, I4, I4, step)
            ref[b] = (m, poses, fixed, desc, rc2, res2)
        ops.merge_batch(hip_ctx, pg, maps)
        hip_ctx.synchronize()
        for b, (m, poses, fixed, desc, rc2, res2) in enumerate(ref):
            if b == 3:
                continue
            got = maps.result[b].cpu().numpy()
            if expect == 0:
                assert rc2 == 0 and (int(got[0]), int(got[1]), int(got[2])) == (res2.n_merged, res2.n_added, res2.flags)
                _assert_map_equal(maps, b, m, poses, step + 1)
            else:
                assert rc2 == om.ERR_HISTORY and int(got[2]) == -7
    # a scene index used twice is refused
    maps2 = ops.MapBatch(0, 1, cap, 2, 4, 700, 700)
    m, poses, fixed, desc, _, _ = ref[0]
    _upload_map(maps2, 0, m, poses)
    corr = np.zeros(3, ob.CORR_DTYPE)
    corr["fixed_idx"] = (1, 2, 1)
    corr["moving_idx"] = (0, 1, 2)
    _upload_frame(maps2, 0, fixed, desc, corr, I4, I4, 3)
    ops.merge_batch(hip_ctx, pg, maps2)
    hip_ctx.synchronize()
    assert int(maps2.result[0, 2].item()) == -9


def test_randomised_sequences(oracle, hip_ctx):
    """a bounded run of tools/fuzz_merge.py: estimators, binning grids, merge targets, history capacities, frame sizes"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_merge
    bad, merges = fuzz_merge.run(12, 20200305, ctx=hip_ctx, verbose=False)
    assert not bad, bad[:3]
    assert merges > 200


def test_fused_smoother_merger_kernel(oracle, monkeypatch):
    """the pose-based smoother merger runs as front | smoother | back kernels by default; PRS_MERGE_FUSED=1 keeps it in
    one kernel (the form the phase stamps time).  Both must equal the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_merge
    monkeypatch.setenv("PRS_MERGE_FUSED", "1")
    ctx = ops.Context(0)
    try:
        bad, merges = fuzz_merge.run(8, 20200307, ctx=ctx, verbose=False)
    finally:
        ctx.close()
    assert not bad, bad[:3]
    assert merges > 100
