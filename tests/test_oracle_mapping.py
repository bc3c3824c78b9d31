"""Oracle pins for SURVEY.md 8f row 1 (landmark estimators + projective mergers): the gates of
tests/test_landmark_estimators.cpp:29-345 (LandmarkWorldNoNoise: every estimator brings the
landmarks to within 1 mm of the truth) and tests/test_mergers.cpp:357-460 (merging a cloud with
its own measurements changes nothing; merging the next frame grows the scene) restated on
seeded synthetic data."""
import numpy as np
import pytest

from oracle import binding as ob
from oracle import binding_mapping as om
from srrg2_proslam_amd import configs, synthetic as syn
from tests import helpers as hp


def _seed_landmarks(K, pts, poses, obs, max_meas):
    """createWorldWithLandmarks (tests/test_landmark_estimators.cpp:268-327)"""
    rng = np.random.default_rng(7)
    m = om.Map(len(obs[0]) + 4, max_meas)
    table = om.pose_table(len(poses))
    for k, T in enumerate(poses):
        om.set_pose(table, k, T)
    index_of = {}
    for wi, pc, z in obs[0]:
        noisy = pc + (1e-4 * rng.uniform(-1, 1, 3)).astype(np.float32)
        world = (poses[0][:3, :3] @ noisy + poses[0][:3, 3]).astype(np.float32)
        meas = np.zeros((), om.MEAS_DTYPE)
        meas["point_in_image"], meas["point_in_camera"], meas["frame"] = (z[0], z[1], pc[2]), pc, 0
        index_of[wi] = m.add_landmark(world, world, 0.01 * np.eye(3), measurement=meas)
    return m, table, index_of


def _run_estimator(params, dim, lis_from_measurement, max_meas=0):
    K, pts, poses, obs = hp.landmark_world()
    m, table, index_of = _seed_landmarks(K, pts, poses, obs, max_meas)
    assert m.n_points > 100
    n_updates = 0
    for k in range(1, len(poses)):
        for wi, pc, z in obs[k]:
            if wi not in index_of:
                continue
            meas = {2: z[:2], 3: np.array([z[0], z[1], pc[2]], np.float32), 4: z}[dim]
            rc = om.landmark_estimate(params, poses[k], poses[k], table, k, m, index_of[wi], meas, pc if lis_from_measurement else None)
            assert rc >= 0
            n_updates += 1
    assert n_updates > 500
    err = np.array([np.linalg.norm(m.state[i, :3] - pts[wi]) for wi, i in index_of.items()])
    assert err.max() < 1e-3, err.max()  # validateLandmarks: 1 mm (:337-339)
    assert np.allclose(m.coords[: m.n_points, :3], m.state[: m.n_points, :3], atol=2e-3)  # scene frame = world here
    return m


def test_L1_weighted_mean():
    K = hp.landmark_world()[0]
    m = _run_estimator(om.estimator_params(om.EST_WEIGHTED_MEAN, 4, K, max_dist2=1.0), 4, True)
    assert m.n_opt[: m.n_points].max() >= 5


@pytest.mark.parametrize("dim", [2, 3, 4])
def test_L2_L4_ekf(dim):
    K = hp.landmark_world()[0]
    p = om.estimator_params(om.EST_EKF, dim, K, baseline_px=(K[0] * 0.5, 0.0), max_dist2=1.0)
    m = _run_estimator(p, dim, False)
    touched = m.n_opt[: m.n_points] > 0
    # the filter shrinks the covariance of every landmark it updated
    assert np.all(np.linalg.norm(m.covariance[: m.n_points][touched], axis=1) < np.linalg.norm(0.01 * np.eye(3)) + 1e-6)


def test_L5_pose_based_smoother():
    K = hp.landmark_world()[0]
    p = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=1.0)
    m = _run_estimator(p, 4, True, max_meas=12)
    assert m.n_meas[: m.n_points].max() <= 10 and m.n_meas[: m.n_points].max() >= 5


def test_L6_history_overflow_is_loud():
    K, pts, poses, obs = hp.landmark_world()
    m, table, index_of = _seed_landmarks(K, pts, poses, obs, 2)
    p = om.estimator_params(om.EST_SMOOTHER, 4, K)
    wi, pc, z = next(o for o in obs[1] if o[0] in index_of)
    assert om.landmark_estimate(p, poses[1], poses[1], table, 1, m, index_of[wi], z, pc) >= 0
    assert om.landmark_estimate(p, poses[2], poses[2], table, 2, m, index_of[wi], z, pc) == om.ERR_HISTORY


# ---- mergers --------------------------------------------------------------------------------------
def merger_params(cfg, variant, est, **kw):
    cam = cfg["camera"]
    p = om.MergerParams()
    p.variant, p.enable_binning = variant, kw.get("enable_binning", 1)
    p.number_of_row_bins, p.number_of_col_bins = kw.get("row_bins", 20), kw.get("col_bins", 60)  # kitti.conf:203-207
    p.canvas_rows, p.canvas_cols = cam["rows"], cam["cols"]
    p.maximum_distance_appearance = kw.get("max_appearance", 50.0)
    p.target_number_of_merges, p.target_merge_ratio = kw.get("target_merges", 100), 0.5
    if cfg.get("triangulator"):
        p.triangulator = hp.oracle_tri_params(ob, cfg)
    p.fx, p.fy, p.cx, p.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    p.estimator = est
    return p


def stereo_scene(seed, n_kp=600):
    """a KITTI-shaped frame: fixed cloud (uL,vL,uR,vR) + descriptors from the oracle's stereo matcher,
    scene = its triangulation (tests/test_mergers.cpp: points_in_camera_00 / measurements[0])"""
    cfg, fr = hp.kitti_frame(seed, n=n_kp)
    corr, _ = ob.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], hp.oracle_stereo_params(ob, cfg["stereo_matcher"]))
    fixed = np.concatenate([fr["uv_left"][corr["fixed_idx"]], fr["uv_right"][corr["moving_idx"]]], axis=1).astype(np.float32)
    desc = fr["desc_left"][corr["fixed_idx"]]
    xyz, valid = ob.triangulate(fixed, hp.oracle_tri_params(ob, cfg))
    keep = valid.astype(bool)
    return cfg, fixed[keep], desc[keep], xyz[keep]


def _scene_map(xyz, desc, fixed, max_meas, extra=600):
    m = om.Map(len(xyz) + extra, max_meas)
    for i in range(len(xyz)):
        meas = np.zeros((), om.MEAS_DTYPE)
        meas["point_in_image"], meas["point_in_camera"], meas["frame"] = fixed[i, :3], xyz[i], 0
        m.add_landmark(xyz[i], xyz[i], np.eye(3), desc=desc[i], measurement=meas)
    return m


def _identity_corr(n):
    c = np.zeros(n, ob.CORR_DTYPE)
    c["fixed_idx"] = c["moving_idx"] = np.arange(n)
    return c


@pytest.mark.parametrize("kind", ["weighted_mean", "smoother", "stereo_ekf"])
def test_M1_merging_a_cloud_with_itself_changes_nothing(kind):
    # tests/test_mergers.cpp:357-460 (00To00): size and order intact, coordinates within 1e-5
    cfg, fixed, desc, xyz = stereo_scene(3)
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    bpx = (configs.baseline_pixels(cfg), 0.0)
    if kind == "weighted_mean":
        est, variant, mm = om.estimator_params(om.EST_WEIGHTED_MEAN, 4, K, max_dist2=25.0), om.MERGER_STEREO_TRIANGULATION, 0
    elif kind == "smoother":
        est, variant, mm = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=25.0), om.MERGER_STEREO_TRIANGULATION, 8
    else:
        est, variant, mm = om.estimator_params(om.EST_EKF, 4, K, baseline_px=bpx, max_dist2=25.0), om.MERGER_STEREO_EKF, 0
    m = _scene_map(xyz, desc, fixed, mm)
    before = m.copy()
    I4 = np.eye(4, dtype=np.float32)
    poses = om.pose_table(4)
    om.set_pose(poses, 0, I4)
    p = merger_params(cfg, variant, est, enable_binning=0, target_merges=0)
    rc, res = om.merge(p, I4, I4, poses, 1, m, fixed, desc, _identity_corr(len(xyz)))
    assert rc == 0 and m.n_points == before.n_points and res.n_added == 0
    assert res.n_merged > 0.9 * len(xyz)
    assert np.abs(m.coords[: m.n_points, :3] - before.coords[: m.n_points, :3]).max() < (1e-5 if kind != "stereo_ekf" else 1e-3)


def test_M2_merging_the_next_frame_grows_the_scene_with_bin_regulation():
    # tests/test_mergers.cpp:136-170 (00To01) + merger_projective_impl.cpp:89-122,210-260
    cfg, fixed0, desc0, xyz0 = stereo_scene(5)
    cfg, fixed1, desc1, xyz1 = stereo_scene(6)
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    est = om.estimator_params(om.EST_WEIGHTED_MEAN, 4, K, max_dist2=25.0)
    m = _scene_map(xyz0, desc0, fixed0, 0, extra=len(fixed1) + 8)
    I4 = np.eye(4, dtype=np.float32)
    poses = om.pose_table(4)
    n0 = m.n_points
    # a handful of (arbitrary) correspondences: frame-1 measurements 0..49 against scene points 0..49
    c = _identity_corr(50)
    c["response"] = np.arange(50) * 2.0  # the upper half exceeds maximum_distance_appearance = 50
    p = merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est, target_merges=10 ** 6)
    rc, res = om.merge(p, I4, I4, poses, 1, m, fixed1, desc1, c)
    assert rc == 0
    assert n0 < m.n_points <= n0 + len(fixed1)  # :158-163
    assert res.n_merged <= 26
    # every added point sits in a bin of its own that no merged measurement blocks
    row_w, col_w = cfg["camera"]["rows"] / 20.0, cfg["camera"]["cols"] / 60.0
    bins_added = set()
    added_desc = m.desc[n0: m.n_points]
    for d in added_desc:
        i = int(np.nonzero((desc1 == d).all(axis=1))[0][0])
        b = (int(np.round(np.float32(fixed1[i, 1]) / np.float32(row_w))), int(np.round(np.float32(fixed1[i, 0]) / np.float32(col_w))))
        assert b not in bins_added
        bins_added.add(b)
    assert m.inlier[n0: m.n_points].all() and (m.n_opt[n0: m.n_points] == 0).all()


def test_M3_no_correspondences_adds_the_binned_measurements():
    # merger_projective_impl.cpp:55-57
    cfg, fixed, desc, xyz = stereo_scene(8)
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    est = om.estimator_params(om.EST_SMOOTHER, 4, K, max_dist2=25.0)
    m = om.Map(len(fixed) + 4, 4)
    I4 = np.eye(4, dtype=np.float32)
    T = I4.copy()
    T[:3, 3] = (1.0, 2.0, 3.0)
    poses = om.pose_table(2)
    rc, res = om.merge(merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est), T, I4, poses, 0, m, fixed, desc, np.zeros(0, ob.CORR_DTYPE))
    assert rc == 0 and 0 < m.n_points <= len(fixed) and res.n_added == m.n_points
    # landmarks are initialised in the world frame (sensor_in_world * p), the scene keeps the local frame
    assert np.allclose(m.state[: m.n_points, :3], m.coords[: m.n_points, :3] + T[:3, 3], atol=1e-4)
    assert (m.n_meas[: m.n_points] == 1).all()


def test_M4_duplicate_scene_index_and_full_scene_are_loud():
    cfg, fixed, desc, xyz = stereo_scene(9)
    K = (cfg["camera"]["fx"], cfg["camera"]["fy"], cfg["camera"]["cx"], cfg["camera"]["cy"])
    est = om.estimator_params(om.EST_WEIGHTED_MEAN, 4, K, max_dist2=25.0)
    I4 = np.eye(4, dtype=np.float32)
    poses = om.pose_table(2)
    m = _scene_map(xyz, desc, fixed, 0)
    c = _identity_corr(4)
    c["fixed_idx"][3] = 0
    rc, _ = om.merge(merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est), I4, I4, poses, 0, m, fixed, desc, c)
    assert rc == om.ERR_DUPLICATE
    m2 = om.Map(3, 0)
    rc, _ = om.merge(merger_params(cfg, om.MERGER_STEREO_TRIANGULATION, est), I4, I4, poses, 0, m2, fixed, desc, np.zeros(0, ob.CORR_DTYPE))
    assert rc == om.ERR_SCENE_FULL
