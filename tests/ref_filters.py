"""The synthetic filter scenarios of the reference's tests/test_projective_point_ekf.cpp (mono, measurement (u, v)) and
tests/test_projective_depth_point_ekf.cpp (RGB-D, measurement (u, v, d)), restated for the landmark estimators that own those
filters (LandmarkEstimatorProjectiveEKF3D / LandmarkEstimatorProjectiveDepthEKF3D, mapping/instances.cpp:29-30,36-37): shared by
the CPU checker test and the HIP test.

What the gtests do (fixtures.hpp:110-286, Synthetic::generateContinousTransitions): a point in front of the first camera, a
sequence of small camera transitions (rotations of at most +-5 degrees per axis and step, kept within +-45 degrees; translations of
at most +-10 cm per axis), the filter fed with the NOISY transition (Gaussian, sigma_motion on angles and translation) and the NOISY
projection (Gaussian, sigma_measurement), and after every step the assertions
    |state - true point in the camera| < bound_error      and      |covariance|_F < bound_covariance
(test_projective_point_ekf.cpp:36-39, 69-71, 104-106, 141-143; test_projective_depth_point_ekf.cpp:44-46).  The reference's random
streams (srand / std::mt19937) are not reproduced: the scenarios are the same distributions from a seeded numpy generator.
Here a scenario runs through the ESTIMATOR (the object a merger owns): state in the world frame (= first camera), measurement
covariance = minimum_state_element_covariance x I and no transition covariance (landmark_estimator_ekf_impl.cpp:10-14,26-27), 64
landmarks at once (one frame = one merge of 64 identity correspondences, no binning, nothing added)."""
import numpy as np

from oracle import binding as ob, binding_mapping as om

# fixtures.hpp:399-404, :318: fx = fy = 450, 640 x 480 image, principal point in the centre
K = (450.0, 450.0, 320.0, 240.0)
ROWS, COLS = 480, 640
N_LANDMARKS, N_TRANSITIONS = 64, 100

# (name, estimator dim, kind of transition, sigma_motion, sigma_measurement, bound on |error| [m], bound on |covariance|_F)
SCENARIOS = [
    ("mono_translations_zero_noise", 2, "translation", 0.0, 0.0, 1e-3, 10.0),        # test_projective_point_ekf.cpp:13-43 (error ~ 0 in double there)
    ("mono_translations_transition_noise", 2, "translation", 0.01, 0.0, 1.0, 10.0),   # :45-76
    ("mono_translations_measurement_noise", 2, "translation", 0.0, 1.0, 1.0, 10.0),   # :78-110
    ("mono_translations_full_noise", 2, "translation", 0.01, 1.0, 1.0, 10.0),         # :112-147
    ("depth_transforms_full_noise", 3, "transform", 0.01, 1.0, 10.0, 100.0),          # test_projective_depth_point_ekf.cpp:13-50
]


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    R = np.eye(3)
    i, j = {0: (1, 2), 1: (2, 0), 2: (0, 1)}[axis]
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def _motion(angles, t):
    T = np.eye(4)
    T[:3, :3] = _rot(0, angles[0]) @ _rot(1, angles[1]) @ _rot(2, angles[2])  # motion.rotate(x); rotate(y); rotate(z) (fixtures.hpp:196-198)
    T[:3, 3] = t
    return T


def make_scenario(kind, sigma_motion, sigma_measurement, dim, seed):
    """-> true / noisy camera-in-world poses of frames 1 .. N, true points in world (= camera 0), noisy measurements per frame"""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-2, 2, N_LANDMARKS), rng.uniform(-1.5, 1.5, N_LANDMARKS), rng.uniform(8, 12, N_LANDMARKS)], axis=1)  # (0, 0, sqrt(100)) there
    acc = np.zeros(3)
    world_in_cam_true, world_in_cam_noisy = np.eye(4), np.eye(4)
    frames = []
    for _ in range(N_TRANSITIONS):
        angles, t = np.zeros(3), np.zeros(3)
        if kind in ("rotation", "transform"):
            angles = rng.uniform(-1, 1, 3) * np.pi / 36.0
            flip = np.abs(acc + angles) > np.pi / 4
            angles[flip] = -angles[flip]
            acc += angles
        if kind in ("translation", "transform"):
            t = rng.uniform(-1, 1, 3) / 10.0
        motion = _motion(angles, t)  # point in camera k = motion * point in camera k - 1
        n_angles = rng.normal(0, sigma_motion, 3) if (sigma_motion > 0 and kind != "translation") else np.zeros(3)
        n_t = rng.normal(0, sigma_motion, 3) if (sigma_motion > 0 and kind != "rotation") else np.zeros(3)
        motion_noisy = motion @ _motion(n_angles, np.zeros(3))
        motion_noisy[:3, 3] += n_t
        world_in_cam_true = motion @ world_in_cam_true
        world_in_cam_noisy = motion_noisy @ world_in_cam_noisy
        pc = pts @ world_in_cam_true[:3, :3].T + world_in_cam_true[:3, 3]
        assert (pc[:, 2] > 1.0).all()
        z = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3], pc[:, 2]], axis=1)[:, :dim]
        if sigma_measurement > 0:
            z = z + rng.normal(0, sigma_measurement, z.shape)
        frames.append({"world_in_cam_noisy": world_in_cam_noisy.copy(), "cam_in_world_noisy": np.linalg.inv(world_in_cam_noisy),
                       "points_in_cam_true": pc, "z": z.astype(np.float32)})
    return pts, frames


def merger_params(dim):
    """an EKF merger that only updates: identity measurement covariance (minimum_state_element_covariance 1), gates wide open"""
    est = om.estimator_params(om.EST_EKF, dim, K, max_dist2=1e6, min_cov=1.0, max_cov_norm2=1e12)
    p = om.MergerParams()
    p.variant, p.enable_binning = om.MERGER_DEPTH_EKF, 0
    p.number_of_row_bins, p.number_of_col_bins = 10, 10
    p.canvas_rows, p.canvas_cols = ROWS, COLS
    p.maximum_distance_appearance, p.target_number_of_merges, p.target_merge_ratio = 1000.0, 0, 0.0  # target 0: no point is ever added
    p.fx, p.fy, p.cx, p.cy = K
    p.estimator = est
    return p


def seed_map(pts):
    m = om.Map(N_LANDMARKS + 8, 0)
    for p in pts:
        w = p.astype(np.float32)
        m.add_landmark(w, w, np.eye(3, dtype=np.float32))  # setState(ground truth), setCovariance(Identity)
    return m


def identity_corr():
    c = np.zeros(N_LANDMARKS, ob.CORR_DTYPE)
    c["fixed_idx"] = c["moving_idx"] = np.arange(N_LANDMARKS)
    return c


def check_step(name, frame, state_world, covariance, bound_error, bound_covariance, step):
    """the gtest's two assertions, in the frame the reference's filter lives in (the sensor the filter was told about)"""
    T = frame["world_in_cam_noisy"]
    in_cam = state_world.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    err = np.linalg.norm(in_cam - frame["points_in_cam_true"], axis=1)
    cov = np.linalg.norm(covariance.astype(np.float64).reshape(-1, 9), axis=1)
    assert err.max() < bound_error, (name, step, float(err.max()))
    assert cov.max() < bound_covariance, (name, step, float(cov.max()))
    return float(err.max()), float(cov.max())
