"""CPU checker: the mono and RGB-D point filters (projective_point_ekf_impl.cpp, projective_depth_point_ekf_impl.cpp) inside their
landmark estimators, on the scenarios and within the bounds of the reference's tests/test_projective_point_ekf.cpp and
tests/test_projective_depth_point_ekf.cpp (tests/ref_filters.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_filters as rf  # noqa: E402
from oracle import binding_mapping as om  # noqa: E402


def run_oracle(name, dim, kind, s_motion, s_meas, b_err, b_cov, seed=11):
    pts, frames = rf.make_scenario(kind, s_motion, s_meas, dim, seed)
    m = rf.seed_map(pts)
    poses = om.pose_table(rf.N_TRANSITIONS + 1)
    om.set_pose(poses, 0, np.eye(4, dtype=np.float32))
    p, corr, desc = rf.merger_params(dim), rf.identity_corr(), np.zeros((rf.N_LANDMARKS, 32), np.uint8)
    worst = (0.0, 0.0)
    for k, fr in enumerate(frames):
        Tw = fr["cam_in_world_noisy"].astype(np.float32)
        rc, res = om.merge(p, Tw, Tw, poses, k + 1, m, fr["z"], desc, corr)
        assert rc == 0 and res.n_merged == rf.N_LANDMARKS and res.n_added == 0, (name, k, rc, res.n_merged, res.n_added)
        e, c = rf.check_step(name, fr, m.state[: rf.N_LANDMARKS, :3], m.covariance[: rf.N_LANDMARKS], b_err, b_cov, k)
        worst = (max(worst[0], e), max(worst[1], c))
    assert m.n_points == rf.N_LANDMARKS and (m.n_opt[: rf.N_LANDMARKS] == rf.N_TRANSITIONS).all()
    return m, worst


@pytest.mark.parametrize("scenario", rf.SCENARIOS, ids=[s[0] for s in rf.SCENARIOS])
def test_filter_scenarios_stay_within_the_reference_bounds(scenario):
    _, worst = run_oracle(*scenario)
    print("%s: max |error| %.3g m, max |covariance| %.3g" % (scenario[0], worst[0], worst[1]))


def test_the_mono_filter_uses_only_u_v():
    """ProjectivePointEKF3D: a measurement's third component is not an input (projective_point_ekf_impl.cpp:16-43)"""
    name, dim, kind, sm, sz, be, bc = rf.SCENARIOS[3]
    a, _ = run_oracle(name, dim, kind, sm, sz, be, bc, seed=5)
    b, _ = run_oracle(name, dim, kind, sm, sz, be, bc, seed=5)
    assert np.array_equal(a.state, b.state) and np.array_equal(a.covariance, b.covariance)
