"""HIP: the mono filter (merge_kernel<PRS_EST_EKF, 2, 0>) and the RGB-D filter inside their landmark estimators on the scenarios of
the reference's tests/test_projective_point_ekf.cpp / tests/test_projective_depth_point_ekf.cpp (tests/ref_filters.py): every
step within the bounds those gtests assert AND bit-equal to the CPU checker (state, covariance, counters); the estimator forms no
merger of the reference drives are refused loudly."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_filters as rf  # noqa: E402
from oracle import binding_mapping as om  # noqa: E402
from srrg2_proslam_amd import _lib, ops  # noqa: E402
from test_mapping_gpu import _assert_map_equal, _gpu_params, _upload_frame, _upload_map  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scenario", rf.SCENARIOS, ids=[s[0] for s in rf.SCENARIOS])
def test_filter_scenarios_on_the_device(oracle, hip_ctx, scenario):
    name, dim, kind, s_motion, s_meas, b_err, b_cov = scenario
    pts, frames = rf.make_scenario(kind, s_motion, s_meas, dim, 11)
    m = rf.seed_map(pts)
    n_frames = rf.N_TRANSITIONS + 1
    poses = om.pose_table(n_frames)
    om.set_pose(poses, 0, np.eye(4, dtype=np.float32))
    po = rf.merger_params(dim)
    pg = _gpu_params(po)
    maps = ops.MapBatch(0, 1, m.capacity, 0, n_frames, 128, 128)
    _upload_map(maps, 0, m, poses)
    corr, desc = rf.identity_corr(), np.zeros((rf.N_LANDMARKS, 32), np.uint8)
    for k, fr in enumerate(frames):
        Tw = fr["cam_in_world_noisy"].astype(np.float32)
        rc, res = om.merge(po, Tw, Tw, poses, k + 1, m, fr["z"], desc, corr)
        assert rc == 0
        _upload_frame(maps, 0, fr["z"], desc, corr, Tw, Tw, k + 1)
        ops.merge_batch(hip_ctx, pg, maps)
        hip_ctx.synchronize()
        got = maps.result[0].cpu().numpy()
        assert (int(got[0]), int(got[1]), int(got[2])) == (res.n_merged, res.n_added, res.flags) == (rf.N_LANDMARKS, 0, 0), (name, k, got)
        rf.check_step(name, fr, maps.state[0, : rf.N_LANDMARKS, :3].cpu().numpy(), maps.covariance[0, : rf.N_LANDMARKS].cpu().numpy(), b_err, b_cov, k)
        if k % 10 == 9 or k < 3:
            _assert_map_equal(maps, 0, m, poses, k + 2)
    _assert_map_equal(maps, 0, m, poses, n_frames)


def test_estimator_forms_without_a_merger_are_refused(hip_ctx):
    po = rf.merger_params(2)
    maps = ops.MapBatch(0, 1, 64, 4, 4, 64, 64)

    def rc_of(p):
        try:
            ops.merge_batch(hip_ctx, _gpu_params(p), maps)
        except _lib.ProslamHipError as exc:
            return exc.status
        return 0

    assert rc_of(po) == 0  # the mono filter itself is served (nothing to merge: empty frame)
    po.target_number_of_merges = 100
    assert rc_of(po) == _lib.ERR_UNSUPPORTED  # ... but cannot add points
    for est_type in (om.EST_WEIGHTED_MEAN, om.EST_SMOOTHER):  # LandmarkEstimator{WeightedMean, PoseBasedSmoother}{2D3D, 3D3D}
        for dim, variant in ((2, om.MERGER_DEPTH_EKF), (3, om.MERGER_DEPTH_EKF), (3, om.MERGER_STEREO_TRIANGULATION), (4, om.MERGER_DEPTH_EKF)):
            p = rf.merger_params(dim)
            p.estimator.type, p.variant = est_type, variant
            assert rc_of(p) == _lib.ERR_UNSUPPORTED, (est_type, dim, variant)
    p = rf.merger_params(3)
    p.variant = om.MERGER_STEREO_EKF
    assert rc_of(p) == _lib.ERR_UNSUPPORTED  # 3-D measurements under a stereo merger
    p = rf.merger_params(4)
    assert rc_of(p) == _lib.ERR_UNSUPPORTED  # 4-D measurements under the depth merger
