"""The readings of the external srrg2_solver arithmetic that prs_aligner_params exposes (kernel_weight_form, damping_form,
translation_weight_form: SURVEY.md section 8 rows a13 / a14) against the CPU checker switched to the same reading
(orc_set_variant).  Through the C-ABI; correspondences, H, b and pose bit-exact for every combination."""
import itertools

import numpy as np
import pytest

from helpers import aligner_params as oracle_aligner_params, corr_equal, make_align_case, pcf_params_from_cfg, rel_frobenius
from srrg2_proslam_amd import ops

pytestmark = pytest.mark.gpu
FORMS = [f for f in itertools.product((0, 1), (0, 1), (0, 1)) if f != (0, 0, 0)]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture
def variant(oracle):
    yield oracle.set_variant
    oracle.set_variant()  # back to the shipped definition whatever the test did


@pytest.mark.parametrize("kernel,damping,weight", FORMS)
def test_non_default_forms_full_frame(oracle, hip_ctx, variant, kernel, damping, weight):
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 40 + 4 * kernel + 2 * damping + weight, 800, 900)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    md = oracle.mean_disparity(fixed)
    variant(kernel_form=kernel, damping_form=damping, idw_form=weight)
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=md), fixed, mp["xyz"], scale, X0)
    Xr = np.array(res.X, np.float32).reshape(4, 4)
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    gap = ops.aligner_params(cfg, kernel_weight_form=kernel, damping_form=damping, translation_weight_form=weight)
    Xg, gcorr, gres, _ = gf.align(gap, X0)
    assert len(rcorr) > 100 and corr_equal(rcorr, gcorr)
    assert rel_frobenius(Xg, Xr) <= 1e-4
    assert np.array_equal(_bits(Xg), _bits(Xr))
    # one linearisation + step at the initial guess: H, b, counts and the damped step
    of2 = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of2.set_fixed(fixed, dfix)
    of2.set_moving(mp["xyz"], mp["desc"])
    of2.set_local_map_in_sensor(X0)
    corr, _ = of2.compute()
    ref = oracle.linearize(oracle_aligner_params(oracle, cfg, mean_disparity=md), X0, corr, fixed, mp["xyz"], scale)
    got = gf.linearize(gap, X0, corr)
    assert np.array_equal(_bits(np.array(ref.H)), _bits(np.array(got.H))) and np.array_equal(_bits(np.array(ref.b)), _bits(np.array(got.b)))
    assert (ref.num_inliers, ref.num_outliers) == (got.num_inliers, got.num_outliers) and ref.num_outliers > 0
    Xs, rc = oracle.gn_step(ref, cfg["aligner"]["damping"], X0)
    Xd, gc = ops.gn_step(hip_ctx, np.array(got.H), np.array(got.b), cfg["aligner"]["damping"], X0, damping_form=damping)
    assert rc == gc == 0 and np.array_equal(_bits(Xs), _bits(Xd))
    for f in (of, of2):
        f.close()
    gf.close()


def test_forms_change_the_result_and_default_is_shipped(oracle, hip_ctx, variant):
    """each switch does something (the non-default pose differs), and all-zero fields reproduce the shipped family"""
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 9, 800, 900)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    poses = {}
    for forms in [(0, 0, 0)] + [(1, 0, 0), (0, 1, 0), (0, 0, 1)]:
        gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
        gf.set_fixed(fixed, dfix)
        gf.set_moving(mp["xyz"], mp["desc"], scale)
        Xg, _, _, _ = gf.align(ops.aligner_params(cfg, kernel_weight_form=forms[0], damping_form=forms[1], translation_weight_form=forms[2]), X0)
        poses[forms] = np.asarray(Xg, np.float32).copy()
        gf.close()
    for forms in [(1, 0, 0), (0, 1, 0), (0, 0, 1)]:
        assert not np.array_equal(_bits(poses[forms]), _bits(poses[(0, 0, 0)])), forms


def test_unknown_form_is_refused(hip_ctx):
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 9, 300, 300)
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], None)
    with pytest.raises(Exception):
        gf.align(ops.aligner_params(cfg, damping_form=7), X0)
    gf.close()


def test_translation_weight_is_finite_for_hostile_disparities(oracle, hip_ctx):
    """a negative disparity over a zero / tiny mean gave a weight of -inf, and -inf * 0 (the K of an invalid correspondence) a NaN
    in the sums of the whole frame (advisor, round 4): the weight is finite now on both sides"""
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 21, 500, 600)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    of.set_local_map_in_sensor(X0)
    corr, _ = of.compute()
    bad = fixed.copy()
    bad[corr["fixed_idx"][:5], 2] = bad[corr["fixed_idx"][:5], 0] + 3.0  # negative disparities
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(bad, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    for md in (0.0, 1e-38):  # (a negative mean_disparity means "compute it on the device")
        ref = oracle.linearize(oracle_aligner_params(oracle, cfg, mean_disparity=md), X0, corr, bad, mp["xyz"], scale)
        got = gf.linearize(ops.aligner_params(cfg, mean_disparity=md), X0, corr)
        assert np.all(np.isfinite(np.array(ref.H))) and np.all(np.isfinite(np.array(got.H)))
        assert np.array_equal(_bits(np.array(ref.H)), _bits(np.array(got.H))) and np.array_equal(_bits(np.array(ref.b)), _bits(np.array(got.b)))
    of.close()
    gf.close()



@pytest.mark.parametrize("weighting", [0, 1])
def test_split_pipeline_with_and_without_translation_weighting(oracle, hip_ctx, weighting):
    """the Gauss-Newton kernel reads the translation weight from the parked operand row (1 when the weighting is off): both settings
    against the checker, whole frame loop"""
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 61 + weighting, 800, 900)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    md = oracle.mean_disparity(fixed)
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=md, enable_inverse_depth_weighting=weighting), fixed, mp["xyz"], scale, X0)
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    Xg, gcorr, _, _ = gf.align(ops.aligner_params(cfg, enable_inverse_depth_weighting=weighting), X0)
    assert len(rcorr) > 100 and corr_equal(rcorr, gcorr)
    assert np.array_equal(_bits(Xg), _bits(np.array(res.X, np.float32).reshape(4, 4)))
    of.close()
    gf.close()
