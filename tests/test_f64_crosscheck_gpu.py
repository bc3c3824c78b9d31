"""The DEVICE's linearisation and Gauss-Newton loop against an order-agnostic float64 evaluation of the same normal equations
(tests/ref_pins.py linearize_f64 / gn_step_f64: numpy, no fused multiply-adds, no fixed-shape sum, no LDL^T) -- directly, without
the CPU checker in between.  "Bit-exact vs the checker" says the two float32 implementations agree with each other; this says the
device agrees with the mathematics (review r04: the float64 check reached the device only transitively)."""
import numpy as np
import pytest

import ref_pins as rp
from helpers import make_align_case
from srrg2_proslam_amd import ops

pytestmark = pytest.mark.gpu


def _P(cfg, md, weighting, **kw):
    cam, al = cfg["camera"], dict(cfg["aligner"])
    al.update(kw)
    return dict(factor_type=al["factor_type"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], cols=cam["cols"], rows=cam["rows"],
                b_lr_x=-cam["fx"] * cam.get("baseline_m", 0.0), info=tuple(al["diagonal_info"]), chi_threshold=al["chi_threshold"],
                weighting=weighting, mean_disparity=md)


@pytest.mark.parametrize("cfg_name,weighting", [("kitti", 1), ("kitti", 0), ("euroc", 1), ("tum", 0), ("icl", 0)])
def test_device_linearisation_against_float64(hip_ctx, cfg_name, weighting):
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, 17, 700, 800)
    scale = ops.info_scale_from_nopt(mp["n_opt"])
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    gf.set_local_map_in_sensor(X0)
    corr, _ = gf.compute()
    assert len(corr) > 50
    md = float(np.mean(fixed[:, 0] - fixed[:, 2])) if fixed.shape[1] == 4 else 0.0
    got = gf.linearize(ops.aligner_params(cfg, mean_disparity=md, enable_inverse_depth_weighting=weighting), X0, corr)
    H, b, chi, inl = rp.linearize_f64(_P(cfg, md, weighting), X0, corr, fixed[:, :3] if fixed.shape[1] == 4 else fixed, mp["xyz"], scale)
    H32, b32 = np.array(got.H, np.float64).reshape(6, 6), np.array(got.b, np.float64)
    assert np.linalg.norm(H32 - H) / np.linalg.norm(H) < 2e-5
    assert np.linalg.norm(b32 - b) / np.linalg.norm(b) < 2e-4
    assert got.num_inliers == inl and abs(got.chi_total - chi) / chi < 1e-4
    gf.close()


@pytest.mark.parametrize("cfg_name", ["kitti", "euroc"])
def test_device_gauss_newton_loop_against_float64(hip_ctx, cfg_name):
    """100 damped iterations on frozen correspondences, device (prs_pcf_linearize + prs_gn_step per iteration) vs float64: the poses
    stay within the 1e-4 relative Frobenius of BASELINE.json"""
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, 23, 600, 700)
    scale = ops.info_scale_from_nopt(mp["n_opt"])
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    gf.set_local_map_in_sensor(T)
    corr, _ = gf.compute()
    assert len(corr) > 50
    md = float(np.mean(fixed[:, 0] - fixed[:, 2]))
    damping = cfg["aligner"]["damping"]
    gap = ops.aligner_params(cfg, mean_disparity=md)
    P = _P(cfg, md, cfg["aligner"]["enable_inverse_depth_weighting"])
    X32 = np.asarray(X0, np.float32).reshape(4, 4).copy()
    X64 = X32.astype(np.float64)
    for _ in range(100):
        got = gf.linearize(gap, X32, corr)
        X32, rc = ops.gn_step(hip_ctx, np.array(got.H), np.array(got.b), damping, X32)
        assert rc == 0
        H, b, _, _ = rp.linearize_f64(P, X64, corr, fixed[:, :3], mp["xyz"], scale)
        X64 = rp.gn_step_f64(H, b, damping, X64)
    assert np.linalg.norm(X32 - X64) / np.linalg.norm(X64) < 1e-4
    gf.close()
