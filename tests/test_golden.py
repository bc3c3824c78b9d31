"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them bit for bit."""
import glob
import os

import numpy as np
import pytest

from helpers import aligner_params, oracle_stereo_params, oracle_tri_params, pcf_params_from_cfg
from srrg2_proslam_amd import configs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEREO = sorted(glob.glob(os.path.join(GOLD, "stereo_*.npz")))
ALIGN = sorted(glob.glob(os.path.join(GOLD, "align_*.npz")))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_fixtures_exist():
    assert len(STEREO) == 3 and len(ALIGN) == 4


@pytest.mark.parametrize("path", STEREO, ids=os.path.basename)
def test_oracle_reproduces_stereo_golden(oracle, path):
    g = np.load(path)
    cfg = configs.get(str(g["cfg_name"]))
    m = dict(cfg["stereo_matcher"])
    m["epipolar_line_thickness_pixels"] = int(g["thickness"])
    corr, flags = oracle.stereo_match(g["uv_left"], g["desc_left"], g["uv_right"], g["desc_right"], oracle_stereo_params(oracle, m))
    assert np.array_equal(corr["fixed_idx"], g["corr_fixed"]) and np.array_equal(corr["moving_idx"], g["corr_moving"])
    assert np.array_equal(corr["response"], g["corr_response"]) and flags == int(g["flags"])
    uvuv, src = oracle.stereo_assemble(g["uv_left"], g["uv_right"], corr)
    xyz, valid = oracle.triangulate(uvuv, oracle_tri_params(oracle, cfg))
    assert np.array_equal(uvuv, g["fixed_uvuv"]) and np.array_equal(_bits(xyz), _bits(g["xyz"])) and np.array_equal(valid, g["valid"])


@pytest.mark.parametrize("path", ALIGN, ids=os.path.basename)
def test_oracle_reproduces_align_golden(oracle, path):
    g = np.load(path)
    cfg = configs.get(str(g["cfg_name"]))
    scale = oracle.info_scale_from_nopt(g["n_opt"])
    f = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg, search_type=int(g["search"])))
    f.set_fixed(g["fixed"], g["desc_fixed"])
    f.set_moving(g["moving"], g["desc_moving"])
    md = oracle.mean_disparity(g["fixed"]) if g["fixed"].shape[1] == 4 else 0.0
    res, corr = oracle.align_frame(f, aligner_params(oracle, cfg, mean_disparity=md), g["fixed"], g["moving"], scale, g["X0"])
    assert np.array_equal(corr["fixed_idx"], g["corr_fixed"]) and np.array_equal(corr["moving_idx"], g["corr_moving"])
    assert np.array_equal(_bits(np.array(res.X)), _bits(g["X"]))
    assert (res.status, res.num_inliers, res.warnings) == (int(g["status"]), int(g["num_inliers"]), int(g["warnings"]))
    assert (f.search_radius, f.iteration, f.has_converged) == (int(g["radius"]), int(g["iteration"]), bool(g["converged"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", STEREO, ids=os.path.basename)
def test_hip_reproduces_stereo_golden(hip_ctx, path):
    from srrg2_proslam_amd import ops
    import torch
    g = np.load(path)
    cfg = configs.get(str(g["cfg_name"]))
    m = dict(cfg["stereo_matcher"])
    m["epipolar_line_thickness_pixels"] = int(g["thickness"])
    sp = ops.stereo_params(m, cfg["camera"]["rows"], cfg["camera"]["cols"])
    corr, flags = ops.stereo_match(hip_ctx, sp, g["uv_left"], g["desc_left"], g["uv_right"], g["desc_right"])
    assert np.array_equal(corr["fixed_idx"], g["corr_fixed"]) and np.array_equal(corr["moving_idx"], g["corr_moving"])
    assert np.array_equal(corr["response"], g["corr_response"]) and flags == int(g["flags"])
    n = len(g["uv_left"])
    frames = ops.StereoFrames(0, 1, n, epilogue=True)
    frames.upload(0, g["uv_left"], g["desc_left"], g["uv_right"], g["desc_right"])
    hip_ctx.use_torch_stream()
    ops.stereo_match_batch(hip_ctx, sp, frames, ops.triangulator_params(cfg))
    torch.cuda.synchronize()
    nf = int(frames.n_fixed[0].item())
    assert nf == len(g["fixed_uvuv"])
    assert np.array_equal(frames.fixed_uvuv[0, :nf].cpu().numpy(), g["fixed_uvuv"])
    assert np.array_equal(frames.fixed_desc[0, :nf].cpu().numpy(), g["desc_left"][g["fixed_src"]])
    out = frames.fixed_xyz[0, :nf].cpu().numpy()
    assert np.array_equal(_bits(out[:, :3]), _bits(g["xyz"])) and np.array_equal(out[:, 3] != 0, g["valid"].astype(bool))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ALIGN, ids=os.path.basename)
def test_hip_reproduces_align_golden(hip_ctx, path):
    from srrg2_proslam_amd import ops
    g = np.load(path)
    cfg = configs.get(str(g["cfg_name"]))
    f = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg, search_type=int(g["search"])))
    f.set_fixed(g["fixed"], g["desc_fixed"])
    f.set_moving(g["moving"], g["desc_moving"], ops.info_scale_from_nopt(g["n_opt"]))
    # first correspondence set + first linearisation
    f.set_local_map_in_sensor(g["X0"])
    corr0, _ = f.compute()
    assert np.array_equal(corr0["fixed_idx"], g["corr0_fixed"]) and np.array_equal(corr0["moving_idx"], g["corr0_moving"])
    lin = f.linearize(ops.aligner_params(cfg), g["X0"], corr0)
    assert np.array_equal(_bits(np.array(lin.H)), _bits(g["H0"])) and np.array_equal(_bits(np.array(lin.b)), _bits(g["b0"]))
    # full loop on a fresh finder
    f2 = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg, search_type=int(g["search"])))
    f2.set_fixed(g["fixed"], g["desc_fixed"])
    f2.set_moving(g["moving"], g["desc_moving"], ops.info_scale_from_nopt(g["n_opt"]))
    X, corr, res, flags = f2.align(ops.aligner_params(cfg), g["X0"])
    assert np.array_equal(corr["fixed_idx"], g["corr_fixed"]) and np.array_equal(corr["moving_idx"], g["corr_moving"])
    assert np.array_equal(corr["response"], g["corr_response"])
    assert np.linalg.norm(X.astype(np.float64) - g["X"].reshape(4, 4)) / np.linalg.norm(g["X"]) <= 1e-4
    assert np.array_equal(_bits(X).ravel(), _bits(g["X"]).ravel())
    assert (res.status, res.num_inliers, flags) == (int(g["status"]), int(g["num_inliers"]), int(g["warnings"]))
    st = f2.state()
    assert (int(st.search_radius_pixels), int(st.current_iteration), bool(st.has_converged)) == (int(g["radius"]), int(g["iteration"]), bool(g["converged"]))
