"""HIP aligner vs the CPU oracle for the rows finished in round 2: ...WithSensor factor variants, the inlier flags of
MultiAligner3DQR (icl.conf / tum.conf), the motion-model prior slice and the constant-velocity prediction -- host-pointer
handle and batched (split and fused) pipelines -- plus the reference's aligner tests on its own KITTI / ICL images."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import ref_pins as rp
from helpers import aligner_params as oracle_aligner_params, corr_equal, make_align_case, pcf_params, pcf_params_from_cfg
from srrg2_proslam_amd import configs, ops
from test_oracle_aligner_ext import _cfg_for, _icl_case, _set_sensor, with_sensor_scene

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("mode", [2, 3, 4])
def test_with_sensor_parity(oracle, hip_ctx, mode):
    pts, desc, K, S, pose, fixed, idx, baseline_px = with_sensor_scene(mode)
    cfg = _cfg_for(K, mode, baseline_px)
    cfg["projector"] = {"range_min": 0.1, "range_max": 1000.0}
    cfg["projective_finder"] = dict(search_type=2, maximum_descriptor_distance=75.0, maximum_distance_ratio_to_second_best=0.5, minimum_matching_ratio=0.25,
                                    minimum_descriptor_distance=25.0, descriptor_distance_step_size_pixels=5.0, maximum_search_radius_pixels=50,
                                    minimum_search_radius_pixels=10, search_radius_step_size_pixels=5, minimum_number_of_iterations=10,
                                    maximum_estimate_change_norm_for_convergence=1e-5, number_of_solver_iterations_per_projection=25)
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, desc[idx])
    of.set_moving(pts, desc)
    oap = oracle_aligner_params(oracle, cfg)
    _set_sensor(oap, S)
    res, rcorr = oracle.align_frame(of, oap, fixed, pts, None, np.eye(4, dtype=np.float32))
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, desc[idx])
    gf.set_moving(pts, desc)
    gap = ops.set_sensor_in_robot(ops.aligner_params(cfg, mean_disparity=0.0, stop_at_fixed_point=0), S)
    Xg, gcorr, gres, _ = gf.align(gap, np.eye(4, dtype=np.float32))
    assert corr_equal(rcorr, gcorr) and len(rcorr) > 40
    assert np.array_equal(_bits(np.array(res.X).reshape(4, 4)), _bits(Xg))
    err = oracle.t2tnq(oracle.se3_mul(Xg, pose.astype(np.float32)))
    assert np.all(np.abs(err[:3]) < 0.15) and np.all(np.abs(err[3:]) < 0.005), err  # tests/test_aligners.cpp:271-278


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 7)])
@pytest.mark.parametrize("stop", [0, 1])
def test_inlier_flags_parity_host_handle(oracle, hip_ctx, flags, stop):
    cfg, fixed, dfix, mp, T, X0, bad = _icl_case()
    kw = dict(enable_inlier_only_runs=flags[0], keep_only_inlier_correspondences=flags[1], inlier_only_iterations=flags[2])
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, **kw), fixed, mp["xyz"], None, X0)
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"])
    Xg, gcorr, gres, _ = gf.align(ops.aligner_params(cfg, stop_at_fixed_point=stop, **kw), X0)
    assert corr_equal(rcorr, gcorr)
    assert np.array_equal(_bits(np.array(res.X).reshape(4, 4)), _bits(Xg))
    assert (res.status, res.num_inliers, res.num_correspondences, res.iterations) == (gres.status, gres.num_inliers, gres.num_correspondences, gres.iterations)
    if flags[1]:
        assert len(gcorr) == gres.num_inliers and not set(gcorr["fixed_idx"].tolist()) & set(bad.tolist())


def _batched_case(oracle, cfg_name, B, seed0, n_kp=320, n_mv=420, **al_kw):
    cases, refs = [], []
    for b in range(B):
        cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, seed0 + b, n_kp, n_mv)
        if b % 3 == 1 and fixed.shape[1] == 3:
            fixed = fixed.copy()
            fixed[:: 9, 2] += 3.0  # gross depth outliers: the kernelised class is populated
        scale = oracle.info_scale_from_nopt(mp["n_opt"])
        cases.append((fixed, dfix, mp, scale, X0))
    return cfg, cases


@pytest.mark.parametrize("cfg_name", ["icl", "tum", "kitti"])
@pytest.mark.parametrize("fused", [0, 1])
def test_inlier_flags_and_prior_parity_batched(oracle, cfg_name, fused):
    """prs_align_batch_run (split pipeline and, with PRS_FUSED_ALIGN=1 in a child process, the fused kernel)"""
    if fused:
        env = dict(os.environ, PRS_FUSED_ALIGN="1", PRS_EXT_CHILD="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__ + "::test_inlier_flags_and_prior_parity_batched", "-k", "0-" + cfg_name, "-m", "gpu"],
                           env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    B = 12
    kw = dict(enable_inlier_only_runs=1, keep_only_inlier_correspondences=1, inlier_only_iterations=9)
    cfg, cases = _batched_case(oracle, cfg_name, B, 300)
    ctx = ops.Context(0)
    ctx.use_torch_stream()
    try:
        fs = max(len(c[0]) for c in cases) + 5
        ms = max(len(c[2]["xyz"]) for c in cases) + 3
        frames = ops.AlignFrames(0, B, fs, ms)
        rng = np.random.default_rng(17)
        Z = np.stack([np.asarray(c[4], np.float32) for c in cases])  # prior mean = the initial guess
        frames.prior_mean = torch.from_numpy(Z.reshape(B, 16).copy()).cuda()
        for b, (fixed, dfix, mp, scale, X0) in enumerate(cases):
            frames.upload(b, fixed, dfix, mp["xyz"], scale, mp["desc"], X0)
        gap = ops.set_motion_prior(ops.aligner_params(cfg, stop_at_fixed_point=0, **kw), (3.0, 3.0, 3.0, 50.0, 50.0, 50.0))
        ops.align_batch(ctx, ops.pcf_params(cfg), gap, frames)
        ctx.synchronize()
        for b, (fixed, dfix, mp, scale, X0) in enumerate(cases):
            of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
            of.set_fixed(fixed, dfix)
            of.set_moving(mp["xyz"], mp["desc"])
            md = oracle.mean_disparity(fixed) if fixed.shape[1] == 4 else 0.0
            oap = oracle_aligner_params(oracle, cfg, mean_disparity=md, **kw)
            oap.enable_motion_prior = 1
            for i, v in enumerate((3.0, 3.0, 3.0, 50.0, 50.0, 50.0)):
                oap.motion_prior_info[i] = v
            res, rcorr = oracle.align_frame(of, oap, fixed, mp["xyz"], scale, X0, prior_mean=Z[b])
            X, gres = frames.X[b].cpu().numpy().reshape(4, 4), frames.result_of(b)
            assert corr_equal(rcorr, frames.corr_of(b)), b
            assert np.array_equal(_bits(np.array(res.X).reshape(4, 4)), _bits(X)), b
            assert (res.status, res.num_inliers, res.num_correspondences, res.iterations) == (gres.status, gres.num_inliers, gres.num_correspondences, gres.iterations), b
    finally:
        ctx.close()


def test_motion_predict_batch(oracle, hip_ctx):
    rng = np.random.default_rng(2)
    from srrg2_proslam_amd import synthetic as syn
    p2 = np.stack([syn.make_transform(rng.normal(0, 1, 3), rng.normal(0, 0.2, 3)) for _ in range(300)]).astype(np.float32)
    p1 = np.stack([(p2[i].astype(np.float64) @ syn.make_transform(rng.normal(0, 0.5, 3), rng.normal(0, 0.05, 3))).astype(np.float32) for i in range(300)])
    hip_ctx.use_torch_stream()
    d2, d1 = torch.from_numpy(p2).cuda(), torch.from_numpy(p1).cuda()
    out = torch.zeros_like(d1)
    ops.motion_predict_batch(hip_ctx, d2, d1, out)
    hip_ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(300):
        assert np.array_equal(_bits(got[i]), _bits(oracle.motion_predict(p2[i], p1[i]))), i


class _Hip:
    """the aligner call of tests/ref_pins.py on the HIP path"""
    name = "hip"

    def __init__(self, ctx):
        from test_ref_pins_gpu import HipBackend
        self._b = HipBackend(ctx)
        self.ctx = ctx

    def __getattr__(self, k):
        return getattr(self._b, k)

    def align(self, cfg, finder, aligner, fixed, dfix, moving, dmov, X0):
        c = dict(cfg)
        c["projective_finder"], c["aligner"] = finder, aligner
        gf = ops.ProjectiveFinder(self.ctx, ops.pcf_params(c))
        gf.set_fixed(fixed, dfix)
        gf.set_moving(moving, dmov)
        X, corr, res, _ = gf.align(ops.aligner_params(c, stop_at_fixed_point=0), X0)
        return X, corr, res.status, res.num_inliers


def test_reference_aligner_tests_on_the_reference_images(oracle, hip_ctx):
    from test_ref_pins import OracleBackend
    B, O = _Hip(hip_ctx), OracleBackend()
    got, ref = rp.kitti_aligner_circle(B), rp.kitti_aligner_circle(O)
    err = got["error"]
    assert got["status"] == 1  # tests/test_aligners.cpp:1246
    assert abs(err[0]) < 0.05 and abs(err[1]) < 0.05 and abs(err[2]) < 0.20 and np.all(np.abs(err[3:]) < 0.01), err  # :1255-1260
    assert np.array_equal(_bits(got["X"]), _bits(ref["X"])) and got["n_corr"] == ref["n_corr"]
    got, ref = rp.icl_aligner_depth(B), rp.icl_aligner_depth(O)
    assert got["status"] == 1 and np.all(np.abs(got["error"]) < 0.01), got["error"]  # :1089, :1098-1103
    assert np.array_equal(_bits(got["X"]), _bits(ref["X"])) and got["n_corr"] == ref["n_corr"]


@pytest.mark.parametrize("sequence", ["city", "highway", "city_gt"])
@pytest.mark.parametrize("weighting", [0, 1])
def test_kitti_factor_level_gn_on_the_reference_images(oracle, hip_ctx, weighting, sequence):
    """tests/test_aligners.cpp:640-759 (city 00 -> 01) and :762-880 (highway 274 -> 275) on the device: circle finder at the perfect
    estimate, then 100 prs_pcf_linearize + prs_gn_step iterations from identity; the reference's bounds on the result and every
    iterate equal to the CPU checker's"""
    from test_ref_pins import OracleBackend
    B, O = _Hip(hip_ctx), OracleBackend()
    fix = rp.highway_fixture(B) if sequence == "highway" else rp.kitti_fixture(B)
    ofix = rp.highway_fixture(O) if sequence == "highway" else rp.kitti_fixture(O)
    relative = rp.highway_relative() if sequence == "highway" else rp.kitti_relative(1, 0)
    m1, d1, p0 = fix["meas"][1], fix["desc"][1], fix["points_in_camera_00"]
    assert np.array_equal(m1, ofix["meas"][1]) and np.array_equal(_bits(p0), _bits(ofix["points_in_camera_00"]))
    fp = rp.finder_params(rp.KITTI_K, rp.CIRCLE, 0.1, 1000.0, max_dist=100.0, min_dist=100.0, ratio=0.5, min_ratio=0.1, max_radius=5, min_radius=5)
    gf, of = B.finder(fp), O.finder(fp)
    for f in (gf, of):
        f.set_fixed(m1, d1)
        f.set_moving(p0, fix["desc"][0])
        f.set_local_map_in_sensor(np.linalg.inv(relative))
    for _ in range(100):
        corr, _ = gf.compute()
        ocorr, _ = of.compute()
    assert corr_equal(corr, ocorr) and len(corr) > 15
    bound_t = 0.1
    if sequence == "city_gt":  # tests/test_aligners.cpp:586-638: ground-truth correspondences, bound 0.15 m
        if weighting:
            pytest.skip("the ground-truth variant runs without disparity weighting")
        corr = ocorr = rp.kitti_gt_correspondences(O)
        bound_t = 0.15
    md = float(np.mean(m1[corr["fixed_idx"], 0] - m1[corr["fixed_idx"], 2]))
    cfg = configs.get("kitti")
    gap = ops.aligner_params(cfg, mean_disparity=md, stop_at_fixed_point=0, chi_threshold=1000.0, enable_inverse_depth_weighting=weighting, damping=0.0)
    oap = oracle_aligner_params(oracle, cfg, mean_disparity=md, chi_threshold=1000.0, enable_inverse_depth_weighting=weighting, damping=0.0)
    Xg, Xo = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
    for _ in range(100):
        res = gf.linearize(gap, Xg, corr)
        Xg, _ = ops.gn_step(hip_ctx, np.array(res.H, np.float32), np.array(res.b, np.float32), 0.0, Xg)
        s = oracle.linearize(oap, Xo, ocorr, m1, p0, None)
        Xo, _ = oracle.gn_step(s, 0.0, Xo)
        assert np.array_equal(_bits(Xg), _bits(Xo))
    err = rp.t2tnq(Xg.astype(np.float64) @ relative)
    assert np.all(np.abs(err[:3]) < bound_t) and np.all(np.abs(err[3:]) < 0.005), err


def test_aligner_with_bruteforce_finder_on_the_reference_images(oracle, hip_ctx):
    """the brute-force-finder aligner tests of the reference on the device: brute-force matcher kernel for the correspondences, then
    100 x (prs_pcf_linearize, prs_gn_step); the reference's bounds, and every iterate equal to the CPU checker's"""
    from test_ref_pins import OracleBackend
    B, O = _Hip(hip_ctx), OracleBackend()
    for case, ocase in zip(rp.aligner_bruteforce_cases(B), rp.aligner_bruteforce_cases(O)):
        assert corr_equal(case["corr"], ocase["corr"]) and np.array_equal(_bits(case["moving"]), _bits(ocase["moving"]))
        cfg, al = case["cfg"], case["cfg"]["aligner"]
        md = oracle.mean_disparity(case["fixed"]) if al["factor_type"] == 4 else 0.0
        gap = ops.aligner_params(cfg, mean_disparity=md, stop_at_fixed_point=0)
        oap = oracle_aligner_params(oracle, cfg, mean_disparity=md)
        gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
        gf.set_fixed(case["fixed"], case["fixed_desc"])
        gf.set_moving(case["moving"], case["moving_desc"])
        Xg, Xo = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
        for _ in range(al["max_iterations"]):
            res = gf.linearize(gap, Xg, case["corr"])
            Xg, _ = ops.gn_step(hip_ctx, np.array(res.H, np.float32), np.array(res.b, np.float32), al["damping"], Xg)
            s = oracle.linearize(oap, Xo, ocase["corr"], case["fixed"], case["moving"], None)
            Xo, _ = oracle.gn_step(s, al["damping"], Xo)
            assert np.array_equal(_bits(Xg), _bits(Xo)), case["name"]
        assert res.num_inliers == s.num_inliers >= al["min_num_inliers"]
        err = rp.t2tnq(Xg.astype(np.float64) @ case["truth"])
        assert np.all(np.abs(err) < case["bound"]), (case["name"], err)
        gf.close()


def test_step_norm_exit_is_opt_in_and_keeps_correspondences(oracle):
    """prs_aligner_params.step_norm_exit (include/proslam_hip.h): 0 = the reference's loop (every iteration).  > 0 leaves the loop once
    the finder has latched and |dx| is below the bound: fewer iterations, the SAME correspondence vector, and a pose within the
    1e-4 relative Frobenius of BASELINE.json of the full run's (which is the checker's bit for bit)."""
    import copy
    import torch
    import bench
    from srrg2_proslam_amd import configs, synthetic as syn
    cfg = configs.get("kitti")
    frames = bench.make_unique_frames(cfg, 13, 1000, 1000, syn.seed_for(1, 0) + 5)
    out = {}
    for name, tol in (("full", 0.0), ("exit", 1e-5)):
        vcfg = copy.deepcopy(cfg)
        vcfg["aligner"]["step_norm_exit"] = tol
        w = bench.FrameWorkload(vcfg, 0, 256, 1000, 1000, 512, len(frames), 0, frames=frames)
        w.step()
        torch.cuda.synchronize()
        snap = w.snapshot()
        ok, it_exec = w.check(snap)
        out[name] = (snap, ok, it_exec)
        w.close()
        del w
    (sf, okf, itf), (se, oke, ite) = out["full"], out["exit"]
    assert okf == 1.0 and oke == 1.0
    assert itf > 90 and ite < 0.6 * itf, (itf, ite)  # the early exit really skips most of the frozen tail
    _, _, poses = bench.cpu_baseline(cfg, frames, len(frames))
    for u, (Xr, c) in enumerate(poses):
        for snap in (sf, se):
            gc = snap["corr"][u]
            assert len(gc) == len(c) and np.array_equal(gc["fixed_idx"], c["fixed_idx"]) and np.array_equal(gc["moving_idx"], c["moving_idx"]), u
        Xr = np.asarray(Xr, np.float32).reshape(4, 4)
        assert np.array_equal(sf["X"][u].reshape(4, 4).view(np.uint32), Xr.view(np.uint32)), u  # the full run is the checker's
        rel = np.linalg.norm(se["X"][u].reshape(4, 4).astype(np.float64) - Xr) / np.linalg.norm(Xr)
        assert rel <= 1e-4, (u, rel)
