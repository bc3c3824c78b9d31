"""HIP projective finder + Gauss-Newton aligner vs the CPU oracle, through the C-ABI (needs an MI355X).

Bars (BASELINE.json north_star): correspondence indices bit-exact, poses within 1e-4 relative
Frobenius.  The kernels are written to reproduce the sequential float evaluation, so the tests
additionally assert bit-exact H, b and pose where that holds by construction.
"""
import numpy as np
import pytest

from helpers import (aligner_params as oracle_aligner_params, corr_equal, make_align_case, pcf_params_from_cfg,
                     rel_frobenius)
from srrg2_proslam_amd import configs, ops, synthetic as syn

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4  # relative Frobenius, BASELINE.json


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _finders(oracle, ctx, cfg, fixed, dfix, mp, scale=None, **kw):
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg, **kw))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    gf = ops.ProjectiveFinder(ctx, ops.pcf_params(cfg, **kw))
    gf.set_fixed(fixed, dfix)
    gf.set_moving(mp["xyz"], mp["desc"], scale)
    return of, gf


def _same_state(of, gf):
    st = gf.state()
    return (of.search_radius == int(st.search_radius_pixels) and of.iteration == int(st.current_iteration)
            and of.has_converged == bool(st.has_converged) and of.num_recomputes == int(st.num_recomputes)
            and np.float32(of.descriptor_distance) == np.float32(st.descriptor_distance))


@pytest.mark.parametrize("search", [0, 1, 2, 3])
@pytest.mark.parametrize("cfg_name", ["kitti", "euroc", "tum", "icl"])
def test_finder_call_by_call_parity_with_moving_estimate(oracle, hip_ctx, search, cfg_name):
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, 31 + search, 400, 500)
    of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp, search_type=search)
    rng = np.random.default_rng(3)
    for call in range(14):
        # an estimate that approaches the truth like the aligner's would
        Tk = syn.perturb(rng, T, 0.04 / (1 + call) ** 2, 0.002 / (1 + call) ** 2) if call < 8 else T
        of.set_local_map_in_sensor(Tk)
        gf.set_local_map_in_sensor(Tk)
        rc, rflags = of.compute()
        gc, gflags = gf.compute()
        assert corr_equal(rc, gc), "call %d: %d vs %d correspondences" % (call, len(rc), len(gc))
        assert rflags == gflags, "call %d flags %d vs %d" % (call, rflags, gflags)
        assert _same_state(of, gf), "call %d state" % call
    assert len(rc) > 30


def test_finder_state_carries_over_frames_and_retry_path(oracle, hip_ctx):
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 77, 600, 700)
    of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp)
    for f in (of, gf):
        f.set_local_map_in_sensor(T)
    for _ in range(12):  # converge: radius shrinks, threshold grows (:277-287)
        of.compute()
        gf.compute()
    assert of.has_converged and _same_state(of, gf) and of.search_radius == 40
    # next frame with a bad guess: low ratio -> reset + internal repeat (:228-263)
    T_bad = (syn.make_transform([0.5, 0.0, 0.0], [0.0, 0.04, 0.0]) @ T).astype(np.float32)
    for f in (of, gf):
        f.set_search_radius(10)
        f.set_descriptor_distance(75.0)
        f.set_fixed(fixed, dfix)
        f.set_local_map_in_sensor(T_bad)
    rc, rflags = of.compute()
    gc, gflags = gf.compute()
    assert rflags & oracle.WARN_RETRIED
    assert corr_equal(rc, gc) and rflags == gflags and _same_state(of, gf)
    assert np.array_equal(_bits(of.local_map_in_sensor()), _bits(gf.local_map_in_sensor()))


@pytest.mark.parametrize("cfg_name,weighting", [("kitti", 0), ("kitti", 1), ("euroc", 1), ("tum", 0), ("icl", 0)])
def test_linearize_is_bit_exact(oracle, hip_ctx, cfg_name, weighting):
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, 5, 700, 800)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    assert np.array_equal(scale, ops.info_scale_from_nopt(mp["n_opt"]))
    of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp, scale)
    of.set_local_map_in_sensor(X0)
    corr, _ = of.compute()
    assert len(corr) > 50
    md = oracle.mean_disparity(fixed) if fixed.shape[1] == 4 else 0.0
    oap = oracle_aligner_params(oracle, cfg, mean_disparity=md, enable_inverse_depth_weighting=weighting)
    ref = oracle.linearize(oap, X0, corr, fixed, mp["xyz"], scale)
    # device computes the mean disparity itself (mean_disparity < 0)
    gap = ops.aligner_params(cfg, mean_disparity=-1.0, enable_inverse_depth_weighting=weighting)
    got = gf.linearize(gap, X0, corr)
    assert np.array_equal(_bits(np.array(ref.H)), _bits(np.array(got.H)))
    assert np.array_equal(_bits(np.array(ref.b)), _bits(np.array(got.b)))
    assert (ref.num_inliers, ref.num_outliers, ref.num_invalid) == (got.num_inliers, got.num_outliers, got.num_invalid)
    assert np.float32(ref.chi_inliers) == np.float32(got.chi_inliers) and np.float32(ref.chi_total) == np.float32(got.chi_total)
    if weighting and fixed.shape[1] == 4:
        assert np.float32(md) == np.float32(got.mean_disparity)
    assert ref.num_outliers > 0  # the saturated kernel is exercised
    # the GN step is the same arithmetic
    Xr, rc = oracle.gn_step(ref, cfg["aligner"]["damping"], X0)
    Xg, gc = ops.gn_step(hip_ctx, np.array(got.H), np.array(got.b), cfg["aligner"]["damping"], X0)
    assert rc == gc == 0 and np.array_equal(_bits(Xr), _bits(Xg))


@pytest.mark.parametrize("cfg_name", ["kitti", "euroc", "tum", "icl"])
@pytest.mark.parametrize("stop", [0, 1])
def test_full_frame_alignment_parity(oracle, hip_ctx, cfg_name, stop):
    cfg, fixed, dfix, mp, T, X0 = make_align_case(cfg_name, 12, 800, 900)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp, scale)
    md = oracle.mean_disparity(fixed) if fixed.shape[1] == 4 else 0.0
    res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=md), fixed, mp["xyz"], scale, X0)
    Xr = np.array(res.X, np.float32).reshape(4, 4)
    Xg, gcorr, gres, gflags = gf.align(ops.aligner_params(cfg, stop_at_fixed_point=stop), X0)
    assert corr_equal(rcorr, gcorr)  # correspondence indices bit-exact
    assert rel_frobenius(Xg, Xr) <= POSE_TOL  # the stated tolerance
    assert np.array_equal(_bits(Xr), _bits(Xg))  # ... and in fact bit-exact
    assert (res.status, res.num_inliers, res.num_correspondences) == (gres.status, gres.num_inliers, gres.num_correspondences)
    assert res.warnings == gflags and gres.iterations == res.iterations  # (icl / tum: + the inlier-only run)
    assert _same_state(of, gf)
    if stop:
        assert gres.iterations_executed <= gres.iterations  # may leave early, never changes the result
    else:
        assert gres.iterations_executed == gres.iterations
    # and the estimate is the true motion to the reference tests' tolerance (tests/test_aligners.cpp:1252-1260)
    err = oracle.t2tnq(oracle.se3_mul(Xg, np.linalg.inv(T).astype(np.float32)))
    assert np.all(np.abs(err[:3]) < 0.05) and np.all(np.abs(err[3:]) < 0.01)


def test_alignment_with_motion_model_prior_and_sequence_of_frames(oracle, hip_ctx):
    cfg = configs.get("kitti")
    of = gf = None
    rng = np.random.default_rng(4)
    X_prev = None
    for frame in range(4):  # finder state (radius / threshold) carries across frames
        _, fixed, dfix, mp, T, X0 = make_align_case("kitti", 200 + frame, 500, 600)
        scale = oracle.info_scale_from_nopt(mp["n_opt"])
        if of is None:
            of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp, scale)
        else:
            of.set_fixed(fixed, dfix)
            of.set_moving(mp["xyz"], mp["desc"])
            gf.set_fixed(fixed, dfix)
            gf.set_moving(mp["xyz"], mp["desc"], scale)
        H0 = (np.eye(6) * rng.uniform(1, 20)).astype(np.float32)
        b0 = rng.normal(0, 0.5, 6).astype(np.float32)
        md = oracle.mean_disparity(fixed)
        res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=md), fixed, mp["xyz"], scale, X0, prior=(H0, b0))
        Xg, gcorr, gres, gflags = gf.align(ops.aligner_params(cfg), X0, prior=(H0, b0))
        assert corr_equal(rcorr, gcorr)
        assert np.array_equal(_bits(np.array(res.X)), _bits(Xg.reshape(-1)))
        assert _same_state(of, gf)
    assert of.search_radius < cfg["projective_finder"]["maximum_search_radius_pixels"]


def test_P10_recover_known_motion_on_gpu(hip_ctx):
    # tests/test_aligners.cpp:15-140,281-584 on the device path: 10 iterations from identity
    from helpers import project_points, synthetic_world
    pts, desc, K = synthetic_world(0)
    pose = syn.make_transform([0, 0, -1], [0.001, 0.001, -0.001])
    W2C = np.linalg.inv(pose)
    uvz, idx = project_points(K, pts @ W2C[:3, :3].T + W2C[:3, 3], 0.1, 1000.0)
    for mode in (2, 3, 4):
        if mode == 2:
            fixed = uvz[:, :2]
        elif mode == 3:
            fixed = uvz
        else:
            fixed = np.stack([uvz[:, 0], uvz[:, 1], uvz[:, 0] - 50.0 / uvz[:, 2], uvz[:, 1]], axis=1)
        cfg = {"camera": {"fx": K["fx"], "fy": K["fy"], "cx": K["cx"], "cy": K["cy"], "cols": K["cols"], "rows": K["rows"], "baseline_m": 50.0 / K["fx"]},
               "projector": {"range_min": 0.1, "range_max": 1000.0},
               "projective_finder": {"search_type": 0, "maximum_descriptor_distance": 75.0, "maximum_distance_ratio_to_second_best": 0.5,
                                     "minimum_matching_ratio": 0.25, "minimum_descriptor_distance": 25.0, "descriptor_distance_step_size_pixels": 5.0,
                                     "maximum_search_radius_pixels": 50, "minimum_search_radius_pixels": 10, "search_radius_step_size_pixels": 5,
                                     "minimum_number_of_iterations": 10, "maximum_estimate_change_norm_for_convergence": 1e-5,
                                     "number_of_solver_iterations_per_projection": 25},
               "aligner": {"factor_type": mode, "diagonal_info": (1.0, 1.0, 1.0), "chi_threshold": 1e4, "enable_inverse_depth_weighting": 0,
                           "damping": 0.0, "max_iterations": 10, "min_num_inliers": 6, "min_num_correspondences": 0}}
        gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
        gf.set_fixed(np.ascontiguousarray(fixed, np.float32), desc[idx])
        gf.set_moving(pts, desc)
        X, corr, res, _ = gf.align(ops.aligner_params(cfg), np.eye(4, dtype=np.float32))
        assert res.status == 1 and len(corr) > 40
        E = X.astype(np.float64) @ pose
        assert np.all(np.abs(E[:3, 3]) < 0.15)
        assert np.all(np.abs(E[:3, :3] - np.eye(3)) < 0.01)


def test_batched_device_api_matches_per_frame_results(oracle, hip_ctx):
    import torch
    cfg = configs.get("kitti")
    cases = [make_align_case("kitti", 300 + b, [600, 350, 64, 700][b % 4], [700, 400, 90, 650][b % 4]) for b in range(8)]
    frames = ops.AlignFrames(0, len(cases), 900, 900)
    for b, (_, fixed, dfix, mp, T, X0) in enumerate(cases):
        frames.upload(b, fixed, dfix, mp["xyz"], oracle.info_scale_from_nopt(mp["n_opt"]), mp["desc"], X0)
    hip_ctx.use_torch_stream()
    ops.align_batch(hip_ctx, ops.pcf_params(cfg), ops.aligner_params(cfg), frames)
    torch.cuda.synchronize()
    for b, (_, fixed, dfix, mp, T, X0) in enumerate(cases):
        scale = oracle.info_scale_from_nopt(mp["n_opt"])
        of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
        of.set_fixed(fixed, dfix)
        of.set_moving(mp["xyz"], mp["desc"])
        res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=oracle.mean_disparity(fixed)), fixed, mp["xyz"], scale, X0)
        assert corr_equal(rcorr, frames.corr_of(b)), "frame %d" % b
        assert np.array_equal(_bits(np.array(res.X)), _bits(frames.X[b].cpu().numpy())), "frame %d" % b
        gres, st = frames.result_of(b), frames.state_of(b)
        assert gres.status == res.status and gres.warnings == res.warnings
        assert int(st.search_radius_pixels) == of.search_radius and bool(st.has_converged) == of.has_converged


def test_P13_finder_error_contract(oracle, hip_ctx):
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 1, 200, 200)
    gf = ops.ProjectiveFinder(hip_ctx, ops.pcf_params(cfg))
    with pytest.raises(ops.ProslamHipError) as ei:
        gf.compute()  # fixed / moving not set -> hard error (bruteforce_impl.cpp:203-216 throws)
    assert ei.value.status == ops._lib.ERR_NULL
    gf.set_fixed(np.zeros((0, 4), np.float32), np.zeros((0, 32), np.uint8))
    gf.set_moving(mp["xyz"], mp["desc"])
    corr, flags = gf.compute()
    assert len(corr) == 0 and flags & ops._lib.WARN_NO_MATCHES
    bad = fixed.copy()
    bad[0, 1] = 5000.0  # outside the projector canvas: loud error, no silent clamp
    gf.set_fixed(bad, dfix)
    with pytest.raises(ops.ProslamHipError) as ei:
        gf.compute()
    assert ei.value.status == ops._lib.ERR_RANGE


def test_split_pipeline_equals_fused_kernel(oracle, monkeypatch):
    """PRS_MODE_ALIGN runs as a search-kernel / GN-kernel pipeline by default; PRS_FUSED_ALIGN=1 selects the single fused
    kernel (the form fixed clouds above 1024 points take).  Both must give the oracle's answer bit for bit."""
    cfg, fixed, dfix, mp, T, X0 = make_align_case("kitti", 91, 700, 800)
    scale = oracle.info_scale_from_nopt(mp["n_opt"])
    outs = []
    for fused in ("0", "1"):
        monkeypatch.setenv("PRS_FUSED_ALIGN", fused)
        ctx = ops.Context(0)
        gf = ops.ProjectiveFinder(ctx, ops.pcf_params(cfg))
        gf.set_fixed(fixed, dfix)
        gf.set_moving(mp["xyz"], mp["desc"], scale)
        X, corr, res, flags = gf.align(ops.aligner_params(cfg, stop_at_fixed_point=0), X0)
        outs.append((X.copy(), corr.copy(), res.num_inliers, res.iterations_executed, flags, gf.search_radius, gf.iteration))
        ctx.close()
    for other in outs[1:]:
        assert np.array_equal(_bits(outs[0][0]), _bits(other[0])) and corr_equal(outs[0][1], other[1])
        assert outs[0][2:] == other[2:]
    of = oracle.ProjectiveFinder(pcf_params_from_cfg(oracle, cfg))
    of.set_fixed(fixed, dfix)
    of.set_moving(mp["xyz"], mp["desc"])
    res, rcorr = oracle.align_frame(of, oracle_aligner_params(oracle, cfg, mean_disparity=oracle.mean_disparity(fixed)), fixed, mp["xyz"], scale, X0)
    assert np.array_equal(_bits(np.array(res.X)), _bits(outs[0][0]).ravel()) and corr_equal(rcorr, outs[0][1])


def test_enqueue_finish_overlaps_two_contexts_and_replays_from_a_graph(oracle):
    """prs_align_batch_enqueue / prs_align_batch_finish: two contexts (own streams) are enqueued back to back from one thread and
    finished afterwards; one of them gets a single round so that finish has to add the rest.  Results equal the blocking call's.
    Then the enqueue sequence is captured in a HIP graph (torch stream capture) and replayed on fresh inputs."""
    import torch
    cases = [make_align_case("kitti", 4100 + i, 500, 600) for i in range(6)]
    cfg = cases[0][0]
    fp, ap = ops.pcf_params(cfg), ops.aligner_params(cfg, stop_at_fixed_point=0)

    def batch():
        fr = ops.AlignFrames(0, len(cases), 700, 700)
        for b, (_, fixed, dfix, mp, T, X0) in enumerate(cases):
            fr.upload(b, fixed, dfix, mp["xyz"], oracle.info_scale_from_nopt(mp["n_opt"]), mp["desc"], X0)
        return fr

    def outputs(fr):
        torch.cuda.synchronize()
        return [(fr.X[b].cpu().numpy().view(np.uint32).copy(), fr.corr_of(b).copy(), fr.result_of(b).num_inliers, int(fr.state_of(b).current_iteration))
                for b in range(len(cases))]

    def same(a, b):
        return all(np.array_equal(x[0], y[0]) and corr_equal(x[1], y[1]) and x[2:] == y[2:] for x, y in zip(a, b))

    ref_ctx = ops.Context(0, stream="own")
    ref_fr = batch()
    ops.align_batch(ref_ctx, fp, ap, ref_fr)
    want = outputs(ref_fr)
    ca, cb = ops.Context(0, stream="own"), ops.Context(0, stream="own")
    fa, fb = batch(), batch()
    torch.cuda.synchronize()
    ops.align_batch_enqueue(ca, fp, ap, fa)             # the nominal five rounds
    ops.align_batch_enqueue(cb, fp, ap, fb, rounds=1)   # one round: finish must add the others
    with pytest.raises(ops.ProslamHipError):            # a context has one batch in flight
        ops.align_batch_enqueue(ca, fp, ap, fa)
    with pytest.raises(ops.ProslamHipError):            # ... whose launches sit on the stream it was enqueued on
        ca.use_torch_stream()
    # other operators of the SAME contexts between the two halves (they take the contexts' shared scratch slots, the enqueued
    # batch owns its working buffers): host-pointer brute-force matcher (slots 0-3), scene clipper, stereo matcher
    rng = np.random.default_rng(5)
    d1, d2 = rng.integers(0, 256, (900, 32), dtype=np.uint8), rng.integers(0, 256, (1100, 32), dtype=np.uint8)
    bf_want = oracle.bruteforce_match(d1, d2, 120.0, 0.95)[0]
    for c in (ca, cb):
        got_bf, _ = ops.bruteforce_match(c, ops.bruteforce_params(120.0, 0.95, 0.0), d1, d2)
        assert corr_equal(got_bf, bf_want)
        xyzw = np.concatenate([rng.uniform(-20, 20, (5000, 3)), np.ones((5000, 1))], axis=1).astype(np.float32)
        ops.scene_clip(c, ops.projector_params(cfg), np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32), xyzw)
    ops.align_batch_finish(cb)
    ops.align_batch_finish(ca)
    assert same(outputs(fa), want) and same(outputs(fb), want)
    ops.align_batch_finish(ca)  # nothing in flight: no-op
    # graph capture of the enqueue sequence (the scratch buffers of `ca` exist by now: nothing is allocated during capture)
    fg = batch()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ca.use_torch_stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ca.use_torch_stream()
            ops.align_batch_enqueue(ca, fp, ap, fg, rounds=6)
        fresh = batch()  # reload the in/out buffers the capture pass has already advanced
        for name in ("X", "state", "corr", "n_corr"):
            getattr(fg, name).copy_(getattr(fresh, name))
        side.synchronize()
        graph.replay()
        side.synchronize()
        ops.align_batch_finish(ca)
        assert same(outputs(fg), want)
        # a second replay: the graph does not pass through the host code, prs_align_batch_rearm arms finish's completion check
        for name in ("X", "state", "corr", "n_corr"):
            getattr(fg, name).copy_(getattr(fresh, name))
        side.synchronize()
        graph.replay()
        ops.align_batch_rearm(ca)
        with pytest.raises(ops.ProslamHipError):  # armed twice
            ops.align_batch_rearm(ca)
        ops.align_batch_finish(ca)
    assert same(outputs(fg), want)
    for c in (ref_ctx, ca, cb):
        c.close()


def test_randomised_configurations_through_the_batched_pipeline(oracle, hip_ctx):
    """a bounded run of tools/fuzz_align.py: search patterns, finder / aligner parameters, cloud sizes, LDS bounds"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_align
    bad, n_corr = fuzz_align.run(16, 20200304, ctx=hip_ctx, oracle=oracle, verbose=False)
    assert not bad, bad[:3]
    assert n_corr > 3000


@pytest.mark.parametrize("ratio", [0.5, 0.8, 0.95])
@pytest.mark.parametrize("min_dd", [10.0, 25.0, 40.0, 60.0])
def test_search_prefilter_keeps_every_candidate_the_filter_can_see(oracle, hip_ctx, ratio, min_dd):
    """The search kernel drops a candidate once its partial descriptor distance reaches the bound beyond which
    _filterCorrespondences' outcome cannot depend on it (csrc/align.hip: irrelevant_distance).  Adversarial inputs for that
    argument: a crowded image (several fixed points inside every search pattern), moving descriptors that are copies of
    fixed ones with 0..90 flipped bits (distances on both sides of the bound, many exactly at it), duplicates of the same
    fixed descriptor on neighbouring points (ties and second-lowest responses that decide the ratio test), over several
    calls so that radius / threshold adaptation and the retry path are part of it.  min_dd = 60 puts the bound above the
    kernel's cut-off for the separate pass (the plain path must give the same answers)."""
    cfg = configs.get("kitti")
    rng = np.random.default_rng(int(1000 * ratio + min_dd))
    cam = cfg["camera"]
    n = 700
    # crowded: keypoints on a jittered grid of ~22 px pitch in a 500 x 300 window
    gx, gy = np.meshgrid(np.arange(28), np.arange(25))
    uv = np.stack([gx.ravel() * 18 + 300, gy.ravel() * 12 + 40], axis=1).astype(np.float32)[:n]
    uv += rng.integers(-3, 4, uv.shape).astype(np.float32)
    depth = rng.uniform(6.0, 40.0, n).astype(np.float32)
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    xyz = np.stack([(uv[:, 0] - cx) / fx * depth, (uv[:, 1] - cy) / fy * depth, depth], axis=1).astype(np.float32)
    disparity = (fx * cam["baseline_m"] / depth).astype(np.float32)
    fixed = np.concatenate([uv, uv - np.stack([disparity, np.zeros(n, np.float32)], axis=1)], axis=1).astype(np.float32)
    base = syn.random_descriptors(rng, 40)  # few distinct descriptors: neighbours share them
    dfix = base[rng.integers(0, 40, n)].copy()
    flips = rng.integers(0, 12, n)
    for i in range(n):
        for b in rng.choice(256, flips[i], replace=False):
            dfix[i, b // 8] ^= np.uint8(1 << (b % 8))
    # the local map: the same points (identity motion) in shuffled order, descriptors = fixed ones with 0..90 flipped bits
    order = rng.permutation(n)
    dmov = dfix[order].copy()
    flips = rng.choice([0, 3, 8, 12, 16, 20, 24, 28, 31, 32, 33, 40, 48, 52, 53, 54, 55, 60, 75, 90], n)
    for i in range(n):
        for b in rng.choice(256, flips[i], replace=False):
            dmov[i, b // 8] ^= np.uint8(1 << (b % 8))
    mp = {"xyz": xyz[order].copy(), "desc": dmov}
    kw = dict(maximum_distance_ratio_to_second_best=ratio, minimum_descriptor_distance=min_dd, maximum_descriptor_distance=max(75.0, min_dd + 15.0))
    of, gf = _finders(oracle, hip_ctx, cfg, fixed, dfix, mp, **kw)
    T = np.eye(4, dtype=np.float32)
    total = 0
    for call in range(10):
        Tk = syn.perturb(rng, T, 0.02 / (1 + call), 0.001 / (1 + call)) if call < 6 else T
        of.set_local_map_in_sensor(Tk)
        gf.set_local_map_in_sensor(Tk)
        rc, rflags = of.compute()
        gc, gflags = gf.compute()
        assert corr_equal(rc, gc), "call %d: %d vs %d correspondences" % (call, len(rc), len(gc))
        assert rflags == gflags and _same_state(of, gf), "call %d" % call
        total += len(rc)
    assert total > 100
