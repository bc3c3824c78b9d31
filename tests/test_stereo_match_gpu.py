"""HIP stereo matcher / triangulator vs the CPU oracle, through the C-ABI (needs an MI355X)."""
import numpy as np
import pytest

from helpers import corr_equal, kitti_frame, oracle_stereo_params, oracle_tri_params
from srrg2_proslam_amd import configs, ops, synthetic as syn

pytestmark = pytest.mark.gpu


def _both(oracle, ctx, fr, m, rows, cols=0):
    """cols is reserved (the column-binned kernel of round 1 was removed)"""
    ref, rflags = oracle.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(oracle, m))
    got, gflags = ops.stereo_match(ctx, ops.stereo_params(m, rows, cols), fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
    return ref, rflags, got, gflags


@pytest.mark.parametrize("n", [1, 7, 64, 200, 513, 1000, 2000, 2048])
@pytest.mark.parametrize("thickness", [0, 1, 2])
@pytest.mark.parametrize("binned", [0])
def test_bit_exact_vs_oracle_kitti_shaped(oracle, hip_ctx, n, thickness, binned):
    cfg, fr = kitti_frame(100 + n + thickness, n, row_jitter_fraction=0.15 if thickness else 0.0)
    m = dict(cfg["stereo_matcher"])
    m["epipolar_line_thickness_pixels"] = thickness
    ref, rflags, got, gflags = _both(oracle, hip_ctx, fr, m, cfg["camera"]["rows"], cfg["camera"]["cols"] if binned else 0)
    assert corr_equal(ref, got), "correspondence vector differs (n=%d thickness=%d): %d vs %d" % (n, thickness, len(ref), len(got))
    assert rflags == gflags


def test_bit_exact_euroc_shaped(oracle, hip_ctx):
    cfg = configs.get("euroc")
    rng = np.random.default_rng(syn.seed_for(2, 0))
    fr = syn.stereo_frame(rng, cfg, 1000)
    for cols in (0,):
        ref, rflags, got, gflags = _both(oracle, hip_ctx, fr, cfg["stereo_matcher"], cfg["camera"]["rows"], cols)
        assert len(ref) > 100 and corr_equal(ref, got) and rflags == gflags


def test_large_frames_take_the_unstaged_variant(oracle, hip_ctx):
    # 4096 and 8192 keypoints per image do not fit the LDS descriptor staging
    for n in (3000, 4096, 8192):
        cfg, fr = kitti_frame(7 + n, n)
        ref, rflags, got, gflags = _both(oracle, hip_ctx, fr, cfg["stereo_matcher"], cfg["camera"]["rows"])
        assert corr_equal(ref, got) and rflags == gflags


def test_ragged_left_right_counts(oracle, hip_ctx):
    cfg, fr = kitti_frame(11, 1500)
    fr2 = dict(fr)
    fr2["uv_right"], fr2["desc_right"] = fr["uv_right"][:400], fr["desc_right"][:400]
    fr3 = dict(fr)
    fr3["uv_left"], fr3["desc_left"] = fr["uv_left"][:33], fr["desc_left"][:33]
    for cols in (0,):
        ref, rflags, got, gflags = _both(oracle, hip_ctx, fr2, cfg["stereo_matcher"], cfg["camera"]["rows"], cols)
        assert corr_equal(ref, got) and rflags == gflags
        ref, rflags, got, gflags = _both(oracle, hip_ctx, fr3, cfg["stereo_matcher"], cfg["camera"]["rows"], cols)
        assert corr_equal(ref, got) and rflags == gflags


def test_collisions_everything_on_one_row_and_duplicate_pixels(oracle, hip_ctx):
    # worst case of the serial chain: all keypoints on a single epipolar line, many identical (row, col)
    rng = np.random.default_rng(5)
    n = 600
    uvl = np.stack([rng.integers(0, 300, n), np.full(n, 17)], axis=1).astype(np.float32)
    uvr = np.stack([rng.integers(0, 300, n), np.full(n, 17)], axis=1).astype(np.float32)
    base = syn.random_descriptors(rng, 40)
    dl = syn.flip_bits(rng, base[rng.integers(0, 40, n)], 0.03)
    dr = syn.flip_bits(rng, base[rng.integers(0, 40, n)], 0.03)
    fr = {"uv_left": uvl, "desc_left": dl, "uv_right": uvr, "desc_right": dr}
    for thickness in (0, 1):
        m = {"maximum_descriptor_distance": 100.0, "maximum_distance_ratio_to_second_best": 0.9,
             "minimum_matching_ratio": 0.3, "maximum_disparity_pixels": 100, "epipolar_line_thickness_pixels": thickness}
        for cols in (0,):
            ref, rflags, got, gflags = _both(oracle, hip_ctx, fr, m, 376, cols)
            assert len(ref) > 10 and corr_equal(ref, got) and rflags == gflags


def test_P1_self_match_identity_on_gpu(hip_ctx):
    cfg, fr = kitti_frame(1, 2000)
    for cols in (0,):
        got, flags = ops.stereo_match(hip_ctx, ops.stereo_params(cfg["stereo_matcher"], 376, cols), fr["uv_left"], fr["desc_left"], fr["uv_left"], fr["desc_left"])
        assert len(got) == 2000 and np.array_equal(got["fixed_idx"], got["moving_idx"]) and np.all(got["response"] == 0)
        assert flags == 0


@pytest.mark.parametrize("cols", [0])
def test_P13_error_contract(oracle, hip_ctx, cols):
    cfg, fr = kitti_frame(13, 64)
    sp = ops.stereo_params(cfg["stereo_matcher"], 376, cols)
    e2, e32 = np.zeros((0, 2), np.float32), np.zeros((0, 32), np.uint8)
    # empty cloud => warning bits + empty output (bruteforce_impl.cpp:217-243)
    got, flags = ops.stereo_match(hip_ctx, sp, e2, e32, fr["uv_right"], fr["desc_right"])
    assert len(got) == 0 and flags & ops._lib.WARN_EMPTY_INPUT and flags & ops._lib.WARN_NO_MATCHES
    got, flags = ops.stereo_match(hip_ctx, sp, fr["uv_left"], fr["desc_left"], e2, e32)
    assert len(got) == 0 and flags & ops._lib.WARN_EMPTY_INPUT
    # unset buffers => hard error (bruteforce_impl.cpp:203-216 throws)
    import ctypes as C
    lib = ops._lib.load()
    n = C.c_int32(0)
    rc = lib.prs_stereo_match(hip_ctx._h, C.byref(sp), None, None, 5, None, None, 5, None, 5, C.byref(n))
    assert rc == ops._lib.ERR_NULL
    # keypoints outside the supported domain => loud error, never a silent wrong answer
    bad = fr["uv_left"].copy()
    bad[3, 1] = 400.0  # row >= image_rows
    with pytest.raises(ops.ProslamHipError) as ei:
        ops.stereo_match(hip_ctx, sp, bad, fr["desc_left"], fr["uv_right"], fr["desc_right"])
    assert ei.value.status == ops._lib.ERR_RANGE
    bad[3, 1] = np.nan
    with pytest.raises(ops.ProslamHipError):
        ops.stereo_match(hip_ctx, sp, bad, fr["desc_left"], fr["uv_right"], fr["desc_right"])


@pytest.mark.parametrize("binned", [0])
def test_batched_device_api_with_fused_adaptor_and_triangulator(oracle, hip_ctx, binned):
    import torch
    cfg = configs.get("kitti")
    m = dict(cfg["stereo_matcher"])
    m["epipolar_line_thickness_pixels"] = 1
    B, stride = 12, 2000
    frames = ops.StereoFrames(0, B, stride, epilogue=True)
    data = []
    for b in range(B):
        rng = np.random.default_rng(syn.seed_for(1, b))
        n = [2000, 1777, 1, 0, 1024, 1025, 300, 2000, 999, 64, 1500, 2000][b]
        fr = syn.stereo_frame(rng, cfg, max(n, 1), row_jitter_fraction=0.2, visible_fraction=0.4)
        if n == 0:
            fr = {k: v[:0] for k, v in fr.items() if k in ("uv_left", "desc_left", "uv_right", "desc_right")}
        frames.upload(b, fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"])
        data.append(fr)
    tp = ops.triangulator_params(cfg)
    hip_ctx.use_torch_stream()
    ops.stereo_match_batch(hip_ctx, ops.stereo_params(m, cfg["camera"]["rows"], cfg["camera"]["cols"] if binned else 0), frames, tp)
    torch.cuda.synchronize()
    otp = oracle_tri_params(oracle, cfg)
    for b in range(B):
        fr = data[b]
        ref, rflags = oracle.stereo_match(fr["uv_left"], fr["desc_left"], fr["uv_right"], fr["desc_right"], oracle_stereo_params(oracle, m))
        got = frames.matches_of(b)
        assert corr_equal(ref, got), "frame %d" % b
        assert int(frames.status[b].item()) == rflags
        uvuv, src = oracle.stereo_assemble(fr["uv_left"], fr["uv_right"], ref)
        nf = int(frames.n_fixed[b].item())
        assert nf == len(uvuv)
        assert np.array_equal(frames.fixed_uvuv[b, :nf].cpu().numpy(), uvuv)
        assert np.array_equal(frames.fixed_desc[b, :nf].cpu().numpy(), fr["desc_left"][src])
        xyz, valid = oracle.triangulate(uvuv, otp)
        g = frames.fixed_xyz[b, :nf].cpu().numpy()
        assert np.array_equal(g[:, :3].view(np.uint32), xyz.view(np.uint32))  # bit-exact float32
        assert np.array_equal(g[:, 3] != 0, valid.astype(bool))


def test_standalone_triangulator_bit_exact(oracle, hip_ctx):
    cfg = configs.get("kitti")
    rng = np.random.default_rng(21)
    pts = syn.sample_landmarks(rng, cfg["camera"], cfg["depth"], 5000)
    u, v, ur = syn.project_left_right(cfg["camera"], pts)
    uvuv = np.stack([u, v, ur, v + rng.integers(-1, 2, 5000)], axis=1).astype(np.float32)
    uvuv[::7, 2] = uvuv[::7, 0]  # zero disparity
    for min_disp in (1.0, 0.0):
        tp = ops.triangulator_params(cfg)
        tp.minimum_disparity_pixels = min_disp
        otp = oracle_tri_params(oracle, cfg)
        otp.minimum_disparity_pixels = min_disp
        xyz, valid = ops.triangulate(hip_ctx, tp, uvuv)
        rxyz, rvalid = oracle.triangulate(uvuv, otp)
        assert np.array_equal(valid, rvalid)
        assert np.array_equal(xyz.view(np.uint32), rxyz.view(np.uint32))
    # P9: triangulate(project(p)) = p
    good = rvalid.astype(bool) & (np.arange(5000) % 7 != 0)
    xyz, valid = ops.triangulate(hip_ctx, ops.triangulator_params(cfg), np.stack([u, v, ur, v], axis=1).astype(np.float32))
    rel = np.linalg.norm(xyz[good] - pts[good], axis=1) / np.linalg.norm(pts[good], axis=1)
    assert rel.max() < 5e-4


def test_first_generation_kernel_and_domain_fallbacks(oracle, monkeypatch):
    """stereo_match_v5.hip serves frames of <= 2048 keypoints; PRS_MATCHER_V3=1, distances above 255 and tall images
    (more padded sorted positions than its 13-bit field holds) take stereo_match.hip: same answers everywhere"""
    cfg, fr = kitti_frame(77, 1800, row_jitter_fraction=0.1)
    ctx_default = ops.Context(0)
    monkeypatch.setenv("PRS_MATCHER_V3", "1")
    ctx_first = ops.Context(0)
    monkeypatch.delenv("PRS_MATCHER_V3")
    try:
        for thickness in (0, 1):
            m = dict(cfg["stereo_matcher"])
            m["epipolar_line_thickness_pixels"] = thickness
            ref, rflags, got, gflags = _both(oracle, ctx_default, fr, m, 376)
            _, _, got1, gflags1 = _both(oracle, ctx_first, fr, m, 376)
            assert len(ref) > 300 and corr_equal(ref, got) and corr_equal(ref, got1) and rflags == gflags == gflags1
        # every distance is acceptable: outside the 8-bit distance records of the v5 kernel
        m = dict(cfg["stereo_matcher"], maximum_descriptor_distance=300.0, maximum_distance_ratio_to_second_best=0.95)
        ref, rflags, got, gflags = _both(oracle, ctx_default, fr, m, 376)
        assert len(ref) > 300 and corr_equal(ref, got) and rflags == gflags
        # 4000 image rows: 1800 + 3 * 1800 padded positions still fit; 2048 keypoints on 4000 rows do not
        tall = dict(fr)
        tall["uv_left"], tall["uv_right"] = fr["uv_left"].copy(), fr["uv_right"].copy()
        tall["uv_left"][:, 1] *= 10.0
        tall["uv_right"][:, 1] *= 10.0
        ref, rflags, got, gflags = _both(oracle, ctx_default, tall, cfg["stereo_matcher"], 4000)
        assert len(ref) > 300 and corr_equal(ref, got) and rflags == gflags
        cfg2, big = kitti_frame(78, 2048)
        big["uv_left"][:, 1] *= 10.0
        big["uv_right"][:, 1] *= 10.0
        ref, rflags, got, gflags = _both(oracle, ctx_default, big, cfg2["stereo_matcher"], 4000)
        assert len(ref) > 300 and corr_equal(ref, got) and rflags == gflags
    finally:
        ctx_default.close()
        ctx_first.close()


def test_rightmost_columns_and_long_rows(oracle, hip_ctx):
    """columns up to 32767 (the key's 15-bit column field) and epipolar rows of more than twelve keypoints
    (the tail loops of the padded-row rank / window counts)"""
    rng = np.random.default_rng(9)
    n = 900
    rows = rng.integers(0, 40, n)  # ~22 keypoints per row
    ul = rng.integers(32000, 32768, n).astype(np.float32) + rng.random(n).astype(np.float32) * 0.9
    disp = rng.integers(0, 60, n)
    uvl = np.stack([ul, rows + 0.5], axis=1).astype(np.float32)
    uvr = np.stack([ul - disp, rows + 0.25], axis=1).astype(np.float32)
    base = syn.random_descriptors(rng, n)
    dl = syn.flip_bits(rng, base, 0.02)
    dr = syn.flip_bits(rng, base, 0.02)
    perm = rng.permutation(n)
    fr = {"uv_left": uvl, "desc_left": dl, "uv_right": uvr[perm], "desc_right": dr[perm]}
    m = {"maximum_descriptor_distance": 100.0, "maximum_distance_ratio_to_second_best": 0.8,
         "minimum_matching_ratio": 0.1, "maximum_disparity_pixels": 100, "epipolar_line_thickness_pixels": 0}
    for thickness in (0, 1):
        m["epipolar_line_thickness_pixels"] = thickness
        ref, rflags, got, gflags = _both(oracle, hip_ctx, fr, m, 376)
        assert len(ref) > 200 and corr_equal(ref, got) and rflags == gflags


def test_randomised_shapes_and_parameters(oracle, hip_ctx):
    """a bounded run of tools/fuzz_matcher.py: crowded rows, duplicate pixels, ragged counts, every kernel variant"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_matcher
    bad, matches = fuzz_matcher.run(60, 20200303, ctx=hip_ctx, oracle=oracle, verbose=False)
    assert not bad, bad[:3]
    assert matches > 2000
