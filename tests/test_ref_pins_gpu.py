"""The HIP path (through the C-ABI) against the golden numbers the reference's own tests assert on its own test images.
Same scenarios as tests/test_ref_pins.py (tests/ref_pins.py); every operator below runs on the GPU, the oracle is only
used by the last test to compare complete outputs."""
import numpy as np
import pytest
import torch

import ref_pins as rp
from srrg2_proslam_amd import _lib, ops

pytestmark = pytest.mark.gpu


class HipBackend:
    name = "hip"

    def __init__(self, ctx):
        self.ctx = ctx
        ctx.use_torch_stream()

    def extract(self, image, thr, target, nv, nh, selection=ops.SELECT_LIBSTDCXX, stride=2048):
        dev = torch.device("cuda", self.ctx.device)
        img = torch.from_numpy(np.ascontiguousarray(image)[None]).to(dev)
        kp = torch.zeros((1, stride, 2), dtype=torch.float32, device=dev)
        desc = torch.zeros((1, stride, 32), dtype=torch.uint8, device=dev)
        inten = torch.zeros((1, stride), dtype=torch.float32, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.extract_features_batch(self.ctx, ops.extractor_params(thr, 1, target, nv, nh, selection, 32768), img, kp, desc, n, st, inten)
        self.ctx.synchronize()
        assert int(st[0]) == 0, int(st[0])
        k = int(n[0])
        return kp[0, :k].cpu().numpy(), desc[0, :k].cpu().numpy(), inten[0, :k].cpu().numpy()

    def stereo_match(self, uvl, dl, uvr, dr, max_dist, ratio, max_disp, thickness, rows=0):
        sp = _lib.StereoParams(max_dist, ratio, 0.0, max_disp, thickness, int(rows), 0)
        return ops.stereo_match(self.ctx, sp, uvl, dl, uvr, dr)[0]

    def bruteforce(self, df, dm, max_dist, ratio):
        return ops.bruteforce_match(self.ctx, ops.bruteforce_params(max_dist, ratio), df, dm)[0]

    def triangulate(self, pts4, K, bx, min_disp):
        return ops.triangulate(self.ctx, _lib.TriangulatorParams(K["fx"], K["fy"], K["cx"], K["cy"], bx, min_disp, 1.84467e19), pts4)

    @staticmethod
    def pcf(p):
        K = p["K"]
        proj = ops.Projector(K["fx"], K["fy"], K["cx"], K["cy"], K["cols"], K["rows"], p["range_min"], p["range_max"])
        return ops.PcfParams(p["maximum_descriptor_distance"], p["maximum_distance_ratio_to_second_best"], p["minimum_matching_ratio"],
                             p["minimum_descriptor_distance"], p["descriptor_distance_step_size_pixels"], p["maximum_search_radius_pixels"],
                             p["minimum_search_radius_pixels"], p["search_radius_step_size_pixels"], p["minimum_number_of_iterations"],
                             p["maximum_estimate_change_norm_for_convergence"], p["number_of_solver_iterations_per_projection"],
                             p["search_type"], proj)

    def finder(self, p):
        f = ops.ProjectiveFinder(self.ctx, self.pcf(p))
        f.make_params = self.pcf
        return f


@pytest.fixture(scope="module")
def B(hip_ctx):
    return HipBackend(hip_ctx)


def test_feature_counts_match_the_reference(B):
    got, want = rp.feature_counts(B)
    assert got == want


def test_kitti_epipolar_matcher_counts(B):
    r = rp.kitti_epipolar(B)
    assert r["n_left"] == 446
    sm = r["self_match"]  # test_correspondence_finders.cpp:176-180
    assert len(sm) == 446 and np.array_equal(sm["fixed_idx"], sm["moving_idx"]) and np.all(sm["response"] == 0.0)
    assert len(r["t0"]) == 150 and np.all(r["t0"]["response"] <= 50.0)  # :274-277
    assert len(r["t1"]) == 241 and np.all(r["t1"]["response"] <= 50.0)  # :290-293


def test_kitti_bruteforce_matcher_counts(B):
    lr, rl = rp.kitti_bruteforce(B)
    assert len(lr) == 237 and np.all(lr["response"] <= 50.0)  # test_correspondence_finders.cpp:214-217
    assert len(rl) == len(lr)
    back = dict(zip(rl["moving_idx"].tolist(), rl["fixed_idx"].tolist()))
    assert all(back[f] == m for f, m in zip(lr["fixed_idx"].tolist(), lr["moving_idx"].tolist()))


@pytest.mark.parametrize("a,b,want", [(0, 0, 319), (0, 1, 226), (0, 50, 117)])
def test_icl_bruteforce_matcher_counts(B, a, b, want):
    ab, ba = rp.icl_bruteforce(B, a, b)
    assert len(ab) == want  # test_correspondence_finders.cpp:37,72,126
    if a == b:
        assert np.array_equal(ab["fixed_idx"], ab["moving_idx"]) and np.all(ab["response"] == 0.0)
    else:
        assert len(ba) == len(ab) and np.all(ab["response"] <= 50.0)


def test_adaptor_counts(B):
    got, want = rp.adaptor_counts(B)
    assert got == want


@pytest.mark.parametrize("search_type", [rp.KDTREE, rp.SQUARE, rp.CIRCLE, rp.RHOMBUS])
def test_icl_projective_identity(B, search_type):
    _, corr = rp.icl_projective(B, search_type, 0, np.eye(4))  # test_correspondence_finders.cpp:297-335
    assert len(corr) == 319 and np.array_equal(corr["fixed_idx"], corr["moving_idx"]) and np.all(corr["response"] == 0.0)


def test_icl_projective_identity_estimate_small_radius(B):
    _, corr = rp.icl_projective(B, rp.KDTREE, 50, np.eye(4), max_radius=10)  # :376-416
    assert len(corr) == 2 and np.all(corr["response"] <= 50.0)


def test_projective_kdtree_counts(B):
    """the counts the reference asserts for its KD-tree finder, on the HIP path (tests/test_ref_pins.py: ..kdtree_counts):
    120, 2 -> 21 (ICL, test_correspondence_finders.cpp:370, :412, :427), 82, 56, 36, 104 (KITTI, :468, :609, :552, :568);
    correspondences equal to the CPU checker's"""
    from test_ref_pins import OracleBackend
    O = OracleBackend()

    def same(a, b):
        return np.array_equal(a["fixed_idx"], b["fixed_idx"]) and np.array_equal(a["moving_idx"], b["moving_idx"]) and np.array_equal(a["response"], b["response"])

    T = np.linalg.inv(rp.icl_relative(50, 0))
    _, perfect = rp.icl_projective(B, rp.KDTREE, 50, T)
    assert len(perfect) == 120 and same(perfect, rp.icl_projective(O, rp.KDTREE, 50, T)[1])
    f, c10 = rp.icl_projective(B, rp.KDTREE, 50, np.eye(4), max_radius=10)
    f.set_params(f.make_params(rp.finder_params(rp.ICL_K, rp.KDTREE, 0.1, 10.0, max_radius=100)))  # :419 param change
    fx, mv = rp.icl_measurements(B, 50), rp.icl_measurements(B, 0)
    f.set_fixed(fx["uv"], fx["desc"])
    f.set_moving(mv["xyz"], mv["desc"])
    c100, _ = f.compute()
    assert len(c10) == 2 and len(c100) == 21
    for frame, radius, Tk, want in ((1, 10, np.linalg.inv(rp.kitti_relative(1, 0)), 82), (2, 10, np.linalg.inv(rp.kitti_relative(2, 0)), 56),
                                    (1, 10, np.eye(4), 36), (1, 100, np.eye(4), 104)):
        _, c = rp.kitti_projective(B, rp.KDTREE, frame, radius, Tk)
        assert len(c) == want and same(c, rp.kitti_projective(O, rp.KDTREE, frame, radius, Tk)[1]), (frame, radius, len(c), want)


def test_kitti_projective_circle_perfect_estimate(B):
    n_fixed, corr = rp.kitti_projective(B, rp.CIRCLE, 1, 10, np.linalg.inv(rp.kitti_relative(1, 0)))  # :474-513
    assert n_fixed == 458
    assert len(corr) == 90 and np.all(corr["response"] <= 50.0)


def test_hip_outputs_equal_the_oracle_on_the_reference_images(B, oracle):
    """beyond the counts: complete outputs, bit for bit, in both selection orders"""
    from oracle import binding_features as of
    for img, thr, target, grid in ((rp.kitti_image("left", 0), 5, 500, (3, 3)), (rp.kitti_image("right", 1), 15, 1000, (3, 3)),
                                   (rp.icl_gray(50), 5, 300, (1, 1)), (rp.load("ref_scene_flow")["left"], 5, 500, (3, 3))):
        for order_g, order_o in ((ops.SELECT_LIBSTDCXX, of.SELECT_LIBSTDCXX), (ops.SELECT_CANONICAL, of.SELECT_CANONICAL)):
            uv, desc, inten = B.extract(img, thr, target, grid[0], grid[1], selection=order_g)
            ouv, ointen, odesc = of.extract_features(of.extractor_params(thr, 1, target, grid[0], grid[1], order_o), img)
            assert np.array_equal(uv, ouv) and np.array_equal(desc, odesc) and np.array_equal(inten, ointen)
    r, ro = rp.kitti_epipolar(B), None
    uvl, dl, _ = B.extract(rp.kitti_image("left", 0), 5, 500, 3, 3)
    uvr, dr, _ = B.extract(rp.kitti_image("right", 0), 5, 500, 3, 3)
    for th, key in ((0, "t0"), (1, "t1")):
        ro = oracle.stereo_match(uvl, dl, uvr, dr, oracle.StereoParams(50.0, 0.9, 0.0, 100, th))[0]
        assert np.array_equal(r[key]["fixed_idx"], ro["fixed_idx"]) and np.array_equal(r[key]["moving_idx"], ro["moving_idx"])
        assert np.array_equal(r[key]["response"], ro["response"])


def test_kitti_bruteforce_versus_projective(B, oracle):
    from test_ref_pins import OracleBackend
    out, n_bf, n_gt = rp.kitti_bruteforce_versus_projective(B)
    ref, r_bf, r_gt = rp.kitti_bruteforce_versus_projective(OracleBackend())
    assert (n_bf, n_gt) == (r_bf, r_gt)
    for (radius, n, overlap_bf, overlap_gt, corr), (_, rn, _, _, rcorr) in zip(out, ref):
        assert n > 30 and overlap_gt > 0.6 and overlap_bf > 0.7, (radius, n, overlap_bf, overlap_gt)  # test_correspondence_finders.cpp:685-686
        assert n == rn and np.array_equal(corr["fixed_idx"], rcorr["fixed_idx"]) and np.array_equal(corr["moving_idx"], rcorr["moving_idx"])
