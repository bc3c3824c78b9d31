"""The CPU oracle against the golden numbers the reference's own tests assert on its own test images
(SURVEY.md section 8c): this is what pins the oracle.  Scenarios and expected values: tests/ref_pins.py.
The same scenarios run on the HIP path in tests/test_ref_pins_gpu.py."""
import numpy as np
import pytest

import ref_pins as rp
from oracle import binding as ob, binding_features as of


class OracleBackend:
    name = "oracle"

    def extract(self, image, thr, target, nv, nh):
        uv, inten, desc = of.extract_features(of.extractor_params(thr, 1, target, nv, nh, of.SELECT_LIBSTDCXX), image, capacity=4096)
        return uv, desc, inten

    def stereo_match(self, uvl, dl, uvr, dr, max_dist, ratio, max_disp, thickness, rows=0):
        return ob.stereo_match(uvl, dl, uvr, dr, ob.StereoParams(max_dist, ratio, 0.0, max_disp, thickness))[0]

    def bruteforce(self, df, dm, max_dist, ratio):
        return ob.bruteforce_match(df, dm, max_dist, ratio)[0]

    def triangulate(self, pts4, K, bx, min_disp):
        return ob.triangulate(pts4, ob.TriangulatorParams(K["fx"], K["fy"], K["cx"], K["cy"], bx, min_disp, 1.84467e19))

    @staticmethod
    def pcf(p):
        K = p["K"]
        proj = ob.Projector(K["fx"], K["fy"], K["cx"], K["cy"], K["cols"], K["rows"], p["range_min"], p["range_max"])
        return ob.PcfParams(p["maximum_descriptor_distance"], p["maximum_distance_ratio_to_second_best"], p["minimum_matching_ratio"],
                            p["minimum_descriptor_distance"], p["descriptor_distance_step_size_pixels"], p["maximum_search_radius_pixels"],
                            p["minimum_search_radius_pixels"], p["search_radius_step_size_pixels"], p["minimum_number_of_iterations"],
                            p["maximum_estimate_change_norm_for_convergence"], p["number_of_solver_iterations_per_projection"],
                            p["search_type"], proj)

    def finder(self, p):
        f = ob.ProjectiveFinder(self.pcf(p))
        f.make_params = self.pcf
        return f

    def align(self, cfg, finder, aligner, fixed, dfix, moving, dmov, X0):
        """MultiAligner3DQR::compute stand-in with the parameter groups of a configs.* dictionary"""
        import helpers as hp
        c = dict(cfg)
        c["projective_finder"], c["aligner"] = finder, aligner
        f = ob.ProjectiveFinder(hp.pcf_params_from_cfg(ob, c))
        f.set_fixed(fixed, dfix)
        f.set_moving(moving, dmov)
        md = ob.mean_disparity(fixed) if aligner["factor_type"] == 4 else 0.0
        ap = hp.aligner_params(ob, c, mean_disparity=md)
        res, corr = ob.align_frame(f, ap, fixed, moving, None, X0)
        return np.array(res.X, np.float32).reshape(4, 4), corr, res.status, res.num_inliers


@pytest.fixture(scope="module")
def B():
    return OracleBackend()


def test_feature_counts_match_the_reference(B):
    got, want = rp.feature_counts(B)
    assert got == want


def test_canonical_tie_break_differs_only_at_the_cut():
    """the default (detection-order) tie-break picks other members of the last response class: counts move by a few"""
    img = rp.kitti_image("left", 0)
    a = of.extract_features(of.extractor_params(5, 1, 500, 3, 3, of.SELECT_LIBSTDCXX), img)[0]
    b = of.extract_features(of.extractor_params(5, 1, 500, 3, 3, of.SELECT_CANONICAL), img)[0]
    sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
    assert abs(len(a) - len(b)) <= 5 and len(sa & sb) >= 0.93 * len(sa)


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
    assert len(rl) == len(lr)  # :225
    back = dict(zip(rl["moving_idx"].tolist(), rl["fixed_idx"].tolist()))
    assert all(back[f] == m for f, m in zip(lr["fixed_idx"].tolist(), lr["moving_idx"].tolist()))  # :226-237


@pytest.mark.parametrize("a,b,want", [(0, 0, 319), (0, 1, 226), (0, 50, 117)])
def test_icl_bruteforce_matcher_counts(B, a, b, want):
    ab, ba = rp.icl_bruteforce(B, a, b)
    assert len(ab) == want  # test_correspondence_finders.cpp:37,72,126
    if a == b:
        assert np.array_equal(ab["fixed_idx"], ab["moving_idx"]) and np.all(ab["response"] == 0.0)  # :38-41
    else:
        assert len(ba) == len(ab) and np.all(ab["response"] <= 50.0)  # :83,137
        back = dict(zip(ba["moving_idx"].tolist(), ba["fixed_idx"].tolist()))
        assert all(back[f] == m for f, m in zip(ab["fixed_idx"].tolist(), ab["moving_idx"].tolist()))


def test_adaptor_counts(B):
    got, want = rp.adaptor_counts(B)
    assert got == want
    m = rp.icl_measurements(B, 0)  # test_measurement_adaptors.cpp:78-88
    rows, cols = np.rint(m["uv"][:, 1]).astype(int), np.rint(m["uv"][:, 0]).astype(int)
    assert np.array_equal(m["intensity"], rp.icl_gray(0)[rows, cols].astype(np.float32))
    assert np.array_equal(m["depth"], rp.icl_depth_m(0)[rows, cols])


def test_numpy_adaptor_equals_oracle_assemble(B):
    uvl, dl, _ = B.extract(rp.kitti_image("left", 0), 5, 500, 3, 3)
    uvr, dr, _ = B.extract(rp.kitti_image("right", 0), 5, 500, 3, 3)
    corr = B.stereo_match(uvl, dl, uvr, dr, 100.0, 0.8, 100, 1)
    pts, src = rp.assemble(uvl, uvr, corr)
    opts, osrc = ob.stereo_assemble(uvl, uvr, corr)
    assert np.array_equal(pts, opts) and np.array_equal(src, osrc) and len(pts) < len(corr)


@pytest.mark.parametrize("search_type", [rp.KDTREE, rp.SQUARE, rp.CIRCLE, rp.RHOMBUS])
def test_icl_projective_identity(B, search_type):
    """test_correspondence_finders.cpp:297-335 asserts 319 mirror matches with the KD-tree finder; the lattice searches of any
    shape with a 100 px radius find the same"""
    _, corr = rp.icl_projective(B, search_type, 0, np.eye(4))
    assert len(corr) == 319 and np.array_equal(corr["fixed_idx"], corr["moving_idx"]) and np.all(corr["response"] == 0.0)


def test_icl_projective_identity_estimate_small_radius(B):
    """test_correspondence_finders.cpp:376-416: 00 -> 50 with an identity guess and a 10 px radius: 2 matches"""
    _, corr = rp.icl_projective(B, rp.KDTREE, 50, np.eye(4), max_radius=10)
    assert len(corr) == 2 and np.all(corr["response"] <= 50.0)


def test_icl_projective_kdtree_counts(B):
    """The KD-tree finder with srrg2_core's tree restated (single-leaf radius queries, leaf extent 3 sigma < search radius):
    120 matches at the perfect estimate (test_correspondence_finders.cpp:370), 2 at the identity estimate with a 10 px radius
    (:412) and 21 after the radius parameter is raised to 100 px on the same finder (:419-427).  An exhaustive radius search
    finds 123 / 2 / 83."""
    _, perfect = rp.icl_projective(B, rp.KDTREE, 50, np.linalg.inv(rp.icl_relative(50, 0)))
    assert len(perfect) == 120 and np.all(perfect["response"] <= 50.0)
    f, c10 = rp.icl_projective(B, rp.KDTREE, 50, np.eye(4), max_radius=10)
    f.set_params(f.make_params(rp.finder_params(rp.ICL_K, rp.KDTREE, 0.1, 10.0, max_radius=100)))  # :419 param change
    fx, mv = rp.icl_measurements(B, 50), rp.icl_measurements(B, 0)
    f.set_fixed(fx["uv"], fx["desc"])
    f.set_moving(mv["xyz"], mv["desc"])
    c100, _ = f.compute()
    assert len(c10) == 2 and len(c100) == 21 and np.all(c100["response"] <= 50.0)


def test_kitti_projective_circle_perfect_estimate(B):
    """test_correspondence_finders.cpp:474-513 (CorrespondenceFinderProjectiveCircle3D3D, radius 10, perfect estimate): 90.
    The whole chain is behind this number: extractor, epipolar matcher, adaptor, triangulator, projector, circle search,
    candidate filter."""
    n_fixed, corr = rp.kitti_projective(B, rp.CIRCLE, 1, 10, np.linalg.inv(rp.kitti_relative(1, 0)))
    assert n_fixed == 458  # :499
    assert len(corr) == 90 and np.all(corr["response"] <= 50.0)  # :509-512


def test_kitti_projective_kdtree_counts(B):
    """the KD-tree finder's counts on the KITTI fixture: 82 (00 -> 01, perfect estimate, 10 px: test_correspondence_finders.cpp:468),
    56 (00 -> 02, :609), 36 and 104 (identity estimate at 10 px and 100 px, :552, :568); exhaustive search: 89 / 64 / 41 / 108"""
    _, c = rp.kitti_projective(B, rp.KDTREE, 1, 10, np.linalg.inv(rp.kitti_relative(1, 0)))
    assert len(c) == 82
    _, c = rp.kitti_projective(B, rp.KDTREE, 2, 10, np.linalg.inv(rp.kitti_relative(2, 0)))
    assert len(c) == 56
    _, c10 = rp.kitti_projective(B, rp.KDTREE, 1, 10, np.eye(4))
    _, c100 = rp.kitti_projective(B, rp.KDTREE, 1, 100, np.eye(4))
    assert len(c10) == 36 and len(c100) == 104


def test_kitti_bruteforce_versus_projective(B):
    """tests/test_correspondence_finders.cpp:615-688: at every search radius more than 70 % of the projective matches are brute-force
    matches too and more than 60 % are ground-truth correspondences"""
    out, n_bf, n_gt = rp.kitti_bruteforce_versus_projective(B)
    assert n_bf > 30 and n_gt > 30
    for radius, n, overlap_bf, overlap_gt, _ in out:
        assert n > 30 and overlap_gt > 0.6 and overlap_bf > 0.7, (radius, n, overlap_bf, overlap_gt)  # :685-686
