"""Scene-clipper and merger scenarios of tests/test_scene_clippers.cpp and tests/test_mergers.cpp on the reference's own
KITTI / ICL data (tests/golden/ref_*.npz); shared by the CPU checker tests and the HIP tests."""
import numpy as np

import ref_pins as rp
from oracle import binding as ob, binding_mapping as om
from srrg2_proslam_amd import configs

I4 = np.eye(4, dtype=np.float32)


def _rot(axis, angle):
    c, s = np.cos(angle), np.sin(angle)
    R = np.eye(4)
    i, j = {"x": (1, 2), "z": (0, 1)}[axis]
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R.astype(np.float32)


def _tz(z):
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = z
    return T


def icl_dense_points():
    """ICL::SetUp (fixtures.hpp:651-656): every pixel of depth image 00 unprojected (307200 points)"""
    if "icl_dense" not in rp._cache:
        K, d = rp.ICL_K, rp.icl_depth_m(0)
        c, r = np.meshgrid(np.arange(d.shape[1], dtype=np.float32), np.arange(d.shape[0], dtype=np.float32))
        rp._cache["icl_dense"] = np.stack([(c - np.float32(K["cx"])) / np.float32(K["fx"]) * d, (r - np.float32(K["cy"])) / np.float32(K["fy"]) * d, d],
                                          axis=-1).reshape(-1, 3).astype(np.float32)
    return rp._cache["icl_dense"]


def clipper_cases(B):
    """(name, K, range_min, range_max, robot_in_local_map, points, expected count | (lower, upper)) per gtest of test_scene_clippers.cpp"""
    sparse, dense = rp.icl_measurements(B, 0)["xyz"], icl_dense_points()
    kitti = rp.kitti_fixture(B)["points_in_camera_00"]
    assert len(sparse) == 321 and len(dense) == 307200 and len(kitti) == 145
    icl, kit = (rp.ICL_K, 0.1, 10.0), (rp.KITTI_K, 0.1, 1000.0)
    # DenseMonocularDepthNoMotion / RotateFullPitch (:7-72) assert "<= 306671" / "<= 306783" and say in a comment that the count
    # depends on the platform's rounding of border pixels (u = 0 / v = 0 exactly); here 4 / 42 border points fall outside
    return [("icl_dense_no_motion", *icl, I4, dense, (306000, 307200)), ("icl_dense_full_pitch", *icl, _rot("z", np.pi), dense, (306000, 307200)),
            ("icl_dense_full_roll", *icl, _rot("x", np.pi), dense, 0), ("icl_dense_quarter_roll", *icl, _rot("x", np.pi / 4), dense, 49872),
            ("icl_dense_backward", *icl, _tz(-1.0), dense, 307200), ("icl_dense_forward", *icl, _tz(1.0), dense, 136022),
            ("icl_sparse_no_motion", *icl, I4, sparse, 321), ("icl_sparse_full_pitch", *icl, _rot("z", np.pi), sparse, 321),
            ("icl_sparse_full_roll", *icl, _rot("x", np.pi), sparse, 0), ("icl_sparse_quarter_roll", *icl, _rot("x", np.pi / 4), sparse, 51),
            ("icl_sparse_backward", *icl, _tz(-1.0), sparse, 321), ("icl_sparse_forward", *icl, _tz(1.0), sparse, 242),
            ("kitti_sparse_no_motion", *kit, I4, kitti, 145), ("kitti_sparse_forward", *kit, _tz(10.0), kitti, 52)]


def _hamming(a, b):
    return float(np.unpackbits(a ^ b).sum())


def icl_ideal_correspondences(m0, m1, camera_1_in_0):
    """fixtures.hpp:665-703: for every point of 00 (moved into 01) the closest unconsumed point of 01 in appearance (< 100) AND
    geometry (< 0.1 m^2), both thresholds tightening as the scan goes"""
    Ti = np.linalg.inv(camera_1_in_0).astype(np.float32)
    p = (m0["xyz"] @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    used, out = set(), []
    for i in range(len(p)):
        best_a, best_g, bi = 100.0, 0.1, -1
        for j in range(len(m1["xyz"])):
            if j in used:
                continue
            da = _hamming(m1["desc"][j], m0["desc"][i])
            dg = float(((m1["xyz"][j] - p[i]).astype(np.float32) ** 2).sum())
            if da < best_a and dg < best_g:
                best_a, best_g, bi = da, dg, j
        if bi != -1:
            out.append((i, bi, best_a))
            used.add(bi)
    c = np.zeros(len(out), ob.CORR_DTYPE)
    c["fixed_idx"], c["moving_idx"], c["response"] = [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]
    return c


def kitti_ideal_correspondences(fix):
    """fixtures.hpp:960-1035 (correspondences_camera_00_from_01): points of 00 projected into image 01 with the known motion; per
    measurement of 01 the nearest projection (< 10 px); kept if < 5 px, descriptor distance < 50 and the 00 point is still free"""
    K = rp.KITTI_K
    T = np.linalg.inv(rp.kitti_relative(1, 0)).astype(np.float32)
    p = (fix["points_in_camera_00"] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    u, v = K["fx"] * p[:, 0] / p[:, 2] + K["cx"], K["fy"] * p[:, 1] / p[:, 2] + K["cy"]
    ok = np.nonzero((p[:, 2] > 0) & (u >= 0) & (u < K["cols"]) & (v >= 0) & (v < K["rows"]))[0]
    picked, out = set(), []
    for i1 in range(len(fix["meas"][1])):
        z = fix["meas"][1][i1]
        best, best_j = 10.0, -1
        for j in ok:
            e = float(np.hypot(z[0] - u[j], z[1] - v[j]))
            if e < best:
                best, best_j = e, int(j)
        if best_j >= 0 and best < 5.0 and best_j not in picked:
            dd = _hamming(fix["desc"][1][i1], fix["desc"][0][best_j])
            if dd < 50:
                out.append((best_j, i1, dd))
                picked.add(best_j)
    c = np.zeros(len(out), ob.CORR_DTYPE)
    c["fixed_idx"], c["moving_idx"], c["response"] = [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]
    return c


def _identity_corr(n):
    c = np.zeros(n, ob.CORR_DTYPE)
    c["fixed_idx"] = c["moving_idx"] = np.arange(n)
    return c


def _merger_params(cfg, variant, est, **kw):
    from test_oracle_mapping import merger_params
    return merger_params(cfg, variant, est, **kw)


def _seed_map(xyz, desc, image_points, max_meas, capacity):
    """scene = points_in_camera_00 with allocated statistics, state = coordinates (test_mergers.cpp:268-271, :377-380); the smoother
    cases also add the first camera measurement (:425-433).  The tests never set a covariance: the landmarks carry whatever
    PointStatisticsField3D::allocate() leaves (srrg2_core, external).  ZERO is used here: the merger sets Matrix3f::Identity()
    explicitly on the points IT creates (mergers/merger_projective_impl.cpp:317-319), which would be redundant if allocate() did,
    and with an identity start the stereo EKF moves a far landmark by 7.5 m on frame 01 where the reference asserts 5
    (test_mergers.cpp:784-786); from zero the estimator's floor (minimum_state_element_covariance 0.01) applies and every
    bound of the reference holds (1.6 m)."""
    m = om.Map(capacity, max_meas)
    for i in range(len(xyz)):
        meas = None
        if max_meas > 0:
            meas = np.zeros((), om.MEAS_DTYPE)
            meas["point_in_image"], meas["point_in_camera"], meas["frame"] = image_points[i], xyz[i], 0
        m.add_landmark(xyz[i], xyz[i], np.zeros((3, 3)), desc=desc[i], measurement=meas)
    return m


def merger_cases(B):
    """one dict per gtest of test_mergers.cpp that uses a merger of SURVEY 8f row 1"""
    icl, kit = dict(configs.get("icl")), dict(configs.get("kitti"))
    Ki = (rp.ICL_K["fx"], rp.ICL_K["fy"], rp.ICL_K["cx"], rp.ICL_K["cy"])
    Kk = (rp.KITTI_K["fx"], rp.KITTI_K["fy"], rp.KITTI_K["cx"], rp.KITTI_K["cy"])
    m0, m1 = rp.icl_measurements(B, 0), rp.icl_measurements(B, 1)
    uvd = lambda m: np.concatenate([m["uv"], m["depth"][:, None]], axis=1).astype(np.float32)  # noqa: E731
    cases = []
    # ICL 00To00 / 00To01_MergerCorrespondenceProjectiveDepthEKF_Sparse (:248-355): MergerProjectiveDepthEKF, 10 x 30 bins, appearance 50;
    # bounds 1e-5 (:293-295) and 0.1 (:351-353)
    for name, meas, corr, d2, want, tol in (("icl_depth_ekf_00_to_00", m0, _identity_corr(321), 1.0, 321, 1e-5),
                                            ("icl_depth_ekf_00_to_01", m1, icl_ideal_correspondences(m0, m1, rp.icl_relative(1, 0)), 0.01, 337, 0.1)):
        est = om.estimator_params(om.EST_EKF, 3, Ki, max_dist2=d2)
        p = _merger_params(icl, om.MERGER_DEPTH_EKF, est, row_bins=10, col_bins=30, max_appearance=50.0, target_merges=1000)
        cases.append(dict(name=name, params=p, map=_seed_map(m0["xyz"], m0["desc"], None, 0, 1024), T=I4, fixed=uvd(meas), desc=meas["desc"], corr=corr,
                          size=want, tol=tol, n_frames=4))
    fix = rp.kitti_fixture(B)
    xyz0, n0 = fix["points_in_camera_00"], len(fix["points_in_camera_00"])
    image00 = np.stack([fix["meas"][0][:, 0], fix["meas"][0][:, 1], xyz0[:, 2]], axis=1).astype(np.float32)  # points_00_in_image_00: (u, v, depth)
    c01 = kitti_ideal_correspondences(fix)
    T01 = rp.kitti_relative(1, 0).astype(np.float32)
    bpx = (rp.KITTI_BX, 0.0)
    ests = {"weighted_mean": (om.MERGER_STEREO_TRIANGULATION, 0, lambda d2: om.estimator_params(om.EST_WEIGHTED_MEAN, 4, Kk, max_dist2=d2)),
            "smoother": (om.MERGER_STEREO_TRIANGULATION, 8, lambda d2: om.estimator_params(om.EST_SMOOTHER, 4, Kk, max_dist2=d2)),
            "stereo_ekf": (om.MERGER_STEREO_EKF, 0, None)}
    for kind, (variant, max_meas, make) in ests.items():
        for step in ("00_to_00", "00_to_01"):
            if kind == "stereo_ekf":  # :675-780: covariance norm / geometry 1 (00To00) and 100 (00To01), minimum covariance 0.01
                lim = 1.0 if step == "00_to_00" else 100.0
                est = om.estimator_params(om.EST_EKF, 4, Kk, baseline_px=bpx, max_dist2=lim, min_cov=0.01, max_cov_norm2=lim)
            else:
                est = make(25.0)  # :370, :420, :478, :537
            p = _merger_params(kit, variant, est, row_bins=10, col_bins=30, max_appearance=50.0, target_merges=100)
            # the fixture's platform: baseline 386.1448 px (fixtures.hpp:811), the value points_in_camera_00 was triangulated with
            p.triangulator = ob.TriangulatorParams(Kk[0], Kk[1], Kk[2], Kk[3], rp.KITTI_BX, 0.0, 1.84467e19)
            first = step == "00_to_00"
            cases.append(dict(name="kitti_%s_%s" % (kind, step), params=p, map=_seed_map(xyz0, fix["desc"][0], image00, max_meas, 1024),
                              T=I4 if first else T01, fixed=fix["meas"][0 if first else 1], desc=fix["desc"][0 if first else 1],
                              corr=_identity_corr(n0) if first else c01, size=n0 if first else None, grows=not first,
                              # 00To00: 1e-5 (:399-403, :456-460, :721-725).  00To01: 5 m per coordinate (:515-517, :580-582, :784-786);
                              # measured here: 3.1 m (weighted mean, smoother), 1.6 m (stereo EKF)
                              tol=1e-5 if first else 5.0, n_frames=4))
    return cases


def check_merge_result(case, n_before, before_xyz, n_points, coords, n_measured):
    """the assertions of the gtest behind `case` on a merge result"""
    if case.get("size") is not None:
        assert n_points == case["size"], (case["name"], n_points, case["size"])
    if case.get("grows"):
        assert n_before < n_points <= n_before + n_measured, (case["name"], n_before, n_points)  # :503-506
    err = np.abs(np.asarray(coords[:n_before, :3], np.float64) - before_xyz[:n_before]).max()
    assert err <= case["tol"] + 1e-12, (case["name"], err)
