"""ctypes binding of the CPU oracle (oracle/libproslam_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (srrg2_proslam_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRS_ORACLE_LIB: another build of the same sources (e.g. `make -C oracle san`: AddressSanitizer + UBSan, see tools/run_san.sh)
_LIB_PATH = os.environ.get("PRS_ORACLE_LIB") or os.path.join(_HERE, "libproslam_oracle.so")

CORR_DTYPE = np.dtype([("fixed_idx", np.int32), ("moving_idx", np.int32), ("response", np.float32)])

WARN_EMPTY_INPUT = 1
WARN_NO_MATCHES = 2
WARN_LOW_RATIO = 4
WARN_RETRIED = 8
WARN_TRACK_LOST = 16
WARN_NO_PROJECTION = 32

SEARCH_KDTREE, SEARCH_SQUARE, SEARCH_CIRCLE, SEARCH_RHOMBUS = 0, 1, 2, 3
FACTOR_MONO, FACTOR_DEPTH, FACTOR_STEREO = 2, 3, 4


class StereoParams(C.Structure):
    _fields_ = [
        ("maximum_descriptor_distance", C.c_float),
        ("maximum_distance_ratio_to_second_best", C.c_float),
        ("minimum_matching_ratio", C.c_float),
        ("maximum_disparity_pixels", C.c_int32),
        ("epipolar_line_thickness_pixels", C.c_int32),
    ]


class TriangulatorParams(C.Structure):
    _fields_ = [
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("b_x", C.c_float),
        ("minimum_disparity_pixels", C.c_float),
        ("infinity_depth_meters", C.c_float),
    ]


class Projector(C.Structure):
    _fields_ = [
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("canvas_cols", C.c_int32),
        ("canvas_rows", C.c_int32),
        ("range_min", C.c_float),
        ("range_max", C.c_float),
    ]


class PcfParams(C.Structure):
    _fields_ = [
        ("maximum_descriptor_distance", C.c_float),
        ("maximum_distance_ratio_to_second_best", C.c_float),
        ("minimum_matching_ratio", C.c_float),
        ("minimum_descriptor_distance", C.c_float),
        ("descriptor_distance_step_size_pixels", C.c_float),
        ("maximum_search_radius_pixels", C.c_uint64),
        ("minimum_search_radius_pixels", C.c_uint64),
        ("search_radius_step_size_pixels", C.c_uint64),
        ("minimum_number_of_iterations", C.c_uint64),
        ("maximum_estimate_change_norm_for_convergence", C.c_float),
        ("number_of_solver_iterations_per_projection", C.c_uint64),
        ("search_type", C.c_int32),
        ("projector", Projector),
        ("minimum_number_of_points_per_cluster", C.c_int32),  # KD-tree finder; 0 = the reference's default (10)
    ]


class AlignerParams(C.Structure):
    _fields_ = [
        ("factor_type", C.c_int32),
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("image_cols", C.c_float),
        ("image_rows", C.c_float),
        ("baseline_left_in_right_px", C.c_float * 3),
        ("diagonal_info", C.c_float * 3),
        ("chi_threshold", C.c_float),
        ("enable_inverse_depth_weighting", C.c_int32),
        ("mean_disparity", C.c_float),
        ("damping", C.c_float),
        ("max_iterations", C.c_int32),
        ("min_num_inliers", C.c_int32),
        ("min_num_correspondences", C.c_int32),
        ("enable_inlier_only_runs", C.c_int32),
        ("keep_only_inlier_correspondences", C.c_int32),
        ("inlier_only_iterations", C.c_int32),
        ("with_sensor", C.c_int32),
        ("sensor_in_robot", C.c_float * 16),
        ("enable_motion_prior", C.c_int32),
        ("motion_prior_info", C.c_float * 6),
    ]


class LinearSystem(C.Structure):
    _fields_ = [
        ("H", C.c_float * 36),
        ("b", C.c_float * 6),
        ("chi_inliers", C.c_float),
        ("chi_total", C.c_float),
        ("num_inliers", C.c_int32),
        ("num_outliers", C.c_int32),
        ("num_invalid", C.c_int32),
    ]


class AlignResult(C.Structure):
    _fields_ = [
        ("X", C.c_float * 16),
        ("status", C.c_int32),
        ("iterations", C.c_int32),
        ("num_inliers", C.c_int32),
        ("num_correspondences", C.c_int32),
        ("warnings", C.c_int32),
    ]


class Variant(C.Structure):
    """orc_variant: sweep switches for the BUILD-DEFINED arithmetic of rows a13 / a14 (tools/sweep_a13.py); all zero = shipped"""
    _fields_ = [("kernel_form", C.c_int32), ("idw_form", C.c_int32), ("damping_form", C.c_int32), ("v_row", C.c_int32),
                ("chi_compare", C.c_int32), ("bounds_form", C.c_int32), ("accum_form", C.c_int32)]


def build(force=False):
    """compile the oracle with its committed Makefile (gcc, seconds)"""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc")) or f == "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        i32p = C.POINTER(C.c_int32)
        vp = C.c_void_p
        L.orc_hamming256.restype = C.c_int
        L.orc_hamming256.argtypes = [vp, vp]
        L.orc_stereo_match.restype = C.c_int
        L.orc_stereo_match.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(StereoParams), vp, C.c_int, i32p]
        L.orc_stereo_assemble.restype = C.c_int
        L.orc_stereo_assemble.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        L.orc_triangulate.restype = None
        L.orc_triangulate.argtypes = [vp, C.c_int, C.POINTER(TriangulatorParams), vp, vp]
        for name in ("orc_se3_identity",):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp]
        L.orc_se3_inverse.restype = None
        L.orc_se3_inverse.argtypes = [vp, vp]
        L.orc_se3_mul.restype = None
        L.orc_se3_mul.argtypes = [vp, vp, vp]
        L.orc_t2tnq.restype = None
        L.orc_t2tnq.argtypes = [vp, vp]
        L.orc_tnq2t.restype = None
        L.orc_tnq2t.argtypes = [vp, vp]
        L.orc_project.restype = C.c_int
        L.orc_project.argtypes = [C.POINTER(Projector), vp, vp, C.c_int, vp, vp]
        L.orc_scene_clip.restype = C.c_int
        L.orc_scene_clip.argtypes = [C.POINTER(Projector), vp, vp, vp, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_int)]
        L.orc_pcf_create.restype = vp
        L.orc_pcf_create.argtypes = [C.POINTER(PcfParams)]
        L.orc_pcf_destroy.restype = None
        L.orc_pcf_destroy.argtypes = [vp]
        L.orc_pcf_set_params.restype = None
        L.orc_pcf_set_params.argtypes = [vp, C.POINTER(PcfParams)]
        L.orc_pcf_set_fixed.restype = None
        L.orc_pcf_set_fixed.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_pcf_set_moving.restype = None
        L.orc_pcf_set_moving.argtypes = [vp, vp, vp, C.c_int]
        L.orc_pcf_set_local_map_in_sensor.restype = None
        L.orc_pcf_set_local_map_in_sensor.argtypes = [vp, vp]
        L.orc_pcf_get_local_map_in_sensor.restype = None
        L.orc_pcf_get_local_map_in_sensor.argtypes = [vp, vp]
        L.orc_pcf_set_search_radius.restype = None
        L.orc_pcf_set_search_radius.argtypes = [vp, C.c_uint64]
        L.orc_pcf_set_descriptor_distance.restype = None
        L.orc_pcf_set_descriptor_distance.argtypes = [vp, C.c_float]
        L.orc_pcf_search_radius.restype = C.c_uint64
        L.orc_pcf_search_radius.argtypes = [vp]
        L.orc_pcf_descriptor_distance.restype = C.c_float
        L.orc_pcf_descriptor_distance.argtypes = [vp]
        L.orc_pcf_iteration.restype = C.c_uint64
        L.orc_pcf_iteration.argtypes = [vp]
        L.orc_pcf_has_converged.restype = C.c_int
        L.orc_pcf_has_converged.argtypes = [vp]
        L.orc_pcf_num_recomputes.restype = C.c_int
        L.orc_pcf_num_recomputes.argtypes = [vp]
        L.orc_pcf_compute.restype = C.c_int
        L.orc_pcf_compute.argtypes = [vp, vp, C.c_int, i32p]
        L.orc_info_scale_from_nopt.restype = None
        L.orc_info_scale_from_nopt.argtypes = [vp, C.c_int, vp]
        L.orc_mean_disparity.restype = C.c_float
        L.orc_mean_disparity.argtypes = [vp, C.c_int]
        L.orc_linearize.restype = None
        L.orc_linearize.argtypes = [C.POINTER(AlignerParams), vp, vp, C.c_int, vp, vp, vp, C.POINTER(LinearSystem)]
        L.orc_gn_step.restype = C.c_int
        L.orc_gn_step.argtypes = [C.POINTER(LinearSystem), C.c_float, vp]
        L.orc_align_frame.restype = None
        L.orc_align_frame.argtypes = [vp, C.POINTER(AlignerParams), vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, i32p, C.POINTER(AlignResult)]
        L.orc_align_frame_ex.restype = None
        L.orc_align_frame_ex.argtypes = [vp, C.POINTER(AlignerParams), vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, i32p, C.POINTER(AlignResult)]
        L.orc_linearize_ex.restype = None
        L.orc_linearize_ex.argtypes = [C.POINTER(AlignerParams), vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.POINTER(LinearSystem)]
        L.orc_add_motion_prior.restype = None
        L.orc_add_motion_prior.argtypes = [C.POINTER(AlignerParams), vp, vp, C.POINTER(LinearSystem)]
        L.orc_motion_predict.restype = None
        L.orc_motion_predict.argtypes = [vp, vp, vp]
        L.orc_bruteforce_match.restype = C.c_int
        L.orc_bruteforce_match.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, vp, C.c_int, i32p]
        L.orc_set_variant.restype = None
        L.orc_set_variant.argtypes = [C.POINTER(Variant)]
        _lib = L
        del fp, u8p
    return _lib


def set_variant(**kw):
    """sweep switches (see Variant); no arguments = the shipped definition"""
    v = Variant(**kw)
    lib().orc_set_variant(C.byref(v))
    return v


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
def hamming256(a, b):
    a, b = _u8(a), _u8(b)
    return int(lib().orc_hamming256(_ptr(a), _ptr(b)))


def stereo_match(uv_left, desc_left, uv_right, desc_right, params):
    """returns (correspondences [CORR_DTYPE], warning flags)"""
    uvl, uvr = _f32(uv_left, (-1, 2)), _f32(uv_right, (-1, 2))
    dl, dr = _u8(desc_left).reshape(-1, 32), _u8(desc_right).reshape(-1, 32)
    nl, nr = uvl.shape[0], uvr.shape[0]
    out = np.zeros(max(nl, 1), dtype=CORR_DTYPE)
    n = C.c_int32(0)
    flags = lib().orc_stereo_match(_ptr(uvl), _ptr(dl), nl, _ptr(uvr), _ptr(dr), nr, C.byref(params), _ptr(out), out.shape[0], C.byref(n))
    if flags < 0:
        raise RuntimeError("orc_stereo_match error %d" % flags)
    return out[: n.value].copy(), flags


def stereo_assemble(uv_left, uv_right, corr):
    uvl, uvr = _f32(uv_left, (-1, 2)), _f32(uv_right, (-1, 2))
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    out = np.zeros((max(len(corr), 1), 4), dtype=np.float32)
    src = np.zeros(max(len(corr), 1), dtype=np.int32)
    n = lib().orc_stereo_assemble(_ptr(uvl), _ptr(uvr), _ptr(corr), len(corr), _ptr(out), _ptr(src))
    return out[:n].copy(), src[:n].copy()


def triangulate(uvuv, params):
    uvuv = _f32(uvuv, (-1, 4))
    n = uvuv.shape[0]
    xyz = np.zeros((max(n, 1), 3), dtype=np.float32)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    lib().orc_triangulate(_ptr(uvuv), n, C.byref(params), _ptr(xyz), _ptr(valid))
    return xyz[:n].copy(), valid[:n].copy()


def se3_inverse(T):
    T = _f32(T, (4, 4))
    out = np.zeros((4, 4), dtype=np.float32)
    lib().orc_se3_inverse(_ptr(T), _ptr(out))
    return out


def se3_mul(A, B):
    A, B = _f32(A, (4, 4)), _f32(B, (4, 4))
    out = np.zeros((4, 4), dtype=np.float32)
    lib().orc_se3_mul(_ptr(A), _ptr(B), _ptr(out))
    return out


def t2tnq(T):
    T = _f32(T, (4, 4))
    out = np.zeros(6, dtype=np.float32)
    lib().orc_t2tnq(_ptr(T), _ptr(out))
    return out


def tnq2t(v):
    v = _f32(v, (6,))
    out = np.zeros((4, 4), dtype=np.float32)
    lib().orc_tnq2t(_ptr(v), _ptr(out))
    return out


def project(projector, camera_pose, xyz):
    xyz = _f32(xyz, (-1, 3))
    T = _f32(camera_pose, (4, 4))
    n = xyz.shape[0]
    uvz = np.zeros((max(n, 1), 3), dtype=np.float32)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    m = lib().orc_project(C.byref(projector), _ptr(T), _ptr(xyz), n, _ptr(uvz), _ptr(idx))
    return uvz[:m].copy(), idx[:m].copy()


class ProjectiveFinder:
    """stateful oracle finder (CorrespondenceFinderProjective{KDTree,Square,Circle,Rhombus})"""

    def __init__(self, params):
        self._h = lib().orc_pcf_create(C.byref(params))
        self._n_fixed = 0

    def close(self):
        if self._h:
            lib().orc_pcf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        lib().orc_pcf_set_params(self._h, C.byref(params))

    def set_fixed(self, coords, desc):
        coords = _f32(coords)
        coords = coords.reshape(coords.shape[0] if coords.ndim == 2 else -1, coords.shape[-1] if coords.ndim == 2 else 2)
        d = _u8(desc).reshape(-1, 32)
        self._n_fixed = coords.shape[0]
        lib().orc_pcf_set_fixed(self._h, _ptr(coords), coords.shape[1], _ptr(d), coords.shape[0])

    def set_moving(self, xyz, desc):
        xyz = _f32(xyz, (-1, 3))
        d = _u8(desc).reshape(-1, 32)
        lib().orc_pcf_set_moving(self._h, _ptr(xyz), _ptr(d), xyz.shape[0])

    def set_local_map_in_sensor(self, T):
        T = _f32(T, (4, 4))
        lib().orc_pcf_set_local_map_in_sensor(self._h, _ptr(T))

    def local_map_in_sensor(self):
        out = np.zeros((4, 4), dtype=np.float32)
        lib().orc_pcf_get_local_map_in_sensor(self._h, _ptr(out))
        return out

    def set_search_radius(self, r):
        lib().orc_pcf_set_search_radius(self._h, int(r))

    def set_descriptor_distance(self, d):
        lib().orc_pcf_set_descriptor_distance(self._h, float(d))

    @property
    def search_radius(self):
        return int(lib().orc_pcf_search_radius(self._h))

    @property
    def descriptor_distance(self):
        return float(lib().orc_pcf_descriptor_distance(self._h))

    @property
    def iteration(self):
        return int(lib().orc_pcf_iteration(self._h))

    @property
    def has_converged(self):
        return bool(lib().orc_pcf_has_converged(self._h))

    @property
    def num_recomputes(self):
        return int(lib().orc_pcf_num_recomputes(self._h))

    def compute(self):
        out = np.zeros(max(self._n_fixed, 1), dtype=CORR_DTYPE)
        n = C.c_int32(0)
        flags = lib().orc_pcf_compute(self._h, _ptr(out), out.shape[0], C.byref(n))
        if flags < 0:
            raise RuntimeError("orc_pcf_compute error %d" % flags)
        return out[: n.value].copy(), flags


def info_scale_from_nopt(n_opt):
    n_opt = np.ascontiguousarray(n_opt, dtype=np.uint32)
    out = np.zeros(max(len(n_opt), 1), dtype=np.float32)
    lib().orc_info_scale_from_nopt(_ptr(n_opt), len(n_opt), _ptr(out))
    return out[: len(n_opt)].copy()


def mean_disparity(fixed_uvuv):
    f = _f32(fixed_uvuv, (-1, 4))
    return float(lib().orc_mean_disparity(_ptr(f), f.shape[0]))


def linearize(params, X, corr, fixed, moving_xyz, info_scale):
    X = _f32(X, (4, 4))
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    fixed = _f32(fixed)
    moving_xyz = _f32(moving_xyz, (-1, 3))
    info_scale = None if info_scale is None else _f32(info_scale)
    sys = LinearSystem()
    lib().orc_linearize(C.byref(params), _ptr(X), _ptr(corr), len(corr), _ptr(fixed), _ptr(moving_xyz), _ptr(info_scale), C.byref(sys))
    return sys


def linearize_ex(params, X, corr, fixed, moving_xyz, info_scale, inlier_only=False):
    """-> (LinearSystem, classes [n_corr] u8: 0 inlier, 1 kernelised, 2 invalid)"""
    X = _f32(X, (4, 4))
    corr = np.ascontiguousarray(corr, dtype=CORR_DTYPE)
    fixed = _f32(fixed)
    moving_xyz = _f32(moving_xyz, (-1, 3))
    info_scale = None if info_scale is None else _f32(info_scale)
    sys = LinearSystem()
    cls = np.zeros(max(len(corr), 1), np.uint8)
    lib().orc_linearize_ex(C.byref(params), _ptr(X), _ptr(corr), len(corr), _ptr(fixed), _ptr(moving_xyz), _ptr(info_scale), int(inlier_only), _ptr(cls), C.byref(sys))
    return sys, cls[: len(corr)]


def add_motion_prior(params, X, prior_mean, sys):
    X = _f32(X, (4, 4))
    Z = None if prior_mean is None else _f32(prior_mean, (4, 4))
    lib().orc_add_motion_prior(C.byref(params), _ptr(X), _ptr(Z), C.byref(sys))
    return sys


def motion_predict(pose_prev2, pose_prev1):
    a, b = _f32(pose_prev2, (4, 4)), _f32(pose_prev1, (4, 4))
    out = np.zeros((4, 4), np.float32)
    lib().orc_motion_predict(_ptr(a), _ptr(b), _ptr(out))
    return out


def gn_step(sys, damping, X):
    X = _f32(X, (4, 4)).copy()
    rc = lib().orc_gn_step(C.byref(sys), float(damping), _ptr(X))
    return X, rc


def align_frame(finder, params, fixed, moving_xyz, info_scale, X_init, prior=None, prior_mean=None):
    fixed = _f32(fixed)
    n_fixed = fixed.shape[0]
    moving_xyz = _f32(moving_xyz, (-1, 3))
    info_scale = None if info_scale is None else _f32(info_scale)
    X_init = _f32(X_init, (4, 4))
    corr = np.zeros(max(n_fixed, 1), dtype=CORR_DTYPE)
    n = C.c_int32(0)
    res = AlignResult()
    pH = pb = None
    if prior is not None:
        pH, pb = _f32(prior[0], (36,)), _f32(prior[1], (6,))
    Z = None if prior_mean is None else _f32(prior_mean, (4, 4))
    lib().orc_align_frame_ex(finder._h, C.byref(params), _ptr(fixed), n_fixed, _ptr(moving_xyz), _ptr(info_scale), moving_xyz.shape[0], _ptr(X_init), _ptr(pH), _ptr(pb), _ptr(Z), _ptr(corr), C.byref(n), C.byref(res))
    return res, corr[: n.value].copy()


def bruteforce_match(desc_fixed, desc_moving, max_distance, max_ratio):
    df, dm = _u8(desc_fixed).reshape(-1, 32), _u8(desc_moving).reshape(-1, 32)
    cap = max(min(df.shape[0], dm.shape[0]), 1)
    out = np.zeros(cap, dtype=CORR_DTYPE)
    n = C.c_int32(0)
    flags = lib().orc_bruteforce_match(_ptr(df), df.shape[0], _ptr(dm), dm.shape[0], float(max_distance), float(max_ratio), _ptr(out), cap, C.byref(n))
    if flags < 0:
        raise RuntimeError("orc_bruteforce_match error %d" % flags)
    return out[: n.value].copy(), flags


def scene_clip(projector, robot_in_local_map, sensor_in_robot, scene_xyzw, scene_desc=None):
    """SceneClipperProjective3D::compute -> (clipped_xyzw, clipped_desc | None, global_indices, flags)"""
    xyzw = _f32(scene_xyzw, (-1, 4))
    n = xyzw.shape[0]
    desc = None if scene_desc is None else _u8(scene_desc).reshape(-1, 32)
    R, S = _f32(robot_in_local_map, (4, 4)), _f32(sensor_in_robot, (4, 4))
    out = np.zeros((max(n, 1), 4), dtype=np.float32)
    odesc = None if desc is None else np.zeros((max(n, 1), 32), dtype=np.uint8)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    m = C.c_int(0)
    flags = lib().orc_scene_clip(C.byref(projector), _ptr(R), _ptr(S), _ptr(xyzw), None if desc is None else _ptr(desc), n,
                                 _ptr(out), None if odesc is None else _ptr(odesc), _ptr(idx), C.byref(m))
    k = m.value
    return out[:k].copy(), (None if odesc is None else odesc[:k].copy()), idx[:k].copy(), flags
