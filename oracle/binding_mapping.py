"""ctypes binding of the mapping part of the CPU oracle (proslam_oracle_mapping.h): landmark
estimators + projective mergers.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import binding as ob

EST_WEIGHTED_MEAN, EST_EKF, EST_SMOOTHER = 0, 1, 2
MERGER_STEREO_TRIANGULATION, MERGER_STEREO_EKF, MERGER_DEPTH_EKF = 0, 1, 2
ERR_HISTORY, ERR_SCENE_FULL, ERR_DUPLICATE = -7, -8, -9

MEAS_DTYPE = np.dtype([("point_in_image", np.float32, 3), ("point_in_camera", np.float32, 3), ("frame", np.int32)])
POSE_DTYPE = np.dtype([("sensor_in_world", np.float32, 12), ("world_in_sensor", np.float32, 12)])


class MapStruct(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("max_measurements", C.c_int32), ("n_points", C.c_int32),
                ("coords", C.c_void_p), ("desc", C.c_void_p), ("state", C.c_void_p), ("covariance", C.c_void_p),
                ("n_opt", C.c_void_p), ("inlier", C.c_void_p), ("n_meas", C.c_void_p), ("meas", C.c_void_p)]


class EstimatorParams(C.Structure):
    _fields_ = [("type", C.c_int32), ("measurement_dim", C.c_int32),
                ("maximum_distance_geometry_meters_squared", C.c_float),
                ("minimum_state_element_covariance", C.c_double), ("maximum_covariance_norm_squared", C.c_double),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("b_x", C.c_double), ("b_y", C.c_double),
                ("maximum_number_of_iterations", C.c_uint32), ("convergence_criterion_minimum_chi2_delta", C.c_float),
                ("maximum_reprojection_error_pixels_squared", C.c_float),
                ("minimum_number_of_measurements_for_optimization", C.c_uint32),
                ("camera_matrix", C.c_float * 9)]


class MergerParams(C.Structure):
    _fields_ = [("variant", C.c_int32), ("enable_binning", C.c_int32),
                ("number_of_row_bins", C.c_uint32), ("number_of_col_bins", C.c_uint32),
                ("canvas_rows", C.c_int32), ("canvas_cols", C.c_int32),
                ("maximum_distance_appearance", C.c_float), ("target_number_of_merges", C.c_uint32),
                ("target_merge_ratio", C.c_float), ("triangulator", ob.TriangulatorParams),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("estimator", EstimatorParams)]


class MergeResult(C.Structure):
    _fields_ = [("n_merged", C.c_int32), ("n_added", C.c_int32), ("flags", C.c_int32)]


def estimator_params(kind, dim, K, baseline_px=(0.0, 0.0), max_dist2=1.0, min_cov=0.01, max_cov_norm2=1.0,
                     max_iterations=100, chi2_delta=1e-5, max_reprojection2=100.0, min_measurements=3):
    """defaults of landmark_estimator_base.hpp:21-25, landmark_estimator_ekf.h:31-47,
    landmark_estimator_pose_based_smoother.h:17-39; K = (fx, fy, cx, cy)"""
    p = EstimatorParams()
    p.type, p.measurement_dim = kind, dim
    p.maximum_distance_geometry_meters_squared = max_dist2
    p.minimum_state_element_covariance, p.maximum_covariance_norm_squared = min_cov, max_cov_norm2
    p.fx, p.fy, p.cx, p.cy = [float(np.float32(v)) for v in K]
    p.b_x, p.b_y = [float(np.float32(v)) for v in baseline_px]
    p.maximum_number_of_iterations, p.convergence_criterion_minimum_chi2_delta = max_iterations, chi2_delta
    p.maximum_reprojection_error_pixels_squared = max_reprojection2
    p.minimum_number_of_measurements_for_optimization = min_measurements
    Km = [K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1]
    for i in range(9):
        p.camera_matrix[i] = Km[i]
    return p


class Map:
    """one local map as numpy arrays (the layout the C-ABI uses on the device)"""

    def __init__(self, capacity, max_measurements):
        self.capacity, self.max_measurements = int(capacity), int(max_measurements)
        self.n_points = 0
        self.coords = np.zeros((capacity, 4), np.float32)
        self.desc = np.zeros((capacity, 32), np.uint8)
        self.state = np.zeros((capacity, 4), np.float32)
        self.covariance = np.zeros((capacity, 9), np.float32)
        self.n_opt = np.zeros(capacity, np.uint32)
        self.inlier = np.zeros(capacity, np.uint8)
        self.n_meas = np.zeros(capacity, np.uint32)
        self.meas = np.zeros((capacity, max(max_measurements, 1)), MEAS_DTYPE)

    def struct(self):
        s = MapStruct()
        s.capacity, s.max_measurements, s.n_points = self.capacity, self.max_measurements, self.n_points
        for name in ("coords", "desc", "state", "covariance", "n_opt", "inlier", "n_meas", "meas"):
            setattr(s, name, getattr(self, name).ctypes.data)
        return s

    def copy(self):
        m = Map(self.capacity, self.max_measurements)
        m.n_points = self.n_points
        for name in ("coords", "desc", "state", "covariance", "n_opt", "inlier", "n_meas", "meas"):
            getattr(m, name)[...] = getattr(self, name)
        return m

    def add_landmark(self, coords_local, state_world, covariance, desc=None, measurement=None):
        i = self.n_points
        self.coords[i, :3], self.state[i, :3] = coords_local, state_world
        self.covariance[i] = np.asarray(covariance, np.float32).ravel()
        if desc is not None:
            self.desc[i] = desc
        if measurement is not None and self.max_measurements > 0:
            self.meas[i, 0] = measurement
            self.n_meas[i] = 1
        self.n_points += 1
        return i


def pose_table(n):
    return np.zeros(n, POSE_DTYPE)


_bound = False


def _lib():
    global _bound
    L = ob.lib()
    if not _bound:
        vp = C.c_void_p
        L.orc_landmark_estimate.restype = C.c_int
        L.orc_landmark_estimate.argtypes = [C.POINTER(EstimatorParams), vp, vp, vp, C.c_int32, C.POINTER(MapStruct), C.c_int32, vp, vp]
        L.orc_merge.restype = C.c_int
        L.orc_merge.argtypes = [C.POINTER(MergerParams), vp, vp, vp, C.c_int32, C.POINTER(MapStruct), vp, vp, C.c_int32, vp, C.c_int32, vp,
                                C.POINTER(MergeResult)]
        _bound = True
    return L


def set_pose(poses, frame, sensor_in_world):
    T = np.asarray(sensor_in_world, np.float32).reshape(4, 4)
    poses[frame]["sensor_in_world"] = T[:3].ravel()
    Ti = np.zeros(16, np.float32)
    ob.lib().orc_se3_inverse(ob._ptr(np.ascontiguousarray(T)), ob._ptr(Ti))
    poses[frame]["world_in_sensor"] = Ti[:12]


def landmark_estimate(params, measurement_in_world, measurement_in_scene, poses, frame, m, index, measurement, landmark_in_sensor=None):
    Tw, Ts = ob._f32(measurement_in_world, (4, 4)), ob._f32(measurement_in_scene, (4, 4))
    z = np.zeros(4, np.float32)
    z[: len(measurement)] = measurement
    lis = ob._f32(landmark_in_sensor if landmark_in_sensor is not None else np.zeros(3), (3,))
    s = m.struct()
    return _lib().orc_landmark_estimate(C.byref(params), ob._ptr(Tw), ob._ptr(Ts), poses.ctypes.data, frame, C.byref(s), index, ob._ptr(z), ob._ptr(lis))


def merge(params, measurement_in_world, measurement_in_scene, poses, frame, m, measurement, measurement_desc, corr, scene_index_map=None):
    """MergerProjective_::compute for one frame -> (rc, MergeResult); the map object is updated in place"""
    dim = params.estimator.measurement_dim
    z = ob._f32(measurement, (-1, dim))
    d = ob._u8(measurement_desc).reshape(-1, 32)
    c = np.ascontiguousarray(corr, dtype=ob.CORR_DTYPE)
    im = None if scene_index_map is None else np.ascontiguousarray(scene_index_map, np.int32)
    Tw, Ts = ob._f32(measurement_in_world, (4, 4)), ob._f32(measurement_in_scene, (4, 4))
    s = m.struct()
    res = MergeResult()
    rc = _lib().orc_merge(C.byref(params), ob._ptr(Tw), ob._ptr(Ts), poses.ctypes.data, frame, C.byref(s), ob._ptr(z), ob._ptr(d),
                          z.shape[0], ob._ptr(c) if len(c) else None, len(c), None if im is None else ob._ptr(im), C.byref(res))
    m.n_points = s.n_points
    return rc, res
