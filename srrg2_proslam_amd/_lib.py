"""ctypes loader for libproslam_hip.so (the HIP/gfx950 product library).

There is NO CPU fallback: if the shared library is missing or cannot be loaded, importing the
operators fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C srrg2_proslam_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libproslam_hip.so")
ABI_VERSION = 102  # PRS_ABI_VERSION of include/proslam_hip.h

# status codes (include/proslam_hip.h)
OK = 0
WARN_EMPTY_INPUT = 1
WARN_NO_MATCHES = 2
WARN_LOW_RATIO = 4
WARN_RETRIED = 8
WARN_TRACK_LOST = 16
WARN_NO_PROJECTION = 32
ERR_NULL = -1
ERR_CAPACITY = -2
ERR_HIP = -3
ERR_RANGE = -4
ERR_UNSUPPORTED = -5
ERR_NO_DEVICE = -6


class StereoParams(C.Structure):
    """prs_stereo_params"""
    _fields_ = [
        ("maximum_descriptor_distance", C.c_float),
        ("maximum_distance_ratio_to_second_best", C.c_float),
        ("minimum_matching_ratio", C.c_float),
        ("maximum_disparity_pixels", C.c_int32),
        ("epipolar_line_thickness_pixels", C.c_int32),
        ("image_rows", C.c_int32),
        ("image_cols", C.c_int32),
    ]


class TriangulatorParams(C.Structure):
    """prs_triangulator_params"""
    _fields_ = [
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("b_x", C.c_float),
        ("minimum_disparity_pixels", C.c_float),
        ("infinity_depth_meters", C.c_float),
    ]


class StereoBatch(C.Structure):
    """prs_stereo_batch (device pointers)"""
    _fields_ = [
        ("batch", C.c_int32),
        ("stride", C.c_int32),
        ("left_kp", C.c_void_p),
        ("left_desc", C.c_void_p),
        ("n_left", C.c_void_p),
        ("right_kp", C.c_void_p),
        ("right_desc", C.c_void_p),
        ("n_right", C.c_void_p),
        ("matches", C.c_void_p),
        ("n_matches", C.c_void_p),
        ("status", C.c_void_p),
        ("fixed_uvuv", C.c_void_p),
        ("fixed_desc", C.c_void_p),
        ("n_fixed", C.c_void_p),
        ("fixed_xyz", C.c_void_p),
        ("triangulator", C.POINTER(TriangulatorParams)),
    ]


class Projector(C.Structure):
    """prs_projector"""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("canvas_cols", C.c_int32), ("canvas_rows", C.c_int32), ("range_min", C.c_float), ("range_max", C.c_float)]


class PcfParams(C.Structure):
    """prs_pcf_params"""
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


class PcfState(C.Structure):
    """prs_pcf_state"""
    _fields_ = [
        ("search_radius_pixels", C.c_uint64),
        ("current_iteration", C.c_uint64),
        ("descriptor_distance", C.c_float),
        ("has_converged", C.c_int32),
        ("config_changed", C.c_int32),
        ("num_recomputes", C.c_int32),
        ("local_map_in_sensor", C.c_float * 16),
        ("local_map_in_sensor_previous", C.c_float * 16),
        ("database_leaf_range", C.c_float),
        ("reserved", C.c_int32),
    ]


class AlignerParams(C.Structure):
    """prs_aligner_params"""
    _fields_ = [
        ("factor_type", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("image_cols", C.c_float), ("image_rows", C.c_float),
        ("baseline_left_in_right_px", C.c_float * 3),
        ("diagonal_info", C.c_float * 3),
        ("chi_threshold", C.c_float),
        ("enable_inverse_depth_weighting", C.c_int32),
        ("mean_disparity", C.c_float),
        ("damping", C.c_float),
        ("max_iterations", C.c_int32),
        ("min_num_inliers", C.c_int32),
        ("min_num_correspondences", C.c_int32),
        ("stop_at_fixed_point", C.c_int32),
        ("enable_inlier_only_runs", C.c_int32),
        ("keep_only_inlier_correspondences", C.c_int32),
        ("inlier_only_iterations", C.c_int32),
        ("with_sensor", C.c_int32),
        ("sensor_in_robot", C.c_float * 16),
        ("enable_motion_prior", C.c_int32),
        ("motion_prior_info", C.c_float * 6),
        ("kernel_weight_form", C.c_int32),       # 0 Omega / chi (shipped), 1 Omega * tau / chi
        ("damping_form", C.c_int32),             # 0 H + lambda diag(H) (shipped), 1 H + lambda I
        ("translation_weight_form", C.c_int32),  # 0 min(0.01 + dn, 1) (shipped), 1 clamp(dn, 0.01, 1)
        ("step_norm_exit", C.c_float),           # opt-in early exit (less work than the reference); 0 = off
    ]


class AlignResult(C.Structure):
    """prs_align_result"""
    _fields_ = [
        ("H", C.c_float * 36),
        ("b", C.c_float * 6),
        ("chi_inliers", C.c_float),
        ("chi_total", C.c_float),
        ("mean_disparity", C.c_float),
        ("num_inliers", C.c_int32),
        ("num_outliers", C.c_int32),
        ("num_invalid", C.c_int32),
        ("num_correspondences", C.c_int32),
        ("status", C.c_int32),
        ("iterations", C.c_int32),
        ("iterations_executed", C.c_int32),
        ("warnings", C.c_int32),
    ]


class AlignBatch(C.Structure):
    """prs_align_batch (device pointers)"""
    _fields_ = [
        ("batch", C.c_int32), ("fixed_stride", C.c_int32), ("moving_stride", C.c_int32),
        ("fixed", C.c_void_p), ("fixed_desc", C.c_void_p), ("n_fixed", C.c_void_p),
        ("moving", C.c_void_p), ("moving_desc", C.c_void_p), ("n_moving", C.c_void_p),
        ("inputs_changed", C.c_void_p), ("state", C.c_void_p), ("X", C.c_void_p),
        ("corr", C.c_void_p), ("n_corr", C.c_void_p), ("result", C.c_void_p), ("prior", C.c_void_p),
        ("prior_mean", C.c_void_p), ("max_fixed", C.c_int32),
    ]


class ClipBatch(C.Structure):
    """prs_clip_batch (device pointers)"""
    _fields_ = [
        ("batch", C.c_int32), ("stride", C.c_int32),
        ("scene_xyzw", C.c_void_p), ("scene_desc", C.c_void_p), ("n_scene", C.c_void_p),
        ("robot_in_local_map", C.c_void_p),
        ("clipped_xyzw", C.c_void_p), ("clipped_desc", C.c_void_p), ("global_indices", C.c_void_p),
        ("n_clipped", C.c_void_p), ("status", C.c_void_p), ("scene_n_opt", C.c_void_p),
    ]


class BruteforceParams(C.Structure):
    """prs_bruteforce_params"""
    _fields_ = [
        ("maximum_descriptor_distance", C.c_float),
        ("maximum_distance_ratio_to_second_best", C.c_float),
        ("minimum_matching_ratio", C.c_float),
    ]


class BruteforceBatch(C.Structure):
    """prs_bruteforce_batch (device pointers)"""
    _fields_ = [
        ("batch", C.c_int32), ("fixed_stride", C.c_int32), ("moving_stride", C.c_int32),
        ("fixed_desc", C.c_void_p), ("n_fixed", C.c_void_p), ("moving_desc", C.c_void_p), ("n_moving", C.c_void_p),
        ("matches", C.c_void_p), ("n_matches", C.c_void_p), ("status", C.c_void_p),
        ("candidate_capacity", C.c_int32),
    ]


class EstimatorParams(C.Structure):
    """prs_estimator_params"""
    _fields_ = [("type", C.c_int32), ("measurement_dim", C.c_int32),
                ("maximum_distance_geometry_meters_squared", C.c_float),
                ("minimum_state_element_covariance", C.c_double), ("maximum_covariance_norm_squared", C.c_double),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("b_x", C.c_double), ("b_y", C.c_double),
                ("maximum_number_of_iterations", C.c_uint32), ("convergence_criterion_minimum_chi2_delta", C.c_float),
                ("maximum_reprojection_error_pixels_squared", C.c_float),
                ("minimum_number_of_measurements_for_optimization", C.c_uint32),
                ("camera_matrix", C.c_float * 9)]


class MergerParams(C.Structure):
    """prs_merger_params"""
    _fields_ = [("variant", C.c_int32), ("enable_binning", C.c_int32),
                ("number_of_row_bins", C.c_uint32), ("number_of_col_bins", C.c_uint32),
                ("canvas_rows", C.c_int32), ("canvas_cols", C.c_int32),
                ("maximum_distance_appearance", C.c_float), ("target_number_of_merges", C.c_uint32),
                ("target_merge_ratio", C.c_float), ("triangulator", TriangulatorParams),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("estimator", EstimatorParams)]


class MergeResult(C.Structure):
    """prs_merge_result"""
    _fields_ = [("n_merged", C.c_int32), ("n_added", C.c_int32), ("status", C.c_int32)]


class MergeBatch(C.Structure):
    """prs_merge_batch (device pointers)"""
    _fields_ = [("batch", C.c_int32), ("capacity", C.c_int32), ("max_measurements", C.c_int32), ("max_frames", C.c_int32),
                ("coords", C.c_void_p), ("desc", C.c_void_p), ("state", C.c_void_p), ("covariance", C.c_void_p),
                ("n_opt", C.c_void_p), ("inlier", C.c_void_p), ("n_meas", C.c_void_p), ("meas", C.c_void_p),
                ("poses", C.c_void_p), ("n_points", C.c_void_p),
                ("measurement_stride", C.c_int32), ("measurement", C.c_void_p), ("measurement_desc", C.c_void_p),
                ("n_measured", C.c_void_p), ("corr_stride", C.c_int32), ("corr", C.c_void_p), ("n_corr", C.c_void_p),
                ("scene_index_map", C.c_void_p), ("measurement_in_world", C.c_void_p), ("measurement_in_scene", C.c_void_p),
                ("frame", C.c_void_p), ("result", C.c_void_p), ("corr_from_aligner", C.c_int32)]


class ExtractorParams(C.Structure):
    """prs_extractor_params"""
    _fields_ = [("detector_threshold", C.c_int32), ("enable_non_maximum_suppression", C.c_int32),
                ("target_number_of_keypoints", C.c_int32), ("number_of_detectors_vertical", C.c_int32),
                ("number_of_detectors_horizontal", C.c_int32), ("selection_order", C.c_int32),
                ("max_raw_detections", C.c_int32)]


class ExtractBatch(C.Structure):
    """prs_extract_batch (device pointers)"""
    _fields_ = [("batch", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("pitch", C.c_int32), ("images", C.c_void_p),
                ("stride", C.c_int32), ("keypoints", C.c_void_p), ("intensity", C.c_void_p), ("descriptors", C.c_void_p),
                ("n_features", C.c_void_p), ("status", C.c_void_p)]


MODE_ALIGN, MODE_FINDER, MODE_LINEARIZE = 0, 1, 2

# every symbol include/proslam_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)
SYMBOLS = {
    "prs_version": (C.c_int, []),
    "prs_abi_check": (C.c_int, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "prs_status_string": (C.c_char_p, [C.c_int]),
    "prs_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "prs_context_destroy": (C.c_int, [_vp]),
    "prs_context_set_stream": (C.c_int, [_vp, _vp]),
    "prs_context_use_own_stream": (C.c_int, [_vp]),
    "prs_context_synchronize": (C.c_int, [_vp]),
    "prs_context_enable_timing": (C.c_int, [_vp, C.c_int32]),
    "prs_context_set_bruteforce_dense_phase": (C.c_int, [_vp, C.c_int32]),
    "prs_context_get_align_timing": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "prs_context_get_align_round_timing": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "prs_last_error": (C.c_char_p, [_vp]),
    "prs_stereo_match": (C.c_int, [_vp, C.POINTER(StereoParams), _vp, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, C.c_int32, _i32p]),
    "prs_stereo_match_batch": (C.c_int, [_vp, C.POINTER(StereoParams), C.POINTER(StereoBatch)]),
    "prs_align_batch_run": (C.c_int, [_vp, C.POINTER(PcfParams), C.POINTER(AlignerParams), C.POINTER(AlignBatch), C.c_int32]),
    "prs_align_batch_enqueue": (C.c_int, [_vp, C.POINTER(PcfParams), C.POINTER(AlignerParams), C.POINTER(AlignBatch), C.c_int32]),
    "prs_align_batch_finish": (C.c_int, [_vp]),
    "prs_align_batch_rearm": (C.c_int, [_vp]),
    "prs_align_batch_rearm_on": (C.c_int, [_vp, _vp]),
    "prs_pcf_create": (C.c_int, [_vp, C.POINTER(PcfParams), C.POINTER(_vp)]),
    "prs_pcf_destroy": (C.c_int, [_vp]),
    "prs_pcf_set_params": (C.c_int, [_vp, C.POINTER(PcfParams)]),
    "prs_pcf_set_fixed": (C.c_int, [_vp, _vp, C.c_int32, _vp, C.c_int32]),
    "prs_pcf_set_moving": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32]),
    "prs_pcf_set_local_map_in_sensor": (C.c_int, [_vp, _vp]),
    "prs_pcf_set_search_radius": (C.c_int, [_vp, C.c_uint64]),
    "prs_pcf_set_descriptor_distance": (C.c_int, [_vp, C.c_float]),
    "prs_pcf_get_state": (C.c_int, [_vp, C.POINTER(PcfState)]),
    "prs_pcf_set_motion_prior_mean": (C.c_int, [_vp, _vp]),
    "prs_pcf_compute": (C.c_int, [_vp, _vp, C.c_int32, _i32p]),
    "prs_pcf_align": (C.c_int, [_vp, C.POINTER(AlignerParams), _vp, _vp, _vp, _vp, C.c_int32, _i32p, C.POINTER(AlignResult)]),
    "prs_pcf_linearize": (C.c_int, [_vp, C.POINTER(AlignerParams), _vp, _vp, C.c_int32, C.POINTER(AlignResult)]),
    "prs_gn_step": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp]),
    "prs_gn_step_ex": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_int32, _vp]),
    "prs_selftest_reciprocal": (C.c_int, [_vp, _vp]),
    "prs_info_scale_from_nopt": (None, [_vp, C.c_int32, _vp]),
    "prs_triangulate": (C.c_int, [_vp, C.POINTER(TriangulatorParams), _vp, C.c_int32, _vp, _vp]),
    "prs_triangulate_dev": (C.c_int, [_vp, C.POINTER(TriangulatorParams), _vp, C.c_int64, _vp]),
    "prs_bruteforce_match_batch": (C.c_int, [_vp, C.POINTER(BruteforceParams), C.POINTER(BruteforceBatch)]),
    "prs_bruteforce_match": (C.c_int, [_vp, C.POINTER(BruteforceParams), _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _i32p]),
    "prs_extract_features_batch": (C.c_int, [_vp, C.POINTER(ExtractorParams), C.POINTER(ExtractBatch)]),
    "prs_selection_order": (C.c_int, [_vp, _vp, C.c_int32, _vp]),
    "prs_extract_features": (C.c_int, [_vp, C.POINTER(ExtractorParams), _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _i32p]),
    "prs_pose_compose_batch": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "prs_motion_predict_batch": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "prs_merge_batch_run": (C.c_int, [_vp, C.POINTER(MergerParams), C.POINTER(MergeBatch)]),
    "prs_map_create": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "prs_map_destroy": (C.c_int, [_vp]),
    "prs_map_clear": (C.c_int, [_vp]),
    "prs_map_reserve": (C.c_int, [_vp, C.c_int32]),
    "prs_map_size": (C.c_int, [_vp, _i32p, _i32p]),
    "prs_map_set_scene": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32]),
    "prs_map_set_frame_pose": (C.c_int, [_vp, C.c_int32, _vp]),
    "prs_map_merge": (C.c_int, [_vp, C.POINTER(MergerParams), _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp]),
    "prs_map_get_scene": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _i32p]),
    "prs_scene_clip_batch": (C.c_int, [_vp, C.POINTER(Projector), _vp, C.POINTER(ClipBatch)]),
    "prs_scene_clip": (C.c_int, [_vp, C.POINTER(Projector), _vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _i32p]),
}

_lib = None


class ProslamHipError(RuntimeError):
    """hard error from the C-ABI (status < 0); the reference throws std::runtime_error here"""

    def __init__(self, status, message):
        super().__init__("libproslam_hip status %d: %s" % (status, message))
        self.status = status


def load():
    """dlopen the product library and bind every declared symbol; raises if it is absent"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libproslam_hip.so is not built (%s). The HIP extension is mandatory: run "
            "`python -c 'import __graft_entry__ as g; g.build()'` -- there is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it FIRST so the
    # dynamic loader resolves our DT_NEEDED against the runtime torch already mapped: one HIP
    # runtime per process, so torch tensors / streams and this library agree.  Two runtime
    # copies in one process fail HSA initialisation for the second one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    # the ctypes mirrors above and the library must describe the same structs (include/proslam_hip.h, PRS_ABI_VERSION)
    rc = lib.prs_abi_check(ABI_VERSION, C.sizeof(StereoParams), C.sizeof(PcfParams), C.sizeof(AlignerParams), C.sizeof(AlignBatch))
    if rc != 0:
        raise ImportError("libproslam_hip.so (version %d) does not match the Python binding (version %d, struct sizes %d/%d/%d/%d): "
                          "rebuild with `python -c 'import __graft_entry__ as g; g.build()'`"
                          % (lib.prs_version(), ABI_VERSION, C.sizeof(StereoParams), C.sizeof(PcfParams), C.sizeof(AlignerParams), C.sizeof(AlignBatch)))
    _lib = lib
    return lib
