"""Hot-path parameter sets of the reference's shipped configurations.

Every value cites the .conf line it was read from (paths relative to /root/reference/configurations/).
These are plain data; nothing here reads the reference at run time.
"""
import copy
import math

SEARCH_KDTREE, SEARCH_SQUARE, SEARCH_CIRCLE, SEARCH_RHOMBUS = 0, 1, 2, 3
FACTOR_MONO, FACTOR_DEPTH, FACTOR_STEREO = 2, 3, 4

_SQRT_FLT_MAX = 1.84467e19  # kitti.conf:40 "infinity_depth_meters"

KITTI = {
    "name": "kitti",
    # tests/fixtures.hpp:810-816,1093-1094
    "camera": {"fx": 718.856, "fy": 718.856, "cx": 607.193, "cy": 185.216, "cols": 1241, "rows": 376,
               "baseline_m": 0.537166},
    "projector": {"range_min": 0.1, "range_max": 1000.0},  # kitti.conf:175-178
    # kitti.conf:484-501
    "stereo_matcher": {"maximum_descriptor_distance": 100.0, "maximum_distance_ratio_to_second_best": 0.5,
                       "minimum_matching_ratio": 0.3, "maximum_disparity_pixels": 100,
                       "epipolar_line_thickness_pixels": 0},
    # kitti.conf:29-49
    "triangulator": {"minimum_disparity_pixels": 1.0, "infinity_depth_meters": _SQRT_FLT_MAX},
    # kitti.conf:834-875
    "projective_finder": {"search_type": SEARCH_CIRCLE, "maximum_descriptor_distance": 75.0,
                          "maximum_distance_ratio_to_second_best": 0.8, "minimum_matching_ratio": 0.1,
                          "minimum_descriptor_distance": 25.0, "descriptor_distance_step_size_pixels": 5.0,
                          "maximum_search_radius_pixels": 50, "minimum_search_radius_pixels": 10,
                          "search_radius_step_size_pixels": 10, "minimum_number_of_iterations": 5,
                          "maximum_estimate_change_norm_for_convergence": 0.01,
                          "number_of_solver_iterations_per_projection": 5},
    # kitti.conf:262-308 (slice), :137-142 (robustifier), :310-315 (damping), :980-1010 (aligner)
    "aligner": {"factor_type": FACTOR_STEREO, "diagonal_info": (1.0, 2.0, 1.0), "chi_threshold": 25.0,
                "enable_inverse_depth_weighting": 1, "damping": 1.0, "max_iterations": 100,
                "min_num_inliers": 6, "min_num_correspondences": 10},
    "depth": {"min": 4.0, "max": 80.0},
}

EUROC = {
    "name": "euroc",
    # apps/example_triangulate_rigid_stereo.cpp:132-133,148-149
    "camera": {"fx": 435.262, "fy": 435.262, "cx": 367.415, "cy": 252.171, "cols": 752, "rows": 480,
               "baseline_m": 0.110078},
    "projector": {"range_min": 0.1, "range_max": 1000.0},  # euroc.conf:218-221
    # euroc.conf:539-551
    "stereo_matcher": {"maximum_descriptor_distance": 75.0, "maximum_distance_ratio_to_second_best": 0.5,
                       "minimum_matching_ratio": 0.3, "maximum_disparity_pixels": 200,
                       "epipolar_line_thickness_pixels": 0},
    "triangulator": {"minimum_disparity_pixels": 1.0, "infinity_depth_meters": _SQRT_FLT_MAX},  # euroc.conf:185
    # euroc.conf:919-957
    "projective_finder": {"search_type": SEARCH_CIRCLE, "maximum_descriptor_distance": 100.0,
                          "maximum_distance_ratio_to_second_best": 0.8, "minimum_matching_ratio": 0.1,
                          "minimum_descriptor_distance": 25.0, "descriptor_distance_step_size_pixels": 5.0,
                          "maximum_search_radius_pixels": 100, "minimum_search_radius_pixels": 25,
                          "search_radius_step_size_pixels": 5, "minimum_number_of_iterations": 5,
                          "maximum_estimate_change_norm_for_convergence": 0.001,
                          "number_of_solver_iterations_per_projection": 5},
    # euroc.conf:463-470 (slice), :231-235 (robustifier), :694-698 (damping), :1-15 (aligner)
    "aligner": {"factor_type": FACTOR_STEREO, "diagonal_info": (1.0, 2.0, 1.0), "chi_threshold": 100.0,
                "enable_inverse_depth_weighting": 1, "damping": 1.0, "max_iterations": 100,
                "min_num_inliers": 6, "min_num_correspondences": 0},  # euroc.conf:493
    "depth": {"min": 1.0, "max": 15.0},
}

ICL = {
    "name": "icl",
    # tests/fixtures.hpp:577,763-764
    "camera": {"fx": 481.2, "fy": -481.0, "cx": 319.5, "cy": 239.5, "cols": 640, "rows": 480, "baseline_m": 0.0},
    "projector": {"range_min": 0.001, "range_max": 100.0},  # icl.conf:313-316
    "stereo_matcher": None,
    "triangulator": None,
    # icl.conf:321-359
    "projective_finder": {"search_type": SEARCH_CIRCLE, "maximum_descriptor_distance": 35.0,
                          "maximum_distance_ratio_to_second_best": 0.9, "minimum_matching_ratio": 0.2,
                          "minimum_descriptor_distance": 30.0, "descriptor_distance_step_size_pixels": 5.0,
                          "maximum_search_radius_pixels": 100, "minimum_search_radius_pixels": 25,
                          "search_radius_step_size_pixels": 5, "minimum_number_of_iterations": 5,
                          "maximum_estimate_change_norm_for_convergence": 0.01,
                          "number_of_solver_iterations_per_projection": 5},
    # icl.conf:566-570 (slice), :459-463 (robustifier), :295-299 (damping), :50-64 (aligner)
    "aligner": {"factor_type": FACTOR_DEPTH, "diagonal_info": (1.0, 1.0, 10.0), "chi_threshold": 10.0,
                "enable_inverse_depth_weighting": 0, "damping": 0.1, "max_iterations": 100,
                "min_num_inliers": 6, "min_num_correspondences": 0,  # icl.conf:584
                "enable_inlier_only_runs": 1, "keep_only_inlier_correspondences": 1},  # icl.conf:50-53, :57-59
    "depth": {"min": 0.5, "max": 6.0},
}

TUM = {
    "name": "tum",
    # intrinsics are not in the reference: the customary fr1 values are used
    "camera": {"fx": 525.0, "fy": 525.0, "cx": 319.5, "cy": 239.5, "cols": 640, "rows": 480, "baseline_m": 0.0},
    "projector": {"range_min": 0.01, "range_max": 7.5},  # tum.conf:133-136
    "stereo_matcher": None,
    "triangulator": None,
    # tum.conf:498-536
    "projective_finder": {"search_type": SEARCH_CIRCLE, "maximum_descriptor_distance": 75.0,
                          "maximum_distance_ratio_to_second_best": 0.7, "minimum_matching_ratio": 0.1,
                          "minimum_descriptor_distance": 35.0, "descriptor_distance_step_size_pixels": 5.0,
                          "maximum_search_radius_pixels": 100, "minimum_search_radius_pixels": 25,
                          "search_radius_step_size_pixels": 5, "minimum_number_of_iterations": 5,
                          "maximum_estimate_change_norm_for_convergence": 1e-5,
                          "number_of_solver_iterations_per_projection": 5},
    # tum.conf:242-246 (slice), :167-171 (robustifier), :146-150 (damping), :90-104 (aligner)
    "aligner": {"factor_type": FACTOR_DEPTH, "diagonal_info": (1.0, 1.0, 10.0), "chi_threshold": 25.0,
                "enable_inverse_depth_weighting": 0, "damping": 0.1, "max_iterations": 100,
                "min_num_inliers": 6, "min_num_correspondences": 0,  # tum.conf:260
                "enable_inlier_only_runs": 1, "keep_only_inlier_correspondences": 1},  # tum.conf:90-93, :97-99
    "depth": {"min": 0.5, "max": 6.0},
}

CONFIGS = {"kitti": KITTI, "euroc": EUROC, "icl": ICL, "tum": TUM}


def get(name):
    return copy.deepcopy(CONFIGS[name])


def baseline_pixels(cfg):
    """b_x = (K * t_right_in_left).x = fx * baseline (triangulator_rigid_stereo.cpp:105-106)"""
    return cfg["camera"]["fx"] * cfg["camera"]["baseline_m"]


def _selfcheck():
    assert abs(baseline_pixels(KITTI) - 386.1448) < 1e-3  # tests/fixtures.hpp:811
    assert math.isclose(_SQRT_FLT_MAX, math.sqrt(3.4028234663852886e38), rel_tol=1e-5)


_selfcheck()
