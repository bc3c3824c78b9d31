/* proslam_oracle_mapping.h -- CPU oracle, SURVEY.md section 8f row 1: landmark estimators + projective
 * mergers.  TEST INFRASTRUCTURE ONLY (same rules as proslam_oracle.h): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED: the reference cannot be built here and its tests for this row
 * (tests/test_mergers.cpp, test_landmark_estimators.cpp, test_stereo_projective_point_ekf.cpp) hold
 * tolerance gates on OpenCV / dataset dependent inputs, no golden vectors.  Restated from
 *   mapping/landmarks/landmark_estimator_base.hpp:7-73
 *   mapping/landmarks/landmark_estimator_weighted_mean_impl.cpp:7-41
 *   mapping/landmarks/landmark_estimator_ekf_impl.cpp:7-82
 *   mapping/landmarks/filters/point_ekf_base.hpp:63-125, stereo_projective_point_ekf_impl.cpp:13-48,
 *   projective_depth_point_ekf_impl.cpp:7-36, projective_point_ekf_impl.cpp:16-43
 *   mapping/landmarks/landmark_estimator_pose_based_smoother_impl.cpp:7-148
 *   mapping/mergers/merger_projective_impl.cpp:8-328, merger_projective_rigid_stereo_impl.cpp:8-77,
 *   merger_projective_rigid_stereo_triangulation_impl.cpp:7-39, merger_projective_depth_ekf_impl.cpp:8-73
 * BUILD-DEFINED (external srrg2_core code or unspecified evaluation order): PointStatisticsField3D
 * (state, covariance, numberOfOptimizations, measurements; addOptimizationResult = set state (+ covariance)
 * and increment the counter), the order of every matrix product (explicit loops, sequential sums), the
 * 4x4 / 3x3 innovation inverse (LDL^T in double), the smoother's 3x3 solve (full-pivot elimination in
 * float), the depth unprojector ((u - cx) / fx * d, (v - cy) / fy * d, d; valid iff d > 0), the bounded
 * measurement history (max_measurements per landmark; exceeding it is a loud error). */
#ifndef PROSLAM_ORACLE_MAPPING_H
#define PROSLAM_ORACLE_MAPPING_H
#include <stdint.h>

#include "proslam_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_EST_WEIGHTED_MEAN = 0, ORC_EST_EKF = 1, ORC_EST_SMOOTHER = 2 };
enum { ORC_MERGER_STEREO_TRIANGULATION = 0, ORC_MERGER_STEREO_EKF = 1, ORC_MERGER_DEPTH_EKF = 2 };
enum { ORC_ERR_HISTORY = -7, ORC_ERR_SCENE_FULL = -8, ORC_ERR_DUPLICATE = -9 };

/* PointStatisticsField3D::CameraMeasurement; the two transforms are per frame, so a measurement
 * stores the index of its frame in the pose table */
typedef struct {
  float point_in_image[3];  /* measurement.head(3) */
  float point_in_camera[3]; /* landmark in the sensor frame at that time */
  int32_t frame;
} orc_camera_measurement;

/* pose table entry: sensor_in_world and world_in_sensor, 3x4 row-major each */
typedef struct {
  float sensor_in_world[12];
  float world_in_sensor[12];
} orc_frame_pose;

/* one local map (scene): structure of arrays, capacity rows */
typedef struct {
  int32_t capacity;
  int32_t max_measurements;
  int32_t n_points;
  float* coords;                /* [capacity][4] xyz in the local map frame (w unused) */
  uint8_t* desc;                /* [capacity][32] */
  float* state;                 /* [capacity][4] statistics().state() (world) */
  float* covariance;            /* [capacity][9] */
  uint32_t* n_opt;              /* numberOfOptimizations */
  uint8_t* inlier;              /* isInlier */
  uint32_t* n_meas;             /* measurements().size() */
  orc_camera_measurement* meas; /* [capacity][max_measurements] */
} orc_map;

typedef struct {
  int32_t type;            /* ORC_EST_* */
  int32_t measurement_dim; /* 2 mono, 3 depth, 4 stereo */
  float maximum_distance_geometry_meters_squared; /* landmark_estimator_base.hpp:21-25 */
  /* EKF (landmark_estimator_ekf.h:31-47) + filter calibration */
  double minimum_state_element_covariance;
  double maximum_covariance_norm_squared;
  double fx, fy, cx, cy, b_x, b_y;
  /* smoother (landmark_estimator_pose_based_smoother.h:17-39) */
  uint32_t maximum_number_of_iterations;
  float convergence_criterion_minimum_chi2_delta;
  float maximum_reprojection_error_pixels_squared;
  uint32_t minimum_number_of_measurements_for_optimization;
  float camera_matrix[9];
} orc_estimator_params;

/* LandmarkEstimator*::compute on landmark `index` of `map` with measurement (measurement_dim floats)
 * and (triangulation merger only) the landmark in the sensor frame; transforms as passed to
 * setTransforms (landmark_estimator_base.hpp:47-56): measurement_in_world, measurement_in_scene (4x4).
 * `frame` is the pose-table slot of this frame (poses[frame] must hold the same transforms).
 * Returns 1 if the landmark ends as inlier, 0 if not, < 0 on error. */
int orc_landmark_estimate(const orc_estimator_params* P,
                          const float* measurement_in_world,
                          const float* measurement_in_scene,
                          const orc_frame_pose* poses,
                          int32_t frame,
                          orc_map* map,
                          int32_t index,
                          const float* measurement,
                          const float* landmark_in_sensor);

typedef struct {
  int32_t variant;          /* ORC_MERGER_* */
  int32_t enable_binning;   /* MergerCorrespondence_ param_enable_binning */
  uint32_t number_of_row_bins, number_of_col_bins; /* merger_projective.h:47-56 */
  int32_t canvas_rows, canvas_cols;                /* projector canvas */
  float maximum_distance_appearance;               /* :42-46 */
  uint32_t target_number_of_merges;
  float target_merge_ratio;
  orc_triangulator_params triangulator;            /* stereo variants */
  float fx, fy, cx, cy;                            /* depth variant: unprojector */
  orc_estimator_params estimator;
} orc_merger_params;

typedef struct {
  int32_t n_merged;
  int32_t n_added;
  int32_t flags; /* ORC_WARN_* bits: NO_MATCHES = all merge attempts failed, LOW_RATIO = low merge ratio */
} orc_merge_result;

/* MergerProjective_::compute (merger_projective_impl.cpp:8-190) for one frame.
 * measurement: [n_meas][measurement_dim] image-space points + descriptors; correspondences index
 * (fixed_idx -> scene, moving_idx -> measurement); scene_index_map optional (clipped -> full scene index).
 * Returns 0 or a negative error. */
int orc_merge(const orc_merger_params* P,
              const float* measurement_in_world,
              const float* measurement_in_scene,
              orc_frame_pose* poses,
              int32_t frame,
              orc_map* map,
              const float* measurement,
              const uint8_t* measurement_desc,
              int32_t n_measured,
              const orc_corr* corr,
              int32_t n_corr,
              const int32_t* scene_index_map,
              orc_merge_result* result);

#ifdef __cplusplus
}
#endif
#endif
