/*
 * proslam_oracle.h -- CPU restatement ("oracle") of srrg2_proslam's per-frame tracking hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as
 * the checker / the reported CPU baseline.  Nothing under srrg2_proslam_amd/ links, includes or
 * calls anything in oracle/.
 *
 * PARITY STATUS: pinned against the data the reference's own tests hold.
 *   The reference (/root/reference, C++11) cannot be compiled here: it needs catkin, srrg2_core,
 *   srrg2_solver, srrg2_slam_interfaces, srrg_hbst, Eigen3 and OpenCV, none of which is present
 *   (srrg2_proslam/package.xml:10-14).  Its gtests assert exact counts and pose tolerances on the
 *   KITTI / ICL / SceneFlow images under /root/reference/test_data; tools/make_ref_fixtures.py
 *   converts those images and ground-truth poses into tests/golden/ref_*.npz (data, no source), and
 *   tests/test_ref_pins.py, tests/test_ref_tracker.py and tests/test_oracle_aligner_ext.py run this
 *   oracle on them (scenarios + expected values in tests/ref_pins.py, each citing its gtest line):
 *     - feature counts 887, 272/280/270/271, 446/444, 458/444 (KITTI), 259/254, 220/228, 321/338/261 (ICL)
 *                                      (tests/test_feature_extractors.cpp, tests/fixtures.hpp)
 *     - epipolar matcher 150 / 241 matches, self-match 446, every response <= 50
 *                                      (tests/test_correspondence_finders.cpp:152-294)
 *     - brute-force matcher 237 (KITTI, both directions), 319 / 226 / 117 (ICL)   (:14-240)
 *     - adaptors 213 / 177 (KITTI), 321 (ICL), SceneFlow 83 points / 43 inliers, 115 / 59
 *                                      (tests/test_measurement_adaptors.cpp, tests/test_triangulators.cpp)
 *     - projective finders 319 (ICL identity, every search shape), 2, 90 (KITTI circle)   (:297-1140)
 *     - scene clipper 49872 / 136022 / 51 / 242 / 52 visible points, depth-EKF merger 321 -> 337 points
 *                                      (tests/test_scene_clippers.cpp:7-462, tests/test_mergers.cpp:248-355)
 *     - aligner and tracker tolerances against the ground-truth poses
 *                                      (tests/test_aligners.cpp:1035-1261, tests/test_trackers.cpp:7-470)
 *   The KD-tree finder counts (120, 21, 82, 36 -> 104, 56) are pinned as bounds only: the reference's
 *   KD-tree (srrg2_core, absent) answers a radius query from a single leaf, the search here is exact.
 *   Arithmetic that lives in the un-vendored dependencies (pinhole projector, t2tnq, error factors,
 *   robustifier, GN step) is restated from first principles following SURVEY.md Appendix A and marked
 *   BUILD-DEFINED below; it is cross-checked in float64 (tests/ref_pins.py linearize_f64 / gn_step_f64)
 *   and through the reference's pose tolerances on its own images.
 *
 * Conventions: all matrices row-major float[16] 4x4 (SE3) unless noted; descriptors are
 * 32-byte rows (256 bit); "fixed"/"moving"/"Correspondence" follow the reference's naming
 * (SURVEY.md section 8).  Compiled as C99, -O2, -ffp-contract=off, no fast-math, one thread.
 */
#ifndef PROSLAM_ORACLE_H
#define PROSLAM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DESC_BYTES 32

typedef struct {
  int32_t fixed_idx;
  int32_t moving_idx;
  float response;
} orc_corr;

/* warning bits (>0) shared by all entry points; <0 = hard error */
enum {
  ORC_OK                 = 0,
  ORC_WARN_EMPTY_INPUT   = 1,
  ORC_WARN_NO_MATCHES    = 2,
  ORC_WARN_LOW_RATIO     = 4,
  ORC_WARN_RETRIED       = 8,
  ORC_WARN_TRACK_LOST    = 16,
  ORC_WARN_NO_PROJECTION = 32,
  ORC_ERR_NULL           = -1,
  ORC_ERR_CAPACITY       = -2
};

/* ---- descriptor distance (srrg2_core PointDescriptorField::distance, call site
 *      correspondence_finder_descriptor_based_epipolar_impl.cpp:157) ---- */
int orc_hamming256(const uint8_t* a, const uint8_t* b);

/* ---- a1+a2: CorrespondenceFinderDescriptorBasedEpipolar::compute
 *      (correspondence_finder_descriptor_based_epipolar_impl.cpp:8-219) ---- */
typedef struct {
  float maximum_descriptor_distance;           /* bruteforce.h:22-26, default 50 */
  float maximum_distance_ratio_to_second_best; /* bruteforce.h:27-31, default 0.9 */
  float minimum_matching_ratio;                /* bruteforce.h:32-36, default 0.25 */
  int32_t maximum_disparity_pixels;            /* epipolar.h:22-26, default 100 */
  int32_t epipolar_line_thickness_pixels;      /* epipolar.h:28-32, default 0 */
} orc_stereo_params;

/* uv_*: [n][2] floats (u = image x, v = image y).  out capacity must be >= n_left.
 * returns warning bits (>=0) or error (<0); *n_out = number of correspondences. */
int orc_stereo_match(const float* uv_left,
                     const uint8_t* desc_left,
                     int n_left,
                     const float* uv_right,
                     const uint8_t* desc_right,
                     int n_right,
                     const orc_stereo_params* params,
                     orc_corr* out,
                     int capacity,
                     int* n_out);

/* ---- a8(caller): RawDataPreprocessorStereoProjective::compute assembly loop
 *      (raw_data_preprocessor_stereo_projective.cpp:107-132): builds (uL,vL,uR,vR) points,
 *      drops horizontal OR vertical disparity < 0.  out_uvuv [<=n_corr][4], out_src = index of
 *      the left keypoint whose descriptor/intensity the point carries.  returns count. */
int orc_stereo_assemble(const float* uv_left,
                        const float* uv_right,
                        const orc_corr* corr,
                        int n_corr,
                        float* out_uvuv,
                        int32_t* out_src_left);

/* ---- a4: TriangulatorRigidStereo (triangulator_rigid_stereo.cpp:7-109) ---- */
typedef struct {
  float fx, fy, cx, cy;
  float b_x;                      /* (K * t_right_in_left).x, triangulator_rigid_stereo.cpp:105-106 */
  float minimum_disparity_pixels; /* .h:34-38 default 1 */
  float infinity_depth_meters;    /* .h:39-43 default sqrt(FLT_MAX) */
} orc_triangulator_params;

/* uvuv [n][4]; xyz [n][3]; valid [n] (1 = Valid, 0 = Invalid: coordinates left at 0). */
void orc_triangulate(const float* uvuv,
                     int n,
                     const orc_triangulator_params* params,
                     float* xyz,
                     uint8_t* valid);

/* ---- SE(3) helpers (srrg2_core geometry3d, BUILD-DEFINED restatement) ---- */
void orc_se3_identity(float* T);
void orc_se3_inverse(const float* T, float* Tinv);       /* [R^T | -R^T t] */
void orc_se3_mul(const float* A, const float* B, float* C);
void orc_t2tnq(const float* T, float* v6);               /* translation + normalized quaternion xyz */
void orc_tnq2t(const float* v6, float* T);               /* inverse of t2tnq */

/* ---- a6: PointProjectorPinhole_::compute (external; BUILD-DEFINED) ---- */
typedef struct {
  float fx, fy, cx, cy;
  int32_t canvas_cols, canvas_rows;
  float range_min, range_max;
} orc_projector;

/* camera_pose = pose of the camera in the frame of the points (the finder passes
 * local_map_in_sensor^-1).  outputs: uvz [<=n][3], indices [<=n]; returns count. */
int orc_project(const orc_projector* proj,
                const float* camera_pose,
                const float* xyz,
                int n,
                float* uvz,
                int32_t* indices);

/* ---- a5,a7-a10: projective correspondence finder (stateful) ---- */
enum { ORC_SEARCH_KDTREE = 0, ORC_SEARCH_SQUARE = 1, ORC_SEARCH_CIRCLE = 2, ORC_SEARCH_RHOMBUS = 3 };

typedef struct {
  /* bruteforce base params (bruteforce.h:22-36) */
  float maximum_descriptor_distance;
  float maximum_distance_ratio_to_second_best;
  float minimum_matching_ratio;
  /* projective base params (projective_base.h:30-74) */
  float minimum_descriptor_distance;
  float descriptor_distance_step_size_pixels;
  uint64_t maximum_search_radius_pixels;
  uint64_t minimum_search_radius_pixels;
  uint64_t search_radius_step_size_pixels;
  uint64_t minimum_number_of_iterations;
  float maximum_estimate_change_norm_for_convergence;
  uint64_t number_of_solver_iterations_per_projection;
  int32_t search_type;
  orc_projector projector;
  /* KD-tree finder only: PARAM minimum_number_of_points_per_cluster (projective_kdtree.h:24-28); 0 = its default, 10 */
  int32_t minimum_number_of_points_per_cluster;
} orc_pcf_params;

typedef struct orc_pcf orc_pcf;

orc_pcf* orc_pcf_create(const orc_pcf_params* params);
void orc_pcf_destroy(orc_pcf* h);
/* param update == PARAM(..., &_config_changed): flags a config change (projective_base.h:30-44) */
void orc_pcf_set_params(orc_pcf* h, const orc_pcf_params* params);
/* fixed: coordinates [n][fixed_dim] (only the first two are used), descriptors [n][32] */
void orc_pcf_set_fixed(orc_pcf* h, const float* coords, int fixed_dim, const uint8_t* desc, int n);
void orc_pcf_set_moving(orc_pcf* h, const float* xyz, const uint8_t* desc, int n);
void orc_pcf_set_local_map_in_sensor(orc_pcf* h, const float* T);
void orc_pcf_get_local_map_in_sensor(const orc_pcf* h, float* T);
void orc_pcf_set_search_radius(orc_pcf* h, uint64_t r);     /* projective_base.h:82-85 */
void orc_pcf_set_descriptor_distance(orc_pcf* h, float d);  /* projective_base.h:94-97 */
uint64_t orc_pcf_search_radius(const orc_pcf* h);
float orc_pcf_descriptor_distance(const orc_pcf* h);
uint64_t orc_pcf_iteration(const orc_pcf* h);
int orc_pcf_has_converged(const orc_pcf* h);
int orc_pcf_num_recomputes(const orc_pcf* h); /* bookkeeping for tests: # of full searches so far */
/* compute(): out capacity >= n_fixed. the correspondence vector persists across calls like
 * the reference's caller-owned CorrespondenceVector: on "nothing new" calls it is untouched. */
int orc_pcf_compute(orc_pcf* h, orc_corr* out, int capacity, int* n_out);

/* ---- a11-a13: aligner slice (setupFactor + errorAndJacobian + robustifier + H/b) ---- */
enum { ORC_FACTOR_MONO = 2, ORC_FACTOR_DEPTH = 3, ORC_FACTOR_STEREO = 4 };

typedef struct {
  int32_t factor_type;          /* fixed dimension: 2 mono, 3 depth, 4 rectified stereo */
  float fx, fy, cx, cy;
  float image_cols, image_rows; /* factor->setImageDim (aligner_slice_processor_projective.cpp:38-39) */
  float baseline_left_in_right_px[3]; /* K * t_left_in_right (.cpp:98-104); stereo only */
  float diagonal_info[3];       /* param_diagonal_info_matrix (.cpp:47) */
  float chi_threshold;          /* RobustifierSaturated chi_threshold */
  int32_t enable_inverse_depth_weighting; /* .cpp:107-112 */
  float mean_disparity;         /* bindFixed (.cpp:76-89), used iff weighting enabled */
  float damping;                /* IterationAlgorithmGN damping (kitti.conf:310-315) */
  int32_t max_iterations;       /* MultiAligner3DQR max_iterations (kitti.conf:990-991) */
  int32_t min_num_inliers;      /* kitti.conf:993-994 */
  int32_t min_num_correspondences; /* slice min_num_correspondences (kitti.conf:286-287) */
  /* MultiAligner3DQR flags of the RGB-D configurations (icl.conf:50-64, tum.conf:90-104; 0 in kitti.conf:980-1010).
   * The class is external and its loop unpinned: BUILD-DEFINED as
   *   enable_inlier_only_runs: after the max_iterations loop, if the last linearisation had >= min_num_inliers
   *     inliers, `inlier_only_iterations` (<= 0: max_iterations) further GN iterations on the frozen correspondence
   *     vector (the finder is not called) in which kernelised factors (chi2 > threshold) are suppressed entirely;
   *   keep_only_inlier_correspondences: the returned vector keeps only the correspondences whose factor was an
   *     inlier in the last linearisation (order preserved). */
  int32_t enable_inlier_only_runs;
  int32_t keep_only_inlier_correspondences;
  int32_t inlier_only_iterations;
  /* ...WithSensor factor variants (aligner_slice_processor_projective.h:80-83,88-91, tests/test_aligners.cpp:142-279):
   * the estimate X is the ROBOT pose (moving in fixed robot frame); points reach the camera through
   * A = sensor_in_robot^-1 * X, which is what the finder and the factor see; the perturbation stays on X. */
  int32_t with_sensor;
  float sensor_in_robot[16];
  /* AlignerSliceMotionModel3D + MotionModelConstantVelocity3D stand-in (kitti.conf:257-260,747-772; external, weights
   * unpinned: the .conf gives the slice no information matrix, so the default identity is assumed): a prior factor
   * e = t2tnq(Z^-1 X) with J = I, H += diag(info), b += info * e, re-evaluated every iteration.  Z = prior mean
   * (the constant-velocity prediction of movingInFixed; identity when the local map was clipped at the prediction). */
  int32_t enable_motion_prior;
  float motion_prior_info[6];
} orc_aligner_params;

/* info scale per moving point: (n_opt > 2 ? 1 + log(n_opt) : 1), aligner_slice_processor_projective.cpp:46-52 */
void orc_info_scale_from_nopt(const uint32_t* n_opt, int n, float* scale);
/* mean disparity over ALL fixed points, aligner_slice_processor_projective.cpp:80-88 */
float orc_mean_disparity(const float* fixed_uvuv, int n);

/* SWEEP SWITCHES -- readings of the BUILD-DEFINED arithmetic of rows a13 / a14 (the srrg2_solver factors, robustifier and
 * damping are not in the reference tree).  Form 0 of every switch is what the oracle and the device implement: the ONE family
 * under which every pose bound the reference's gtests assert on its own images holds (tools/sweep_a13.py, table in
 * profiles/r04/sweep_a13_grid.txt).  Only the sweep sets anything else. */
typedef struct {
  int32_t kernel_form;  /* kernelised factor: 0 Omega / chi; 1 Omega * tau / chi; 2 Omega * sqrt(tau / chi); 3 Omega * 0 */
  int32_t idw_form;     /* translation weight, dn = d / mean disparity: 0 min(0.01 + dn, 1); 1 clamp(dn, 0.01, 1); 2 sqrt of 1;
                           3 form 1 on Omega instead of J; 4 clamp(1 / dn, 0.01, 1); 5 max(dn, 0.01); 6 square of 1; 7 off */
  int32_t damping_form; /* 0 H + lambda diag(H); 1 H + lambda I */
  int32_t v_row;        /* stereo row 1 measurement: 0 vL; 1 (vL + vR) / 2 */
  int32_t chi_compare;  /* 0 kernel active when chi > tau; 1 when chi >= tau */
  int32_t bounds_form;  /* 0 prediction must lie in [0, cols] x [0, rows]; 1 no image test; 2 [0, cols) x [0, rows) */
  int32_t accum_form;   /* 0 camera-frame sums + one rotation of the summed system (shipped); 1 J^T Omega J entry by entry (rounds 1-3) */
} orc_variant;
void orc_set_variant(const orc_variant* v);

/* one linearization: H [36] row-major, b [6] (b = sum J^T Omega e), chi2 sum, counts.
 * fixed [n_f][factor_type], moving [n_m][3], info_scale [n_m]. */
typedef struct {
  float H[36];
  float b[6];
  float chi_inliers;    /* sum of chi2 over inliers */
  float chi_total;      /* sum of (kernelized) chi2 over all valid terms */
  int32_t num_inliers;  /* chi2 <= threshold */
  int32_t num_outliers; /* chi2 > threshold (kernelized) */
  int32_t num_invalid;  /* behind camera / outside image: skipped */
} orc_linear_system;

void orc_linearize(const orc_aligner_params* p,
                   const float* X,
                   const orc_corr* corr,
                   int n_corr,
                   const float* fixed,
                   const float* moving_xyz,
                   const float* info_scale,
                   orc_linear_system* out);
/* the same with the sensor transform applied (p->with_sensor), optionally suppressing kernelised factors
 * (inlier-only run) and reporting the class of every correspondence: 0 inlier, 1 kernelised, 2 invalid */
void orc_linearize_ex(const orc_aligner_params* p,
                      const float* X,
                      const orc_corr* corr,
                      int n_corr,
                      const float* fixed,
                      const float* moving_xyz,
                      const float* info_scale,
                      int inlier_only,
                      uint8_t* cls_out,
                      orc_linear_system* out);
/* adds the motion prior (p->enable_motion_prior) at X to sys; prior_mean NULL = identity */
void orc_add_motion_prior(const orc_aligner_params* p, const float* X, const float* prior_mean, orc_linear_system* sys);
/* constant-velocity prediction: pose_pred = pose_prev1 * (pose_prev2^-1 * pose_prev1) (MotionModelConstantVelocity3D) */
void orc_motion_predict(const float* pose_prev2, const float* pose_prev1, float* pose_pred);

/* (H + damping diag(H)) dx = -b; X <- X * exp(dx). returns 0 ok, 1 if the system was not SPD (X unchanged) */
int orc_gn_step(const orc_linear_system* sys, float damping, float* X);

/* ---- a14: the per-frame loop MultiAligner3DQR::compute drives (external; restated minimal) ---- */
typedef struct {
  float X[16];           /* final movingInFixed */
  int32_t status;        /* 1 Success, 0 Fail */
  int32_t iterations;
  int32_t num_inliers;
  int32_t num_correspondences;
  int32_t warnings;      /* OR of finder warnings over the loop */
} orc_align_result;

/* runs max_iterations of: finder.setLocalMapInSensor(X); finder.compute(); setupFactor;
 * linearize; step.  corr_out (capacity n_fixed) receives the final correspondences.
 * prior_H/prior_b optional additive prior (motion-model slice), may be NULL. */
void orc_align_frame(orc_pcf* finder,
                     const orc_aligner_params* p,
                     const float* fixed,
                     int n_fixed,
                     const float* moving_xyz,
                     const float* info_scale,
                     int n_moving,
                     const float* X_init,
                     const float* prior_H,
                     const float* prior_b,
                     orc_corr* corr_out,
                     int* n_corr_out,
                     orc_align_result* result);
/* the same with a motion-prior mean (NULL = identity) */
void orc_align_frame_ex(orc_pcf* finder,
                        const orc_aligner_params* p,
                        const float* fixed,
                        int n_fixed,
                        const float* moving_xyz,
                        const float* info_scale,
                        int n_moving,
                        const float* X_init,
                        const float* prior_H,
                        const float* prior_b,
                        const float* prior_mean,
                        orc_corr* corr_out,
                        int* n_corr_out,
                        orc_align_result* result);

/* ---- section 8f next #4: bijective brute-force matcher (bruteforce_impl.cpp:8-293) ---- */
int orc_bruteforce_match(const uint8_t* desc_fixed,
                         int n_fixed,
                         const uint8_t* desc_moving,
                         int n_moving,
                         float maximum_descriptor_distance,
                         float maximum_distance_ratio,
                         orc_corr* out,
                         int capacity,
                         int* n_out);

/* ---- section 8f next #2: SceneClipperProjective3D::compute (mapping/scene_clipper_projective_3d.cpp:9-67)
 * camera pose = robot_in_local_map * sensor_in_robot (:46); the projector keeps the points inside
 * range and canvas and returns them IN THE CAMERA FRAME with their source indices (:53); if
 * sensor_in_robot is not exactly the identity the kept points are moved into the robot frame (:61-63).
 * scene_xyzw: [n][4], w is carried through (the per-landmark information scale column);
 * returns ORC_WARN_EMPTY_INPUT for an empty scene (outputs untouched, :21-28), ORC_WARN_NO_PROJECTION
 * when nothing survives (:55-58), else 0. */
int orc_scene_clip(const orc_projector* proj,
                   const float* robot_in_local_map,
                   const float* sensor_in_robot,
                   const float* scene_xyzw,
                   const uint8_t* scene_desc,
                   int n,
                   float* clipped_xyzw,
                   uint8_t* clipped_desc,
                   int32_t* global_indices,
                   int* n_clipped);

#ifdef __cplusplus
}
#endif
#endif
