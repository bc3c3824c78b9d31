/*
 * proslam_hip.h -- C-ABI of libproslam_hip.so: the MI355X (gfx950) implementation of
 * srrg2_proslam's per-frame tracking hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference/srrg2_proslam/src/srrg2_proslam/; CF/ = registration/correspondence_finders/).
 * The boundary is plain C: pointers, sizes, POD structs.  No torch / Eigen / OpenCV types.
 *
 * Two flavours of every operator:
 *   - "host" calls take host pointers for ONE frame and are what a srrg2 plugin adapter's
 *     compute() binds; they upload, launch, download and synchronise.
 *   - "_batch" calls take DEVICE pointers for B independent frames (one per sequence), enqueue
 *     on the context's HIP stream and return without synchronising.  Per-frame status words are
 *     written to device memory.
 *
 * Status convention (mirrors the reference's error behaviour):
 *   0      ok
 *   < 0    hard error (the reference throws std::runtime_error: CF/..bruteforce_impl.cpp:203-216)
 *   > 0    OR of warning bits (the reference prints a warning and returns:
 *          CF/..bruteforce_impl.cpp:217-226,237-242; CF/..epipolar_impl.cpp:211-216;
 *          CF/..projective_base_impl.cpp:228-263)
 *
 * Supported domain of the device kernels (checked; violations are hard errors, never silent):
 *   keypoint coordinates 0 <= u < 32768, 0 <= v < image_rows <= 4096; keypoints per image <= 8192
 *   (stereo matcher) / fixed points <= 32767 (lattice finder, the reference's own int16 limit,
 *   CF/correspondence_finder_projective_square_impl.cpp:20-22).
 */
#ifndef PROSLAM_HIP_H
#define PROSLAM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRS_API __attribute__((visibility("default")))
#define PRS_DESC_BYTES 32 /* 256-bit binary descriptor row (cv::Mat 1x32 CV_8U in the reference) */

/* ---- status ---- */
enum {
  PRS_OK                 = 0,
  PRS_WARN_EMPTY_INPUT   = 1,  /* CF/..bruteforce_impl.cpp:217-226 */
  PRS_WARN_NO_MATCHES    = 2,  /* CF/..bruteforce_impl.cpp:237-242 */
  PRS_WARN_LOW_RATIO     = 4,  /* CF/..epipolar_impl.cpp:211-216, CF/..projective_base_impl.cpp:228-232 */
  PRS_WARN_RETRIED       = 8,  /* CF/..projective_base_impl.cpp:235-249 */
  PRS_WARN_TRACK_LOST    = 16, /* CF/..projective_base_impl.cpp:251-259 */
  PRS_WARN_NO_PROJECTION = 32, /* CF/..projective_base_impl.cpp:167-171 */
  PRS_ERR_NULL           = -1, /* unset input (reference: throw) */
  PRS_ERR_CAPACITY       = -2, /* output buffer too small */
  PRS_ERR_HIP            = -3, /* HIP runtime failure; see prs_last_error() */
  PRS_ERR_RANGE          = -4, /* input outside the supported domain (see above) */
  PRS_ERR_UNSUPPORTED    = -5, /* size beyond kernel limits */
  PRS_ERR_NO_DEVICE      = -6  /* no MI355X-class device visible */
};

/* Correspondence{int fixed_idx, int moving_idx, float response}
 * (srrg2_core, emitted at CF/..epipolar_impl.cpp:177-178) */
typedef struct {
  int32_t fixed_idx;
  int32_t moving_idx;
  float response;
} prs_corr;

/* image-plane keypoint (coordinates()(0), coordinates()(1)) */
typedef struct {
  float u;
  float v;
} prs_kp2;

/* ---- context: one device, one stream, scratch memory.  Not re-entrant (the reference's
 *      finders are not either); different contexts are independent. ---- */
typedef struct prs_context prs_context;
PRS_API int prs_context_create(int device_id, prs_context** ctx);
PRS_API int prs_context_destroy(prs_context* ctx);
/* enqueue on a caller-owned hipStream_t (e.g. torch's current stream); NULL = HIP's default stream */
PRS_API int prs_context_set_stream(prs_context* ctx, void* hip_stream);
/* go back to the non-blocking stream the context created for itself (the initial state) */
PRS_API int prs_context_use_own_stream(prs_context* ctx);
PRS_API int prs_context_synchronize(prs_context* ctx);
/* measurement: when on, prs_align_batch_run brackets every launch of its two kernels (projective search, Gauss-Newton
 * rounds) with HIP events on the context's stream and accumulates their durations; enabling resets the sums */
PRS_API int prs_context_enable_timing(prs_context* ctx, int32_t on);
PRS_API int prs_context_get_align_timing(prs_context* ctx, double* search_ms, double* gn_ms, int64_t* search_launches, int64_t* gn_launches);
/* the same sums by round of the batch: search_ms16[r] / gn_ms16[r] = total time of the r-th search / Gauss-Newton launch over
 * `batches` timed batches (round 15 collects every later round) */
PRS_API int prs_context_get_align_round_timing(prs_context* ctx, double* search_ms16, double* gn_ms16, int64_t* batches);
/* Dense phase of the brute-force matcher (prs_bruteforce_match_batch): which kernels score the N_f x N_m pairs.  Results are
 * identical; only the cost differs.
 *   PRS_BF_DENSE_MATRIX_WHEN_FULL (the default): a batch of 32 or more cloud pairs (of at least 256 x 64 points) runs one workgroup
 *     per pair with the distances from v_mfma_i32_16x16x64_i8 (exact: integer products) and the registration state in LDS; fewer pairs
 *     run the popcount kernels, which spread a pair over several workgroups.  On real cloud pairs (KITTI stereo pairs, ~750 points a
 *     side, 1.6 % of the pairs within 50 bits) 1.8x the popcount kernels at 1024 pairs and 2x at 128, on 1024 pairs of uniform random
 *     rows 2.7x; 32 - 63 pairs of uniform random rows are the one shape it loses on (0.23 against 0.14 ms) (profiles/r06/README.md).
 *   PRS_BF_DENSE_POPCOUNT: v_xor / v_bcnt on the vector units for every batch size.
 *   PRS_BF_DENSE_MATRIX: always the matrix cores (the fused shape where it applies, else a split matrix-core kernel + a registration
 *     launch that re-scores what it selects: fast on uniform random rows, slow on real ones; tests, A-B runs).
 * The environment variable PRS_BF_MFMA (0 / auto / 1), read when the context is created, sets the initial mode. */
#define PRS_BF_DENSE_POPCOUNT 0
#define PRS_BF_DENSE_MATRIX_WHEN_FULL 1
#define PRS_BF_DENSE_MATRIX 2
PRS_API int prs_context_set_bruteforce_dense_phase(prs_context* ctx, int32_t mode);
PRS_API const char* prs_last_error(const prs_context* ctx);
PRS_API const char* prs_status_string(int status);
PRS_API int prs_version(void);
/* The parameter structs below carry no size field and grow at their END between versions (round 5 added three int32 fields to
 * prs_aligner_params).  PRS_ABI_VERSION is what this header describes, prs_version() what the loaded library was built from; a
 * client checks that they agree once (prs_abi_check: also the sizes of the structs it will pass, as the client's compiler laid
 * them out) instead of finding out through a library that reads past a shorter struct (101 -> 102: step_norm_exit at the end of
 * prs_aligner_params; 102 also adds the entry points prs_abi_check and prs_context_set_bruteforce_dense_phase).  Callers memset() parameter structs before
 * filling them, so that fields they do not know select the shipped defaults (all zero). */
#define PRS_ABI_VERSION 102
PRS_API int prs_abi_check(int32_t header_version, uint64_t sizeof_stereo_params, uint64_t sizeof_pcf_params, uint64_t sizeof_aligner_params,
                          uint64_t sizeof_align_batch);
#define PRS_ABI_CHECK() prs_abi_check(PRS_ABI_VERSION, sizeof(prs_stereo_params), sizeof(prs_pcf_params), sizeof(prs_aligner_params), sizeof(prs_align_batch))

/* ================================================================================================
 * Stereo epipolar matcher
 * replaces CorrespondenceFinderDescriptorBasedEpipolar<..>::compute (CF/..epipolar_impl.cpp:46-219)
 * incl. Feature/_sortFeatureVector (:8-42) and the pre/post contract (CF/..bruteforce_impl.cpp:203-243)
 * ============================================================================================== */
typedef struct {
  float maximum_descriptor_distance;           /* CF/..bruteforce.h:22-26 */
  float maximum_distance_ratio_to_second_best; /* CF/..bruteforce.h:27-31 */
  float minimum_matching_ratio;                /* CF/..bruteforce.h:32-36 */
  int32_t maximum_disparity_pixels;            /* CF/..epipolar.h:22-26 */
  int32_t epipolar_line_thickness_pixels;      /* CF/..epipolar.h:28-32 */
  int32_t image_rows;                          /* extent of the row table: 0 <= v < image_rows */
  int32_t image_cols;                          /* reserved (the column-binned kernel of round 1 was removed); ignored */
} prs_stereo_params;

/* triangulation parameters, TriangulatorRigidStereo (mapping/triangulator_rigid_stereo.h:34-58,
 * .cpp:88-109): b_x = (K * t_right_in_left).x */
typedef struct {
  float fx, fy, cx, cy;
  float b_x;
  float minimum_disparity_pixels;
  float infinity_depth_meters;
} prs_triangulator_params;

/* host, one frame.  fixed = left keypoints, moving = right keypoints
 * (sensor_processing/raw_data_preprocessor_stereo_projective.cpp:99-102).
 * out: capacity >= n_left; order = sorted-left traversal per offset pass like the reference. */
PRS_API int prs_stereo_match(prs_context* ctx,
                             const prs_stereo_params* params,
                             const prs_kp2* left,
                             const uint8_t* desc_left,
                             int32_t n_left,
                             const prs_kp2* right,
                             const uint8_t* desc_right,
                             int32_t n_right,
                             prs_corr* out,
                             int32_t capacity,
                             int32_t* n_out);

/* device-resident batch: frame b uses element range [b*stride, b*stride + n[b]) of every array.
 * Optional fused "adaptor + triangulator" epilogue (all four pointers non-NULL to enable):
 * replaces the assembly loop of RawDataPreprocessorStereoProjective::compute
 * (sensor_processing/raw_data_preprocessor_stereo_projective.cpp:107-132: (uL,vL,uR,vR) points,
 * matches with negative horizontal or vertical disparity dropped, left descriptor kept) and
 * TriangulatorRigidStereo::compute (mapping/triangulator_rigid_stereo.cpp:7-56) on its output. */
typedef struct {
  int32_t batch;
  int32_t stride;            /* keypoint capacity per image and frame (elements) */
  const prs_kp2* left_kp;    /* [batch][stride] */
  const uint8_t* left_desc;  /* [batch][stride][32], 16-byte aligned */
  const int32_t* n_left;     /* [batch] */
  const prs_kp2* right_kp;
  const uint8_t* right_desc;
  const int32_t* n_right;
  prs_corr* matches;         /* [batch][stride] */
  int32_t* n_matches;        /* [batch] */
  int32_t* status;           /* [batch] */
  /* optional epilogue outputs (NULL to skip) */
  float* fixed_uvuv;         /* [batch][stride][4]  (uL,vL,uR,vR) */
  uint8_t* fixed_desc;       /* [batch][stride][32] left descriptor of the match */
  int32_t* n_fixed;          /* [batch] */
  float* fixed_xyz;          /* [batch][stride][4]  triangulated (x,y,z, valid ? 1 : 0) */
  const prs_triangulator_params* triangulator; /* HOST pointer, read at enqueue */
} prs_stereo_batch;

PRS_API int prs_stereo_match_batch(prs_context* ctx,
                                   const prs_stereo_params* params,
                                   const prs_stereo_batch* batch);

/* ================================================================================================
 * Rectified stereo triangulation
 * replaces TriangulatorRigidStereo::compute / triangulateRectifiedMidpoint
 * (mapping/triangulator_rigid_stereo.cpp:7-56,60-85).  Output size == input size; points with
 * uL - uR < minimum_disparity_pixels are flagged invalid (valid[i] = 0, xyz = 0).
 * ============================================================================================== */
PRS_API int prs_triangulate(prs_context* ctx,
                            const prs_triangulator_params* params,
                            const float* uvuv, /* host [n][4] */
                            int32_t n,
                            float* xyz,        /* host [n][3] */
                            uint8_t* valid);   /* host [n] */

/* device: uvuv [n][4] -> xyz4 [n][4] = (x, y, z, valid) */
PRS_API int prs_triangulate_dev(prs_context* ctx,
                                const prs_triangulator_params* params,
                                const float* d_uvuv,
                                int64_t n,
                                float* d_xyz4);

/* ================================================================================================
 * Projective correspondence finder + reprojection-error Gauss-Newton aligner
 * replaces CorrespondenceFinderProjectiveBase<..>::compute and its Square / Circle / Rhombus /
 * KDTree search patterns (CF/correspondence_finder_projective_base_impl.cpp:105-293,
 * CF/..square_impl.cpp:8-118, CF/..circle_impl.cpp:8-94, CF/..rhombus_impl.cpp:8-93,
 * CF/..kdtree_impl.cpp:8-80), AlignerSliceProcessorProjective{,Depth,Stereo}::setupFactor
 * (registration/aligner_slice_processor_projective.cpp:28-112) and the external arithmetic those
 * configure (srrg2_solver SE3*ProjectiveErrorFactor::errorAndJacobian, RobustifierSaturated,
 * H/b accumulation, damped GN step, MultiAligner3DQR iteration loop).
 * ============================================================================================== */
/* PRS_SEARCH_KDTREE = CorrespondenceFinderProjectiveKDTree (CF/..projective_kdtree_impl.cpp:8-80) over srrg2_core::KDTree<float, 2>
 * (external), restated from its published construction with the constants the reference's own pinned results fix: a cluster
 * is split at its mean along the direction of largest variance until it holds fewer than minimum_number_of_points_per_cluster
 * points or 3 * sqrt(largest eigenvalue of its covariance) < leaf range (= the search radius when _initializeDatabase ran,
 * kept in prs_pcf_state.database_leaf_range); findNeighbors(query, r^2) returns the members of the ONE leaf the query descends
 * to that lie within r.  This reproduces the eight counts the reference asserts for the finder (319, 2, 120, 21, 82, 36, 104, 56:
 * tests/test_correspondence_finders.cpp:330-609); an exhaustive radius search would return supersets.
 * Arithmetic (BUILD-DEFINED): cluster sums as exact integers of the coordinates in 1/16 px, mean / covariance / eigenvector in
 * double, node = (mean, unit normal) in float, side (x - mean_x) * n_x + (y - mean_y) * n_y < 0 -> left in float, leaf members in
 * ascending fixed index; |u|, |v| < 32768.  Clusters beyond the LDS carve of max_fixed (about max_fixed / 3 inner nodes) give
 * PRS_ERR_CAPACITY for that frame. */
enum { PRS_SEARCH_KDTREE = 0, PRS_SEARCH_SQUARE = 1, PRS_SEARCH_CIRCLE = 2, PRS_SEARCH_RHOMBUS = 3 };
enum { PRS_FACTOR_MONO = 2, PRS_FACTOR_DEPTH = 3, PRS_FACTOR_STEREO = 4 }; /* = fixed dimension */

/* PointProjectorPinhole_ parameters reached through param_projector (CF/..projective_base.h:55-59) */
typedef struct {
  float fx, fy, cx, cy;
  int32_t canvas_cols, canvas_rows;
  float range_min, range_max;
} prs_projector;

typedef struct {
  float maximum_descriptor_distance;           /* CF/..bruteforce.h:22-26 */
  float maximum_distance_ratio_to_second_best; /* CF/..bruteforce.h:27-31 */
  float minimum_matching_ratio;                /* CF/..bruteforce.h:32-36 */
  float minimum_descriptor_distance;           /* CF/..projective_base.h:30-34 */
  float descriptor_distance_step_size_pixels;  /* :35-39 */
  uint64_t maximum_search_radius_pixels;       /* :40-44 */
  uint64_t minimum_search_radius_pixels;       /* :45-49 */
  uint64_t search_radius_step_size_pixels;     /* :50-54 */
  uint64_t minimum_number_of_iterations;       /* :60-64 */
  float maximum_estimate_change_norm_for_convergence; /* :65-69 */
  uint64_t number_of_solver_iterations_per_projection; /* :70-74 */
  int32_t search_type;                         /* which subclass: PRS_SEARCH_* */
  prs_projector projector;
  int32_t minimum_number_of_points_per_cluster; /* KD-tree finder only (CF/..projective_kdtree.h:24-28); 0 = its default, 10 */
} prs_pcf_params;

/* the finder members that live across calls and frames (CF/..projective_base.h:132-154).  POD so
 * it can sit in device memory for the batched path; zero-initialise, then set config_changed = 1. */
typedef struct {
  uint64_t search_radius_pixels;
  uint64_t current_iteration;
  float descriptor_distance;
  int32_t has_converged;
  int32_t config_changed;
  int32_t num_recomputes; /* bookkeeping only: number of full searches so far */
  float local_map_in_sensor[16];
  float local_map_in_sensor_previous[16];
  float database_leaf_range; /* KD-tree finder: _search_radius_pixels when _initializeDatabase last ran (the tree's leaf range) */
  int32_t reserved; /* internal hint, no result depends on it: bit 0 = the search has stopped pruning candidates by partial descriptor
                       distances for this finder (its rows are correlated: more than an eighth of a search's queries overflowed) */
} prs_pcf_state;

typedef struct {
  int32_t factor_type;                    /* PRS_FACTOR_*: which error factor / fixed dimension */
  float fx, fy, cx, cy;                   /* factor->setCameraMatrix (aligner_slice_processor_projective.cpp:36-37) */
  float image_cols, image_rows;           /* factor->setImageDim (:38-39) */
  float baseline_left_in_right_px[3];     /* K * t_left_in_right (:98-104), stereo only */
  float diagonal_info[3];                 /* param_diagonal_info_matrix (:47) */
  float chi_threshold;                    /* RobustifierSaturated chi_threshold: a factor with chi > threshold is kernelised,
                                             its Omega scaled by 1 / chi, its chi reported as the threshold (*) */
  int32_t enable_inverse_depth_weighting; /* :107-112: translation columns of J scaled by min(0.01 + d / mean disparity, 1) (*) */
  float mean_disparity;                   /* bindFixed (:76-89); < 0 = compute it on the device */
  float damping;                          /* IterationAlgorithmGN damping: (H + damping diag(H)) dx = -b (*)
                                             (*) srrg2_solver is not in the reference tree; these three readings are the family under
                                             which every pose bound of the reference's gtests holds */
  int32_t max_iterations;                 /* MultiAligner3DQR max_iterations */
  int32_t min_num_inliers;
  int32_t min_num_correspondences;
  int32_t stop_at_fixed_point;            /* 1: leave the loop once the finder has converged and a GN
                                             step reproduces the estimate bit-for-bit (every remaining
                                             iteration would repeat it exactly); 0: always run all */
  /* MultiAligner3DQR flags of the RGB-D configurations (configurations/icl.conf:50-64, tum.conf:90-104; both 0 in
   * kitti.conf:980-1010 / euroc.conf).  The class is external and its loop is not pinned by anything in the reference
   * tree, so the semantics are BUILD-DEFINED:
   *   enable_inlier_only_runs ("toggles additional inlier only runs if sufficient inliers are available"): after the
   *     max_iterations loop, if the last linearisation had >= min_num_inliers inliers, inlier_only_iterations
   *     (<= 0: max_iterations) further Gauss-Newton iterations on the frozen correspondence vector (the finder is not
   *     called) in which kernelised factors (chi2 > threshold) are suppressed instead of saturated;
   *   keep_only_inlier_correspondences ("toggles removal of correspondences which factors are not inliers in the last
   *     iteration"): the returned vector keeps the inliers of the last linearisation, order preserved. */
  int32_t enable_inlier_only_runs;
  int32_t keep_only_inlier_correspondences;
  int32_t inlier_only_iterations;
  /* AlignerSliceProcessorProjective{,Depth,Stereo}WithSensor (aligner_slice_processor_projective.h:80-83,88-91,
   * tests/test_aligners.cpp:142-584): the estimate X is the ROBOT's movingInFixed; points reach the camera through
   * A = sensor_in_robot^-1 * X, which is what the finder projects with and the factor linearises at; the
   * perturbation stays on X.  sensor_in_robot: row-major 4x4 (Platform::getTransform(frame_id, base_frame_id)). */
  int32_t with_sensor;
  float sensor_in_robot[16];
  /* AlignerSliceMotionModel3D + MotionModelConstantVelocity3D (configurations/kitti.conf:257-260,747-772; external,
   * the .conf gives the slice no information matrix: identity assumed, BUILD-DEFINED): prior factor
   * e = t2tnq(Z^-1 X), J = I, H += diag(motion_prior_info), b += motion_prior_info * e, re-evaluated every iteration.
   * Z = prs_align_batch.prior_mean (NULL = identity: the local map was clipped at the motion-model prediction). */
  int32_t enable_motion_prior;
  float motion_prior_info[6];
  /* Readings of the un-vendored srrg2_solver arithmetic, selectable per call.  0 everywhere
   * (what memset gives) = the shipped family, the one under which all 17 pose bounds of the reference's gtests hold
   * (DESIGN.md section 2).  The other values are the readings a maintainer with the real srrg2_solver sources may need instead
   * ("Which upstream lines decide the a13 forms"); the CPU checker switches the same forms (orc_set_variant). */
  int32_t kernel_weight_form;      /* RobustifierSaturated, chi > chi_threshold: PRS_KERNEL_WEIGHT_INV_CHI Omega / chi (shipped);
                                      PRS_KERNEL_WEIGHT_TAU_OVER_CHI Omega * chi_threshold / chi (the form of the in-repo smoother,
                                      mapping/landmarks/landmark_estimator_pose_based_smoother_impl.cpp:81-84) */
  int32_t damping_form;            /* IterationAlgorithmGN: PRS_DAMPING_DIAG (H + damping diag(H)) dx = -b (shipped);
                                      PRS_DAMPING_IDENTITY (H + damping I) dx = -b */
  int32_t translation_weight_form; /* aligner_slice_processor_projective.cpp:107-112, dn = d / mean disparity:
                                      PRS_TRANSLATION_WEIGHT_OFFSET min(0.01 + dn, 1) (shipped, the literal "(0.01+d,1)*I");
                                      PRS_TRANSLATION_WEIGHT_CLAMP clamp(dn, 0.01, 1).  Non-finite results count as 1 (0.01 for a NaN
                                      under CLAMP). */
  /* OPT-IN, does LESS work than the reference (MultiAligner3DQR has no termination criterion in kitti.conf:1006-1009 / euroc.conf:
   * every frame runs max_iterations): > 0 = leave the loop once the finder has latched (has_converged: the correspondences are
   * frozen) and a Gauss-Newton step's 6-vector dx = (translation, normalised quaternion part) has |dx| below this bound.  The
   * correspondence vector is the full run's (it froze before the exit); the pose differs from the max_iterations pose by about the
   * last step (with lambda = 1 on diag(H) the steps halve: |dx| < 1e-5 leaves ~1e-5).  0 (what memset gives) = off. */
  float step_norm_exit;
} prs_aligner_params;
enum { PRS_KERNEL_WEIGHT_INV_CHI = 0, PRS_KERNEL_WEIGHT_TAU_OVER_CHI = 1 };
enum { PRS_DAMPING_DIAG = 0, PRS_DAMPING_IDENTITY = 1 };
enum { PRS_TRANSLATION_WEIGHT_OFFSET = 0, PRS_TRANSLATION_WEIGHT_CLAMP = 1 };

/* The normal equations of one linearisation.  Every one of the 29 sums (21 entries of the upper triangle of H, 6 of b,
 * the two chi) is a FIXED-SHAPE float reduction over the correspondence vector (BUILD-DEFINED: the upstream factor loop
 * and its summation order live in srrg2_solver and are not pinned by anything in the reference tree; the shape is the
 * "LDS tree-reduced" one the hot path is specified with, chosen so that gfx950 evaluates it with register exchanges):
 *   leaf l (0..127)  = ((+0 + t_l) + t_{l+128}) + t_{l+256} + ...   terms of the correspondences l, l + 128, ... in order
 *   seven levels      v[l] <- v[l] + v[l ^ m]  for m = 32, 16, 8, 7, 2, 1, 64 (a balanced binary tree over the 128 leaves)
 *   sum               = v[0] + 0.0f
 * What is summed (round 4) are the CAMERA-FRAME normal equations: with [R | t] the transform a point goes through and
 * D = d(image point) / d(point in camera), J = D R [ wt I | -2 [p]x ] = D G_c Rt with G_c = [ wt I | -[y]x ], y = 2 R p and
 * Rt = blockdiag(R, R) -- the same for every correspondence --, so the terms are those of G_c^T (D^T Omega D) G_c and
 * G_c^T D^T Omega e, and H = Rt^T (sum) Rt, b = Rt^T (sum) is evaluated ONCE per linearisation on the summed system (row by
 * row; the lower triangle of the rotated matrix is the system and is mirrored).  Operation order: csrc/align.hip
 * factor_accumulate, csrc/prs_se3.h rotate_normal_equations; restated in oracle/proslam_oracle.c.
 * Same inputs give the same bits on every launch, batch size and entry point (fused, split, prs_pcf_linearize). */
typedef struct {
  float H[36];          /* last linearisation, row-major, without prior */
  float b[6];           /* sum J^T Omega e */
  float chi_inliers;
  float chi_total;
  float mean_disparity; /* value used by the factor */
  int32_t num_inliers;
  int32_t num_outliers;
  int32_t num_invalid;
  int32_t num_correspondences;
  int32_t status;       /* 1 Success, 0 Fail (tests/test_aligners.cpp:117-121) */
  int32_t iterations;   /* aligner iterations accounted for (= max_iterations in align mode) */
  int32_t iterations_executed; /* < iterations when stop_at_fixed_point cut the loop */
  int32_t warnings;     /* OR of PRS_WARN_* over the call, or a PRS_ERR_* code */
} prs_align_result;

enum {
  PRS_MODE_ALIGN     = 0, /* per iteration: finder.compute(); setupFactor; linearize; GN step */
  PRS_MODE_FINDER    = 1, /* ONE CorrespondenceFinderProjective::compute() with local_map_in_sensor = X */
  PRS_MODE_LINEARIZE = 2  /* ONE linearisation at X of the correspondences passed in */
};

/* device-resident batch of B independent frames (one per sequence).
 * fixed:  [batch][fixed_stride][4] floats: (u,v,-,-) mono, (u,v,d,-) depth, (uL,vL,uR,vR) stereo
 * moving: [batch][moving_stride][4] floats: (x,y,z, information scale of the point:
 *         1 + log(numberOfOptimizations) if > 2 else 1, aligner_slice_processor_projective.cpp:46-52) */
typedef struct {
  int32_t batch;
  int32_t fixed_stride;
  int32_t moving_stride;
  const float* fixed;
  const uint8_t* fixed_desc;     /* [batch][fixed_stride][32] */
  const int32_t* n_fixed;        /* [batch] */
  const float* moving;
  const uint8_t* moving_desc;    /* [batch][moving_stride][32] */
  const int32_t* n_moving;       /* [batch] */
  const uint8_t* inputs_changed; /* [batch] setFixed/setMoving since the last call; NULL = all changed */
  prs_pcf_state* state;          /* [batch] in/out */
  float* X;                      /* [batch][16] in: movingInFixed guess, out: estimate (row-major 4x4) */
  prs_corr* corr;                /* [batch][fixed_stride] in/out: the caller-owned CorrespondenceVector */
  int32_t* n_corr;               /* [batch] in/out */
  prs_align_result* result;      /* [batch] */
  const float* prior;            /* optional [batch][42]: additive H0 (36) and b0 (6) (an externally linearised slice) */
  const float* prior_mean;       /* optional [batch][16]: mean Z of the motion prior (NULL = identity) */
  int32_t max_fixed;             /* 0 = fixed_stride; else an upper bound on n_fixed[] the kernel sizes its LDS
                                    for (fewer bytes per frame = more frames per CU); a frame exceeding it
                                    gets PRS_ERR_CAPACITY in result[].warnings */
} prs_align_batch;

/* mode PRS_MODE_FINDER and PRS_MODE_LINEARIZE only enqueue.  mode PRS_MODE_ALIGN alternates a search launch and a Gauss-Newton
 * launch over the frames that are still pending (4-5 rounds at kitti.conf settings) and BLOCKS until the batch is done
 * (= prs_align_batch_enqueue + prs_align_batch_finish); results are complete in device memory when it returns. */
PRS_API int prs_align_batch_run(prs_context* ctx,
                                const prs_pcf_params* finder,
                                const prs_aligner_params* aligner,
                                const prs_align_batch* batch,
                                int32_t mode);
/* The two halves of mode PRS_MODE_ALIGN, for callers that pipeline (several contexts from one thread, or other work between
 * the two calls).  enqueue: `rounds` (0 = the nominal 5) x (search launch, Gauss-Newton launch) on the context's stream and
 * nothing else -- no host synchronisation, no readback; every launch skips the frames that are finished or not waiting for
 * it.  Once the context's scratch buffers exist (after a first batch of the same shape) the sequence allocates nothing and
 * can be captured in a HIP graph.  finish: one 4-byte readback; while frames are still pending (finder retries shift the
 * nominal schedule) four more rounds and another readback.  Between the two calls the context may run any OTHER operator
 * (matcher, scene clipper, extractor, brute-force matcher, merger: the enqueued batch owns its working buffers), but must not
 * start another aligner batch nor change its stream (PRS_ERR_UNSUPPORTED), and `batch`'s buffers must stay valid; the structs
 * themselves are copied.  finish without an enqueued batch is a no-op.
 * rearm: a HIP graph captured around enqueue replays the launches without passing through the host code; call rearm after
 * each graph launch so that finish performs its completion check (and its extra rounds) for the replayed batch.  The graph
 * bakes in the batch shape: a later enqueue with a larger batch may move the working buffers -- capture again after it. */
PRS_API int prs_align_batch_enqueue(prs_context* ctx,
                                    const prs_pcf_params* finder,
                                    const prs_aligner_params* aligner,
                                    const prs_align_batch* batch,
                                    int32_t rounds);
PRS_API int prs_align_batch_finish(prs_context* ctx);
PRS_API int prs_align_batch_rearm(prs_context* ctx);
/* rearm for a graph that is replayed on ANOTHER stream than the one the batch was enqueued / captured on: finish then synchronises
 * `hip_stream` (a hipStream_t) and enqueues its extra rounds there.  prs_align_batch_rearm assumes the capture stream; replaying
 * elsewhere without telling the library would let finish read the completion word before the replay has run.  Either way the
 * number of rounds the graph holds is the `rounds` the captured enqueue was called with (remembered by the context). */
PRS_API int prs_align_batch_rearm_on(prs_context* ctx, void* hip_stream);

/* ---- host, one frame: stateful finder handle mirroring the reference object -------------------
 * setFixed / setMoving / setLocalMapInSensor / compute (tests/test_correspondence_finders.cpp:314,
 * tests/test_aligners.cpp:658-666) */
typedef struct prs_pcf prs_pcf;
PRS_API int prs_pcf_create(prs_context* ctx, const prs_pcf_params* params, prs_pcf** out);
PRS_API int prs_pcf_destroy(prs_pcf* h);
PRS_API int prs_pcf_set_params(prs_pcf* h, const prs_pcf_params* params); /* flags a config change */
/* coords: [n][fixed_dim] floats */
PRS_API int prs_pcf_set_fixed(prs_pcf* h, const float* coords, int32_t fixed_dim, const uint8_t* desc, int32_t n);
/* xyz [n][3]; info_scale [n] or NULL (= 1) */
PRS_API int prs_pcf_set_moving(prs_pcf* h, const float* xyz, const float* info_scale, const uint8_t* desc, int32_t n);
PRS_API int prs_pcf_set_local_map_in_sensor(prs_pcf* h, const float* T16);
PRS_API int prs_pcf_set_search_radius(prs_pcf* h, uint64_t radius_pixels);      /* CF/..projective_base.h:82-85 */
PRS_API int prs_pcf_set_descriptor_distance(prs_pcf* h, float distance);        /* CF/..projective_base.h:94-97 */
PRS_API int prs_pcf_get_state(prs_pcf* h, prs_pcf_state* out);
/* mean Z (row-major 4x4) of the motion prior prs_pcf_align applies when prs_aligner_params.enable_motion_prior is set;
 * NULL = identity */
PRS_API int prs_pcf_set_motion_prior_mean(prs_pcf* h, const float* Z16);
/* out capacity >= n_fixed; untouched calls ("nothing new", converged) return the previous vector */
PRS_API int prs_pcf_compute(prs_pcf* h, prs_corr* out, int32_t capacity, int32_t* n_out);
/* the full per-frame loop on the handle's fixed/moving clouds (MultiAligner3DQR::compute stand-in) */
PRS_API int prs_pcf_align(prs_pcf* h,
                          const prs_aligner_params* aligner,
                          const float* X_init16,
                          const float* prior42, /* optional */
                          float* X_out16,
                          prs_corr* corr_out,
                          int32_t capacity,
                          int32_t* n_corr_out,
                          prs_align_result* result);
/* one linearisation of given correspondences on the handle's clouds (factor-level use,
 * tests/test_aligners.cpp:586-638) */
PRS_API int prs_pcf_linearize(prs_pcf* h,
                              const prs_aligner_params* aligner,
                              const float* X16,
                              const prs_corr* corr,
                              int32_t n_corr,
                              prs_align_result* result);

/* (H + damping diag(H)) dx = -b, X <- X * exp(dx) on the device (same arithmetic as the aligner loops: LDL^T without square roots,
 * csrc/prs_se3.h ldlt_solve6; H row-major, its LOWER triangle is read; a pivot that is not positive leaves X untouched) */
PRS_API int prs_gn_step(prs_context* ctx, const float* H36, const float* b6, float damping, float* X16);
/* the same with the damping form chosen (prs_aligner_params.damping_form: PRS_DAMPING_DIAG or PRS_DAMPING_IDENTITY) */
PRS_API int prs_gn_step_ex(prs_context* ctx, const float* H36, const float* b6, float damping, int32_t damping_form, float* X16);

/* Self-test of the reciprocal the aligner kernels use (csrc/prs_device.h, recip_exact: v_rcp_f32 + one Newton step in fused
 * multiply-adds where that equals the IEEE quotient, the compiler's division elsewhere): every one of the 2^32 float bit patterns
 * through the function as shipped, compared with 1.0f / x of the same device (NaNs compare equal).  counts[0] = operands that differ
 * (must be 0), counts[1] = operands that went through the short form (waves that held only operands of 2^-126 <= |x| < 2^126).
 * About a second on an MI355X; not part of the tracking path. */
PRS_API int prs_selftest_reciprocal(prs_context* ctx, uint64_t counts[2]);

/* host helper: information scale column from landmark ages
 * (aligner_slice_processor_projective.cpp:46-52: n > 2 ? 1 + log(n) : 1) */
PRS_API void prs_info_scale_from_nopt(const uint32_t* n_opt, int32_t n, float* scale);

/* ================================================================================================
 * Scene clipper
 * replaces SceneClipperProjective3D::compute (mapping/scene_clipper_projective_3d.cpp:9-67):
 * projector->setCameraPose(robot_in_local_map * sensor_in_robot) (:46), the projector keeps the
 * local-map points inside [range_min, range_max] and the canvas and returns them in the camera
 * frame together with their indices into the full scene (:53, globalIndices()); when
 * sensor_in_robot is not exactly the identity the kept points are moved to the robot frame
 * (:61-63).  Survivors keep ascending source order.  The clipped cloud has the layout of the
 * aligner's `moving` arrays, so it can be consumed in place.
 * Status per scene: PRS_WARN_EMPTY_INPUT for an empty full scene (outputs AND n_clipped left
 * untouched, like the reference :21-28), PRS_WARN_NO_PROJECTION when nothing survives (:55-58).
 * ============================================================================================== */
typedef struct {
  int32_t batch;                   /* independent scenes (one per sequence) */
  int32_t stride;                  /* row stride (points) of every per-scene array */
  const float* scene_xyzw;         /* [batch][stride][4] local-map points; w is carried through (information scale) */
  const uint8_t* scene_desc;       /* [batch][stride][32] or NULL (then clipped_desc must be NULL too) */
  const int32_t* n_scene;          /* [batch] */
  const float* robot_in_local_map; /* [batch][16] row-major */
  float* clipped_xyzw;             /* out [batch][stride][4] */
  uint8_t* clipped_desc;           /* out [batch][stride][32] or NULL */
  int32_t* global_indices;         /* out [batch][stride]: clipped index -> full-scene index */
  int32_t* n_clipped;              /* out [batch] */
  int32_t* status;                 /* out [batch] */
  const uint32_t* scene_n_opt;     /* optional [batch][stride] numberOfOptimizations of the scene points: the clipped
                                      w column then is the aligner's information scale n > 2 ? 1 + log(n) : 1
                                      (aligner_slice_processor_projective.cpp:46-52), taken from a table the host
                                      evaluates with the same double log (n clamped to 4095) */
} prs_clip_batch;

/* device pointers, asynchronous on the context's stream */
PRS_API int prs_scene_clip_batch(prs_context* ctx,
                                 const prs_projector* projector,
                                 const float* sensor_in_robot16, /* host */
                                 const prs_clip_batch* batch);

/* host pointers, one scene; capacity = rows available in the three output arrays (>= n) */
PRS_API int prs_scene_clip(prs_context* ctx,
                           const prs_projector* projector,
                           const float* robot_in_local_map16,
                           const float* sensor_in_robot16,
                           const float* scene_xyzw,
                           const uint8_t* scene_desc, /* or NULL */
                           int32_t n,
                           float* clipped_xyzw,
                           uint8_t* clipped_desc, /* or NULL */
                           int32_t* global_indices,
                           int32_t capacity,
                           int32_t* n_clipped);

/* ================================================================================================
 * Bijective brute-force descriptor matcher
 * replaces CorrespondenceFinderDescriptorBasedBruteforce::compute
 * (CF/correspondence_finder_descriptor_based_bruteforce_impl.cpp:8-155,157-199,247-293): all
 * N_f x N_m Hamming distances, candidates below maximum_descriptor_distance, registration pool by
 * pool (one pool per distinct distance) with the in-pool uniqueness test and Lowe's ratio on the
 * fixed AND the moving side.  Output order: ascending (response, fixed index) -- the reference's
 * std::sort by response alone (:94-97) leaves ties unspecified.
 * Status per pair of clouds: PRS_WARN_EMPTY_INPUT (:217-226), PRS_WARN_NO_MATCHES (:237-242),
 * PRS_ERR_CAPACITY when more candidates pass the threshold than candidate_capacity.
 * ============================================================================================== */
typedef struct {
  float maximum_descriptor_distance;           /* CF/..bruteforce.h:22-26 (default 50); must be <= 256 */
  float maximum_distance_ratio_to_second_best; /* CF/..bruteforce.h:27-31 (default 0.9) */
  float minimum_matching_ratio;                /* CF/..bruteforce.h:32-36; not read by compute() */
} prs_bruteforce_params;

typedef struct {
  int32_t batch;               /* independent (fixed, moving) cloud pairs */
  int32_t fixed_stride;        /* <= 8192 */
  int32_t moving_stride;       /* <= 65535 */
  const uint8_t* fixed_desc;   /* [batch][fixed_stride][32] */
  const int32_t* n_fixed;      /* [batch] */
  const uint8_t* moving_desc;  /* [batch][moving_stride][32] */
  const int32_t* n_moving;     /* [batch] */
  prs_corr* matches;           /* out [batch][min(fixed_stride, moving_stride)] */
  int32_t* n_matches;          /* out [batch] */
  int32_t* status;             /* out [batch] */
  int32_t candidate_capacity;  /* 0 = 16 * max(stride): pairs below the threshold the kernel can hold per cloud pair */
} prs_bruteforce_batch;

/* device pointers, asynchronous on the context's stream */
PRS_API int prs_bruteforce_match_batch(prs_context* ctx,
                                       const prs_bruteforce_params* params,
                                       const prs_bruteforce_batch* batch);

/* host pointers, one pair of clouds; capacity >= min(n_fixed, n_moving) */
PRS_API int prs_bruteforce_match(prs_context* ctx,
                                 const prs_bruteforce_params* params,
                                 const uint8_t* fixed_desc,
                                 int32_t n_fixed,
                                 const uint8_t* moving_desc,
                                 int32_t n_moving,
                                 prs_corr* correspondences,
                                 int32_t capacity,
                                 int32_t* n_correspondences);

/* ================================================================================================
 * Landmark estimators + projective mergers
 * replaces MergerProjective_::compute with its RigidStereoTriangulation / RigidStereoProjectiveEKF /
 * ProjectiveDepthEKF specialisations (mapping/mergers/merger_projective_impl.cpp:8-328,
 * merger_projective_rigid_stereo_impl.cpp:8-77, .._triangulation_impl.cpp:7-39,
 * merger_projective_depth_ekf_impl.cpp:8-73) and the estimator each one drives:
 * LandmarkEstimatorWeightedMean_ (mapping/landmarks/landmark_estimator_weighted_mean_impl.cpp:7-41),
 * LandmarkEstimatorEKF_ + PointEKFBase + the stereo / depth / mono measurement models in double
 * (landmark_estimator_ekf_impl.cpp:7-82, filters/point_ekf_base.hpp:63-125,
 * filters/stereo_projective_point_ekf_impl.cpp:13-48, projective_depth_point_ekf_impl.cpp:7-36,
 * projective_point_ekf_impl.cpp:16-43), LandmarkEstimatorPoseBasedSmoother_
 * (landmark_estimator_pose_based_smoother_impl.cpp:7-148).
 * The local map lives on the device as a structure of arrays; one workgroup merges one frame into
 * one map: first-come bin blocking by correspondence order (:89-122), one estimator update per
 * surviving correspondence, then the binned addition of new landmarks (:210-305) in the
 * reference's order (bins in order of their first measurement, best disparity / depth per bin).
 * Every scene index may appear in at most one correspondence (the finder's output is bijective).
 * ============================================================================================== */
enum { PRS_EST_WEIGHTED_MEAN = 0, PRS_EST_EKF = 1, PRS_EST_SMOOTHER = 2 };
enum { PRS_MERGER_STEREO_TRIANGULATION = 0, PRS_MERGER_STEREO_EKF = 1, PRS_MERGER_DEPTH_EKF = 2 };
enum {
  PRS_ERR_HISTORY    = -7, /* a landmark's measurement history is full (max_measurements) */
  PRS_ERR_SCENE_FULL = -8, /* the map has no room for the landmarks to add */
  PRS_ERR_DUPLICATE  = -9  /* a scene index appears in two correspondences */
};

/* PointStatisticsField3D::CameraMeasurement; the two transforms of a measurement are per frame and
 * sit in the map's pose table */
typedef struct {
  float point_in_image[3];
  float point_in_camera[3];
  int32_t frame;
} prs_camera_measurement;

typedef struct {
  float sensor_in_world[12]; /* 3x4 row-major */
  float world_in_sensor[12];
} prs_frame_pose;

typedef struct {
  int32_t type;            /* PRS_EST_* */
  int32_t measurement_dim; /* 2 mono (estimator only), 3 depth, 4 stereo */
  float maximum_distance_geometry_meters_squared; /* landmark_estimator_base.hpp:21-25 */
  double minimum_state_element_covariance;        /* landmark_estimator_ekf.h:37-41 */
  double maximum_covariance_norm_squared;         /* :43-47 */
  double fx, fy, cx, cy, b_x, b_y;                /* filter calibration (setCameraMatrix / setBaseline) */
  uint32_t maximum_number_of_iterations;          /* landmark_estimator_pose_based_smoother.h:17-21 */
  float convergence_criterion_minimum_chi2_delta; /* :23-27 */
  float maximum_reprojection_error_pixels_squared; /* :29-33 */
  uint32_t minimum_number_of_measurements_for_optimization; /* :35-39 */
  float camera_matrix[9];                         /* smoother: setCameraMatrix */
} prs_estimator_params;

typedef struct {
  int32_t variant;        /* PRS_MERGER_* */
  int32_t enable_binning; /* MergerCorrespondence_::param_enable_binning */
  uint32_t number_of_row_bins, number_of_col_bins; /* merger_projective.h:47-56 */
  int32_t canvas_rows, canvas_cols;                /* projector canvas (:30-33) */
  float maximum_distance_appearance;               /* merger_projective.h:42-46 */
  uint32_t target_number_of_merges;                /* MergerCorrespondence_::param_target_number_of_merges */
  float target_merge_ratio;                        /* merger_projective.h:57-61 (warning only) */
  prs_triangulator_params triangulator;            /* stereo variants */
  float fx, fy, cx, cy;                            /* depth variant: unprojector */
  prs_estimator_params estimator;
} prs_merger_params;

typedef struct {
  int32_t n_merged;
  int32_t n_added;
  int32_t status; /* PRS_WARN_NO_MATCHES = all merge attempts failed (:141-144), PRS_WARN_LOW_RATIO = low merge
                     ratio (:145-150), or a PRS_ERR_* code */
} prs_merge_result;

/* B local maps + the frames merged into them; device pointers */
typedef struct {
  int32_t batch;
  int32_t capacity;           /* landmarks per map (row stride of the per-landmark arrays) */
  int32_t max_measurements;   /* history slots per landmark (0 = none kept: not with the smoother) */
  int32_t max_frames;         /* rows of the pose table per map */
  /* the map */
  float* coords;              /* [batch][capacity][4] xyz in the local map frame */
  uint8_t* desc;              /* [batch][capacity][32] */
  float* state;               /* [batch][capacity][4] statistics().state(), world frame */
  float* covariance;          /* [batch][capacity][9] */
  uint32_t* n_opt;            /* [batch][capacity] numberOfOptimizations */
  uint8_t* inlier;            /* [batch][capacity] */
  uint32_t* n_meas;           /* [batch][capacity] */
  prs_camera_measurement* meas; /* [batch][capacity][max_measurements] */
  prs_frame_pose* poses;      /* [batch][max_frames] */
  int32_t* n_points;          /* [batch] in/out */
  /* the frame */
  int32_t measurement_stride;
  const float* measurement;   /* [batch][measurement_stride][4] image-space points ((uL,vL,uR,vR) / (u,v,d,-)) */
  const uint8_t* measurement_desc; /* [batch][measurement_stride][32] */
  const int32_t* n_measured;  /* [batch] */
  int32_t corr_stride;
  const prs_corr* corr;       /* [batch][corr_stride]: fixed_idx -> scene, moving_idx -> measurement */
  const int32_t* n_corr;      /* [batch] */
  const int32_t* scene_index_map; /* optional [batch][capacity]: clipped index -> scene index (prs_clip_batch.global_indices) */
  const float* measurement_in_world; /* [batch][16] */
  const float* measurement_in_scene; /* [batch][16] */
  const int32_t* frame;       /* [batch] pose-table slot of this frame */
  prs_merge_result* result;   /* [batch] */
  int32_t corr_from_aligner;  /* 0: fixed_idx -> scene, moving_idx -> measurement (the merger's own convention,
                                 merger_projective_impl.cpp:60-77); 1: the vector comes straight from
                                 prs_align_batch.corr (fixed_idx -> measurement, moving_idx -> clipped scene) */
} prs_merge_batch;

PRS_API int prs_merge_batch_run(prs_context* ctx, const prs_merger_params* params, const prs_merge_batch* batch);

/* ---- host, one local map: stateful handle mirroring the reference's merger object ---------------------------------
 * setScene / setMeasurement / setCorrespondences / setMeasurementInScene / setMeasurementInWorld / compute
 * (mapping/mergers/merger_projective.h, tests/test_mergers.cpp:248-780).  The map lives on the device in the layout of
 * prs_merge_batch; every merge uploads the frame, runs the merge kernel(s) with batch = 1 and synchronises. */
typedef struct prs_map prs_map;
/* capacity: landmarks; max_measurements: history slots per landmark (0 = none; the pose-based smoother needs them);
 * max_frames: frames merged between two prs_map_clear calls (rows of the pose table); max_measured: measurements and
 * correspondences per frame */
PRS_API int prs_map_create(prs_context* ctx, int32_t capacity, int32_t max_measurements, int32_t max_frames, int32_t max_measured, prs_map** out);
PRS_API int prs_map_destroy(prs_map* h);
PRS_API int prs_map_clear(prs_map* h); /* a new local map: no landmarks, frame counter 0 */
/* grows the landmark arrays to `capacity` (no-op if already that large): every landmark, its state, covariance, counters and
 * measurement history are kept */
PRS_API int prs_map_reserve(prs_map* h, int32_t capacity);
PRS_API int prs_map_size(prs_map* h, int32_t* n_points, int32_t* frames_merged /* may be NULL */);
/* setScene with allocated statistics: coords_in_scene [n][3]; state_in_world [n][3] or NULL (= the coordinates,
 * tests/test_mergers.cpp:268-271); covariance [n][9] or NULL (zero: what statistics().allocate() leaves when no covariance is
 * set, tests/ref_mapping.py:_seed_map; points the merger creates get the identity, merger_projective_impl.cpp:319); desc [n][32]; n_opt [n] or NULL (0);
 * first_measurement [n] or NULL: the camera measurement the landmark was created from (its .frame names a pose-table row set
 * with prs_map_set_frame_pose; tests/test_mergers.cpp:425-433) */
PRS_API int prs_map_set_scene(prs_map* h,
                              const float* coords_in_scene,
                              const float* state_in_world,
                              const float* covariance,
                              const uint8_t* desc,
                              const uint32_t* n_opt,
                              const prs_camera_measurement* first_measurement,
                              int32_t n);
/* pose of an earlier frame the scene's measurements refer to; frames merged afterwards continue behind the highest row set */
PRS_API int prs_map_set_frame_pose(prs_map* h, int32_t frame, const float* sensor_in_world16);
/* MergerProjective_::compute for one frame.  measurement: [n_measured][estimator.measurement_dim] image-space points;
 * corr: fixed_idx -> scene, moving_idx -> measurement (corr_from_aligner = 0) or the aligner's vector (= 1, with
 * scene_index_map = the clipper's global indices, or NULL when the aligner ran on the whole scene).  Returns the warning
 * bits of result->status (>= 0) or a PRS_ERR_* code. */
PRS_API int prs_map_merge(prs_map* h,
                          const prs_merger_params* params,
                          const float* measurement_in_world16,
                          const float* measurement_in_scene16,
                          const float* measurement,
                          const uint8_t* measurement_desc,
                          int32_t n_measured,
                          const prs_corr* corr,
                          int32_t n_corr,
                          const int32_t* scene_index_map,
                          int32_t corr_from_aligner,
                          prs_merge_result* result);
/* the scene after merging; any output may be NULL: coords_in_scene / state_in_world [capacity][3], desc [capacity][32],
 * n_opt / inlier [capacity] */
PRS_API int prs_map_get_scene(prs_map* h,
                              int32_t capacity,
                              float* coords_in_scene,
                              float* state_in_world,
                              uint8_t* desc,
                              uint32_t* n_opt,
                              uint8_t* inlier,
                              int32_t* n_points);

/* pose bookkeeping between the aligner and the merger / next clip, on the device:
 * pose_out[b] = prediction[b] * X[b]^-1.  The clipper expresses the local map in the predicted sensor
 * frame (scene_clipper_projective_3d.cpp:46-53), so the aligner's estimate X (moving in fixed) is the
 * motion relative to the prediction; the tracker's new sensor pose in the map is prediction * X^-1.
 * All pointers are device arrays of [batch][16] row-major float; pose_out may alias prediction. */
PRS_API int prs_pose_compose_batch(prs_context* ctx, int32_t batch, const float* prediction, const float* X, float* pose_out);

/* MotionModelConstantVelocity3D (external; configurations/kitti.conf:257-260): the tracker's guess for the next pose
 * repeats the last inter-frame motion, pose_pred[b] = pose_prev1[b] * (pose_prev2[b]^-1 * pose_prev1[b]), with the
 * rotation block renormalised through its unit quaternion (the recursion amplifies rounding otherwise).
 * Device arrays of [batch][16] row-major float; pose_pred may alias pose_prev2. */
PRS_API int prs_motion_predict_batch(prs_context* ctx, int32_t batch, const float* pose_prev2, const float* pose_prev1, float* pose_pred);

/* ================================================================================================
 * Intensity feature extraction
 * replaces IntensityFeatureExtractorBinned_::compute (sensor_processing/feature_extractors/
 * intensity_feature_extractor_binned.cpp:7-208, intensity_feature_extractor_base.cpp:56-95): FAST
 * keypoints with non-maximum suppression, the detection-region grid with "keep all below the
 * per-region target, else the best by response" (:47-196, in-repo, restated exactly), then one
 * 256-bit binary descriptor per keypoint.
 * The two OpenCV calls of the reference (cv::FastFeatureDetector::detect, cv::ORB::compute) are NOT part of the
 * reference tree; they are restated from OpenCV's published algorithms:
 *  - FAST-9 segment test on the 16-pixel circle of radius 3, response = largest threshold that still detects
 *    (cornerScore), strict 8-neighbour non-maximum suppression, outermost 3 pixels not examined, raster order;
 *  - cv::ORB::compute on provided keypoints: keypoints closer than 31 px (edgeThreshold) to the border are removed, no
 *    orientation (FAST keypoints carry angle -1: the pattern is used unrotated), the image is smoothed with the
 *    8-bit fixed-point GaussianBlur(7x7, sigma 2) and bit i compares the smoothed pixels of pair i of ORB's learned
 *    pattern (bit_pattern_31_).
 * With selection_order = PRS_SELECT_LIBSTDCXX this reproduces every feature / match count the reference's own tests
 * assert on its own KITTI / ICL / SceneFlow images (tests/test_ref_pins_gpu.py).  Outputs have the layout of
 * prs_stereo_batch's inputs.
 * Status per image: PRS_WARN_NO_MATCHES (no keypoints, :126-131), PRS_ERR_CAPACITY (more than max_raw_detections
 * raw detections or more features than `stride`).
 * ============================================================================================== */
/* which of several EQUAL responses survive the per-region cut (intensity_feature_extractor_binned.cpp:179-195 uses
 * std::sort with a response-only comparator, so the answer is implementation defined):
 *   PRS_SELECT_CANONICAL  ties in detection (raster) order; one bitonic sort of all detections of the image
 *   PRS_SELECT_LIBSTDCXX  the permutation GNU libstdc++'s std::sort produces: its introsort replayed by the waves of a
 *                         workgroup (only the ranges that reach a region's kept prefix); results identical to a reference
 *                         built with GCC, and since round 5 the faster of the two (profiles/r05/features_kitti_images.txt) */
enum { PRS_SELECT_CANONICAL = 0, PRS_SELECT_LIBSTDCXX = 1 };

typedef struct {
  int32_t detector_threshold;             /* intensity_feature_extractor_base.h:36-40; in [1, 254] */
  int32_t enable_non_maximum_suppression; /* :48-52 */
  int32_t target_number_of_keypoints;     /* :54-58 */
  int32_t number_of_detectors_vertical;   /* intensity_feature_extractor_binned.h:17-22 */
  int32_t number_of_detectors_horizontal; /* :23-28 */
  int32_t selection_order;                /* PRS_SELECT_* */
  int32_t max_raw_detections;             /* FAST detections per image the selection can hold: 0 = 8192, at most 32768 */
} prs_extractor_params;

typedef struct {
  int32_t batch;
  int32_t rows, cols;     /* image size */
  int32_t pitch;          /* bytes between image rows (>= cols) */
  const uint8_t* images;  /* [batch][rows][pitch] 8-bit intensity */
  int32_t stride;         /* feature capacity per image = row stride of the outputs */
  prs_kp2* keypoints;     /* out [batch][stride] (u, v) = (column, row) */
  float* intensity;       /* out [batch][stride] or NULL */
  uint8_t* descriptors;   /* out [batch][stride][32] */
  int32_t* n_features;    /* out [batch] */
  int32_t* status;        /* out [batch] */
} prs_extract_batch;

PRS_API int prs_extract_features_batch(prs_context* ctx, const prs_extractor_params* params, const prs_extract_batch* batch);

/* The order in which the reference's selection leaves ONE region's keypoints (PRS_SELECT_LIBSTDCXX): the permutation GNU
 * libstdc++'s std::sort produces for the comparator `a.response > b.response`
 * (intensity_feature_extractor_binned.cpp:182-186) on keypoints whose responses are response[0..n), in detection order.
 * order[k] = index of the keypoint that ends up at position k.  Host pointers; runs the same device code as the
 * extractor's selection (work queue over the workgroup's waves, ranges of <= 64 items in registers, heapsort at the depth
 * limit), synchronises.  Responses are 1..255 (a detected corner never scores 0: PRS_ERR_RANGE); n <= 32768. */
PRS_API int prs_selection_order(prs_context* ctx, const uint8_t* response, int32_t n, int32_t* order);

/* host pointers, one image: what an adapter's IntensityFeatureExtractorBase_::compute(const cv::Mat&) binds
 * (intensity_feature_extractor_binned.cpp:47-196): uploads the image, runs the three kernels, downloads the features,
 * synchronises.  image: rows x cols 8-bit pixels, `pitch` bytes between rows; keypoints [capacity][2] (u, v) floats,
 * intensity [capacity] or NULL, descriptors [capacity][32]; *n_features = features written.  More features than
 * `capacity` or more raw detections than max_raw_detections: PRS_ERR_CAPACITY (nothing is returned). */
PRS_API int prs_extract_features(prs_context* ctx,
                                 const prs_extractor_params* params,
                                 const uint8_t* image,
                                 int32_t rows,
                                 int32_t cols,
                                 int32_t pitch,
                                 float* keypoints,
                                 float* intensity,
                                 uint8_t* descriptors,
                                 int32_t capacity,
                                 int32_t* n_features);

#ifdef __cplusplus
}
#endif
#endif
