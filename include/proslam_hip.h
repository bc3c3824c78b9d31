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
PRS_API const char* prs_last_error(const prs_context* ctx);
PRS_API const char* prs_status_string(int status);
PRS_API int prs_version(void);

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

#ifdef __cplusplus
}
#endif
#endif
