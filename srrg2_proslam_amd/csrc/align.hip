// align.hip -- projective correspondence finder + reprojection-error Gauss-Newton aligner (gfx950).
//
// The per-frame loop the external MultiAligner3DQR drives in the reference, up to max_iterations of
//     finder.setLocalMapInSensor(X); finder.compute(); slice.setupFactor(); linearize; GN step
// for `batch` independent frames (sequences) per launch.  Two forms, bit-identical:
//   * the split pipeline (default for PRS_MODE_ALIGN): align_kernel<512, true, pattern, slots> performs ONE projective
//     search for every frame that waits for it (three workgroups per CU) and commits the correspondence vector, gn_kernel gathers its
//     operand rows through that vector and runs Gauss-Newton iterations
//     until the finder needs the next search (two waves per frame, eight frames per CU); the host enqueues five
//     rounds (align_batch_launch) and confirms completion with one 4-byte readback (align_batch_finish);
//   * the fused kernel align_kernel<256, false, -1>: one 256-thread workgroup owns a frame for the whole loop
//     (finder-only mode, frames whose fixed cloud exceeds the split pipeline's bound, PRS_FUSED_ALIGN=1, phase stamps).
//
// Reference code replaced (CF/ = registration/correspondence_finders/):
//   CorrespondenceFinderProjectiveBase::compute        CF/correspondence_finder_projective_base_impl.cpp:105-293
//   _addCorrespondenceCandidate / _filterCorrespondences                                   ...:8-37, 41-102
//   Square/Circle/Rhombus _initializeDatabase, _findNearestNeighbors
//        CF/..square_impl.cpp:8-118, CF/..circle_impl.cpp:8-94, CF/..rhombus_impl.cpp:8-93
//   KDTree _initializeDatabase / _findNearestNeighbors (srrg2_core's tree restated: single-leaf radius queries)  CF/..kdtree_impl.cpp:8-80
//   AlignerSliceProcessorProjective{,Stereo}::setupFactor / bindFixed
//        registration/aligner_slice_processor_projective.cpp:28-112
//   + the un-vendored pinhole projector, error factors, saturated robustifier, H/b accumulation and
//     damped GN step, restated per SURVEY.md Appendix A.
//
// Determinism / parity: candidates are reduced with order-independent LDS atomicMin on
// (response, insertion order) keys, correspondences are emitted in ascending fixed index, and the
// 21+6+2 normal-equation sums are fixed-shape reductions (128 interleaved leaves, seven pairwise levels:
// include/proslam_hip.h at prs_align_result) that every entry point evaluates with the same code
// (factor_accumulate + wave_sum_slots), so correspondences AND poses are bit-identical to the CPU
// checker's evaluation of the same definition.
#include "prs_device.h"
#include "prs_host.h"
#include "prs_se3.h"

#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

namespace prs {

constexpr int kAlignThreads = 256;
constexpr int kTerms        = 29;  // 21 H (upper) + 6 b + chi_inliers + chi_total
constexpr uint32_t kNoneU32 = 0xffffffffu;
constexpr float kFltMax     = 3.402823466e+38f;

typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));

struct AlignShared {
  float X[16];
  float A[16];      // points -> camera: X, or sensor_in_robot^-1 * X (...WithSensor factors)
  float Sinv[16];   // sensor_in_robot^-1
  float T[16];      // finder's local_map_in_sensor
  float Tprev[16];  // _local_map_in_sensor_previous
  float W[16];      // points -> camera used by the projector
  float H[36];
  float b[6];
  float chi_in, chi_tot, mean_disp, change_norm, dd;
  float kd_range;   // KD-tree finder: the search radius _initializeDatabase last ran with (leaf range of the tree)
  int kd_nodes, kd_leaves, kd_open_next;  // KD-tree build counters
  unsigned long long radius, it;
  unsigned long long stamp_acc[9], stamp_mark, stamp_sub;  // PRS_STAMPS phase timers of thread 0 (in LDS: no registers when they are off)
  int converged, config_changed, num_recomputes;
  int n_corr, n_filtered, n_projected, decision, flags, error;
  int n_inl, n_out, n_inv;
  int corr_changed, have_terms;
  int n_overflow;   // queries of this search that met more survivors of the irrelevance bound than a thread can park
  int no_prune;     // the finder's hint (prs_pcf_state.reserved bit 0): this sequence's descriptor rows are correlated, do not prune
  int have_cls;     // the last executed iteration linearised (cls[] holds its factor classes)
  int wave_tot[16];
};

static_assert(offsetof(AlignShared, b) == offsetof(AlignShared, H) + 36 * sizeof(float) && offsetof(AlignShared, chi_in) == offsetof(AlignShared, b) + 6 * sizeof(float) &&
                offsetof(AlignShared, chi_tot) == offsetof(AlignShared, chi_in) + sizeof(float),
              "the summed slots are written through AlignShared::H");

// split pipeline (search kernel + GN kernel alternate until every frame is done): per-frame control word
struct FrameCtl {
  int it_align;     // aligner iterations whose GN step has been applied
  int need_search;  // 1: the finder.compute() of iteration it_align needs a projective search (search kernel's turn)
  int done;
  int executed;
  int flags;        // OR of finder warnings over the frame loop
  int n_inl, n_out, n_inv;
  int db_ready;     // the lattice of this frame's fixed cloud is in AlignArgs::dbcache (built by an earlier search launch)
};
constexpr int kModeSplitSearch = 100;  // internal: align_kernel<512> performing ONE finder.compute() for frames that wait for it
constexpr int kSearchThreads   = 512;

struct AlignArgs {
  prs_pcf_params f;
  prs_aligner_params a;
  prs_align_batch b;
  int mode;
  int no_prefilter;  // diagnostic (PRS_NO_PREFILTER=1)
  int surv_slots;  // LDS slots per thread for survivors of the irrelevance bound (align_batch_launch: 8 where that costs no resident workgroup, else 4)
  int narrow_prefilter_limit;  // largest irrelevance bound tested on 96 instead of 128 bits (32; PRS_PREFILTER_96_LIMIT overrides: diagnostic)
  int rows_table;  // R = projector canvas rows (lattice row table extent)
  int lut_cap;     // entries of the circle width table
  int cell_sy, cell_sx, cell_ncx, cell_ncy, ncells;  // 2-D cell grid over the canvas (cells of 2^sy rows x 2^sx cols)
  float4* ops;     // split pipeline: [batch][max_fixed][2] operand rows (fixed measurement, moving point) of the correspondences BEYOND the ones
                   // the Gauss-Newton kernel parks in LDS: gathered by that kernel once per launch, read back at every iteration
  FrameCtl* ctl;   // split pipeline: [batch]
  int* pending;    // split pipeline: number of frames the last GN launch left unfinished
  unsigned char* dbcache;  // split pipeline: [batch][db_blob] image of the LDS lattice (db | inv | cellstart)
  uint32_t db_blob;        // bytes of that image (multiple of 16)
  unsigned long long* stamps;  // diagnostic: [batch][16] accumulated shader clocks per phase (NULL = off)
  int max_fixed;   // LDS capacity in fixed points (frames with more are rejected loudly)
  uint32_t off_db, off_inv, off_cellstart, off_cfix, off_cmov, off_cls, off_sh;  // persistent for the whole frame loop
  const float* prior_mean;  // optional [batch][16]: mean of the motion prior (NULL = identity)
  uint32_t off_u;                                              // phase-exclusive union region:
  uint32_t off_fdesc, off_fuv, off_best, off_second, off_lut, off_surv;  //   search phase (relative to the LDS base)
  uint32_t off_terms;                                          //   GN phase / database build / disparity column
};

enum { kDecisionCommit = 0, kDecisionRetry = 1, kDecisionReturn = 2 };



// floor(sqrt(n)) for 0 <= n < 2^31, exact
__device__ __forceinline__ int isqrt_exact(int n) {
  int s = (int) sqrtf((float) n);
  while (s * s > n) {
    --s;
  }
  while ((s + 1) * (s + 1) <= n) {
    ++s;
  }
  return s;
}

__device__ __forceinline__ int hamming_regs(const au32x4& a0, const au32x4& a1, const au32x4& b0, const au32x4& b1) {
  uint32_t d = (uint32_t) __popc(a0.x ^ b0.x);
  d = popc_acc(a0.y ^ b0.y, d);
  d = popc_acc(a0.z ^ b0.z, d);
  d = popc_acc(a0.w ^ b0.w, d);
  d = popc_acc(a1.x ^ b1.x, d);
  d = popc_acc(a1.y ^ b1.y, d);
  d = popc_acc(a1.z ^ b1.z, d);
  d = popc_acc(a1.w ^ b1.w, d);
  return (int) d;
}

// An opaque copy of a thread index for code that runs once per launch (or never): addresses and masks derived from it cannot be
// hoisted to the top of the kernel, where they would sit in registers -- spilled ones -- across the whole search loop.
__device__ __forceinline__ int cold_copy(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ int hamming_half(const au32x4& a0, const au32x4& b0) {
  uint32_t d = (uint32_t) __popc(a0.x ^ b0.x);
  d = popc_acc(a0.y ^ b0.y, d);
  d = popc_acc(a0.z ^ b0.z, d);
  d = popc_acc(a0.w ^ b0.w, d);
  return (int) d;
}
// the first 96 bits only: what the lattice scan compares with the irrelevance bound.  ANY partial distance that reaches the bound
// proves the candidate irrelevant; 96 bits of unrelated rows differ in 48 +- 5, so a bound of <= 36 still rejects all but ~1 %
// of them at three quarters of the instructions
__device__ __forceinline__ int hamming_96(const au32x4& a0, const au32x4& b0) {
  uint32_t d = (uint32_t) __popc(a0.x ^ b0.x);
  d = popc_acc(a0.y ^ b0.y, d);
  d = popc_acc(a0.z ^ b0.z, d);
  return (int) d;
}

// _filterCorrespondences accepts a fixed point's best candidate when response < dd and response / second-lowest < ratio
// (correspondence_finder_projective_base_impl.cpp:57-99).  A candidate whose descriptor distance is >= B, B the smallest
// integer with B >= dd and (largest acceptable response) / B < ratio (evaluated in float like the filter does), cannot win
// (its response is not below dd) and, as a second-lowest response, cannot make the ratio test fail: whether such a candidate is
// handed to the filter at all, and with which response, does not change the outcome.  The search may therefore drop every
// candidate as soon as a PARTIAL distance reaches B.  Returns 0 when nothing can be dropped this way.
__device__ __forceinline__ int irrelevant_distance(const float dd, const float ratio) {
  if (!(dd > 0.0f) || !(ratio > 0.0f) || !(dd <= 256.0f)) {
    return 0;
  }
  const int r_max = (int) ceilf(dd) - 1;  // responses are integers: the largest one below dd
  if (r_max <= 0) {
    return 1;
  }
  float guess = (float) r_max / ratio;
  int b       = guess < 1024.0f ? (int) guess : 1024;
  b           = b > r_max ? b : r_max + 1;
  while (b < 2048 && !((float) r_max / (float) b < ratio)) {
    ++b;
  }
  return b < 2048 ? b : 0;
}

// the pose the finder and the factors see: X, or sensor_in_robot^-1 * X for the ...WithSensor variants
// (registration/aligner_slice_processor_projective.h:80-83,88-91); the perturbation stays on X
__device__ __forceinline__ void pose_to_camera(const prs_aligner_params& a, const float* Sinv, const float* X, float* A) {
  if (a.with_sensor) {
    se3_mul(Sinv, X, A);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      A[i] = X[i];
    }
  }
}

// AlignerSliceMotionModel3D stand-in: prior factor e = t2tnq(Z^-1 X), J = I: H += diag(info), b += info * e
// (Z = prior_mean, NULL = identity); re-evaluated at every iteration
__device__ __forceinline__ void add_motion_prior(const prs_aligner_params& a, const float* Z, const float* X, float* H, float* b) {
  float e[6];
  if (Z) {
    float z[16], Zi[16], D[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      z[i] = Z[i];
    }
    se3_inverse(z, Zi);
    se3_mul(Zi, X, D);
    t2tnq(D, e);
  } else {
    t2tnq(X, e);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    H[7 * i] += a.motion_prior_info[i];
    b[i] += a.motion_prior_info[i] * e[i];
  }
}

// iterations of the inlier-only run that follows the max_iterations loop (prs_aligner_params.enable_inlier_only_runs)
__host__ __device__ __forceinline__ int inlier_run_length(const prs_aligner_params& a) {
  return a.enable_inlier_only_runs ? (a.inlier_only_iterations > 0 ? a.inlier_only_iterations : a.max_iterations) : 0;
}

// One correspondence of SE3{,Depth,RectifiedStereo}ProjectiveErrorFactor::errorAndJacobian + saturated
// robustifier: the 21 upper-triangle entries of J^T Omega J, the 6 of J^T Omega e, chi (inliers only) and
// chi (kernelised).  z = fixed measurement, p = moving point (w = information scale).
// cls: 0 inlier, 1 outlier (kernelised), 2 invalid (behind the camera / outside the image: all terms +0).
struct PoseRegs {
  float R00, R01, R02, t0, R10, R11, R12, t1, R20, R21, R22, t2;
};
// factor_accumulate runs its arithmetic on every lane; a pose with a NaN / inf entry (every correspondence is then
// invalid: no term, no update) is kept away from it by its callers
__device__ __forceinline__ bool pose_is_finite(const PoseRegs& X) {
  const float s = ((((X.R00 + X.R01) + (X.R02 + X.t0)) + ((X.R10 + X.R11) + (X.R12 + X.t1))) + ((X.R20 + X.R21) + (X.R22 + X.t2)));
  return (s - s) == 0.0f;
}
// DIM: the factor type as a compile-time constant (0 = read it from the parameters)
// inverse-depth weight of a stereo measurement (aligner_slice_processor_projective.cpp:107-112): a function of the measurement and
// the frame's mean disparity only, so callers that iterate over fixed correspondences evaluate it once (PRE_WT: z.w carries it)
// "(0.01 + d, 1) * I": wt = min(0.01 + d / mean disparity, 1) (round 4: tools/sweep_a13.py); a weight that is
// NaN (0 / 0), +-inf or so negative that its square is not finite (a negative disparity over a zero / denormal mean) counts as 1:
// wt and wt^2 are always finite, so the zero K of an invalid correspondence cannot turn them into a NaN in the sums of the whole
// frame.  clamp_form: prs_aligner_params.translation_weight_form, clamp(d / mean disparity, 0.01, 1), NaN -> 0.01.
constexpr float kWeightFloor = -1.0e19f;  // (-1e19)^2 = 1e38 < FLT_MAX
__device__ __forceinline__ float inverse_depth_weight(const float4 z, const float mean_dsp, const int clamp_form) {
  const float dn = (z.x - z.z) / mean_dsp;
  if (clamp_form) {
    return !(dn >= 0.01f) ? 0.01f : (dn > 1.0f ? 1.0f : dn);
  }
  const float wt = 0.01f + dn;
  return wt < 1.0f ? (wt >= kWeightFloor ? wt : 1.0f) : 1.0f;
}
// what the Gauss-Newton kernel parks in the fourth component of a measurement (no factor reads it): the translation weight the stereo
// factor applies, 1 when the weighting is off
__device__ __forceinline__ float parked_translation_weight(const prs_aligner_params& a, const float4 z, const float mean_dsp, const int form) {
  return a.enable_inverse_depth_weighting ? inverse_depth_weight(z, mean_dsp, form) : 1.0f;
}
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kPairs = 16;  // the 29 sums + 3 class counts of one linearisation travel as 16 float pairs (32 "slots")
// slot (= 2 * pair + component) -> meaning: the upper triangle of the CAMERA-FRAME normal equations row by row (factor_accumulate),
// rows starting on even slots; the three slots in between are free, the first of them carries the class counts into the reduction:
// #inliers + kClsOutUnit * #kernelised as one float (an exact integer below 2^24, so its sum does not depend on the order).
//    0..5   H00 H01 H02 H03 H04 H05      6  class counts  7..11  H11 H12 H13 H14 H15
//   12..15  H22 H23 H24 H25             16  -            17..19  H33 H34 H35
//   20..21  H44 H45                     22  -            23      H55
//   24..29  b0..b5                      30  chi (inliers only)   31  chi (all, kernelised ones saturated)
constexpr float kClsOutUnit = 2048.0f;  // > the largest number of correspondences of a frame (align_batch_launch refuses max_fixed >= 2048)
constexpr int kSlotCls = 6, kSlotUnusedA = 16, kSlotUnusedB = 22;

// One correspondence of SE3{,Depth,RectifiedStereo}ProjectiveErrorFactor::errorAndJacobian + saturated robustifier,
// ACCUMULATED into the 16 pairs above (the running sums of this lane's leaf of the fixed-shape sum).
// cls: 0 inlier, 1 kernelised, 2 invalid (behind the camera / outside the image), 3 inactive.
// CAMERA-FRAME sums (round 4): with [R | t] the transform the point goes through and D = d(image point) / d(point in camera),
//   J = D R [ wt I | -2 [p]x ] = D G_c Rt,   G_c = [ wt I | -[y]x ],  y = 2 R p,  Rt = blockdiag(R, R)      (R [a]x = [R a]x R)
// so J^T Omega J = Rt^T (G_c^T K G_c) Rt with K = D^T Omega D and J^T Omega e = Rt^T G_c^T (D^T Omega e): what is summed over the
// correspondences is the bracket (K has five distinct non-zero entries, no product with R), and Rt, the same for all of them, is
// applied ONCE to the summed system (prs_se3.h, rotate_normal_equations).  ~85 instead of ~160 issue slots per correspondence.
//   D rows: (alpha, 0, beta0), (0, gamma, beta1), (d20, 0, d22): stereo (alpha, beta2), depth (0, 1), mono (0, 0)
//   K00 = o0 alpha^2 + o2 d20^2, K02 = o0 alpha beta0 + o2 d20 d22, K11 = o1 gamma^2, K12 = o1 gamma beta1, K22 = sum o_i d_i2^2, K01 = 0
//   N = K [y]x;  H_tt += wt^2 K,  H_tr -= wt N,  H_rr += [y]x^T N,  b_t += wt r,  b_r += y x r,  r = D^T Omega e
// Every entry enters its running sum through fused multiply-adds in the order written below (the CPU checker performs the same).
template <int DIM = 0, bool PRE_WT = false>
__device__ __forceinline__ void factor_accumulate(const prs_aligner_params& a, const PoseRegs& X, const float4 z, const float4 p_in,
                                                  const float mean_dsp, const bool active, f2* acc, float& code, int& cls, const bool inlier_only = false) {
  // Straight-line on purpose: with the terms live out of nested divergent branches the compiler re-materialises
  // all zeros at every nesting level.  A correspondence that is inactive (slot past the end) or invalid (behind the
  // camera / outside the image) runs the same arithmetic on harmless stand-in values (inverse depth 0, y = 0,
  // zero information): every product is then +-0, and a fixed-shape sum that is normalised with + 0.0f at its root does
  // not see the sign of a zero.  Valid correspondences execute exactly the operations of the sequential evaluation.
  const float fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
  const int dim = DIM ? DIM : a.factor_type;
  // explicit fused multiply-adds (one rounding each): the factor arithmetic is defined this way on
  // both sides of the parity test
  const float pcx = fmaf(X.R02, p_in.z, fmaf(X.R01, p_in.y, fmaf(X.R00, p_in.x, X.t0)));
  const float pcy = fmaf(X.R12, p_in.z, fmaf(X.R11, p_in.y, fmaf(X.R10, p_in.x, X.t1)));
  const float pcz = fmaf(X.R22, p_in.z, fmaf(X.R21, p_in.y, fmaf(X.R20, p_in.x, X.t2)));
  const float hx_r = fmaf(fx, pcx, cx * pcz);
  const float hy_r = fmaf(fy, pcy, cy * pcz);
  const float iz_r = recip_exact(pcz);  // (= 1.0f / pcz, bit for bit: prs_device.h)
  const float u_r = hx_r * iz_r, v_r = hy_r * iz_r;
  const bool valid = active && pcz > 0.0f && !(u_r < 0.0f || u_r > a.image_cols || v_r < 0.0f || v_r > a.image_rows);
  // the stand-in of an invalid / inactive correspondence: inverse depth 0 (the predicted image point and D are then zeros: hx_r,
  // hy_r, the measurement and the weight are finite), information 0 and y = 0 (a non-finite map point is "behind the camera"
  // for the reference, and 0 * NaN would poison the sums)
  const float iz = valid ? iz_r : 0.0f;
  const float u_pred = hx_r * iz, v_pred = hy_r * iz;  // = u_r, v_r of a valid correspondence (same operations), 0 otherwise
  const float e0 = u_pred - z.x, e1 = v_pred - z.y;
  float e2 = 0.0f;
  float ur = 0.0f;  // predicted column in the right image
  if (dim == PRS_FACTOR_STEREO) {
    const float hrx = hx_r + a.baseline_left_in_right_px[0];
    e2              = fmaf(hrx, iz, -z.z);
    ur              = hrx * iz;
  } else if (dim == PRS_FACTOR_DEPTH) {
    e2 = (valid ? pcz : 0.0f) - z.z;
  }
  float wt = 1.0f;
  if (dim == PRS_FACTOR_STEREO) {
    if (PRE_WT) {
      wt = z.w;  // (whoever parked the row wrote 1 there when the weighting is off: parked_translation_weight)
    } else if (a.enable_inverse_depth_weighting) {
      wt = inverse_depth_weight(z, mean_dsp, a.translation_weight_form);  // always finite
    }
  }
  const float alpha = fx * iz, gamma = fy * iz;
  const float beta0 = (cx - u_pred) * iz;
  const float beta1 = (cy - v_pred) * iz;
  const float beta2 = (cx - ur) * iz;
  const float d20   = dim == PRS_FACTOR_STEREO ? alpha : 0.0f;
  const float d22   = dim == PRS_FACTOR_STEREO ? beta2 : (dim == PRS_FACTOR_DEPTH ? (valid ? 1.0f : 0.0f) : 0.0f);
  const float yx = valid ? 2.0f * (pcx - X.t0) : 0.0f, yy = valid ? 2.0f * (pcy - X.t1) : 0.0f, yz = valid ? 2.0f * (pcz - X.t2) : 0.0f;
  // Omega = diag(info) * scale(moving point) (aligner_slice_processor_projective.cpp:46-56)
  const float s = valid ? p_in.w : 0.0f;
  float o0 = a.diagonal_info[0] * s;
  float o1 = a.diagonal_info[1] * s;
  float o2 = dim == PRS_FACTOR_MONO ? 0.0f : a.diagonal_info[2] * s;
  float chi = fmaf(o2 * e2, e2, fmaf(o1 * e1, e1, (o0 * e0) * e0));
  // saturated kernel: a kernelised factor is weighted 1 / chi (round 4: tools/sweep_a13.py; a factor of exactly 1 leaves the
  // unsaturated weights untouched)
  const bool saturated = valid && chi > a.chi_threshold;
  // (prs_aligner_params.kernel_weight_form: tau / chi instead.  The instantiations with a compile-time factor type are the shipped
  // family's: align_batch_launch sends every other reading to the generic ones)
  float ratio;  // inlier-only run: kernelised factors are suppressed
  if (DIM != 0) {
    ratio = inlier_only ? 0.0f : recip_exact(chi);  // (= 1.0f / chi, bit for bit)
  } else {
    ratio = inlier_only ? 0.0f : (a.kernel_weight_form == PRS_KERNEL_WEIGHT_TAU_OVER_CHI ? a.chi_threshold : 1.0f) / chi;
  }
  const float scale    = saturated ? ratio : 1.0f;
  o0 *= scale;
  o1 *= scale;
  o2 *= scale;
  chi = saturated ? a.chi_threshold : chi;
  cls = !active ? 3 : (!valid ? 2 : (saturated ? 1 : 0));
  acc[15] += f2{saturated ? 0.0f : chi, chi};
  // class counts: one exact small-integer code per correspondence, 1 for an inlier, kClsOutUnit for a kernelised one
  code += valid ? (saturated ? kClsOutUnit : 1.0f) : 0.0f;
  // K = D^T Omega D, r = D^T Omega e
  const float p0 = o0 * alpha, p2 = o2 * d20, g1 = o1 * gamma, q0 = o0 * beta0, q1 = o1 * beta1, q2 = o2 * d22;
  const float K00 = fmaf(p2, d20, p0 * alpha), K02 = fmaf(p2, d22, p0 * beta0), K11 = g1 * gamma, K12 = g1 * beta1;
  const float K22 = fmaf(q2, d22, fmaf(q1, beta1, q0 * beta0));
  const float w0 = o0 * e0, w1 = o1 * e1, w2 = o2 * e2;
  const float r0 = fmaf(d20, w2, alpha * w0), r1 = gamma * w1, r2 = fmaf(d22, w2, fmaf(beta1, w1, beta0 * w0));
  // N = K [y]x (K01 = 0)
  const float N00 = -(K02 * yy), N01 = fmaf(K02, yx, -(K00 * yz)), N02 = K00 * yy;
  const float N10 = fmaf(K11, yz, -(K12 * yy)), N11 = K12 * yx, N12 = -(K11 * yx);
  const float N20 = fmaf(K12, yz, -(K22 * yy)), N21 = fmaf(K22, yx, -(K02 * yz)), N22 = fmaf(K02, yy, -(K12 * yx));
  const float wt2 = wt * wt, nwt = -wt;
  // slot = 2 * pair + component; slot 1 (H01 of the camera-frame sums) receives nothing: K01 = 0
  acc[0].x  = fmaf(wt2, K00, acc[0].x);                      //  0: (0,0)
  acc[1].x  = fmaf(wt2, K02, acc[1].x);                      //  2: (0,2)
  acc[3].y  = fmaf(wt2, K11, acc[3].y);                      //  7: (1,1)
  acc[4].x  = fmaf(wt2, K12, acc[4].x);                      //  8: (1,2)
  acc[6].x  = fmaf(wt2, K22, acc[6].x);                      // 12: (2,2)
  acc[1].y  = fmaf(nwt, N00, acc[1].y);                      //  3: (0,3)
  acc[2].x  = fmaf(nwt, N01, acc[2].x);                      //  4: (0,4)
  acc[2].y  = fmaf(nwt, N02, acc[2].y);                      //  5: (0,5)
  acc[4].y  = fmaf(nwt, N10, acc[4].y);                      //  9: (1,3)
  acc[5].x  = fmaf(nwt, N11, acc[5].x);                      // 10: (1,4)
  acc[5].y  = fmaf(nwt, N12, acc[5].y);                      // 11: (1,5)
  acc[6].y  = fmaf(nwt, N20, acc[6].y);                      // 13: (2,3)
  acc[7].x  = fmaf(nwt, N21, acc[7].x);                      // 14: (2,4)
  acc[7].y  = fmaf(nwt, N22, acc[7].y);                      // 15: (2,5)
  acc[8].y  = fmaf(yz, N10, fmaf(-yy, N20, acc[8].y));       // 17: (3,3)
  acc[9].x  = fmaf(yz, N11, fmaf(-yy, N21, acc[9].x));       // 18: (3,4)
  acc[9].y  = fmaf(yz, N12, fmaf(-yy, N22, acc[9].y));       // 19: (3,5)
  acc[10].x = fmaf(yx, N21, fmaf(-yz, N01, acc[10].x));      // 20: (4,4)
  acc[10].y = fmaf(yx, N22, fmaf(-yz, N02, acc[10].y));      // 21: (4,5)
  acc[11].y = fmaf(yy, N02, fmaf(-yx, N12, acc[11].y));      // 23: (5,5)
  acc[12].x = fmaf(wt, r0, acc[12].x);                       // 24..29: b
  acc[12].y = fmaf(wt, r1, acc[12].y);
  acc[13].x = fmaf(wt, r2, acc[13].x);
  acc[13].y = fmaf(yy, r2, fmaf(-yz, r1, acc[13].y));
  acc[14].x = fmaf(yz, r0, fmaf(-yx, r2, acc[14].x));
  acc[14].y = fmaf(yx, r1, fmaf(-yy, r0, acc[14].y));
}

// ---- the fixed-shape sum of the normal equations (defined in include/proslam_hip.h, prs_align_result) --------------
//   leaf l (0..127) = terms of index l, l + 128, l + 256, ... added in that order;
//   seven pairwise levels v[l] <- v[l] + v[l ^ m], m = 32, 16, 8, 7, 2, 1, 64; result = v[0] + 0.0f.
// On the device leaf l is lane (l & 63) of wave (l >> 6) of a 128-thread workgroup: the levels 32 and 16 are
// v_permlane32_swap / v_permlane16_swap exchanges, 8 / 7 / 2 / 1 are DPP reads inside a row of 16 lanes (row_ror:8,
// row_half_mirror, two quad_perm), 64 is the add across the two waves.  The reduction is "transposed": a lane gives
// away half of its slots at every level, so 32 slots cost 16 + 8 + 4 + 2 + 1 + 1 adds per lane instead of 32 * 6.

template <int CTRL>
__device__ __forceinline__ float dpp_read(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void swap_halves32(f2& a, f2& b) {  // lanes 32..63 of a <-> lanes 0..31 of b (both components)
  const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a             = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b             = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
__device__ __forceinline__ void swap_rows16(f2& a, f2& b) {  // odd rows (of 16 lanes) of a <-> even rows of b
  const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a             = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b             = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
// sums the 32 slots of `acc` over the 64 lanes of the wave (levels 32, 16, 8, 7, 2, 1); lane L returns slot (L >> 1) & 31
__device__ __forceinline__ float wave_sum_slots(f2* acc, const int lane) {
  f2 w[8], x[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    swap_halves32(acc[i], acc[i + 8]);
    w[i] = acc[i] + acc[i + 8];  // lanes < 32: pairs 0..7, lanes >= 32: pairs 8..15
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    swap_rows16(w[i], w[i + 4]);
    x[i] = w[i] + w[i + 4];  // even rows: pairs i, odd rows: pairs i + 4 (of this half's eight)
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
  float y[4], zz[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // float slots j and j + 4 of this row's eight
    const float lo = (j & 1) ? x[j >> 1].y : x[j >> 1].x;
    const float hi = (j & 1) ? x[(j >> 1) + 2].y : x[(j >> 1) + 2].x;
    y[j]           = (b3 ? hi : lo) + dpp_read<0x128>(b3 ? lo : hi);  // row_ror:8 = lane ^ 8
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    zz[j] = (b2 ? y[j + 2] : y[j]) + dpp_read<0x141>(b2 ? y[j] : y[j + 2]);  // row_half_mirror = lane ^ 7
  }
  const float q = (b1 ? zz[1] : zz[0]) + dpp_read<0x4E>(b1 ? zz[0] : zz[1]);  // quad_perm [2,3,0,1] = lane ^ 2
  return q + dpp_read<0xB1>(q);                                                // quad_perm [1,0,3,2] = lane ^ 1
}

// where the lane that ends up with summed slot `slot` puts it, as a float index from the H member of the kernel's shared state (H[36], b[6], chi of the inliers, chi of all, then GnShared's three class counts) (`mirror`: the lower-triangle twin)
__device__ __forceinline__ int gn_slot_destination(const int slot, const bool mirror) {
  if (slot >= 24) {
    return 36 + (slot - 24);  // b0..b5, chi_in, chi_tot
  }
  if (slot == kSlotCls) {
    return 44;  // the class-count code
  }
  if (slot == kSlotUnusedA) {
    return 45;  // (mirrored entries: parked next to it, never read)
  }
  if (slot == kSlotUnusedB) {
    return 46;
  }
  // rows of the upper triangle start at slots 0, 6 (+1), 12, 16 (+1), 20, 22 (+1): row r holds columns (r & ~1) .. 5
  const int r     = slot < 6 ? 0 : (slot < 12 ? 1 : (slot < 16 ? 2 : (slot < 20 ? 3 : (slot < 22 ? 4 : 5))));
  const int first = r == 0 ? 0 : (r == 1 ? 6 : (r == 2 ? 12 : (r == 3 ? 16 : (r == 4 ? 20 : 22))));
  const int c     = (r & ~1) + (slot - first);
  return mirror ? 6 * c + r : 6 * r + c;
}

// ---- srrg2_core::KDTree<float, 2> as the KD-tree finder uses it (CF/correspondence_finder_projective_kdtree_impl.cpp:8-26,39-50;
// the class is external): split a cluster at its mean along the direction of largest variance until it holds fewer than
// minimum_number_of_points_per_cluster points or its extent 3 * sqrt(largest eigenvalue) is below the leaf range (the search
// radius when _initializeDatabase ran); a radius query is answered from the ONE leaf the query point descends to.  With these
// constants the eight counts the reference asserts for this finder come out exactly (tests/test_ref_pins.py).
// Arithmetic (defined in include/proslam_hip.h, PRS_SEARCH_KDTREE): exact integer sums of the coordinates in 1/16 px (order
// independent, exact in double), mean / covariance / eigen-decomposition in double, node = (mean, unit normal) in float, side
// test (x - mean_x) * n_x + (y - mean_y) * n_y < 0 -> left in float, leaf members in ascending fixed index.
struct KdNode {
  float mx, my, nx, ny;
  short child[2];  // >= 0: internal node, < 0: leaf ~child
  int pad;
};
static_assert(sizeof(KdNode) == 24, "KdNode layout");

__device__ __forceinline__ bool kd_decide(const long long* st, const int n, const double leaf_range, const int min_points, float* mean, float* normal) {
  const double dn = (double) n;
  const double sx = (double) st[0], sy = (double) st[1];
  mean[0]          = (float) (sx / (16.0 * dn));
  mean[1]          = (float) (sy / (16.0 * dn));
  const double cxx = ((double) st[2] - sx * sx / dn) / dn / 256.0;
  const double cxy = ((double) st[3] - sx * sy / dn) / dn / 256.0;
  const double cyy = ((double) st[4] - sy * sy / dn) / dn / 256.0;
  const double half_diff = 0.5 * (cxx - cyy);
  double lambda          = 0.5 * (cxx + cyy) + sqrt(half_diff * half_diff + cxy * cxy);
  if (!(lambda > 0.0)) {
    lambda = 0.0;
  }
  double vx = lambda - cyy, vy = cxy;
  const double norm = sqrt(vx * vx + vy * vy);
  if (norm > 0.0) {
    vx = vx / norm;
    vy = vy / norm;
  } else {
    vx = cxx >= cyy ? 1.0 : 0.0;
    vy = cxx >= cyy ? 0.0 : 1.0;
  }
  normal[0] = (float) vx;
  normal[1] = (float) vy;
  return n < min_points || 3.0 * sqrt(lambda) < leaf_range;
}
__device__ __forceinline__ float kd_side(const KdNode& nd, const float x, const float y) {
  const float dx = x - nd.mx, dy = y - nd.my;
  return dx * nd.nx + dy * nd.ny;
}
__host__ __device__ __forceinline__ int kd_open_capacity(const int max_fixed) {
  return max_fixed / 5 + 4;  // open clusters of one level: a cluster that is split holds >= 10 points
}

// SPLIT = the search half of the split pipeline (mode kModeSplitSearch): a compile-time flag, so the
// Gauss-Newton code (and its registers) is not part of that instantiation
// STYPE: the search pattern as a compile-time constant (-1 = read it from the parameters), so that the search
// half only carries the code and registers of its own pattern
// SLOTS: LDS slots per thread for survivors of the irrelevance bound (4, or 8 where they cost no resident workgroup): a compile-time
// constant like the pattern -- as run-time values the slot count and its mask pushed the kernel (at its 106 scalar registers)
// into scalar spills inside the scan: 13.99 -> 14.43 ms on the headline
template <int T, bool SPLIT, int STYPE, int SLOTS = 4>
__global__ __launch_bounds__(T, SPLIT ? 6 : 1) void align_kernel(const AlignArgs g) {  // search half: <= 128 VGPRs = two workgroups per CU
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid   = threadIdx.x;
  const int lane  = tid & 63;
  const int wave  = tid >> 6;
  const int tid_hot = tid;  // (cold blocks shadow tid / lane / wave with cold_copy(tid_hot))
  const int frame = blockIdx.x;

  int nF = g.b.n_fixed[frame];
  int nM = g.b.n_moving[frame];
  nF     = nF < 0 ? 0 : (nF > g.b.fixed_stride ? g.b.fixed_stride : nF);
  nM     = nM < 0 ? 0 : (nM > g.b.moving_stride ? g.b.moving_stride : nM);
  const size_t fbase = (size_t) frame * (size_t) g.b.fixed_stride;
  const size_t mbase = (size_t) frame * (size_t) g.b.moving_stride;
  const float4* __restrict__ gfix = reinterpret_cast<const float4*>(g.b.fixed) + fbase;
  const float4* __restrict__ gmov = reinterpret_cast<const float4*>(g.b.moving) + mbase;
  const au32x4* __restrict__ gfd  = reinterpret_cast<const au32x4*>(g.b.fixed_desc + fbase * PRS_DESC_BYTES);
  const au32x4* __restrict__ gmd  = reinterpret_cast<const au32x4*>(g.b.moving_desc + mbase * PRS_DESC_BYTES);
  prs_corr* __restrict__ gcorr    = g.b.corr + fbase;
  prs_pcf_state* gstate           = g.b.state + frame;
  prs_align_result* gres          = g.b.result + frame;

  uint2* db           = reinterpret_cast<uint2*>(smem + g.off_db);        // cell-sorted lattice: x = row | col << 16, y = fixed index | canonical position << 16
  uint16_t* inv       = reinterpret_cast<uint16_t*>(smem + g.off_inv);     // canonical position -> fixed index
  uint16_t* cellstart = reinterpret_cast<uint16_t*>(smem + g.off_cellstart);
  float4* cfix        = reinterpret_cast<float4*>(smem + g.off_cfix);      // per-correspondence fixed measurement
  float4* cmov        = reinterpret_cast<float4*>(smem + g.off_cmov);      // per-correspondence moving point + information scale
  au32x4* fdesc       = reinterpret_cast<au32x4*>(smem + g.off_fdesc);    // fixed descriptor rows (search phase only): first halves of all rows, then second halves
  au32x4* fdesc_hi    = fdesc + g.max_fixed;                               // (16-byte stride: a gather of first halves spreads over all 32 LDS banks; 32-byte rows hit 16)
  // Lattice patterns keep the rows in LATTICE order (row of db[pos] at fdesc[pos]): the scan reads an entry and its half row
  // with two independent LDS reads instead of entry -> fixed index -> row.  The KD-tree finder keeps them in fixed-index order.
  float2* fuv         = reinterpret_cast<float2*>(smem + g.off_fuv);      // fixed (u,v) in index order (KD-tree variant)
  uint32_t* bestkey   = reinterpret_cast<uint32_t*>(smem + g.off_best);
  uint32_t* second    = reinterpret_cast<uint32_t*>(smem + g.off_second);
  uint16_t* lut       = reinterpret_cast<uint16_t*>(smem + g.off_lut);
  constexpr int kSurvivors = SLOTS;                                        // survivors of the irrelevance bound a query may park (lattice scan): 4, or 8 where the LDS is free
  uint16_t* surv      = reinterpret_cast<uint16_t*>(smem + g.off_surv) + kSurvivors * tid;
  float* terms        = reinterpret_cast<float*>(smem + g.off_terms);     // aliases the search-phase arrays
  uint8_t* clsbuf     = smem + g.off_cls;                                  // factor class per correspondence (last linearisation)
  AlignShared& sh     = *reinterpret_cast<AlignShared*>(smem + g.off_sh);
  const int R         = g.rows_table;
  const int stype     = STYPE >= 0 ? STYPE : g.f.search_type;
  const bool lattice  = stype != PRS_SEARCH_KDTREE;

  constexpr bool split_search = SPLIT;
  FrameCtl* ctl           = split_search ? g.ctl + frame : nullptr;
  if (split_search && (ctl->done || !ctl->need_search)) {
    return;  // this frame does not wait for a search (block-uniform)
  }

  // ---- load the persistent state ---------------------------------------------------------------
  if (tid < 16) {
    sh.X[tid]     = g.b.X[(size_t) frame * 16 + tid];
    sh.T[tid]     = gstate->local_map_in_sensor[tid];
    sh.Tprev[tid] = gstate->local_map_in_sensor_previous[tid];
  }
  if (tid == 0) {
    sh.radius         = gstate->search_radius_pixels;
    sh.it             = gstate->current_iteration;
    sh.dd             = gstate->descriptor_distance;
    sh.kd_range       = gstate->database_leaf_range;
    sh.converged      = gstate->has_converged;
    sh.config_changed = gstate->config_changed;
    sh.num_recomputes = gstate->num_recomputes;
    sh.no_prune       = gstate->reserved & 1;
    sh.n_overflow     = 0;
    sh.n_corr         = g.b.n_corr[frame];
    sh.flags          = 0;
    sh.error          = 0;
    sh.n_inl = sh.n_out = sh.n_inv = 0;
    sh.chi_in = sh.chi_tot = 0.0f;
    sh.mean_disp           = (split_search && ctl->it_align != 0) ? gres->mean_disparity : g.a.mean_disparity;
    sh.corr_changed        = 1;
    sh.have_terms          = 0;
    sh.have_cls            = 0;
    if (sh.n_corr < 0 || sh.n_corr > nF) {
      sh.n_corr = 0;
    }
    {
      float x[16], a16[16], si[16];
      const float* gx = g.b.X + (size_t) frame * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = gx[i];
      }
      se3_inverse(g.a.sensor_in_robot, si);
      pose_to_camera(g.a, si, x, a16);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sh.Sinv[i] = si[i];
        sh.A[i]    = a16[i];
      }
    }
    if (nF > g.max_fixed) {
      sh.error = PRS_ERR_CAPACITY;  // the caller's max_fixed hint was too small for this frame
    }
  }
  for (int i = tid; i < 36; i += T) {
    sh.H[i] = 0.0f;
  }
  if (tid < 6) {
    sh.b[tid] = 0.0f;
  }
  __syncthreads();

  bool inputs_changed = g.b.inputs_changed ? (g.b.inputs_changed[frame] != 0) : true;
  if (split_search && ctl->it_align != 0) {
    inputs_changed = false;  // later searches of the same frame loop
  }
  bool db_built       = false;
  if (inputs_changed && g.mode != PRS_MODE_LINEARIZE) {
    // a new fixed/moving cloud invalidates the previous frame's correspondence vector
    if (tid == 0) {
      sh.n_corr = 0;
    }
    __syncthreads();
  }

  // ---- bindFixed: mean disparity over ALL fixed points, sequential float sum -------------------
  // (aligner_slice_processor_projective.cpp:80-88)
  if (g.mode != PRS_MODE_FINDER && (!split_search || ctl->it_align == 0) && g.a.factor_type == PRS_FACTOR_STEREO &&
      g.a.enable_inverse_depth_weighting && g.a.mean_disparity < 0.0f && !sh.error) {
    for (int i = tid; i < nF; i += T) {
      const float4 c = gfix[i];
      terms[i]       = c.x - c.z;
    }
    __syncthreads();
    if (tid == 0) {
      float acc = 0.0f;
      int i     = 0;
      const float4* t4 = reinterpret_cast<const float4*>(terms);
      float4 n0 = t4[0], n1 = t4[1], n2 = t4[2], n3 = t4[3];  // the column is padded with zeros up to a multiple of 16
      for (; i + 16 <= nF; i += 16) {
        const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3;
        const int k = (i >> 2) + 4;
        n0 = t4[k];
        n1 = t4[k + 1];
        n2 = t4[k + 2];
        n3 = t4[k + 3];
        acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w;
        acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
        acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w;
        acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
      }
      for (; i < nF; ++i) {
        acc += terms[i];
      }
      sh.mean_disp = nF > 0 ? acc / (float) (size_t) nF : 0.0f;
    }
    __syncthreads();
  }

  // the caller-owned correspondence vector persists across calls: "nothing new" finder calls keep
  // using it, and linearize-only calls receive it as input
  if (g.mode != PRS_MODE_FINDER && !split_search && !sh.error) {
    for (int c = tid; c < sh.n_corr; c += T) {
      const prs_corr cr = gcorr[c];
      if (cr.fixed_idx < 0 || cr.fixed_idx >= nF || cr.moving_idx < 0 || cr.moving_idx >= nM) {
        sh.error = PRS_ERR_RANGE;
      } else {
        cfix[c] = gfix[cr.fixed_idx];
        cmov[c] = gmov[cr.moving_idx];
      }
    }
    __syncthreads();
  }

  enum { acc_finder = 0, acc_lin, acc_sum, acc_solve, acc_db, acc_search, acc_pass2, acc_filter, acc_commit };
  if (g.stamps && tid == 0) {
    for (int i = 0; i < 9; ++i) {
      sh.stamp_acc[i] = 0;
    }
    sh.stamp_mark = sh.stamp_sub = 0;
  }
#define SUB_MARK() do { if (g.stamps && tid == 0) { sh.stamp_sub = (unsigned long long) clock64(); } } while (0)
#define SUB_ACC(v) do { if (g.stamps && tid == 0) { const unsigned long long now_ = (unsigned long long) clock64(); sh.stamp_acc[v] += now_ - sh.stamp_sub; sh.stamp_sub = now_; } } while (0)
#define ALIGN_MARK() do { if (g.stamps && tid == 0) { sh.stamp_mark = (unsigned long long) clock64(); } } while (0)
#define ALIGN_ACC(v) do { if (g.stamps && tid == 0) { sh.stamp_acc[v] += (unsigned long long) clock64() - sh.stamp_mark; } } while (0)
  // PRS_MODE_ALIGN: max_iterations of finder + GN, then the inlier-only run (frozen correspondences, no finder)
  const int max_it = (g.mode == PRS_MODE_ALIGN && !sh.error) ? g.a.max_iterations + inlier_run_length(g.a) : (sh.error ? 0 : 1);
  if (split_search && sh.error && tid == 0) {
    ctl->done = 1;  // loud per-frame error, the GN kernel never touches this frame
  }
  int executed     = 0;
  int it_align     = 0;
  bool ran_inlier_phase = false;
  for (; it_align < max_it; ++it_align) {
    const bool inlier_run = g.mode == PRS_MODE_ALIGN && it_align >= g.a.max_iterations;
    if (g.mode == PRS_MODE_ALIGN && it_align == g.a.max_iterations) {
      if (sh.n_inl < g.a.min_num_inliers) {
        break;  // not enough inliers for an inlier-only run (block-uniform: sh.n_inl was written before the last barrier)
      }
      ran_inlier_phase = true;
    }
    ++executed;
    // ============================================================================================
    // finder.setLocalMapInSensor(X); finder.compute()
    // ============================================================================================
    ALIGN_MARK();
    if (g.mode != PRS_MODE_LINEARIZE && !inlier_run) {
      if (tid < 16) {
        sh.T[tid] = sh.A[tid];
      }
      __syncthreads();
      for (;;) {  // the reference re-enters compute() recursively (projective_base_impl.cpp:262)
        // -- new optimisation when fixed / moving / config changed (:109-134)
        const bool reset = inputs_changed || sh.config_changed;
        __syncthreads();
        if (reset) {
          if (tid == 0) {
            if ((sh.radius == 0 && sh.dd == 0.0f) || sh.config_changed) {
              sh.radius = g.f.maximum_search_radius_pixels;
              sh.dd     = g.f.minimum_descriptor_distance;
            }
            if (sh.config_changed) {
              sh.no_prune = 0;
            }
            sh.converged = 0;
            sh.it        = 0;
            se3_identity(sh.Tprev);
            sh.config_changed = 0;
            sh.kd_range       = (float) sh.radius;  // _initializeDatabase(): KDTree(fixed, _search_radius_pixels, ..) (kdtree_impl.cpp:21-24)
          }
          inputs_changed = false;
          db_built       = false;  // _initializeDatabase() is part of the reset (:131)
          __syncthreads();
        }
        if (sh.error) {
          break;
        }

        // -- converged: correspondences are not touched (:138-142)
        if (tid == 0) {
          sh.decision = kDecisionCommit;
          if (sh.converged) {
            sh.decision = kDecisionReturn;
          } else {
            // setCameraPose(local_map_in_sensor^-1) (:158); reproject periodically, always for it 0 and 1 (:162-178)
            float cam[16];
            se3_inverse(sh.T, cam);
            const unsigned long long k = g.f.number_of_solver_iterations_per_projection;
            if (k == 0 || sh.it % k == 0 || sh.it == 1) {
              se3_inverse(cam, sh.W);
              float delta[16], v6[6];
              se3_mul(cam, sh.Tprev, delta);
              t2tnq(delta, v6);
              sh.change_norm =
                sqrtf(((((v6[0] * v6[0] + v6[1] * v6[1]) + v6[2] * v6[2]) + v6[3] * v6[3]) + v6[4] * v6[4]) + v6[5] * v6[5]);
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                sh.Tprev[i] = sh.T[i];
              }
              sh.n_projected = 0;
              sh.n_overflow  = 0;
              ++sh.num_recomputes;
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                sh.Tprev[i] = sh.T[i];
              }
              ++sh.it;
              sh.decision = kDecisionReturn;
            }
          }
        }
        __syncthreads();
        if (sh.decision == kDecisionReturn) {
          break;
        }

        SUB_MARK();
        // -- the lattice lives in LDS: (re)build it on the first search of a launch and after every reset
        if (!db_built && split_search && g.dbcache && ctl->db_ready) {
          // the lattice only depends on the fixed cloud: the first search launch of a frame built it and left
          // an image of the three LDS arrays in global memory; later launches copy it back (11 kB, coalesced)
          const uint4* src = reinterpret_cast<const uint4*>(g.dbcache + (size_t) frame * g.db_blob);
          uint4* dst       = reinterpret_cast<uint4*>(smem + g.off_db);
          for (int i = tid; i < (int) (g.db_blob >> 4); i += T) {
            dst[i] = src[i];
          }
          __syncthreads();
          db_built = true;
        }
        if (!db_built && !lattice) {
          const int tid = cold_copy(tid_hot), lane = tid & 63, wave = tid >> 6;
          (void) lane;
          (void) wave;
          // _initializeDatabase of the KD-tree finder (kdtree_impl.cpp:8-26), level by level: the open clusters of a level get
          // their exact coordinate sums (LDS 64-bit atomics), one lane per cluster decides leaf / split, every point moves to
          // its child.  Node and leaf numbers depend on the scheduling, the tree does not.
          KdNode* nodes        = reinterpret_cast<KdNode*>(db);
          uint16_t* order      = inv;
          short* kdhead        = reinterpret_cast<short*>(cellstart);  // [0] root code, [1] number of leaves
          uint16_t* leaf_start = cellstart + 2;
          const int node_cap   = (int) ((size_t) g.max_fixed * sizeof(uint2) / sizeof(KdNode));
          const int open_cap   = kd_open_capacity(g.max_fixed);
          const int min_points = g.f.minimum_number_of_points_per_cluster > 0 ? g.f.minimum_number_of_points_per_cluster : 10;
          unsigned long long* st = reinterpret_cast<unsigned long long*>(terms);  // [open_cap][5]
          int* cnt        = reinterpret_cast<int*>(st + (size_t) open_cap * 5);
          int* kind       = cnt + open_cap;
          int* child_open = kind + open_cap;
          int* par_cur    = child_open + open_cap;  // parent node, side << 30 folded in below
          int* pcnt_cur   = par_cur + open_cap;
          int* par_next   = pcnt_cur + open_cap;
          int* pcnt_next  = par_next + open_cap;
          uint16_t* node_of = reinterpret_cast<uint16_t*>(pcnt_next + open_cap);
          uint16_t* leaf_of = node_of + g.max_fixed + 2;
          uint16_t* bucket  = leaf_of + g.max_fixed + 2;
          for (int i = tid; i < nF; i += T) {
            const float4 c = gfix[i];
            if (!(c.x > -32768.0f && c.x < 32768.0f && c.y > -32768.0f && c.y < 32768.0f)) {
              sh.error = PRS_ERR_RANGE;
            }
            node_of[i] = 0;
          }
          if (tid == 0) {
            sh.kd_nodes = sh.kd_leaves = 0;
            par_cur[0]  = -1;
            pcnt_cur[0] = -1;
            if (nF == 0) {  // one empty leaf
              kdhead[0]     = (short) ~0;
              sh.kd_leaves  = 1;
            }
          }
          __syncthreads();
          if (sh.error) {
            break;
          }
          int n_open = nF > 0 ? 1 : 0;
          while (n_open > 0) {
            for (int k = tid; k < n_open; k += T) {
#pragma unroll
              for (int j = 0; j < 5; ++j) {
                st[5 * k + j] = 0ull;
              }
              cnt[k] = 0;
            }
            if (tid == 0) {
              sh.kd_open_next = 0;
            }
            __syncthreads();
            for (int i = tid; i < nF; i += T) {
              const int k = node_of[i];
              if (k != 0xffff) {
                const float4 c     = gfix[i];
                const long long qx = __float2ll_rn(c.x * 16.0f), qy = __float2ll_rn(c.y * 16.0f);
                atomicAdd(&st[5 * k + 0], (unsigned long long) qx);
                atomicAdd(&st[5 * k + 1], (unsigned long long) qy);
                atomicAdd(&st[5 * k + 2], (unsigned long long) (qx * qx));
                atomicAdd(&st[5 * k + 3], (unsigned long long) (qx * qy));
                atomicAdd(&st[5 * k + 4], (unsigned long long) (qy * qy));
                atomicAdd(&cnt[k], 1);
              }
            }
            __syncthreads();
            for (int k = tid; k < n_open; k += T) {
              const int n = cnt[k];
              int code    = 0x7fffffff;  // nothing here (the empty side of a split that could not separate its points)
              if (n > 0) {
                const int P       = par_cur[k] < 0 ? -1 : (par_cur[k] & 0x3fffffff);
                const int side    = par_cur[k] < 0 ? 0 : ((par_cur[k] >> 30) & 1);
                const bool forced = pcnt_cur[k] == n;  // every point of the parent went to this side: the parent IS this leaf
                float mean[2], normal[2];
                bool leaf = kd_decide(reinterpret_cast<const long long*>(st + 5 * k), n, (double) sh.kd_range, min_points, mean, normal) || forced;
                int N     = -1;
                if (!leaf) {
                  N = atomicAdd(&sh.kd_nodes, 1);
                  const int c0 = atomicAdd(&sh.kd_open_next, 2);
                  if (N >= node_cap || N > 32767 || c0 + 2 > open_cap) {
                    sh.error = PRS_ERR_CAPACITY;  // more clusters than the LDS carve of this max_fixed holds
                    leaf     = true;
                  } else {
                    KdNode nd;
                    nd.mx = mean[0];
                    nd.my = mean[1];
                    nd.nx = normal[0];
                    nd.ny = normal[1];
                    nd.child[0] = nd.child[1] = (short) ~0;
                    nd.pad   = 0;
                    nodes[N] = nd;
                    child_open[k]     = c0;
                    par_next[c0]      = N;
                    par_next[c0 + 1]  = N | (1 << 30);
                    pcnt_next[c0]     = n;
                    pcnt_next[c0 + 1] = n;
                    code              = N;
                  }
                }
                if (leaf) {
                  code = ~atomicAdd(&sh.kd_leaves, 1);
                }
                if (P < 0) {
                  kdhead[0] = (short) code;
                } else if (forced) {
                  nodes[P].child[0] = nodes[P].child[1] = (short) code;
                } else {
                  nodes[P].child[side] = (short) code;
                }
              }
              kind[k] = code;
            }
            __syncthreads();
            if (sh.error) {
              break;
            }
            const int next_open = sh.kd_open_next;
            for (int i = tid; i < nF; i += T) {
              const int k = node_of[i];
              if (k != 0xffff) {
                const int code = kind[k];
                if (code < 0) {
                  leaf_of[i] = (uint16_t) ~code;
                  node_of[i] = 0xffff;
                } else {
                  const float4 c = gfix[i];
                  node_of[i]     = (uint16_t) (child_open[k] + (kd_side(nodes[code], c.x, c.y) < 0.0f ? 0 : 1));
                }
              }
            }
            n_open = next_open;
            {
              int* t0 = par_cur;
              par_cur = par_next;
              par_next = t0;
              t0       = pcnt_cur;
              pcnt_cur = pcnt_next;
              pcnt_next = t0;
            }
            __syncthreads();
          }
          if (sh.error) {
            break;
          }
          // leaf members in ascending fixed index: counting sort by leaf, rank inside the leaf
          const int n_leaves = sh.kd_leaves;
          uint32_t* lh       = reinterpret_cast<uint32_t*>(st);
          for (int l = tid; l <= n_leaves; l += T) {
            lh[l] = 0;
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            node_of[i] = (uint16_t) atomicAdd(&lh[leaf_of[i]], 1u);  // (node_of is free again: slot inside the leaf)
          }
          __syncthreads();
          if (wave == 0) {
            const int n     = n_leaves + 1;
            const int chunk = (n + 63) >> 6;
            uint32_t sum    = 0;
            for (int j = 0; j < chunk; ++j) {
              const int r = lane * chunk + j;
              sum += r < n ? lh[r] : 0u;
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
              const uint32_t o = __shfl_up(incl, d, 64);
              if (lane >= d) {
                incl += o;
              }
            }
            uint32_t run = incl - sum;
            for (int j = 0; j < chunk; ++j) {
              const int r = lane * chunk + j;
              if (r < n) {
                leaf_start[r] = (uint16_t) run;
                run += lh[r];
              }
            }
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            bucket[leaf_start[leaf_of[i]] + node_of[i]] = (uint16_t) i;
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            const int l0 = leaf_start[leaf_of[i]], l1 = leaf_start[leaf_of[i] + 1];
            int rank = 0;
            for (int j = l0; j < l1; ++j) {
              rank += bucket[j] < (uint16_t) i ? 1 : 0;
            }
            order[l0 + rank] = (uint16_t) i;
          }
          if (tid == 0) {
            kdhead[1] = (short) n_leaves;
          }
          __syncthreads();
          db_built = true;
          if (split_search && g.dbcache) {
            const uint4* src = reinterpret_cast<const uint4*>(smem + g.off_db);
            uint4* dst       = reinterpret_cast<uint4*>(g.dbcache + (size_t) frame * g.db_blob);
            for (int i = tid; i < (int) (g.db_blob >> 4); i += T) {
              dst[i] = src[i];
            }
            if (tid == 0) {
              ctl->db_ready = 1;  // read by the NEXT search launch of this frame
            }
          }
        }
        if (!db_built) {
          const int tid = cold_copy(tid_hot), lane = tid & 63, wave = tid >> 6;
          (void) lane;
          (void) wave;
          // _initializeDatabase (square_impl.cpp:8-31).  The reference scans a row-sorted vector; its
          // scan position only matters for tie-breaks ("first wins"), so every fixed point gets its
          // CANONICAL position (stable row sort, ascending index inside a row) and the lattice itself
          // is bucketed into a 2-D cell grid: a query visits the cells under its search region
          // instead of every entry of ~2r+1 rows.
          const int histcap  = (R > g.ncells ? R : g.ncells) + 2;
          uint32_t* hist     = reinterpret_cast<uint32_t*>(terms);
          uint16_t* slot     = reinterpret_cast<uint16_t*>(hist + histcap);
          uint16_t* bucket   = slot + g.max_fixed + 2;
          uint16_t* canon    = bucket + g.max_fixed + 2;
          uint16_t* rowfirst = canon + g.max_fixed + 2;
          for (int i = tid; i <= R; i += T) {
            hist[i] = 0;
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            const float4 c = gfix[i];
            if (c.x >= 0.0f && c.x < 32767.0f && c.y >= 0.0f && c.y < (float) R) {
              const int row = (int) (int16_t) c.y;  // Element(coordinates(1), coordinates(0), i)
              slot[i]       = (uint16_t) atomicAdd(&hist[row], 1u);
            } else {
              sh.error = PRS_ERR_RANGE;
            }
          }
          __syncthreads();
          if (sh.error) {
            break;
          }
          {
            if (wave == 0) {
              const int n     = R + 1;
              const int chunk = (n + 63) >> 6;
              uint32_t sum    = 0;
              for (int j = 0; j < chunk; ++j) {
                const int r = lane * chunk + j;
                sum += r < n ? hist[r] : 0u;
              }
              uint32_t incl = sum;
#pragma unroll
              for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if (lane >= d) {
                  incl += o;
                }
              }
              uint32_t run = incl - sum;
              for (int j = 0; j < chunk; ++j) {
                const int r = lane * chunk + j;
                if (r < n) {
                  rowfirst[r] = (uint16_t) run;
                  run += hist[r];
                }
              }
            }
            __syncthreads();
            for (int i = tid; i < nF; i += T) {
              const int row = (int) (int16_t) gfix[i].y;
              bucket[rowfirst[row] + slot[i]] = (uint16_t) i;
            }
            __syncthreads();
            for (int i = tid; i < nF; i += T) {
              const int row = (int) (int16_t) gfix[i].y;
              const int s = rowfirst[row], e = rowfirst[row + 1];
              int rank = 0;
              for (int j = s; j < e; ++j) {
                rank += bucket[j] < (uint16_t) i ? 1 : 0;
              }
              canon[i] = (uint16_t) (s + rank);
            }
          }
          __syncthreads();
          // bucket by cell (order inside a cell is irrelevant: ties are resolved on the canonical position)
          for (int i = tid; i <= g.ncells; i += T) {
            hist[i] = 0;
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            const float4 c = gfix[i];
            const int row  = (int) (int16_t) c.y;
            int cxi        = ((int) (int16_t) c.x) >> g.cell_sx;
            cxi            = cxi < g.cell_ncx ? cxi : g.cell_ncx - 1;
            slot[i]        = (uint16_t) atomicAdd(&hist[(row >> g.cell_sy) * g.cell_ncx + cxi], 1u);
          }
          __syncthreads();
          if (wave == 0) {
            const int n     = g.ncells + 1;
            const int chunk = (n + 63) >> 6;
            uint32_t sum    = 0;
            for (int j = 0; j < chunk; ++j) {
              const int r = lane * chunk + j;
              sum += r < n ? hist[r] : 0u;
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
              const uint32_t o = __shfl_up(incl, d, 64);
              if (lane >= d) {
                incl += o;
              }
            }
            uint32_t run = incl - sum;
            for (int j = 0; j < chunk; ++j) {
              const int r = lane * chunk + j;
              if (r < n) {
                cellstart[r] = (uint16_t) run;
                run += hist[r];
              }
            }
          }
          __syncthreads();
          for (int i = tid; i < nF; i += T) {
            const float4 c = gfix[i];
            const int row  = (int) (int16_t) c.y;
            const int col  = (int) (int16_t) c.x;
            int cxi        = col >> g.cell_sx;
            cxi            = cxi < g.cell_ncx ? cxi : g.cell_ncx - 1;
            db[cellstart[(row >> g.cell_sy) * g.cell_ncx + cxi] + slot[i]] =
              make_uint2(((uint32_t) row & 0xffffu) | ((uint32_t) col << 16), (uint32_t) i | ((uint32_t) canon[i] << 16));
            inv[canon[i]] = (uint16_t) i;
          }
          if (tid == 0) {
            db[nF] = make_uint2(0x7fff7fffu, 0xffffffffu);  // sentinel behind the last entry: row / column no search pattern accepts
          }
          __syncthreads();
          db_built = true;
          if (split_search && g.dbcache) {
            const uint4* src = reinterpret_cast<const uint4*>(smem + g.off_db);
            uint4* dst       = reinterpret_cast<uint4*>(g.dbcache + (size_t) frame * g.db_blob);
            for (int i = tid; i < (int) (g.db_blob >> 4); i += T) {
              dst[i] = src[i];
            }
            if (tid == 0) {
              ctl->db_ready = 1;  // read by the NEXT search launch of this frame
            }
          }
        }

        SUB_ACC(acc_db);
        // -- projection + candidate search (:165-166, :192-200)
        const int rad = (int) sh.radius;
        {
          const int tid = cold_copy(tid_hot);
          for (int i = tid; i < nF; i += T) {
            bestkey[i] = kNoneU32;
            second[i]  = kNoneU32;
          }
          if (lattice) {
            // rows in lattice order (the entry at db[pos] owns fdesc[pos] / fdesc_hi[pos]); the fixed cloud of the stereo adaptor is
            // row-sorted like the cells, so neighbouring positions read neighbouring rows
            for (int pos = tid; pos < nF; pos += T) {
              const int fi  = (int) (db[pos].y & 0xffffu);
              fdesc[pos]    = gfd[2 * fi];
              fdesc_hi[pos] = gfd[2 * fi + 1];
            }
          } else {
            for (int i = tid; i < 2 * nF; i += T) {
              (i & 1 ? fdesc_hi : fdesc)[i >> 1] = gfd[i];  // coalesced 16 B/lane
            }
            for (int i = tid; i < nF; i += T) {
              const float4 c = gfix[i];
              fuv[i]         = make_float2(c.x, c.y);
            }
          }
          if (stype == PRS_SEARCH_CIRCLE) {
            // width = int(sqrt(r^2 - h^2) + 1) per row offset h (circle_impl.cpp:51-53), exact in integers
            for (int i = tid; i < 2 * rad + 1 && i < g.lut_cap; i += T) {
              const int h = i - rad;
              lut[i]      = (uint16_t) (isqrt_exact(rad * rad - h * h) + 1);
            }
          }
        }
        __syncthreads();
        SUB_ACC(acc_pass2);  // staging: candidate keys, fixed descriptor rows -> LDS
        {
          const float W0 = sh.W[0], W1 = sh.W[1], W2 = sh.W[2], W3 = sh.W[3];
          const float W4 = sh.W[4], W5 = sh.W[5], W6 = sh.W[6], W7 = sh.W[7];
          const float W8 = sh.W[8], W9 = sh.W[9], W10 = sh.W[10], W11 = sh.W[11];
          const float cols = (float) g.f.projector.canvas_cols, rows = (float) g.f.projector.canvas_rows;
          const float max_dd = g.f.maximum_descriptor_distance;
          const float r2f    = (float) (sh.radius * sh.radius);
          const int rad2     = rad * rad;
          // candidates at or beyond this descriptor distance cannot change what the filter below accepts; only worth a
          // separate pass while the bound is well below what half a random descriptor pair differs by (64 +- 6 bits)
          const int irrelevant = irrelevant_distance(sh.dd, g.f.maximum_distance_ratio_to_second_best);
          // (... and not at all on sequences whose rows have shown themselves correlated: on the reference's real KITTI keypoints a
          // third of the scanned entries survive a bound of 31, every query overflows its slots and is scanned twice; unpruned the
          // real-keypoint leg runs 2.25 ms per 6144 frames against 3.05 pruned.  The hint is set below, lives in the finder's state
          // and changes no result)
          const int prune_at   = (lattice && irrelevant > 0 && irrelevant <= 54 && !g.no_prefilter && !sh.no_prune) ? irrelevant : 0;
          // the int16 row / column arithmetic of the reference cannot wrap on this canvas
          const bool circle_exact = rad < 8192 && R + rad < 32000 && (g.cell_ncx << g.cell_sx) + rad < 32000;
          int projected      = 0;
          // four queries per thread are fetched together (point + 256-bit row: 48 B each) so the
          // global-memory latency is paid once per block of queries, not once per query
          constexpr int kPre = SPLIT ? 1 : 4;  // queries fetched together
          for (int m0 = tid; m0 < nM; m0 += kPre * T) {
           float4 pre_p[kPre];
           au32x4 pre_q0[kPre], pre_q1[kPre];
#pragma unroll
           for (int qi = 0; qi < kPre; ++qi) {
             const int mm = m0 + qi * T < nM ? m0 + qi * T : m0;
             pre_p[qi]     = gmov[mm];
             pre_q0[qi]    = gmd[2 * mm];
             pre_q1[qi]    = gmd[2 * mm + 1];
           }
#pragma unroll
           for (int qi = 0; qi < kPre; ++qi) {
            const int m = m0 + qi * T;
            if (m >= nM) {
              continue;
            }
            const float4 p = pre_p[qi];
            // PointProjectorPinhole_::compute (external)
            const float x = ((W0 * p.x + W1 * p.y) + W2 * p.z) + W3;
            const float y = ((W4 * p.x + W5 * p.y) + W6 * p.z) + W7;
            const float z = ((W8 * p.x + W9 * p.y) + W10 * p.z) + W11;
            uint2 cd      = make_uint2(kNoneU32, kNoneU32);
            bool visible  = !(z < g.f.projector.range_min || z > g.f.projector.range_max);
            float u = 0.0f, v = 0.0f;
            if (visible) {
              const float hx = g.f.projector.fx * x + g.f.projector.cx * z;
              const float hy = g.f.projector.fy * y + g.f.projector.cy * z;
              u              = hx / z;
              v              = hy / z;
              visible        = !(u < 0.0f || u >= cols || v < 0.0f || v >= rows);
            }
            if (visible) {
              ++projected;
              const au32x4 q0 = pre_q0[qi], q1 = pre_q1[qi];
              // top-2 on (distance, canonical lattice position) keys: identical to the reference's
              // sequential "strictly smaller wins" scan, but independent of the visiting order
              uint32_t bestk = kNoneU32, seck = kNoneU32;
              int row = 0, col = 0, rmin = 0, rmax = 0, cmin = 0, cmax = 0;
              int r0, r1, cb0, cb1;
              if (lattice) {
                row  = (int) (int16_t) roundf(v);           // circle_impl.cpp:15-16
                col  = (int) (int16_t) roundf(u);
                rmin = (int) (int16_t) (row - rad);         // :25
                rmax = (int) (int16_t) (row + rad + 1);     // :26
                cmin = (int) (int16_t) (col - rad - 1);     // square_impl.cpp:56
                cmax = (int) (int16_t) (col + rad + 1);     // square_impl.cpp:57
                r0   = rmin;
                r1   = rmax - 1;
                cb0  = col - rad;                           // every pattern accepts only |dcol - col| <= rad
                cb1  = col + rad;
              } else {
                // KDTree::findNeighbors(query, r^2) (kdtree_impl.cpp:39-50): the leaf the query descends to, its members
                // within the radius in ascending fixed index, best initialised to maximum_descriptor_distance (:54)
                const KdNode* nodes        = reinterpret_cast<const KdNode*>(db);
                const uint16_t* leaf_start = cellstart + 2;
                int code                   = (int) reinterpret_cast<const short*>(cellstart)[0];
                while (code >= 0) {
                  const KdNode nd = nodes[code];
                  code            = (int) nd.child[kd_side(nd, u, v) < 0.0f ? 0 : 1];
                }
                const int leaf = ~code;
                for (int o = leaf_start[leaf]; o < (int) leaf_start[leaf + 1]; ++o) {
                  const int fi   = inv[o];
                  const float2 c = fuv[fi];
                  const float du = c.x - u, dv = c.y - v;
                  if (du * du + dv * dv < r2f) {
                    const uint32_t d   = (uint32_t) hamming_regs(fdesc[fi], fdesc_hi[fi], q0, q1);
                    const uint32_t key = (float) d < max_dd ? ((d << 16) | (uint32_t) fi) : kNoneU32;
                    bestk              = key < bestk ? key : bestk;
                  }
                }
                r0 = 1;  // (no lattice scan)
                r1 = cb0 = cb1 = 0;
              }
              r0  = (lattice && r0 < 0) ? 0 : r0;
              r1  = r1 > R - 1 ? R - 1 : r1;
              cb0 = cb0 < 0 ? 0 : cb0;
              const int colmax = (g.cell_ncx << g.cell_sx) - 1;
              cb1 = cb1 > colmax ? colmax : cb1;
              if (r0 <= r1 && cb0 <= cb1) {
                const int cx0 = cb0 >> g.cell_sx, cx1 = cb1 >> g.cell_sx;
                const uint32_t query_rc = ((uint32_t) row & 0xffffu) | ((uint32_t) col << 16);
                // whether the lattice entry e lies inside the search pattern of this query.  Circle on a canvas where the int16
                // arithmetic of the reference cannot wrap: rows [row - r, row + r] (circle_impl.cpp:25-26,40-47) and
                // col - w < dcol < col + w with w = int(sqrt(r^2 - h^2) + 1) (:51-56) is |dc| <= isqrt(r^2 - h^2), i.e. the integer
                // test dc^2 + h^2 <= r^2 (which also implies |h| <= r): no width table, no row compare.  Both differences in one
                // packed 16-bit subtract on the entry's (row | col << 16) word, their squares summed by one dot product.
                auto accepts_circle = [&](const uint2 e) -> bool {
                  typedef short i16x2 __attribute__((ext_vector_type(2)));
                  const i16x2 d = __builtin_bit_cast(i16x2, e.x) - __builtin_bit_cast(i16x2, query_rc);
                  return __builtin_amdgcn_sdot2(d, d, 0, false) <= rad2;
                };
                auto accepts_any = [&](const uint2 e) -> bool {
                  const int drow = (int) (int16_t) (e.x & 0xffffu);
                  const int dcol = (int) (int16_t) (e.x >> 16);
                  if (drow < rmin || drow >= rmax) {
                    return false;  // outside the scanned rows (circle_impl.cpp:40-47)
                  } else if (stype == PRS_SEARCH_SQUARE) {
                    return dcol > cmin && dcol < cmax;  // square_impl.cpp:80
                  } else if (stype == PRS_SEARCH_CIRCLE) {
                    const int h     = drow - row;
                    const int li    = h + rad;
                    const int width = li < g.lut_cap ? (int) lut[li] : isqrt_exact(rad * rad - h * h) + 1;
                    return dcol > col - width && dcol < col + width;  // circle_impl.cpp:56
                  } else {
                    int width = (int) (int16_t) (drow - rmin + 1);  // rhombus_impl.cpp:49-52
                    if (width > (int) (int16_t) rad) {
                      width = (int) (int16_t) (rmax - drow);
                    }
                    return dcol > col - width && dcol < col + width;
                  }
                };
                // the lattice scan: two entries per trip (the entry behind the last one of a segment is read but not used).  ONE loop over
                // the entries of all cell rows of the query: a lane that has finished a cell row's segment moves on to the next row
                // while its neighbours are still inside theirs, so a wave makes max over lanes of (sum over rows) trips instead of
                // sum over rows of (max over lanes) -- the segments hold 3 +- 2 entries, which nearly halves the trips.
                // Round 6 (the kernel issues 0.84 vector instructions per cycle and CU: only their number counts).  With the
                // irrelevance bound on, a trip reads the two HALF ROWS only (one LDS round trip, the address depends on the position
                // alone) and compares their distances with the bound; the entry itself -- row, column, indices -- and the pattern test
                // are looked at for the survivors only (a true match and next to nothing else: 1.4 % of the scanned entries).  The
                // set of candidates that reach the full score is the same as with the pattern test first: both tests are conjuncts.
                const bool lean_circle = stype == PRS_SEARCH_CIRCLE && circle_exact;
                auto score_at = [&](const int p) {  // full distance of the entry at lattice position p
                  const uint32_t ey = db[p].y;       // fixed index | canonical lattice position << 16
                  const uint32_t d  = (uint32_t) hamming_regs(fdesc[p], fdesc_hi[p], q0, q1);
                  // best / second best (circle_impl.cpp:64-72) as min / second-min of unique keys
                  const uint32_t key = (d << 16) | (ey >> 16);
                  const uint32_t hi  = key > bestk ? key : bestk;
                  seck               = hi < seck ? hi : seck;
                  bestk              = key < bestk ? key : bestk;
                };
                typedef __attribute__((address_space(3))) unsigned char lds_byte;
                const uint32_t fd_lds = (uint32_t) (uintptr_t) (lds_byte*) fdesc;  // LDS byte offset of the half rows
                // (PRUNED as a compile-time flag: the loop body is one basic block up to the rare survivor.)  Survivors of the
                // irrelevance bound go to this thread's four LDS slots (lattice positions); the only loop-carried register they touch
                // is their count, which the straight line updates: a survivor kept in registers -- with the pattern test and the full
                // score of an evicted one inside the loop -- made the compiler copy best / second best / survivors around every trip.
                // Pattern test + full score after the scan; a fifth survivor (never on random rows) sends the query through the
                // unpruned scan instead.
                // pruned_tag: 0 = unpruned (every accepted entry is scored in full), 3 / 4 = the bound is tested on the first 96 / 128 bits
                // (two copies of the loop, chosen per query by the frame's bound: a uniform branch INSIDE one loop cost the headline 2 %).
                // 96 bits of unrelated rows differ in 48 +- 5, so a bound of up to 32 (kitti.conf / euroc.conf's first searches: 31) still
                // rejects all but ~1e-4 of them at three quarters of the instructions; a larger bound (tum.conf's 49) needs all 128 bits
                // (64 +- 6) or nearly every entry survives and the query falls through to the unpruned scan
                auto scan = [&](auto accepts, auto pruned_tag) {
                  constexpr int WORDS = decltype(pruned_tag)::value;
                  constexpr bool PRUNED = WORDS != 0;
                  const uint16_t* cs = cellstart + (r0 >> g.cell_sy) * g.cell_ncx + cx0;  // bounds of the next cell row's segment
                  const int width    = cx1 + 1 - cx0;
                  int rows_left      = (r1 >> g.cell_sy) - (r0 >> g.cell_sy) + 1;
                  int pos = 0, seg1 = 0;
                  int n_surv = 0;
                  for (;;) {
                    if (pos >= seg1) {
                      if (rows_left <= 0) {
                        break;
                      }
                      pos  = cs[0];
                      seg1 = cs[width];
                      cs += g.cell_ncx;
                      --rows_left;
                    }
                    if (pos < seg1) {
                      if (PRUNED) {
                        // (the two reads back to back behind one wait; the half row behind the last entry of a segment is read, not used)
                        au32x4 la, lb;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(la), "=&v"(lb)
                                     : "v"(fd_lds + 16u * (uint32_t) pos));
                        // (no "pos + 1 < seg1" test on the second slot: the entry behind a segment is another cell's.  If it survives the
                        // bound and lies in the pattern it is in a scanned cell too -- the cells cover the pattern's bounding box -- and is
                        // met again there: survivors are de-duplicated before they are scored.  Behind the last entry sits a sentinel no
                        // pattern accepts.)
                        const bool hit_a = (WORDS == 3 ? hamming_96(la, q0) : hamming_half(la, q0)) < prune_at;
                        const bool hit_b = (WORDS == 3 ? hamming_96(lb, q0) : hamming_half(lb, q0)) < prune_at;
                        if (hit_a) {
                          surv[n_surv & (kSurvivors - 1)] = (uint16_t) pos;  // (slots are a power of two; a wrapped write belongs to a query that is rescanned anyway)
                          ++n_surv;
                        }
                        if (hit_b) {
                          surv[n_surv & (kSurvivors - 1)] = (uint16_t) (pos + 1);
                          ++n_surv;
                        }
                      } else {
                        const unsigned long long wa = reinterpret_cast<const unsigned long long*>(db)[pos];
                        const unsigned long long wb = reinterpret_cast<const unsigned long long*>(db)[pos + 1];
                        const uint2 ea = make_uint2((uint32_t) wa, (uint32_t) (wa >> 32)), eb = make_uint2((uint32_t) wb, (uint32_t) (wb >> 32));
                        const bool in_a = accepts(ea);
                        const bool in_b = (pos + 1 < seg1) & accepts(eb);
                        if (in_a) {
                          score_at(pos);
                        }
                        if (in_b) {
                          score_at(pos + 1);
                        }
                      }
                      pos += 2;
                    }
                  }
                  return n_surv;
                };
                auto scan_pattern = [&](auto pruned_tag) {
                  return lean_circle ? scan(accepts_circle, pruned_tag) : scan(accepts_any, pruned_tag);
                };
                if (prune_at > 0) {
                  const int n_surv = (prune_at <= g.narrow_prefilter_limit) ? scan_pattern(std::integral_constant<int, 3>{}) : scan_pattern(std::integral_constant<int, 4>{});
                  if (n_surv > kSurvivors) {
                    atomicAdd(&sh.n_overflow, 1);
                    scan_pattern(std::integral_constant<int, 0>{});  // (nothing has been scored yet)
                  } else {
                    for (int i = 0; i < n_surv; ++i) {
                      const int p = (int) surv[i];
                      bool seen   = false;  // (an entry read as "one past a segment" and again in its own)
                      for (int j = 0; j < i; ++j) {
                        seen = seen | ((int) surv[j] == p);
                      }
                      if (!seen && (lean_circle ? accepts_circle(db[p]) : accepts_any(db[p]))) {
                        score_at(p);
                      }
                    }
                  }
                } else {
                  scan_pattern(std::integral_constant<int, 0>{});
                }
              }
              if (bestk != kNoneU32) {  // circle_impl.cpp:78-92 / kdtree_impl.cpp:72-78 (best only)
                cd.x = (lattice ? (uint32_t) inv[bestk & 0xffffu] : (bestk & 0xffffu)) | ((bestk >> 16) << 16);
                if (lattice && seck != kNoneU32) {
                  cd.y = (uint32_t) inv[seck & 0xffffu] | ((seck >> 16) << 16);
                }
              }
              // _addCorrespondenceCandidate (:8-37) + the per-fixed best / second-lowest response of _filterCorrespondences
              // (:57-68), order-independent: keys are (response, insertion order = (moving index, best before second best));
              // atomicMin returns the value it met, and whichever of the two is not the new minimum has lost for good, so
              // second[] ends up as the minimum response over everything but the winner without a second pass
              auto emit = [&](const uint32_t c, const uint32_t tag) {
                const uint32_t f   = c & 0xffffu;
                const uint32_t key = ((c >> 16) << 17) | ((uint32_t) m << 1) | tag;
                const uint32_t old = atomicMin(&bestkey[f], key);
                if (old != kNoneU32) {
                  atomicMin(&second[f], (old > key ? old : key) >> 17);
                }
              };
              if (cd.x != kNoneU32) {
                emit(cd.x, 0u);
              }
              if (cd.y != kNoneU32) {
                emit(cd.y, 1u);
              }
            }
           }
          }
          if (projected) {
            atomicAdd(&sh.n_projected, projected);
          }
        }
        __syncthreads();
        SUB_ACC(acc_search);
        // -- _filterCorrespondences (:41-102) in ascending fixed index; second[] is recycled to hold
        //    the output slot of accepted entries
        {
          const float ratio = g.f.maximum_distance_ratio_to_second_best;
          const float dd    = sh.dd;
          int base          = 0;
          for (int f0 = 0; f0 < nF; f0 += T) {
            const int f = f0 + tid;
            bool acc    = false;
            if (f < nF) {
              const uint32_t bk = bestkey[f];
              if (bk != kNoneU32) {
                const float resp = (float) (bk >> 17);
                const uint32_t s = second[f];
                const float fs   = s == kNoneU32 ? kFltMax : (float) s;
                // bijection (:82-99): the winner must be the moving point's own best emission
                acc = resp < dd && resp / fs < ratio && (bk & 1u) == 0u;
              }
            }
            const unsigned long long bal = __ballot(acc);
            const int pre                = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) {
              sh.wave_tot[wave] = __popcll(bal);
            }
            __syncthreads();
            int wbase = base;
            for (int w = 0; w < wave; ++w) {
              wbase += sh.wave_tot[w];
            }
            if (f < nF) {
              second[f] = acc ? (uint32_t) (wbase + pre) : kNoneU32;
            }
            for (int w = 0; w < T / 64; ++w) {
              base += sh.wave_tot[w];
            }
            __syncthreads();
          }
          if (tid == 0) {
            sh.n_filtered = base;
          }
        }
        __syncthreads();
        // -- matching ratio, reset + internal repeat, convergence latch (:215-291)
        if (tid == 0) {
          if (sh.n_projected == 0) {
            sh.flags |= PRS_WARN_NO_PROJECTION;
          }
          if (8 * sh.n_overflow > sh.n_projected) {
            sh.no_prune = 1;  // more than an eighth of the queries were scanned twice: this finder stops pruning (until its configuration changes)
          }
          const float matching_ratio = (float) sh.n_filtered / (float) (size_t) nF;
          sh.decision                = kDecisionCommit;
          if (matching_ratio < g.f.minimum_matching_ratio) {
            sh.flags |= PRS_WARN_LOW_RATIO;
            if (sh.radius < g.f.maximum_search_radius_pixels || sh.dd > g.f.minimum_descriptor_distance) {
              sh.radius = g.f.maximum_search_radius_pixels;
              sh.dd     = g.f.minimum_descriptor_distance;
              sh.flags |= PRS_WARN_RETRIED;
              if (matching_ratio == 0.0f) {
                sh.flags |= PRS_WARN_TRACK_LOST;
                se3_identity(sh.T);
                sh.it = 0;
              } else {
                ++sh.it;
              }
              sh.decision = kDecisionRetry;
            }
          }
          if (sh.decision == kDecisionCommit) {
            sh.n_corr       = sh.n_filtered;
            sh.corr_changed = 1;
            if (sh.change_norm < g.f.maximum_estimate_change_norm_for_convergence &&
                sh.it > g.f.minimum_number_of_iterations) {
              sh.converged = 1;
              if (matching_ratio > g.f.minimum_matching_ratio) {
                const unsigned long long reduced = sh.radius - g.f.search_radius_step_size_pixels;  // may wrap like size_t
                sh.radius = reduced > g.f.minimum_search_radius_pixels ? reduced : g.f.minimum_search_radius_pixels;
                const float increased = sh.dd + g.f.descriptor_distance_step_size_pixels;
                sh.dd = increased < g.f.maximum_descriptor_distance ? increased : g.f.maximum_descriptor_distance;
              }
            }
            ++sh.it;
          }
        }
        __syncthreads();
        if (sh.decision == kDecisionRetry) {
          continue;
        }
        SUB_ACC(acc_filter);
        // -- commit: correspondences->swap(filtered) (:268) (+ the operands of the factor, fused kernel only)
        for (int f = cold_copy(tid_hot); f < nF; f += T) {
          const uint32_t slot = second[f];
          if (slot != kNoneU32) {
            const uint32_t bk = bestkey[f];
            const int m       = (int) ((bk >> 1) & 0xffffu);
            prs_corr cr;
            cr.fixed_idx  = f;
            cr.moving_idx = m;
            cr.response   = (float) (bk >> 17);
            gcorr[slot]   = cr;
            if (g.mode == PRS_MODE_ALIGN) {
              cfix[slot] = gfix[f];
              cmov[slot] = gmov[m];
            }
            // (split pipeline: the Gauss-Newton kernel gathers its operand rows through the correspondence vector itself;
            // round 5 wrote them here, 32 B per correspondence and search, behind two gathers that missed the L2)
          }
        }
        __syncthreads();
        SUB_ACC(acc_commit);
        break;
      }  // finder compute()
      if (sh.error) {
        break;
      }
      // _postCompute (bruteforce_impl.cpp:231-243)
      if (tid == 0 && sh.n_corr == 0) {
        sh.flags |= PRS_WARN_NO_MATCHES;
      }
      __syncthreads();
    }
    ALIGN_ACC(acc_finder);
    if (g.mode == PRS_MODE_FINDER || split_search) {
      break;
    }

    // ==============================================================================================
    // setupFactor + errorAndJacobian + robustifier + H/b (aligner_slice_processor_projective.cpp:28-112
    // and the restated srrg2_solver arithmetic)
    // ==============================================================================================
    const int nc = sh.n_corr;
    __syncthreads();
    if (g.mode == PRS_MODE_ALIGN && nc < g.a.min_num_correspondences) {
      if (tid == 0) {
        sh.n_inl = sh.n_out = sh.n_inv = 0;
        sh.chi_in = sh.chi_tot = 0.0f;
        sh.have_cls = 0;
      }
      for (int i = tid; i < 36; i += T) {
        sh.H[i] = 0.0f;
      }
      if (tid < 6) {
        sh.b[tid] = 0.0f;
      }
      __syncthreads();
      if (g.a.stop_at_fixed_point && (sh.converged || inlier_run)) {
        // nothing can change any more; the inlier-only run (if any) needs n_inl = 0 >= min_num_inliers and would not move X either
        ran_inlier_phase = ran_inlier_phase || (inlier_run_length(g.a) > 0 && 0 >= g.a.min_num_inliers);
        break;
      }
      continue;  // slice has too few correspondences: no update this iteration
    }
    {
      const PoseRegs pose  = {sh.A[0], sh.A[1], sh.A[2], sh.A[3], sh.A[4], sh.A[5], sh.A[6], sh.A[7], sh.A[8], sh.A[9], sh.A[10], sh.A[11]};
      const float mean_dsp = sh.mean_disp;
      // a pose with a NaN / inf entry: every correspondence is invalid, all sums stay zero
      const bool pose_ok   = __all(pose_is_finite(pose));  // (uniform by construction; __all makes the branch scalar)
      if (tid == 0) {
        sh.have_cls = 1;
      }
      if (!pose_ok) {
        for (int c = tid; c < nc; c += T) {
          clsbuf[c] = 2;
        }
      }
      // the fixed-shape sum of the normal equations (include/proslam_hip.h, prs_align_result), exactly as the split
      // pipeline's Gauss-Newton kernel evaluates it: the first 128 threads own one leaf each (running sums in registers,
      // correspondences l, l + 128, ... in order), six exchange levels inside each of the two waves, the seventh through LDS
      ALIGN_MARK();
      float tot = 0.0f;
      if (tid < 128) {
        f2 acc[kPairs];
#pragma unroll
        for (int i = 0; i < kPairs; ++i) {
          acc[i] = f2{0.0f, 0.0f};
        }
        float code = 0.0f;
        if (pose_ok) {
          for (int c0 = 0; c0 < nc; c0 += 128) {
            const int c  = c0 + tid;
            const int cc = c < nc ? c : 0;
            int cls;
            factor_accumulate(g.a, pose, cfix[cc], cmov[cc], mean_dsp, c < nc, acc, code, cls, inlier_run);
            if (c < nc) {
              clsbuf[c] = (uint8_t) cls;
            }
          }
        }
        acc[kSlotCls >> 1].x = code;  // (a pose that is not finite: no inlier, no kernelised factor, everything invalid)
        tot = wave_sum_slots(acc, lane);
        if (wave == 1) {
          terms[(lane >> 1) & 31] = tot;
        }
      }
      ALIGN_ACC(acc_lin);
      __syncthreads();
      ALIGN_MARK();
      if (wave == 0) {
        const int slot = (lane >> 1) & 31;
        tot            = tot + terms[slot];
        tot            = tot + 0.0f;  // the root: a zero sum is +0
        if (slot == kSlotCls) {
          if ((lane & 1) == 0) {
            const int cc = (int) tot;  // #inliers + kClsOutUnit * #kernelised
            sh.n_inl     = cc & ((int) kClsOutUnit - 1);
            sh.n_out     = cc / (int) kClsOutUnit;
            sh.n_inv     = nc - sh.n_inl - sh.n_out;
          }
        } else if (slot != kSlotUnusedA && slot != kSlotUnusedB) {
          (&sh.H[0])[gn_slot_destination(slot, (lane & 1) != 0)] = tot;  // H (both triangles), b, chi_in, chi_tot
        }
      }
      __syncthreads();
      if (tid == 0 && pose_ok) {
        rotate_normal_equations(sh.A, sh.H, sh.b);  // camera frame -> tangent space of X (factor_accumulate)
      }
      __syncthreads();
      ALIGN_ACC(acc_sum);
    }
    if (g.mode == PRS_MODE_LINEARIZE) {
      break;
    }
    // ---- damped GN step on one lane (IterationAlgorithmGN + dense LDL^T + X <- X * exp(dx)) ---
    ALIGN_MARK();
    if (tid == 0) {
      float H[36], b[6], X[16];
#pragma unroll
      for (int i = 0; i < 36; ++i) {
        H[i] = sh.H[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        b[i] = sh.b[i];
      }
      if (g.b.prior) {
        const float* pr = g.b.prior + (size_t) frame * 42;
#pragma unroll
        for (int i = 0; i < 36; ++i) {
          H[i] += pr[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          b[i] += pr[36 + i];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        X[i] = sh.X[i];
      }
      if (g.a.enable_motion_prior) {
        add_motion_prior(g.a, g.prior_mean ? g.prior_mean + (size_t) frame * 16 : nullptr, X, H, b);
      }
      float dx6[6];
      const bool step_ok = gn_step(H, b, g.a.damping, X, g.a.damping_form == PRS_DAMPING_IDENTITY, dx6);
      // opt-in early exit (prs_aligner_params.step_norm_exit: less work than the reference)
      const float dn2 = ((((dx6[0] * dx6[0] + dx6[1] * dx6[1]) + dx6[2] * dx6[2]) + dx6[3] * dx6[3]) + dx6[4] * dx6[4]) + dx6[5] * dx6[5];
      const bool small_step = g.a.step_norm_exit > 0.0f && step_ok && dn2 < g.a.step_norm_exit * g.a.step_norm_exit;
      uint32_t changed_bits = 0;  // (bitwise, no short-circuit: sixteen compares would be sixteen branches)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        changed_bits |= __float_as_uint(X[i]) ^ __float_as_uint(sh.X[i]);
        sh.X[i] = X[i];
      }
      float a16[16];
      pose_to_camera(g.a, sh.Sinv, X, a16);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sh.A[i] = a16[i];
      }
      // fixed point: finder latched (or out of the loop: inlier-only run) + pose reproduced bit-for-bit
      // => every later iteration of this phase repeats this one
      sh.decision = ((g.a.stop_at_fixed_point && changed_bits == 0u && (sh.converged || inlier_run)) || (small_step && (sh.converged || inlier_run))) ? 1 : 0;
    }
    __syncthreads();
    ALIGN_ACC(acc_solve);
    if (sh.decision) {
      if (!inlier_run && inlier_run_length(g.a) > 0) {
        it_align = g.a.max_iterations - 1;  // the rest of the first phase repeats this iteration: go on with the inlier-only run
        continue;
      }
      ++it_align;
      break;
    }
  }

  // ---- keep_only_inlier_correspondences: the returned vector keeps the inliers of the last linearisation, in order ----
  __syncthreads();
  if (g.mode == PRS_MODE_ALIGN && !split_search && g.a.keep_only_inlier_correspondences && sh.have_cls && !sh.error) {
    const int nc = sh.n_corr;
    int base     = 0;
    for (int c0 = 0; c0 < nc; c0 += T) {
      const int c     = c0 + tid;
      const bool keep = c < nc && clsbuf[c] == 0;
      prs_corr cr;
      if (keep) {
        cr = gcorr[c];
      }
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) {
        sh.wave_tot[wave] = __popcll(bal);
      }
      __syncthreads();  // (also: every entry of this chunk has been read before any slot <= c is overwritten)
      int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      for (int w = 0; w < T / 64; ++w) {
        pos += w < wave ? sh.wave_tot[w] : 0;
        base += sh.wave_tot[w];
      }
      if (keep) {
        gcorr[pos] = cr;
      }
      __syncthreads();
    }
    if (tid == 0) {
      sh.n_corr = base;
    }
    __syncthreads();
  }

  // ---- write back --------------------------------------------------------------------------------
  __syncthreads();
  if (tid < 16) {
    if (!split_search) {
      g.b.X[(size_t) frame * 16 + tid] = sh.X[tid];
    }
    gstate->local_map_in_sensor[tid]          = sh.T[tid];
    gstate->local_map_in_sensor_previous[tid] = sh.Tprev[tid];
  }
  for (int i = tid; i < 36; i += T) {
    gres->H[i] = sh.H[i];
  }
  if (tid < 6) {
    gres->b[tid] = sh.b[tid];
  }
  if (tid == 0) {
    gstate->search_radius_pixels = sh.radius;
    gstate->current_iteration    = sh.it;
    gstate->descriptor_distance  = sh.dd;
    gstate->database_leaf_range  = sh.kd_range;
    gstate->has_converged        = sh.converged;
    gstate->config_changed       = sh.config_changed;
    gstate->num_recomputes       = sh.num_recomputes;
    gstate->reserved             = sh.no_prune;
    g.b.n_corr[frame]            = sh.n_corr;
    gres->chi_inliers            = sh.chi_in;
    gres->chi_total              = sh.chi_tot;
    gres->mean_disparity         = sh.mean_disp;
    gres->num_inliers            = sh.n_inl;
    gres->num_outliers           = sh.n_out;
    gres->num_invalid            = sh.n_inv;
    gres->num_correspondences    = sh.n_corr;
    gres->status                 = sh.n_inl >= g.a.min_num_inliers ? 1 : 0;
    gres->iterations             = g.mode == PRS_MODE_ALIGN ? (ran_inlier_phase ? max_it : g.a.max_iterations) : 1;
    gres->iterations_executed    = executed;
    gres->warnings               = sh.error ? sh.error : sh.flags;
    if (split_search) {
      ctl->flags |= sh.flags;
      ctl->need_search = 0;
      gres->warnings   = sh.error ? sh.error : ctl->flags;
    }
    if (g.stamps) {
      unsigned long long* st = g.stamps + (size_t) frame * 16;
      st[0] = 0;
      for (int i = 0; i < 9; ++i) {  // cumulative: finder, + linearise, + sums, + solve, + lattice, + search, + staging, + filter, + commit
        st[i + 1] = st[i] + sh.stamp_acc[i];
      }
    }
  }
}

// ---- split pipeline, GN half -------------------------------------------------------------------------
// One workgroup per frame runs aligner iterations (linearize + damped GN step, with the
// finder's "nothing new" bookkeeping in between) until the finder needs a projective search again or
// max_iterations is reached.  The per-correspondence operands (gathered through the committed correspondence vector) sit in
// registers (<= 8 per thread), so a workgroup needs only the 29 x THREADS term matrix in LDS and many
// frames share a CU.  Arithmetic and summation order are those of the fused kernel.
// Instantiated for 128 threads x {4, 6, 7, 8} correspondences per thread (the default) and 256 x {2, 3, 4}.
// The row stride of the term matrix is THREADS + 4 floats, so that the 29 summing lanes (one row each,
// 16-B reads) start 16 B apart in the bank space instead of all on the same four banks.
constexpr int kGnThreads = 128;  // two waves per frame
constexpr int kGnSlots   = 8;    // the split pipeline serves max_fixed <= 1024

struct GnShared {
  float X[16], T[16], Tprev[16];
  // destinations of the 32 summed slots, consecutive: H (both triangles), b, chi (inliers), chi (all), the three class counts
  float H[36], b[6];
  float chi_in, chi_tot;
  float fcnt[3];   // [0]: #inliers + kClsOutUnit * #kernelised of the last linearisation (an exact integer); [1], [2]: not used
  float pad0;
  float A[16];     // points -> camera: X, or sensor_in_robot^-1 * X
  float Sinv[16];
  float wsum[32];  // the 32 slot sums of the wave that does not solve
  unsigned long long it;
  int converged, need_search, n_inl, n_out, n_inv, stop, flags;
  int pose_ok;  // every entry of A is finite (kept by whoever writes X)
  int wave_tot[4];
};
static_assert(offsetof(GnShared, b) == offsetof(GnShared, H) + 36 * sizeof(float) && offsetof(GnShared, chi_in) == offsetof(GnShared, b) + 6 * sizeof(float) &&
                offsetof(GnShared, chi_tot) == offsetof(GnShared, chi_in) + sizeof(float) && offsetof(GnShared, fcnt) == offsetof(GnShared, chi_tot) + sizeof(float),
              "the summed slots are written through GnShared::H");

// the aligner loop after iteration `it_align` has been executed: the next iteration's index, and whether the frame is
// finished (MultiAligner3DQR stand-in: max_iterations, then the inlier-only run if there are enough inliers)
__device__ __forceinline__ bool gn_advance(const prs_aligner_params& a, const int extra, int& it_align, bool fixed_point, const int n_inl) {
  const bool was_inlier_run = it_align >= a.max_iterations;
  ++it_align;
  if (fixed_point && !was_inlier_run && extra > 0) {
    it_align    = a.max_iterations;  // the rest of the first phase repeats this iteration
    fixed_point = false;
  }
  if (fixed_point) {
    return true;
  }
  if (it_align == a.max_iterations && extra > 0) {
    return n_inl < a.min_num_inliers;  // not enough inliers for an inlier-only run
  }
  return it_align >= a.max_iterations + extra;
}

// The single-wave phase of a Gauss-Newton iteration once the 32 summed slots of the linearisation are in sh.H / b / chi / fcnt:
// camera frame -> tangent space of X, class counts, (H + damping diag(H)) dx = -b, X <- X * exp(dx); `stid` = lane of the solving wave.
// PLAIN (what align_batch_launch found in the parameters): 1 = no additive prior, no motion-model prior, no sensor offset, no
// inlier-only runs; 2 = the same with the motion-model prior on (the aligner of kitti.conf's tracker: its AlignerSliceMotionModel3D);
// 0 = anything.  The branches a parameter set does not take, and their registers, are not part of its instantiation.
template <bool SHIPPED_FORMS = false, int PLAIN = 0>
__device__ __forceinline__ void gn_solve_wave(const AlignArgs& g, GnShared& sh, const int frame, const int nc, const int stid, const bool inlier_run) {
  const int lane = stid;
  if (sh.pose_ok) {  // (a pose that is not finite: all sums are zero and stay zero)
    // camera frame -> tangent space of X: H <- Rt^T H Rt, b <- Rt^T b, Rt = blockdiag(R, R) (prs_se3.h, rotate_normal_equations:
    // the same expressions); lane r < 6 owns row r, the lower triangle is the system and is mirrored
    const int rrow = lane < 6 ? lane : 5;
    const int blk  = rrow >= 3 ? 1 : 0;
    const int ci   = rrow - 3 * blk;
    const float* Yb = &sh.H[18 * blk];
    const float Ri0 = sh.A[ci], Ri1 = sh.A[4 + ci], Ri2 = sh.A[8 + ci];
    float o[6];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = fmaf(Ri2, Yb[12 + 3 * cb + c], fmaf(Ri1, Yb[6 + 3 * cb + c], Ri0 * Yb[3 * cb + c]));
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        o[3 * cb + j] = fmaf(v[2], sh.A[8 + j], fmaf(v[1], sh.A[4 + j], v[0] * sh.A[j]));
      }
    }
    const float bb = fmaf(Ri2, sh.b[3 * blk + 2], fmaf(Ri1, sh.b[3 * blk + 1], Ri0 * sh.b[3 * blk]));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 6) {
      // (the whole row: the solve reads the lower triangle only; the upper one is mirrored when the kernel stores H)
      float2* hw = reinterpret_cast<float2*>(&sh.H[6 * rrow]);
      hw[0]      = make_float2(o[0], o[1]);
      hw[1]      = make_float2(o[2], o[3]);
      hw[2]      = make_float2(o[4], o[5]);
      sh.b[rrow] = bb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (stid == 0) {
    const int cc = (int) sh.fcnt[0];  // #inliers + kClsOutUnit * #kernelised
    sh.n_inl     = cc & ((int) kClsOutUnit - 1);
    sh.n_out     = cc / (int) kClsOutUnit;
    sh.n_inv     = nc - sh.n_inl - sh.n_out;
  }
  // ---- (H + damping diag(H)) dx = -b, X <- X * exp(dx): every lane evaluates the (uniform) 6 x 6 solve of prs_se3.h, lane r < 3
  // then owns row r of the pose.  Same operations as gn_step on one lane, so the pose is bit-identical to that evaluation.
  {
    const int prow  = lane < 3 ? lane : 2;
    const float4 xr = *reinterpret_cast<const float4*>(&sh.X[4 * prow]);
    float H[36], b[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      H[i] = sh.H[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      b[i] = sh.b[i];
    }
    if (PLAIN == 0 && g.b.prior) {
      const float* pr = g.b.prior + (size_t) frame * 42;
#pragma unroll
      for (int i = 0; i < 36; ++i) {
        H[i] += pr[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        b[i] += pr[36 + i];
      }
    }
    if (PLAIN == 2 || (PLAIN == 0 && g.a.enable_motion_prior)) {
      float X[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        X[i] = sh.X[i];
      }
      add_motion_prior(g.a, g.prior_mean ? g.prior_mean + (size_t) frame * 16 : nullptr, X, H, b);
    }
    auto bcast = [](const float v, const int l) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    float dx[6];
    const bool ok = ldlt_solve6(H, b, g.a.damping, dx, !SHIPPED_FORMS && g.a.damping_form == PRS_DAMPING_IDENTITY);
    // opt-in early exit (prs_aligner_params.step_norm_exit: less work than the reference): the step is small and the finder has latched
    bool small_step = false;
    if (g.a.step_norm_exit > 0.0f) {  // (uniform)
      const float n2 = ((((dx[0] * dx[0] + dx[1] * dx[1]) + dx[2] * dx[2]) + dx[3] * dx[3]) + dx[4] * dx[4]) + dx[5] * dx[5];
      small_step     = ok && n2 < g.a.step_norm_exit * g.a.step_norm_exit;
    }
    float D[16];
    tnq2t(dx, D);
    float4 xn;
    xn.x = (xr.x * D[0] + xr.y * D[4]) + xr.z * D[8];
    xn.y = (xr.x * D[1] + xr.y * D[5]) + xr.z * D[9];
    xn.z = (xr.x * D[2] + xr.y * D[6]) + xr.z * D[10];
    xn.w = ((xr.x * D[3] + xr.y * D[7]) + xr.z * D[11]) + xr.w;
    xn.x = ok ? xn.x : xr.x;  // (component by component: a select between the two structs goes through scratch)
    xn.y = ok ? xn.y : xr.y;
    xn.z = ok ? xn.z : xr.z;
    xn.w = ok ? xn.w : xr.w;
    const uint32_t changed = (__float_as_uint(xn.x) ^ __float_as_uint(xr.x)) | (__float_as_uint(xn.y) ^ __float_as_uint(xr.y)) |
                             (__float_as_uint(xn.z) ^ __float_as_uint(xr.z)) | (__float_as_uint(xn.w) ^ __float_as_uint(xr.w));
    const bool any_changed = (__ballot(changed != 0u) & 7ull) != 0ull;
    // points -> camera: X, or sensor_in_robot^-1 * X
    float4 an = xn;
    if (PLAIN == 0 && g.a.with_sensor) {
      const float4 sr = *reinterpret_cast<const float4*>(&sh.Sinv[4 * prow]);
      const float4 x0 = {bcast(xn.x, 0), bcast(xn.y, 0), bcast(xn.z, 0), bcast(xn.w, 0)};
      const float4 x1 = {bcast(xn.x, 1), bcast(xn.y, 1), bcast(xn.z, 1), bcast(xn.w, 1)};
      const float4 x2 = {bcast(xn.x, 2), bcast(xn.y, 2), bcast(xn.z, 2), bcast(xn.w, 2)};
      an.x = (sr.x * x0.x + sr.y * x1.x) + sr.z * x2.x;
      an.y = (sr.x * x0.y + sr.y * x1.y) + sr.z * x2.y;
      an.z = (sr.x * x0.z + sr.y * x1.z) + sr.z * x2.z;
      an.w = ((sr.x * x0.w + sr.y * x1.w) + sr.z * x2.w) + sr.w;
    }
    const float rs = (an.x + an.y) + (an.z + an.w);
    const float fs = (bcast(rs, 0) + bcast(rs, 1)) + bcast(rs, 2);  // pose_is_finite's sum
    if (lane < 3) {
      *reinterpret_cast<float4*>(&sh.X[4 * lane]) = xn;
      *reinterpret_cast<float4*>(&sh.A[4 * lane]) = an;
    }
    if (stid == 0) {
      sh.stop    = ((g.a.stop_at_fixed_point && !any_changed && (sh.converged || inlier_run)) || (small_step && (sh.converged || inlier_run))) ? 1 : 0;
      sh.pose_ok = (fs - fs) == 0.0f ? 1 : 0;
    }
  }
}

// What the solving wave does once an iteration's pose is final (uniform control flow inside the wave): decide whether the
// loop goes on and, if the next iteration belongs to the first phase, perform its finder.setLocalMapInSensor(X);
// finder.compute() as long as that needs no projective search (CF/correspondence_finder_projective_base_impl.cpp:138-178)
__device__ __forceinline__ void gn_finder_bookkeeping(const AlignArgs& g, GnShared& sh, const int stid, const int nc, const int it_align, const int extra,
                                                      const bool fixed_point) {
    int next        = it_align;
    const bool over = gn_advance(g.a, extra, next, fixed_point, sh.n_inl);
    if (!over && next < g.a.max_iterations) {
      if (stid < 16) {
        sh.T[stid] = sh.A[stid];
      }
      if (!sh.converged) {
        const unsigned long long k  = g.f.number_of_solver_iterations_per_projection;
        const unsigned long long it = sh.it;
        if (k == 0 || it % k == 0 || it == 1) {
          if (stid == 0) {
            sh.need_search = 1;
          }
        } else {
          if (stid < 16) {
            sh.Tprev[stid] = sh.A[stid];
          }
          if (stid == 0) {
            sh.it = it + 1;
            if (nc == 0) {
              sh.flags |= PRS_WARN_NO_MATCHES;  // _postCompute (bruteforce_impl.cpp:237-242)
            }
          }
        }
      } else if (stid == 0 && nc == 0) {
        sh.flags |= PRS_WARN_NO_MATCHES;
      }
    }
}

// SLOTS = correspondences per thread the instantiation can hold (ceil(max_fixed / 128)).  The operand rows of the first
// kGnLdsSlots * 128 correspondences (measurement + inverse-depth weight, moving point + information scale: 32 B each) are
// parked in LDS for the whole launch (16 KB per frame; registers are needed for the 32 running sums), later ones (rare: more
// than 512 correspondences) are fetched from global memory at every iteration.  KEEP_CLS: the factor classes are remembered
// (keep_only_inlier_correspondences).
constexpr int kGnLdsSlots = 4;
// ... plus the rows of ONE more wave (64 correspondences, 2 KB): a frame with a few correspondences more than 512 (the TUM-shaped
// workload: 524) otherwise waits for a global-memory round trip in every one of its 200 iterations (round 4: 9.0 -> 4.6 ms per
// 4608 frames).  8 frames x (16 + 2 + 0.7) KB = 150 KB of the CU's 160 KB.
// LDS_SLOTS / WAVES: the instantiation of the headline configuration (stereo factor, up to 1024 fixed points) parks 448 rows and is
// built for five waves per SIMD (96 VGPRs, ten frames per CU: 15 KB each): -3.5 % on the headline, -5 % on real KITTI frames; the
// generic ones lose with it (spills, streamed rows) and keep four waves per SIMD and 576 parked rows
constexpr size_t gn_lds_bytes(const int lds_slots) {
  return ((sizeof(GnShared) + 15) / 16) * 16 + (size_t) (lds_slots * 128 + 64) * 2 * sizeof(float4);  // shared state + parked operand rows
}
template <int SLOTS, int DIM, bool KEEP_CLS, int LDS_SLOTS = kGnLdsSlots, int WAVES = 4, int PLAIN = 0>
__global__ __launch_bounds__(128, WAVES) void gn_kernel(const AlignArgs g) {
  constexpr int THREADS = 128;
  constexpr int kGnLdsSlots = LDS_SLOTS;  // (shadow the defaults)
  constexpr int kGnLdsRows  = LDS_SLOTS * 128 + 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid   = threadIdx.x;
  const int lane  = tid & 63;
  const int wave  = tid >> 6;
  const int frame = blockIdx.x;
  // the single-wave phase (cross-wave add, 6x6 solve, finder bookkeeping) runs on a different wave -- hence a different
  // SIMD -- from frame to frame, so that the workgroups sharing a CU do not queue it on one SIMD
  const int solver = frame & 1;
  const int stid   = tid - 64 * solver;  // 0..63 on the solving wave
  FrameCtl* ctl    = g.ctl + frame;
  // Round 6: the operand rows are gathered through the correspondence vector the search kernel committed.  The index pairs of this
  // thread's rows are requested before anything else -- together with the control word and the count, which say whether and how
  // many of them mean something (every slot below fixed_stride is addressable) -- so that the gather is the second trip to memory
  // of a launch, as the coalesced operand rows of round 5 were.
  constexpr int LS_IDX = (SLOTS < LDS_SLOTS ? SLOTS : LDS_SLOTS) + 1;
  const size_t fbase                    = (size_t) frame * (size_t) g.b.fixed_stride;
  const prs_corr* __restrict__ gcorr_in = g.b.corr + fbase;
  int fi[LS_IDX], mi[LS_IDX];
#pragma unroll
  for (int k = 0; k < LS_IDX; ++k) {
    const int c  = k * 128 + tid;
    const int cc = c < g.b.fixed_stride ? c : 0;
    fi[k]        = gcorr_in[cc].fixed_idx;
    mi[k]        = gcorr_in[cc].moving_idx;
  }
  if (ctl->done || ctl->need_search) {
    return;  // finished, or waiting for the search kernel (block-uniform)
  }
  GnShared& sh             = *reinterpret_cast<GnShared*>(smem);
  prs_pcf_state* gstate    = g.b.state + frame;
  prs_align_result* gres   = g.b.result + frame;
  const float4* gops       = g.ops + (size_t) frame * (size_t) g.max_fixed * 2;
  const int nc             = g.b.n_corr[frame];
  const float mean_dsp     = gres->mean_disparity;

  if (tid < 16) {
    sh.X[tid]     = g.b.X[(size_t) frame * 16 + tid];
    sh.T[tid]     = gstate->local_map_in_sensor[tid];
    sh.Tprev[tid] = gstate->local_map_in_sensor_previous[tid];
  }
  if (tid == 0) {
    sh.it          = gstate->current_iteration;
    sh.converged   = gstate->has_converged;
    sh.need_search = 0;
    sh.flags       = 0;
    sh.n_inl       = ctl->n_inl;
    sh.n_out       = ctl->n_out;
    sh.n_inv       = ctl->n_inv;
    sh.chi_in      = gres->chi_inliers;
    sh.chi_tot     = gres->chi_total;
    sh.stop        = 0;
    const float* gx = g.b.X + (size_t) frame * 16;
    float x[16], a16[16], si[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] = gx[i];
    }
    se3_inverse(g.a.sensor_in_robot, si);
    pose_to_camera(g.a, si, x, a16);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sh.Sinv[i] = si[i];
      sh.A[i]    = a16[i];
    }
    const PoseRegs x0 = {a16[0], a16[1], a16[2], a16[3], a16[4], a16[5], a16[6], a16[7], a16[8], a16[9], a16[10], a16[11]};
    sh.pose_ok        = pose_is_finite(x0) ? 1 : 0;
  }
  for (int i = tid; i < 36; i += THREADS) {
    sh.H[i] = gres->H[i];
  }
  if (tid < 6) {
    sh.b[tid] = gres->b[tid];
  }
  // operand rows of this thread's first correspondences: parked in LDS ([slot][thread]: consecutive lanes, consecutive 16 B)
  constexpr int LS = SLOTS < kGnLdsSlots ? SLOTS : kGnLdsSlots;
  float4* lz = reinterpret_cast<float4*>(smem + ((sizeof(GnShared) + 15) / 16) * 16);
  constexpr int LROWS = SLOTS <= kGnLdsSlots ? SLOTS * THREADS : kGnLdsRows;  // rows parked in LDS
  float4* lp = lz + LROWS;
  // Round 6: the rows are GATHERED here through the correspondence vector the search kernel committed (fixed measurement of
  // fixed_idx, moving point + information scale of moving_idx): the search kernel no longer writes 32 B per correspondence that
  // this kernel read back, and its own two gathers are gone with them.
  const float4* __restrict__ gfix = reinterpret_cast<const float4*>(g.b.fixed) + fbase;
  const float4* __restrict__ gmov = reinterpret_cast<const float4*>(g.b.moving) + (size_t) frame * (size_t) g.b.moving_stride;
  if (nc > 0) {
    static_assert(LS_IDX == LS + 1, "index pairs of the parked rows");
    float4 zr[LS + 1], pr[LS + 1];
#pragma unroll
    for (int k = 0; k < LS + 1; ++k) {
      const bool live = k * THREADS + tid < nc;  // (rows past the end read point 0 of both clouds: what their slots hold is not an index)
      zr[k]           = gfix[live ? fi[k] : 0];
      pr[k]           = gmov[live ? mi[k] : 0];
    }
#pragma unroll
    for (int k = 0; k < LS + 1; ++k) {
      const int c = k * THREADS + tid;
      if (k * THREADS < nc && c < LROWS) {
        float4 z = c < nc ? zr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        z.w      = parked_translation_weight(g.a, z, mean_dsp, DIM != 0 ? 0 : g.a.translation_weight_form);  // (the factors read x, y, z of the measurement only)
        lz[c]    = z;
        lp[c]    = c < nc ? pr[k] : make_float4(0.f, 0.f, 1.f, 1.f);
      }
    }
  }
  // rows beyond the parked ones are streamed from global memory at every iteration: gathered once per launch into this frame's
  // operand rows, the translation weight in the fourth component of the measurement (no factor reads that component; the same
  // thread reads the row back)
  if (SLOTS > kGnLdsSlots) {
    float4* wops = g.ops + (size_t) frame * (size_t) g.max_fixed * 2;
#pragma unroll
    for (int k = LS; k < SLOTS; ++k) {
      const int c = k * THREADS + tid;  // (the thread that reads row c back in pass k)
      if (c >= LROWS && c < nc) {
        const int fi    = gcorr_in[c].fixed_idx, mi = gcorr_in[c].moving_idx;
        float4 z        = gfix[fi];
        z.w             = parked_translation_weight(g.a, z, mean_dsp, DIM != 0 ? 0 : g.a.translation_weight_form);
        wops[2 * c]     = z;
        wops[2 * c + 1] = gmov[mi];
      }
    }
  }
  // the summed slot this lane ends up with, and where it goes
  const int my_slot = (lane >> 1) & 31;
  const int sum_dst = gn_slot_destination(my_slot, (lane & 1) != 0);
  __syncthreads();

  int it_align = ctl->it_align;
  int executed = ctl->executed;
  bool done    = false;
  const int extra = PLAIN != 0 ? 0 : inlier_run_length(g.a);  // iterations of the inlier-only run after the max_iterations loop
  uint32_t cls_bits = 0;                     // factor classes of this thread's correspondences in the last linearisation (2 bits each)
  bool have_cls     = false;                 // (block-uniform) the last executed iteration linearised


  while (true) {
    ++executed;
    if (nc < g.a.min_num_correspondences) {
      // slice has too few correspondences: no update this iteration
      if (tid == 0) {
        sh.n_inl = sh.n_out = sh.n_inv = 0;
        sh.chi_in = sh.chi_tot = 0.0f;
        sh.stop = (g.a.stop_at_fixed_point && (sh.converged || it_align >= g.a.max_iterations)) ? 1 : 0;
      }
      for (int i = tid; i < 36; i += THREADS) {
        sh.H[i] = 0.0f;
      }
      if (tid < 6) {
        sh.b[tid] = 0.0f;
      }
      have_cls = false;
      __syncthreads();
      if (wave == solver) {
        gn_finder_bookkeeping(g, sh, stid, nc, it_align, extra, sh.stop != 0);
      }
      __syncthreads();
    } else {
      const bool inlier_run = PLAIN == 0 && it_align >= g.a.max_iterations;  // (PLAIN: no inlier-only runs, so never true)
      have_cls              = true;
      cls_bits              = 0;
      const PoseRegs pose = {sh.A[0], sh.A[1], sh.A[2], sh.A[3], sh.A[4], sh.A[5], sh.A[6], sh.A[7], sh.A[8], sh.A[9], sh.A[10], sh.A[11]};
      // a pose with a NaN / inf entry: every correspondence is invalid, all sums stay zero
      const bool pose_ok  = __builtin_amdgcn_readfirstlane(sh.pose_ok) != 0;
      f2 acc[kPairs];
#pragma unroll
      for (int i = 0; i < kPairs; ++i) {
        acc[i] = f2{0.0f, 0.0f};
      }
      float code = 0.0f;  // class counts of this lane's correspondences (factor_accumulate)
      if (pose_ok) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
          const int c0 = k * THREADS;
          // a wave whose 64 lanes are all past the end skips the pass (its terms would all be +-0)
          if (k == 0 || c0 + 64 * wave < nc) {
            const int c = c0 + tid;
            float4 z, p;
            if (k < LS || (k == LS && c0 + 64 * wave < LROWS)) {  // (wave-uniform)
              z = lz[c];
              p = lp[c];
            } else {
              // (an opaque copy of the row number: the addresses of the streamed rows are formed here, not kept -- spilled -- across
              // the iteration loop of the frames that never stream a row)
              const int cs = cold_copy(c);
              z   = c < nc ? gops[2 * cs] : make_float4(0.f, 0.f, 0.f, 1.f);
              p   = c < nc ? gops[2 * cs + 1] : make_float4(0.f, 0.f, 1.f, 1.f);
            }
            int cls;
            factor_accumulate<DIM, true>(g.a, pose, z, p, mean_dsp, c < nc, acc, code, cls, inlier_run);
            if (KEEP_CLS) {
              cls_bits |= (uint32_t) (cls & 3) << (2 * k);
            }
          }
        }
      } else if (KEEP_CLS) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
          if (k * THREADS + tid < nc) {
            cls_bits |= 2u << (2 * k);  // a pose that is not finite: everything is invalid
          }
        }
      }
      acc[kSlotCls >> 1].x = code;
      // levels 32, 16, 8, 7, 2, 1 of the fixed-shape sum inside the wave; level 64 across the two waves through LDS
      float tot = wave_sum_slots(acc, lane);
      if (wave != solver) {
        sh.wsum[my_slot] = tot;
      }
      __syncthreads();
      if (wave == solver) {
        tot = tot + sh.wsum[my_slot];
        tot = tot + 0.0f;  // the root: a zero sum is +0
        (&sh.H[0])[sum_dst] = tot;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        gn_solve_wave<DIM != 0, PLAIN>(g, sh, frame, nc, stid, inlier_run);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        gn_finder_bookkeeping(g, sh, stid, nc, it_align, extra, sh.stop != 0);
      }
      __syncthreads();
    }
    if (gn_advance(g.a, extra, it_align, sh.stop != 0, sh.n_inl)) {
      done = true;
      break;
    }
    if (it_align < g.a.max_iterations && sh.need_search) {
      break;
    }
  }

  // ---- keep_only_inlier_correspondences: the returned vector keeps the inliers of the last linearisation, in order ----
  int nc_out = nc;
  __syncthreads();
  if (KEEP_CLS && done && g.a.keep_only_inlier_correspondences && have_cls) {
    prs_corr* __restrict__ gcorr = g.b.corr + (size_t) frame * (size_t) g.b.fixed_stride;
    int base                     = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c0 = k * THREADS;
      if (c0 < nc) {
        const int c     = c0 + tid;
        const bool keep = c < nc && ((cls_bits >> (2 * k)) & 3u) == 0u;
        prs_corr cr;
        if (keep) {
          cr = gcorr[c];
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) {
          sh.wave_tot[wave] = __popcll(bal);
        }
        __syncthreads();  // (every entry of this chunk has been read before any slot <= c is overwritten)
        int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) {
          pos += w < wave ? sh.wave_tot[w] : 0;
          base += sh.wave_tot[w];
        }
        if (keep) {
          gcorr[pos] = cr;
        }
        __syncthreads();
      }
    }
    nc_out = base;
  }

  __syncthreads();
  if (tid < 16) {
    g.b.X[(size_t) frame * 16 + tid]           = sh.X[tid];
    gstate->local_map_in_sensor[tid]          = sh.T[tid];
    gstate->local_map_in_sensor_previous[tid] = sh.Tprev[tid];
  }
  for (int i = tid; i < 36; i += THREADS) {
    const int r = i / 6, c = i - 6 * r;
    gres->H[i]  = sh.H[r >= c ? i : 6 * c + r];  // the lower triangle is the system (gn_solve_wave)
  }
  if (tid < 6) {
    gres->b[tid] = sh.b[tid];
  }
  if (tid == 0) {
    gstate->current_iteration = sh.it;
    gres->chi_inliers         = sh.chi_in;
    gres->chi_total           = sh.chi_tot;
    gres->num_inliers         = sh.n_inl;
    gres->num_outliers        = sh.n_out;
    gres->num_invalid         = sh.n_inv;
    gres->num_correspondences = nc_out;
    if (nc_out != nc) {
      g.b.n_corr[frame] = nc_out;
    }
    gres->status              = sh.n_inl >= g.a.min_num_inliers ? 1 : 0;
    gres->iterations          = it_align > g.a.max_iterations ? g.a.max_iterations + extra : g.a.max_iterations;
    gres->iterations_executed = executed;
    ctl->flags |= sh.flags;
    gres->warnings   = ctl->flags;
    ctl->it_align    = it_align;
    ctl->executed    = executed;
    ctl->n_inl       = sh.n_inl;
    ctl->n_out       = sh.n_out;
    ctl->n_inv       = sh.n_inv;
    ctl->need_search = done ? 0 : 1;
    ctl->done        = done ? 1 : 0;
    if (!done) {
      atomicAdd(g.pending, 1);
    }
  }
}

__global__ void split_init_kernel(FrameCtl* ctl, prs_align_result* res, int* pending, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) {
    FrameCtl c;
    c.it_align    = 0;
    c.need_search = 1;
    c.done        = 0;
    c.executed    = 0;
    c.flags       = 0;
    c.n_inl = c.n_out = c.n_inv = 0;
    c.db_ready                  = 0;
    ctl[i]                      = c;
    prs_align_result r;
    for (int k = 0; k < 36; ++k) {
      r.H[k] = 0.0f;
    }
    for (int k = 0; k < 6; ++k) {
      r.b[k] = 0.0f;
    }
    r.chi_inliers = r.chi_total = r.mean_disparity = 0.0f;
    r.num_inliers = r.num_outliers = r.num_invalid = r.num_correspondences = 0;
    r.status = r.iterations = r.iterations_executed = r.warnings = 0;
    res[i]                                                        = r;
  }
  if (i == 0) {
    *pending = 0;
  }
}

__global__ void gn_step_kernel(const float* H, const float* b, float damping, int identity_damping, float* X, int* ok) {
  float h[36], bb[6], x[16];
  for (int i = 0; i < 36; ++i) {
    h[i] = H[i];
  }
  for (int i = 0; i < 6; ++i) {
    bb[i] = b[i];
  }
  for (int i = 0; i < 16; ++i) {
    x[i] = X[i];
  }
  const bool r = gn_step(h, bb, damping, x, identity_damping != 0);
  for (int i = 0; i < 16; ++i) {
    X[i] = x[i];
  }
  *ok = r ? 1 : 0;
}

static inline uint32_t align_up16(uint32_t v) {
  return (v + 15u) / 16u * 16u;
}

// an enqueued batch of the split pipeline (kept in the context between align_batch_launch and align_batch_finish)
struct SplitJob {
  bool active = false;
  // The job OWNS the buffers its launches read and write between enqueue and finish (frame control + pending counter, operand
  // rows, lattice images): the context's shared scratch slots belong to whatever call runs next (scene clipper, extractor,
  // brute-force matcher, merger), and a caller may run any of them between the two halves.  Grow-only; freed with the context.
  void* own[3]        = {nullptr, nullptr, nullptr};
  size_t own_bytes[3] = {0, 0, 0};
  hipStream_t stream  = nullptr;  // the stream the rounds run on: where they were enqueued, or where the captured graph was last replayed (finish synchronises this one)
  hipStream_t capture_stream = nullptr;  // the stream align_batch_launch enqueued / captured on (what a rearm without a stream goes back to)
  ~SplitJob() {
    for (void* p : own) {
      if (p) {
        (void) hipFree(p);
      }
    }
  }
  void* buffer(hipStream_t s, int i, size_t bytes) {
    if (bytes <= own_bytes[i]) {
      return own[i];
    }
    if (own[i]) {
      (void) hipStreamSynchronize(s);
      (void) hipFree(own[i]);
      own[i]       = nullptr;
      own_bytes[i] = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&own[i], want) != hipSuccess) {
      own[i] = nullptr;
      return nullptr;
    }
    own_bytes[i] = want;
    return own[i];
  }
  AlignArgs g, gs;                      // Gauss-Newton / search kernel arguments
  void (*search)(AlignArgs) = nullptr;  // kernel instantiations of this batch
  void (*gn)(AlignArgs)     = nullptr;
  size_t lds_search = 0, lds_gn = 0;
  int total = 0, limit = 0;  // rounds launched so far / upper bound
  int enqueued = 0;          // rounds align_batch_launch enqueued (what a captured graph replays)
  int ev_used = 0;
  int rounds_timed = 0;
  bool stamps = false;
};
static int split_rounds(prs_context* ctx, SplitJob* job, int rounds);

// mode PRS_MODE_ALIGN with the split pipeline only ENQUEUES `rounds` rounds (0 = the nominal five); align_batch_finish completes it
int align_batch_launch(prs_context* ctx, const prs_pcf_params* finder, const prs_aligner_params* aligner, const prs_align_batch* batch, int mode, int rounds) {
  if (ctx->align_job && static_cast<SplitJob*>(ctx->align_job)->active) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_*: the previous batch of this context has not been finished (prs_align_batch_finish)");
  }
  if (!finder || !aligner || !batch || !batch->fixed || !batch->fixed_desc || !batch->n_fixed || !batch->moving ||
      !batch->moving_desc || !batch->n_moving || !batch->state || !batch->X || !batch->corr || !batch->n_corr || !batch->result) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_align_batch_run: fixed, moving, correspondences, state or result not set");
  }
  if (mode < PRS_MODE_ALIGN || mode > PRS_MODE_LINEARIZE) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: unknown mode");
  }
  if (batch->batch <= 0) {
    return PRS_OK;
  }
  if (batch->fixed_stride <= 0 || batch->fixed_stride > 32767 || batch->moving_stride <= 0 || batch->moving_stride > 65535) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: fixed_stride must be in [1,32767], moving_stride in [1,65535]");
  }
  if (finder->projector.canvas_rows <= 0 || finder->projector.canvas_rows > 8192 || finder->projector.canvas_cols <= 0 ||
      finder->projector.canvas_cols > 32767) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: projector canvas must be within 32767 x 8192");
  }
  if (finder->maximum_search_radius_pixels > 16383 || finder->search_type < PRS_SEARCH_KDTREE || finder->search_type > PRS_SEARCH_RHOMBUS) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: search radius > 16383 px or unknown search type");
  }
  if (aligner->factor_type != PRS_FACTOR_MONO && aligner->factor_type != PRS_FACTOR_DEPTH && aligner->factor_type != PRS_FACTOR_STEREO) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: factor_type must be 2, 3 or 4");
  }
  if ((aligner->kernel_weight_form != PRS_KERNEL_WEIGHT_INV_CHI && aligner->kernel_weight_form != PRS_KERNEL_WEIGHT_TAU_OVER_CHI) ||
      (aligner->damping_form != PRS_DAMPING_DIAG && aligner->damping_form != PRS_DAMPING_IDENTITY) ||
      (aligner->translation_weight_form != PRS_TRANSLATION_WEIGHT_OFFSET && aligner->translation_weight_form != PRS_TRANSLATION_WEIGHT_CLAMP)) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: unknown kernel_weight_form / damping_form / translation_weight_form");
  }
  AlignArgs g;
  g.f          = *finder;
  g.a          = *aligner;
  g.b          = *batch;
  g.mode       = mode;
  g.no_prefilter = ctx->no_prefilter ? 1 : 0;
  g.narrow_prefilter_limit = ctx->prefilter_96_limit;
  g.prior_mean = batch->prior_mean;
  g.rows_table = finder->projector.canvas_rows;
  const int max_fixed = batch->max_fixed > 0 && batch->max_fixed < batch->fixed_stride ? batch->max_fixed : batch->fixed_stride;
  g.max_fixed       = max_fixed;
  const uint32_t nf = (uint32_t) max_fixed;
  const uint32_t R  = (uint32_t) g.rows_table;
  uint32_t lut_cap  = 2u * (uint32_t) finder->maximum_search_radius_pixels + 1u;
  if (lut_cap > 2048u) {
    lut_cap = 2048u;
  }
  g.lut_cap = (int) lut_cap;
  // 2-D cell grid over the canvas: 16 x 16 px cells, coarsened until there are at most 2048 of them
  int sy = 4, sx = 4;
  auto cells_of = [&](int shift_y, int shift_x, int& ncy, int& ncx) {
    ncy = ((int) R + (1 << shift_y) - 1) >> shift_y;
    ncx = (finder->projector.canvas_cols + (1 << shift_x) - 1) >> shift_x;
    return ncy * ncx;
  };
  int ncy = 0, ncx = 0;
  while (cells_of(sy, sx, ncy, ncx) > 2048) {
    if (ncy > ncx) {
      ++sy;
    } else {
      ++sx;
    }
  }
  g.cell_sy  = sy;
  g.cell_sx  = sx;
  g.cell_ncy = ncy;
  g.cell_ncx = ncx;
  g.ncells   = ncy * ncx;
  // LDS carve.  with_operands = false: the search half of the split pipeline, which keeps no operand rows (the GN kernel gathers
  // them through the committed correspondence vector)
  // the KD-tree finder keeps its tree where the lattice finders keep theirs: nodes in `db`, leaf members in `inv`, root + leaf
  // offsets in `cellstart` (2 + n_leaves + 1 <= max_fixed + 3 entries)
  const bool kdtree         = finder->search_type == PRS_SEARCH_KDTREE;
  const uint32_t cs_entries = kdtree && nf + 4u > (uint32_t) g.ncells + 2u ? nf + 4u : (uint32_t) g.ncells + 2u;
  auto carve = [&](AlignArgs& g, bool with_operands) -> size_t {
    uint32_t off = 0;
    g.off_db       = off; off = align_up16(off + (nf + 1) * 8);  // (+ the sentinel entry behind the last one: the scan reads one entry past a segment)
    g.off_inv      = off; off = align_up16(off + nf * 2);
    g.off_cellstart = off; off = align_up16(off + cs_entries * 2);
    g.off_cfix     = off; off = align_up16(off + (with_operands ? nf * 16 : 0));
    g.off_cmov     = off; off = align_up16(off + (with_operands ? nf * 16 : 0));
    g.off_cls      = off; off = align_up16(off + (with_operands ? nf : 0));
    g.off_sh       = off; off = align_up16(off + (uint32_t) sizeof(AlignShared));
    g.off_u        = off;
    // search-phase arrays
    uint32_t u = off;
    g.off_fdesc  = u; u = align_up16(u + nf * 32);
    g.off_fuv    = u; u = align_up16(u + (finder->search_type == PRS_SEARCH_KDTREE ? nf * 8 : 0));
    g.off_best   = u; u = align_up16(u + nf * 4);
    g.off_second = u; u = align_up16(u + nf * 4);
    g.off_lut    = u; u = align_up16(u + lut_cap * 2);
    g.off_surv   = u; u = align_up16(u + (uint32_t) (with_operands ? kAlignThreads : kSearchThreads) * (uint32_t) g.surv_slots * 2);  // slots x u16 per thread
    // GN-phase terms; the region also serves the database build (hist + slot + bucket) and the disparity column
    g.off_terms = off;
    uint32_t terms_bytes       = kTerms * kAlignThreads * 4;
    const uint32_t histcap     = (R > (uint32_t) g.ncells ? R : (uint32_t) g.ncells) + 2;
    const uint32_t build_bytes = histcap * 4 + (nf + 2) * 2 * 3 + (R + 2) * 2 + 16;
    if (build_bytes > terms_bytes) {
      terms_bytes = build_bytes;
    }
    if (nf * 4 + 160 > terms_bytes) {
      terms_bytes = nf * 4 + 160;
    }
    const uint32_t kd_bytes = (uint32_t) kd_open_capacity((int) nf) * (40 + 7 * 4) + (nf + 2) * 2 * 3 + 64;  // KD-tree build scratch
    if (kdtree && kd_bytes > terms_bytes) {
      terms_bytes = kd_bytes;
    }
    const uint32_t t_end = align_up16(off + terms_bytes);
    return u > t_end ? u : t_end;
  };
  g.surv_slots     = 4;
  const size_t lds = carve(g, true);
  if (lds > 160 * 1024) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: fixed cloud does not fit the 160 KiB LDS (lower max_fixed)");
  }
  g.stamps = ctx_stamps(ctx, (size_t) batch->batch * 16 * sizeof(unsigned long long));
  g.ops     = nullptr;
  g.ctl     = nullptr;
  g.pending = nullptr;
  g.dbcache = nullptr;
  g.db_blob = 0;
  hipStream_t stream = ctx_stream(ctx);
  hipError_t e;
  // diagnostic: PRS_STAMPS=1 times the phases of the fused kernel; with PRS_STAMPS_SPLIT=1 the split pipeline
  // runs instead and the finder phases of every search launch are reported
  if (max_fixed >= (int) kClsOutUnit) {
    // the class counts of a linearisation travel as ONE exact float code (#inliers + kClsOutUnit * #kernelised)
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_run: max_fixed must be below 2048");
  }
  const bool stamps_split = g.stamps && ctx->stamps_split;
  const bool split = mode == PRS_MODE_ALIGN && !ctx_fused_align(ctx) && max_fixed <= kGnThreads * kGnSlots && (!g.stamps || stamps_split);
  if (!split) {
    auto kernel = align_kernel<kAlignThreads, false, -1>;
    e           = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) {
      return ctx_fail_hip(ctx, e, "prs_align_batch_run attribute");
    }
    hipLaunchKernelGGL(kernel, dim3(batch->batch), dim3(kAlignThreads), lds, stream, g);
    e = hipGetLastError();
    if (e != hipSuccess) {
      return ctx_fail_hip(ctx, e, "prs_align_batch_run launch");
    }
    if (g.stamps) {
      ctx_report_stamps(ctx, batch->batch, 10, "align (fused, 256 threads): finder | linearize + in-wave reduction | cross-wave sum | GN solve || of finder: lattice build | projection+search | staging (keys, fixed rows -> LDS) | filter | commit");
    }
    return PRS_OK;
  }
  // ---- split pipeline: search kernel (512 threads/frame) and GN kernel (operands in registers,
  //      ~30 KB LDS, many frames per CU) alternate; every launch skips frames that do not wait for it.
  SplitJob* job = static_cast<SplitJob*>(ctx->align_job);
  if (!job) {
    job                 = new SplitJob();
    ctx->align_job      = job;
    ctx->align_job_free = [](void* p) { delete static_cast<SplitJob*>(p); };
  }
  const size_t ops_bytes = (size_t) batch->batch * (size_t) max_fixed * 2 * sizeof(float4);
  const size_t ctl_bytes = (size_t) batch->batch * sizeof(FrameCtl) + 256;
  unsigned char* small   = static_cast<unsigned char*>(job->buffer(stream, 0, ctl_bytes));
  g.ops                  = static_cast<float4*>(job->buffer(stream, 1, ops_bytes));
  if (!small || !g.ops) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_align_batch_run: split-pipeline scratch allocation failed");
  }
  g.pending = reinterpret_cast<int*>(small);
  g.ctl     = reinterpret_cast<FrameCtl*>(small + 256);
  AlignArgs gs = g;
  gs.mode      = kModeSplitSearch;
  // eight survivor slots per thread where they cost no resident workgroup (a wide search radius or correlated rows overflow four:
  // tum.conf's shape 1.02 -> 0.93 ms per 6144 frames), four otherwise (the headline's 49.6 + 4 KB keep three workgroups per CU)
  gs.surv_slots = 8;  // (the KD-tree finder parks nothing)
  size_t lds_search = carve(gs, false);
  gs.surv_slots     = 4;
  {
    const size_t lds4 = carve(gs, false);
    if ((160u * 1024u) / lds_search == (160u * 1024u) / lds4 && finder->search_type != PRS_SEARCH_KDTREE) {
      gs.surv_slots = 8;
      lds_search    = carve(gs, false);
    } else {
      lds_search = lds4;
    }
  }
  // image of the lattice arrays (contiguous in LDS: db | inv | cellstart), kept per frame between search launches
  gs.db_blob = align_up16(gs.off_cellstart + cs_entries * 2) - gs.off_db;
  gs.dbcache = static_cast<unsigned char*>(job->buffer(stream, 2, (size_t) batch->batch * gs.db_blob));
  if (!gs.dbcache) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_align_batch_run: lattice cache allocation failed");
  }
  const bool wide_slots = gs.surv_slots == 8;
  auto skernel = finder->search_type == PRS_SEARCH_CIRCLE
                   ? (wide_slots ? align_kernel<kSearchThreads, true, PRS_SEARCH_CIRCLE, 8> : align_kernel<kSearchThreads, true, PRS_SEARCH_CIRCLE, 4>)
                   : (finder->search_type == PRS_SEARCH_SQUARE
                        ? (wide_slots ? align_kernel<kSearchThreads, true, PRS_SEARCH_SQUARE, 8> : align_kernel<kSearchThreads, true, PRS_SEARCH_SQUARE, 4>)
                        : (finder->search_type == PRS_SEARCH_RHOMBUS
                             ? (wide_slots ? align_kernel<kSearchThreads, true, PRS_SEARCH_RHOMBUS, 8> : align_kernel<kSearchThreads, true, PRS_SEARCH_RHOMBUS, 4>)
                             : align_kernel<kSearchThreads, true, PRS_SEARCH_KDTREE, 4>));
  // two waves per frame (eight frames resident per CU: the kernel is bound by its single-wave phases and by
  // instruction issue, more independent frames fill the idle slots)
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(skernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_search);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_align_batch_run attribute");
  }
  hipLaunchKernelGGL(split_init_kernel, dim3((batch->batch + 255) / 256), dim3(256), 0, stream, g.ctl, batch->result, g.pending, batch->batch);
  // (the rectified-stereo factor, the one kitti.conf / euroc.conf use, has its own instantiation: the factor type as a
  // compile-time constant removes ~10 selects per linearised correspondence; so does not remembering the factor classes)
  const bool fast = aligner->factor_type == PRS_FACTOR_STEREO && !aligner->keep_only_inlier_correspondences && aligner->kernel_weight_form == PRS_KERNEL_WEIGHT_INV_CHI &&
                    aligner->damping_form == PRS_DAMPING_DIAG && aligner->translation_weight_form == PRS_TRANSLATION_WEIGHT_OFFSET;
  // a batch that cannot fill the chip (fewer frames than two per CU) is bound by the serial chain of each frame, not by occupancy:
  // its instantiation may use the whole register file (WAVES = 1: no spills in the solve, the compiler schedules for latency)
  int n_cu = 0;
  (void) hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const bool lone = fast && batch->batch <= 2 * (n_cu > 0 ? n_cu : 256) && !ctx->no_lone_gn;
  const bool five = fast && !lone && max_fixed > 4 * 128;  // (gn_kernel: LDS_SLOTS / WAVES)
  const bool lean  = !batch->prior && !aligner->with_sensor && !aligner->enable_inlier_only_runs;
  const bool plain = lean && !aligner->enable_motion_prior;   // (gn_kernel / gn_solve_wave: PLAIN = 1)
  const bool mprior = lean && aligner->enable_motion_prior;   // (PLAIN = 2)
  // (the depth factor of the RGB-D configurations -- tum.conf / icl.conf: kept classes, inlier-only runs -- as a compile-time constant too)
  const bool depth = aligner->factor_type == PRS_FACTOR_DEPTH && aligner->kernel_weight_form == PRS_KERNEL_WEIGHT_INV_CHI && aligner->damping_form == PRS_DAMPING_DIAG;
  auto gnk        = lone ? (max_fixed <= 4 * 128 ? (plain ? gn_kernel<4, PRS_FACTOR_STEREO, false, kGnLdsSlots, 1, 1> : gn_kernel<4, PRS_FACTOR_STEREO, false, kGnLdsSlots, 1>)
                                                 : (plain ? gn_kernel<8, PRS_FACTOR_STEREO, false, kGnLdsSlots, 1, 1> : gn_kernel<8, PRS_FACTOR_STEREO, false, kGnLdsSlots, 1>))
                         : (max_fixed <= 4 * 128 ? (fast ? (plain ? gn_kernel<4, PRS_FACTOR_STEREO, false, kGnLdsSlots, 4, 1> : (mprior ? gn_kernel<4, PRS_FACTOR_STEREO, false, kGnLdsSlots, 4, 2> : gn_kernel<4, PRS_FACTOR_STEREO, false>))
                                                         : (depth ? gn_kernel<4, PRS_FACTOR_DEPTH, true> : gn_kernel<4, 0, true>))
                                                 : (fast ? (plain ? gn_kernel<8, PRS_FACTOR_STEREO, false, 3, 5, 1> : (mprior ? gn_kernel<8, PRS_FACTOR_STEREO, false, 3, 5, 2> : gn_kernel<8, PRS_FACTOR_STEREO, false, 3, 5>))
                                                         : (depth ? gn_kernel<8, PRS_FACTOR_DEPTH, true> : gn_kernel<8, 0, true>)));
  const size_t lds_gn = gn_lds_bytes(five ? 3 : kGnLdsSlots);
  // The job lives in the context until align_batch_finish: the rounds are plain launches on the context's stream (no host
  // synchronisation here, the sequence can be captured in a HIP graph once the scratch buffers exist).
  job->active      = true;
  job->stream      = stream;
  job->capture_stream = stream;
  job->g           = g;
  job->gs          = gs;
  job->search      = reinterpret_cast<void (*)(AlignArgs)>(skernel);
  job->gn          = reinterpret_cast<void (*)(AlignArgs)>(gnk);
  job->lds_search  = lds_search;
  job->lds_gn      = lds_gn;
  job->total       = 0;
  job->limit       = 2 * (aligner->max_iterations + inlier_run_length(*aligner)) + 8;
  job->ev_used     = 0;
  job->rounds_timed = 0;
  job->stamps      = stamps_split;
  job->enqueued    = rounds > 0 ? rounds : 5;
  return split_rounds(ctx, job, job->enqueued);
}

// `rounds` x (search launch, Gauss-Newton launch) over the frames that are still pending; every launch skips the others
static int split_rounds(prs_context* ctx, SplitJob* job, int rounds) {
  hipStream_t stream = job->stream;
  auto tick = [&]() {  // measurement only (prs_context_enable_timing): one event per kernel boundary
    if (ctx->timing && job->ev_used < 3 * 64) {
      if (!ctx->timing_ev[job->ev_used]) {
        (void) hipEventCreate(&ctx->timing_ev[job->ev_used]);
      }
      (void) hipEventRecord(ctx->timing_ev[job->ev_used], stream);
      ++job->ev_used;
    }
  };
  const int batch = job->g.b.batch;
  for (int r = 0; r < rounds; ++r) {
    (void) hipMemsetAsync(job->g.pending, 0, sizeof(int), stream);
    tick();
    hipLaunchKernelGGL(job->search, dim3(batch), dim3(kSearchThreads), job->lds_search, stream, job->gs);
    tick();
    if (job->stamps) {
      ctx_report_stamps(ctx, batch, 10, "search launch (split): - | - | - | - || lattice build | projection+search | staging (keys, fixed rows -> LDS) | filter | commit");
    }
    hipLaunchKernelGGL(job->gn, dim3(batch), dim3(kGnThreads), job->lds_gn, stream, job->g);
    tick();
    ++job->total;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    job->active = false;
    return ctx_fail_hip(ctx, e, "prs_align_batch_run split launch");
  }
  return PRS_OK;
}

// blocks until the enqueued batch is complete: one 4-byte readback per group of rounds; more rounds while frames are pending
int align_batch_finish(prs_context* ctx) {
  SplitJob* job = static_cast<SplitJob*>(ctx->align_job);
  if (!job || !job->active) {
    return PRS_OK;
  }
  hipStream_t stream = job->stream;
  auto collect = [&]() {  // after a stream synchronisation: add up (search, GN) pairs of [e0, e1, e2] triples
    for (int i = 0; i + 2 < job->ev_used; i += 3) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, ctx->timing_ev[i], ctx->timing_ev[i + 1]) == hipSuccess &&
          hipEventElapsedTime(&b, ctx->timing_ev[i + 1], ctx->timing_ev[i + 2]) == hipSuccess) {
        ctx->t_search_ms += a;
        ctx->t_gn_ms += b;
        ++ctx->n_search;
        ++ctx->n_gn;
        const int round = job->rounds_timed < 15 ? job->rounds_timed : 15;
        ctx->t_search_round[round] += a;
        ctx->t_gn_round[round] += b;
      }
      ++job->rounds_timed;
    }
    job->ev_used = 0;
  };
  for (;;) {
    int pending  = 0;
    hipError_t e = hipMemcpyAsync(&pending, job->g.pending, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) {
      e = hipStreamSynchronize(stream);
    }
    if (e != hipSuccess) {
      job->active = false;
      return ctx_fail_hip(ctx, e, "prs_align_batch_finish: completion check");
    }
    collect();
    if (pending == 0) {
      break;
    }
    if (job->total > job->limit) {
      job->active = false;
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_align_batch_finish: split pipeline did not finish");
    }
    const int rc = split_rounds(ctx, job, 4);
    if (rc != PRS_OK) {
      return rc;
    }
  }
  job->active = false;
  if (ctx->timing) {
    ++ctx->n_batches_timed;
  }
  return PRS_OK;
}

// a HIP graph captured around align_batch_launch replays the rounds without passing through the host code: re-arm the job so
// that align_batch_finish checks completion (and adds rounds) for the replayed batch too.  `replay_stream`: the stream the graph
// was launched on (NULL = the stream the batch was enqueued / captured on); finish synchronises THAT stream and enqueues its extra
// rounds there, so a replay on another stream is not raced.
int align_batch_rearm(prs_context* ctx, hipStream_t replay_stream) {
  SplitJob* job = static_cast<SplitJob*>(ctx->align_job);
  if (!job || !job->gn) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_rearm: no batch has been enqueued on this context");
  }
  if (job->active) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_align_batch_rearm: the previous batch has not been finished");
  }
  job->stream       = replay_stream ? replay_stream : job->capture_stream;  // (NULL: back to the capture stream, whatever the last replay used)
  job->active       = true;
  job->total        = job->enqueued;  // the rounds the captured enqueue holds
  job->ev_used      = 0;
  job->rounds_timed = 0;
  return PRS_OK;
}

bool align_job_active(const prs_context* ctx) {
  const SplitJob* job = static_cast<const SplitJob*>(ctx->align_job);
  return job && job->active;
}

// prs_selftest_reciprocal: all 2^32 operands through recip_exact as the kernels above use it
__global__ __launch_bounds__(256) void recip_selftest_kernel(unsigned long long* counts) {
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long differ = 0, fast = 0;
  for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x  = __uint_as_float((uint32_t) i);
    const float ax = __builtin_fabsf(x);
    const bool short_form = __ballot(!(ax >= 0x1p-126f && ax < 0x1p126f)) == 0ull;  // (the test recip_exact makes)
    const uint32_t got = __float_as_uint(recip_exact(x)), want = __float_as_uint(1.0f / x);
    const bool both_nan = (got & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u;
    differ += (got != want && !both_nan) ? 1ull : 0ull;
    fast += short_form ? 1ull : 0ull;
  }
  if (differ) {
    atomicAdd(&counts[0], differ);
  }
  atomicAdd(&counts[1], fast);
}

int recip_selftest_launch(prs_context* ctx, unsigned long long* d_counts) {
  hipLaunchKernelGGL(recip_selftest_kernel, dim3(256 * 32), dim3(256), 0, ctx_stream(ctx), d_counts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_selftest_reciprocal launch");
  }
  return PRS_OK;
}

int gn_step_launch(prs_context* ctx, const float* dH, const float* db, float damping, int damping_form, float* dX, int* dok) {
  hipLaunchKernelGGL(gn_step_kernel, dim3(1), dim3(1), 0, ctx_stream(ctx), dH, db, damping, damping_form == PRS_DAMPING_IDENTITY ? 1 : 0, dX, dok);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_gn_step launch");
  }
  return PRS_OK;
}

}  // namespace prs
