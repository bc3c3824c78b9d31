// mapping.hip -- landmark estimators + projective mergers on the device.
//
// Reference: MergerProjective_::compute (mapping/mergers/merger_projective_impl.cpp:8-190), its
// _updatePoint (:192-208, merger_projective_rigid_stereo_triangulation_impl.cpp:7-39), _addPoints
// (:210-305), _initializeLandmark (:308-326), and the estimators it drives:
// LandmarkEstimatorWeightedMean_ (mapping/landmarks/landmark_estimator_weighted_mean_impl.cpp:7-41),
// LandmarkEstimatorEKF_ (landmark_estimator_ekf_impl.cpp:17-82) with PointEKFBase::_predict/_correct
// (filters/point_ekf_base.hpp:63-125) and the three measurement models, all in double,
// LandmarkEstimatorPoseBasedSmoother_ (landmark_estimator_pose_based_smoother_impl.cpp:7-148).
//
// One 256-thread workgroup merges one frame into one map.  The reference's sequential loop over the
// correspondence vector blocks a bin for the FIRST correspondence that reaches it (:89-122); that is
// an atomicMin on the correspondence index.  The surviving correspondences touch distinct landmarks,
// so each one is updated by its own lane with exactly the arithmetic (order of every product and
// sum, double where the reference is double) a sequential evaluation performs.  New landmarks are
// appended in the reference's order: bins in the order their first measurement appears, the best
// candidate (largest disparity / smallest depth, earliest on ties) of each bin.
//
// For the weighted mean and the EKF the per-landmark work is tiny and the kernel is bound by the scattered
// reads and writes of the landmark rows (coordinates 16 B, state 16 B, covariance 36 B, descriptor 32 B,
// history 28 B per measurement).  The pose-based smoother is different: up to 100 Gauss-Newton iterations per
// landmark, a long serial per-lane computation.  It runs as four kernels (merge_batch_launch):
//   merge_kernel<.., 1>    correspondences, updates that need no iteration, smoother work items queued
//   smoother_kernel        one wave per frame, rounds of 8 iterations with compaction, exact cycle short-cut
//   smoother_tail_kernel   the few landmarks per frame that are still iterating, 64 per wave across frames
//   merge_kernel<.., 2>    additions and the frame's result
#include "prs_device.h"
#include "prs_host.h"
#include "prs_se3.h"

namespace prs {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves   = kMergeThreads / 64;
constexpr uint32_t kNoBin   = 0xffffffffu;

struct MergeArgs {
  prs_merger_params p;
  prs_merge_batch b;
  float row_w, col_w;  // bin widths in pixels (merger_projective_impl.cpp:30-33)
  int nbr, nbc;        // bin table extent
  uint32_t off_owner, off_first, off_best, off_seen, off_sh, off_wave, off_pose_cache;
  void* work;          // smoother: [batch][2][corr_stride] work items (two lists, swapped every round)
  unsigned long long* stamps;  // diagnostic (PRS_STAMPS=1): [batch][16] shader-clock stamps of thread 0
  struct MergeCarry* carry;    // smoother merger run as three kernels: [batch] state handed from one to the next
  struct TailItem* tail;       // landmarks still iterating after the smoother kernel's rounds (all frames), [tail_capacity]
  int* tail_count;
  int tail_capacity;
};

struct SmootherItem;
// pose-based smoother as front kernel | smoother kernel | back kernel (see merge_batch_launch)
struct MergeCarry {
  int n_merged;  // landmarks merged so far
  int n_work;    // smoother work items the front kernel queued
  int error;     // error raised while updating points (the frame's additions are skipped, like in the fused kernel)
  int returned;  // the front kernel has written the frame's result already (error before any update)
};

struct MergeShared {
  float sensor_in_world[16];
  float world_in_sensor[16];
  float world_in_local_map[16];
  float measurement_in_scene[16];
  int error;
  int n_merged;
  int base;
  int n_work, n_next;  // smoother work items of this / the next round
};

__device__ __forceinline__ void apply_rows(const float* T, const float* p, float* out) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[i] = ((T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3];
  }
}
__device__ __forceinline__ float sqnorm3(const float* a, const float* b) {
  const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return (d0 * d0 + d1 * d1) + d2 * d2;
}

// one landmark row of one map
struct Landmark {
  float* coords;
  uint8_t* desc;
  float* state;
  float* cov;
  uint32_t* n_opt;
  uint8_t* inlier;
  uint32_t* n_meas;
  prs_camera_measurement* meas;
};

__device__ __forceinline__ Landmark landmark_at(const prs_merge_batch& b, int map, int idx) {
  const size_t r = (size_t) map * (size_t) b.capacity + (size_t) idx;
  Landmark l;
  l.coords = b.coords + 4 * r;
  l.desc   = b.desc + 32 * r;
  l.state  = b.state + 4 * r;
  l.cov    = b.covariance + 9 * r;
  l.n_opt  = b.n_opt + r;
  l.inlier = b.inlier + r;
  l.n_meas = b.n_meas + r;
  l.meas   = b.meas ? b.meas + r * (size_t) b.max_measurements : nullptr;
  return l;
}

// PointStatisticsField3D::addOptimizationResult
__device__ __forceinline__ void add_optimization_result(const Landmark& l, const float* cw, const float* cov9) {
  l.state[0] = cw[0];
  l.state[1] = cw[1];
  l.state[2] = cw[2];
  if (cov9) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      l.cov[i] = cov9[i];
    }
  }
  *l.n_opt = *l.n_opt + 1u;
}

// ---- LandmarkEstimatorWeightedMean_::compute -----------------------------------------------------
__device__ int estimate_weighted_mean(const prs_estimator_params& P, const MergeShared& sh, const Landmark& l, const float* landmark_in_sensor) {
  *l.inlier           = 0;  // :13
  const float init[3] = {l.state[0], l.state[1], l.state[2]};
  float upd[3];
  apply_rows(sh.sensor_in_world, landmark_in_sensor, upd);  // :20-21
  const float npo = (float) (*l.n_opt + 1u);                // :23
  float cw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    cw[i] = (npo * init[i] + upd[i]) / (npo + 1.0f);  // :25-27
  }
  if (sqnorm3(cw, init) > P.maximum_distance_geometry_meters_squared) {  // :30-34
    return 0;
  }
  add_optimization_result(l, cw, nullptr);  // :37
  *l.inlier = 1;
  float loc[3];
  apply_rows(sh.world_in_local_map, cw, loc);  // :41
  l.coords[0] = loc[0];
  l.coords[1] = loc[1];
  l.coords[2] = loc[2];
  return 1;
}

// ---- PointEKFBase in double -----------------------------------------------------------------------
// inverse of the symmetric positive definite innovation covariance through LDL^T
template <int N>
__device__ void spd_inverse(const double* S, double* Sinv) {
  double L[N][N], D[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = S[N * j + j];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      if (k < j) {
        d -= (L[j][k] * L[j][k]) * D[k];
      }
    }
    D[j] = d;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i > j) {
        double v = S[N * i + j];
#pragma unroll
        for (int k = 0; k < N; ++k) {
          if (k < j) {
            v -= (L[i][k] * L[j][k]) * D[k];
          }
        }
        L[i][j] = v / d;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double y[N], x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double v = i == c ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (k < i) {
          v -= L[i][k] * y[k];
        }
      }
      y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      y[i] = y[i] / D[i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double v = y[i];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (k > i) {
          v -= L[k][i] * x[k];
        }
      }
      x[i] = v;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      Sinv[N * i + c] = x[i];
    }
  }
}

// LandmarkEstimatorEKF_::compute (landmark_estimator_ekf_impl.cpp:17-82), measurement dimension N
template <int N>
__device__ int estimate_ekf(const prs_estimator_params& P, const MergeShared& sh, const Landmark& l, const float* measurement) {
  *l.inlier           = 0;  // :24
  const float init[3] = {l.state[0], l.state[1], l.state[2]};
  double st[3], cov[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    st[i] = (double) init[i];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    cov[i] = (double) l.cov[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // :48-50
    cov[4 * i] = cov[4 * i] > P.minimum_state_element_covariance ? cov[4 * i] : P.minimum_state_element_covariance;
  }
  // _predict (filters/point_ekf_base.hpp:63-78)
  double F[9], tr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      F[3 * i + j] = (double) sh.world_in_sensor[4 * i + j];
    }
    tr[i] = (double) sh.world_in_sensor[4 * i + 3];
  }
  double M[9], covp[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      M[3 * i + j] = (F[3 * i + 0] * cov[0 + j] + F[3 * i + 1] * cov[3 + j]) + F[3 * i + 2] * cov[6 + j];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      covp[3 * i + j] = (M[3 * i + 0] * F[3 * j + 0] + M[3 * i + 1] * F[3 * j + 1]) + M[3 * i + 2] * F[3 * j + 2];
    }
  }
  double sp[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sp[i] = ((F[3 * i + 0] * st[0] + F[3 * i + 1] * st[1]) + F[3 * i + 2] * st[2]) + tr[i];
  }
  // _computeMeasurementPrediction (stereo :13-48, depth :7-36, mono :16-43)
  double h[N], J[N * 3];
  {
    const double x = sp[0], y = sp[1], z = sp[2];
    const double z_2     = z * z;
    const double fx_x    = P.fx * x;
    const double fy_y    = P.fy * y;
    const double fx_by_z = P.fx / z;
    const double fy_by_z = P.fy / z;
    if (N == 4) {
      const double x_h = fx_x + P.cx * z;
      const double y_h = fy_y + P.cy * z;
      h[0] = x_h / z;
      h[1] = y_h / z;
      h[2 % N] = (x_h - P.b_x) / z;
      h[3 % N] = (y_h - P.b_y) / z;
      J[0] = fx_by_z; J[1] = 0.0;     J[2] = -fx_x / z_2;
      J[3] = 0.0;     J[4] = fy_by_z; J[5] = -fy_y / z_2;
      J[6 % (3 * N)] = fx_by_z; J[7 % (3 * N)] = 0.0;      J[8 % (3 * N)]  = -(fx_x - P.b_x) / z_2;
      J[9 % (3 * N)] = 0.0;     J[10 % (3 * N)] = fy_by_z; J[11 % (3 * N)] = -(fy_y - P.b_y) / z_2;
    } else {
      h[0] = fx_by_z * x + P.cx;
      h[1] = fy_by_z * y + P.cy;
      J[0] = fx_by_z; J[1] = 0.0;     J[2] = -fx_x / z_2;
      J[3] = 0.0;     J[4] = fy_by_z; J[5] = -fy_y / z_2;
      if (N == 3) {
        h[2 % N] = z;
        J[6 % (3 * N)] = 0.0;
        J[7 % (3 * N)] = 0.0;
        J[8 % (3 * N)] = 1.0;
      }
    }
  }
  // _correct (filters/point_ekf_base.hpp:81-123)
  double A[N * 3];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[3 * i + j] = (J[3 * i + 0] * covp[0 + j] + J[3 * i + 1] * covp[3 + j]) + J[3 * i + 2] * covp[6 + j];
    }
  }
  double S[N * N], Sinv[N * N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double a = (A[3 * i + 0] * J[3 * j + 0] + A[3 * i + 1] * J[3 * j + 1]) + A[3 * i + 2] * J[3 * j + 2];
      S[N * i + j]   = (i == j ? P.minimum_state_element_covariance : 0.0) + a;  // landmark_estimator_ekf_impl.cpp:28-29
    }
  }
  spd_inverse<N>(S, Sinv);
  double B[3 * N];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      B[N * i + j] = (covp[3 * i + 0] * J[3 * j + 0] + covp[3 * i + 1] * J[3 * j + 1]) + covp[3 * i + 2] * J[3 * j + 2];
    }
  }
  double K[3 * N];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        s += B[N * i + k] * Sinv[N * k + j];
      }
      K[N * i + j] = s;
    }
  }
  double r[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    r[k] = (double) measurement[k] - h[k];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      s += K[N * i + k] * r[k];
    }
    sp[i] += s;
  }
  double C[9], covc[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        s += K[N * i + k] * J[3 * k + j];
      }
      C[3 * i + j] = (i == j ? 1.0 : 0.0) - s;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      covc[3 * i + j] = (C[3 * i + 0] * covp[0 + j] + C[3 * i + 1] * covp[3 + j]) + C[3 * i + 2] * covp[6 + j];
    }
  }
  double n2 = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    n2 += covc[i] * covc[i];
  }
  if (sp[2] <= 0.0 || n2 > P.maximum_covariance_norm_squared) {  // landmark_estimator_ekf_impl.cpp:60-64
    return 0;
  }
  const float sf[3] = {(float) sp[0], (float) sp[1], (float) sp[2]};
  float cw[3];
  apply_rows(sh.sensor_in_world, sf, cw);  // :68-69
  if (sqnorm3(cw, init) > P.maximum_distance_geometry_meters_squared) {  // :70-74
    return 0;
  }
  float covf[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    covf[i] = (float) covc[i];
  }
  add_optimization_result(l, cw, covf);  // :77-78
  *l.inlier = 1;
  float loc[3];
  apply_rows(sh.world_in_local_map, cw, loc);  // :82-83
  l.coords[0] = loc[0];
  l.coords[1] = loc[1];
  l.coords[2] = loc[2];
  return 1;
}

// ---- LandmarkEstimatorPoseBasedSmoother_ -------------------------------------------------------------
__device__ void apply_pose(const float* T12, const float* p, float* out) {
  apply_rows(T12, p, out);
}

// _setMeanCoordinatesInWorld (:138-147)
__device__ void mean_in_world(const prs_camera_measurement* m, uint32_t n, const prs_frame_pose* poses, float* out) {
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
  for (uint32_t k = 0; k < n; ++k) {
    float w[3];
    apply_pose(poses[m[k].frame].sensor_in_world, m[k].point_in_camera, w);
    acc0 += w[0];
    acc1 += w[1];
    acc2 += w[2];
  }
  const float fn = (float) n;
  out[0] = acc0 / fn;
  out[1] = acc1 / fn;
  out[2] = acc2 / fn;
}

// 3x3 solve by elimination with full pivoting (stand-in for Eigen's fullPivLu, :108)
__device__ void solve3_full_pivot(const float* A_in, const float* rhs, float* x) {
  float A[3][3], b[3];
  int col_of[3] = {0, 1, 2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[i][j] = A_in[3 * i + j];
    }
    b[i] = rhs[i];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int pr = k, pc = k;
    float best = fabsf(A[k][k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (i >= k && j >= k) {
          const float v = fabsf(A[i][j]);
          if (v > best) {
            best = v;
            pr   = i;
            pc   = j;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // row swap k <-> pr
      if (i > k && i == pr) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float tmp = A[k][j];
          A[k][j]         = A[i][j];
          A[i][j]         = tmp;
        }
        const float tb = b[k];
        b[k]           = b[i];
        b[i]           = tb;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // column swap k <-> pc
      if (j > k && j == pc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float tmp = A[i][k];
          A[i][k]         = A[i][j];
          A[i][j]         = tmp;
        }
        const int tc = col_of[k];
        col_of[k]    = col_of[j];
        col_of[j]    = tc;
      }
    }
    const float piv = A[k][k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > k) {
        const float f = A[i][k] / piv;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (j > k) {
            A[i][j] -= f * A[k][j];
          }
        }
        b[i] -= f * b[k];
      }
    }
  }
  float y[3];
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    float v = b[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j > i) {
        v -= A[i][j] * y[j];
      }
    }
    y[i] = v / A[i][i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (col_of[i] == c) {
        x[c] = y[i];
      }
    }
  }
}

// pose_cache: per frame of the map's pose table 21 floats in LDS: world_in_sensor (3x4) and camera_matrix * R (3x3, the
// "jacobian_linear" of :89, which depends on the frame only) -- evaluated once per workgroup instead of once per
// measurement and iteration, with the same operations in the same order.
//
// The Gauss-Newton loop of a landmark either repeats its chi2 after a handful of iterations or runs all
// maximum_number_of_iterations (kitti.conf's delta of 1e-6 is below the float resolution of the chi2 sums), so
// one lane per landmark for the whole loop would make every wave as slow as its slowest lane.  The loop state
// therefore lives in a work item; lanes run a few iterations per round and the unfinished items are compacted
// between rounds, so that only as many waves as there are unfinished landmarks keep iterating.
struct SmootherItem {
  int s, m;               // scene index, measurement index
  float world[3];         // the variable being optimised (:26)
  float total_previous;   // :46
  uint32_t it;            // iterations done
  uint32_t n_inliers;     // of the last iteration (:110)
};
// a landmark that is still iterating after the per-frame smoother kernel's rounds: continued by smoother_tail_kernel,
// where the stragglers of ALL frames share waves (a frame keeps ~4 of them, and a wave per frame would idle 60 lanes
// for another ~75 iterations)
struct TailItem {
  int map;
  SmootherItem item;
};
constexpr int kTailRounds   = 3;   // rounds of 8 iterations the per-frame kernel runs before handing over
constexpr int kTailPerFrame = 64;  // hand-over only when at most this many landmarks of the frame are left (list capacity)

// everything before the loop (:13-43).  Returns < 0 on error, 0 / 1 when the landmark is finished already
// (fewer measurements than minimum_number_of_measurements_for_optimization: averaging), 2 when `item` is ready to iterate
__device__ int smoother_begin(const prs_estimator_params& P, const MergeShared& sh, const prs_frame_pose* poses, int frame, int max_measurements,
                              const Landmark& l, const float* measurement, const float* landmark_in_sensor, SmootherItem& item) {
  *l.inlier = 0;  // :13
  if (!l.meas || *l.n_meas >= (uint32_t) max_measurements) {
    return PRS_ERR_HISTORY;
  }
  prs_camera_measurement* M = l.meas;
  {  // addMeasurement (:16-20)
    prs_camera_measurement nm;
    nm.point_in_image[0]  = measurement[0];
    nm.point_in_image[1]  = measurement[1];
    nm.point_in_image[2]  = measurement[2];
    nm.point_in_camera[0] = landmark_in_sensor[0];
    nm.point_in_camera[1] = landmark_in_sensor[1];
    nm.point_in_camera[2] = landmark_in_sensor[2];
    nm.frame              = frame;
    M[*l.n_meas]          = nm;
    *l.n_meas             = *l.n_meas + 1u;
  }
  const uint32_t n    = *l.n_meas;
  const float init[3] = {l.state[0], l.state[1], l.state[2]};
  float world[3]      = {init[0], init[1], init[2]};
  if (n < P.minimum_number_of_measurements_for_optimization) {  // :29-43
    mean_in_world(M, n, poses, world);
    if (sqnorm3(world, init) < P.maximum_distance_geometry_meters_squared) {
      float loc[3];
      apply_rows(sh.world_in_local_map, world, loc);
      l.coords[0] = loc[0];
      l.coords[1] = loc[1];
      l.coords[2] = loc[2];
      l.state[0]  = world[0];
      l.state[1]  = world[1];
      l.state[2]  = world[2];
      *l.inlier   = 1;
      *l.n_opt    = n;
    }
    return *l.inlier;
  }
  item.world[0]       = world[0];
  item.world[1]       = world[1];
  item.world[2]       = world[2];
  item.total_previous = 0.0f;  // :46
  item.it             = 0;
  item.n_inliers      = 0;
  return 2;
}

// up to `budget` iterations of the loop (:49-119); true when the loop has ended (converged or out of iterations)
// `ring` (optional, LDS, [kSmootherRing][5][64] floats, this lane's column `ring_lane`): the states of this call's
// iterations, used to end a loop that has entered a short cycle.  The iteration is a deterministic map of `world`, so
// once a state repeats (bit for bit) everything after it is known: in float arithmetic the chi2 sums of about a
// third of the landmarks never settle within kitti.conf's delta of 1e-6 -- they alternate between 2..8 states from
// the first ten iterations on and would run all 100.  The result is exactly the one the full loop produces.
// POSES_IN_LDS: `pose_cache` holds [frame][21] floats (world_in_sensor 3x4, camera_matrix * R 3x3) of the landmark's map
// in LDS; otherwise it points at the map's prs_frame_pose table in global memory and the 3x3 product is formed per
// use with the same expression (the tail kernel, whose lanes belong to different maps).
constexpr int kSmootherRing = 6;
template <bool POSES_IN_LDS = true>
__device__ bool smoother_iterate(const prs_estimator_params& P, const float* pose_cache, const Landmark& l, SmootherItem& item, int budget,
                                 float* ring = nullptr, int ring_lane = 0) {
  const prs_camera_measurement* M = l.meas;
  const uint32_t n           = *l.n_meas;
  const float* Km            = P.camera_matrix;
  const float max_kernel     = P.maximum_reprojection_error_pixels_squared;
  float world[3]             = {item.world[0], item.world[1], item.world[2]};
  float total_previous       = item.total_previous;
  uint32_t number_of_inliers = item.n_inliers;
  uint32_t it                = item.it;
  bool ended                 = it >= P.maximum_number_of_iterations;
  // The history of the landmark does not change during the optimisation: its first measurements are read
  // once per call (all loads in flight together) instead of once per iteration -- the loop is a serial
  // per-lane chain and a global load inside it costs its full latency every time.
  constexpr int kCached = 8;
  float mu[kCached], mv[kCached], md[kCached];
  int mf[kCached];
#pragma unroll
  for (int k = 0; k < kCached; ++k) {
    const uint32_t kk = (uint32_t) k < n ? (uint32_t) k : 0u;
    mu[k] = M[kk].point_in_image[0];
    mv[k] = M[kk].point_in_image[1];
    md[k] = M[kk].point_in_camera[2];
    mf[k] = M[kk].frame;
  }
  int slot = 0;  // iterations of this call (= states in the ring)
  auto ring_at = [&](int state, int field) -> float& { return ring[(state * 5 + field) * 64 + ring_lane]; };
  while (!ended && budget > 0) {
    if (ring && slot < kSmootherRing) {
      ring_at(slot, 0) = world[0];
      ring_at(slot, 1) = world[1];
      ring_at(slot, 2) = world[2];
    }
    float H[9], b[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      H[i] = 0.0f;
    }
    b[0] = b[1] = b[2] = 0.0f;
    float total_error_squared   = 0.0f;
    uint32_t number_of_outliers = 0;
    // one measurement (image point u, v, depth d, taken at `frame`): :59-104
    auto accumulate = [&](const float u, const float v, const float d, const int frame) {
      float omega[3] = {1.0f, 1.0f, 10.0f};  // :59-60
      float Wg[12], Jg[9];
      const float* W  = pose_cache + 21 * frame;
      const float* Jl = W + 12;
      if (!POSES_IN_LDS) {
        const float* src = reinterpret_cast<const prs_frame_pose*>(pose_cache)[frame].world_in_sensor;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          Wg[i] = src[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {  // landmark_estimator_pose_based_smoother_impl.cpp:89
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            Jg[3 * i + j] = (Km[3 * i + 0] * Wg[0 + j] + Km[3 * i + 1] * Wg[4 + j]) + Km[3 * i + 2] * Wg[8 + j];
          }
        }
        W  = Wg;
        Jl = Jg;
      }
      float pc[3];
      apply_pose(W, world, pc);  // :63
      if (pc[2] <= 0.0f) {
        ++number_of_outliers;
        return;
      }
      float ph[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ph[i] = (Km[3 * i + 0] * pc[0] + Km[3 * i + 1] * pc[1]) + Km[3 * i + 2] * pc[2];
      }
      const float c      = ph[2];
      const float inv_c  = recip_exact(c);  // (= 1.0f / c, bit for bit: prs_device.h)
      const float inv_c2 = inv_c * inv_c;
      const float pi0 = ph[0] / c, pi1 = ph[1] / c;  // :71
      const float e[3] = {pi0 - u, pi1 - v, c - d};  // :74-76
      const float error_squared = (e[0] * (omega[0] * e[0]) + e[1] * (omega[1] * e[1])) + e[2] * (omega[2] * e[2]);
      total_error_squared += error_squared;
      if (error_squared > max_kernel) {  // :83-86
        const float s = max_kernel / error_squared;
        omega[0] *= s;
        omega[1] *= s;
        omega[2] *= s;
        ++number_of_outliers;
      }
      float Jh[9], J[9];
      Jh[0] = inv_c; Jh[1] = 0.0f;  Jh[2] = -ph[0] * inv_c2;  // :94-98
      Jh[3] = 0.0f;  Jh[4] = inv_c; Jh[5] = -ph[1] * inv_c2;
      Jh[6] = 0.0f;  Jh[7] = 0.0f;  Jh[8] = 1.0f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          J[3 * i + j] = (Jh[3 * i + 0] * Jl[0 + j] + Jh[3 * i + 1] * Jl[3 + j]) + Jh[3 * i + 2] * Jl[6 + j];
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {  // :103-104
#pragma unroll
        for (int c2 = 0; c2 < 3; ++c2) {
          H[3 * a + c2] += (J[0 + a] * (omega[0] * J[0 + c2]) + J[3 + a] * (omega[1] * J[3 + c2])) + J[6 + a] * (omega[2] * J[6 + c2]);
        }
        b[a] += (J[0 + a] * (omega[0] * e[0]) + J[3 + a] * (omega[1] * e[1])) + J[6 + a] * (omega[2] * e[2]);
      }
    };
#pragma unroll
    for (int k = 0; k < kCached; ++k) {
      if ((uint32_t) k < n) {
        accumulate(mu[k], mv[k], md[k], mf[k]);
      }
    }
    for (uint32_t k = kCached; k < n; ++k) {
      accumulate(M[k].point_in_image[0], M[k].point_in_image[1], M[k].point_in_camera[2], M[k].frame);
    }
    const float nb[3] = {-b[0], -b[1], -b[2]};
    float dx[3];
    solve3_full_pivot(H, nb, dx);  // :108
    world[0] += dx[0];
    world[1] += dx[1];
    world[2] += dx[2];
    number_of_inliers = n - number_of_outliers;
    ++it;
    --budget;
    if (ring && slot < kSmootherRing) {
      ring_at(slot, 3) = total_error_squared;
      ring_at(slot, 4) = __uint_as_float(number_of_inliers);
    }
    if (fabsf(total_error_squared - total_previous) < P.convergence_criterion_minimum_chi2_delta) {  // :113-117
      ended = true;
    } else {
      total_previous = total_error_squared;
      ended          = it >= P.maximum_number_of_iterations;
      if (!ended && ring && slot < kSmootherRing) {
        // the state the NEXT iteration (index `it`) starts from: has this call been there already?
        int q = -1;
        for (int k = slot; k >= 0; --k) {
          if (__float_as_uint(ring_at(k, 0)) == __float_as_uint(world[0]) && __float_as_uint(ring_at(k, 1)) == __float_as_uint(world[1]) &&
              __float_as_uint(ring_at(k, 2)) == __float_as_uint(world[2])) {
            q = k;
          }
        }
        if (q >= 0) {
          // iterations (it - period) and it start from the same state: iteration it + m repeats iteration it - period + m.
          // Every later convergence test compares a pair of chi2 values that has been compared before and did not end
          // the loop -- except the one of iteration `it` itself (chi2 of state q against the chi2 just computed).
          const int period = slot + 1 - q;
          if (fabsf(ring_at(q, 3) - total_previous) < P.convergence_criterion_minimum_chi2_delta) {
            // iteration `it` runs (state q -> state q + 1) and ends the loop
            if (q + 1 <= slot) {
              world[0] = ring_at(q + 1, 0);
              world[1] = ring_at(q + 1, 1);
              world[2] = ring_at(q + 1, 2);
            }  // (period 1: a fixed point, the state after it is the current one)
            number_of_inliers = __float_as_uint(ring_at(q, 4));
            ++it;
          } else {
            // the loop runs to the iteration limit: it ends in the state (limit - first) mod period of the cycle, with the
            // inlier count of the iteration before
            const uint32_t first = it - (uint32_t) period;
            const int r_world    = (int) ((P.maximum_number_of_iterations - first) % (uint32_t) period);
            const int r_inliers  = (int) ((P.maximum_number_of_iterations - 1u - first) % (uint32_t) period);
            world[0]             = ring_at(q + r_world, 0);
            world[1]             = ring_at(q + r_world, 1);
            world[2]             = ring_at(q + r_world, 2);
            number_of_inliers    = __float_as_uint(ring_at(q + r_inliers, 4));
            it                   = P.maximum_number_of_iterations;
          }
          ended = true;
        }
      }
    }
    ++slot;
  }
  item.world[0]       = world[0];
  item.world[1]       = world[1];
  item.world[2]       = world[2];
  item.total_previous = total_previous;
  item.n_inliers      = number_of_inliers;
  item.it             = it;
  return ended;
}

// everything after the loop (:121-138)
__device__ int smoother_finish(const float* world_in_local_map, const prs_frame_pose* poses, const Landmark& l, const SmootherItem& item) {
  float world[3] = {item.world[0], item.world[1], item.world[2]};
  if (item.n_inliers > *l.n_opt) {  // :122-127
    add_optimization_result(l, world, nullptr);
    *l.inlier = 1;
  } else {  // :130-135
    mean_in_world(l.meas, *l.n_meas, poses, world);
    l.state[0] = world[0];
    l.state[1] = world[1];
    l.state[2] = world[2];
  }
  float loc[3];
  apply_rows(world_in_local_map, world, loc);  // :138
  l.coords[0] = loc[0];
  l.coords[1] = loc[1];
  l.coords[2] = loc[2];
  return *l.inlier;
}

// triangulateRectifiedMidpoint (mapping/triangulator_rigid_stereo.cpp:60-85)
__device__ __forceinline__ bool triangulate_one(const prs_triangulator_params& tp, const float4 z, float* p) {
  const float x_L = z.x, y_L = z.y, x_R = z.z, y_R = z.w;
  p[0] = p[1] = p[2] = 0.0f;
  if (x_L - x_R < tp.minimum_disparity_pixels) {
    return false;
  }
  float depth_meters = tp.infinity_depth_meters;
  if (x_L > x_R) {
    depth_meters = tp.b_x / (x_L - x_R);
  }
  p[2] = depth_meters;
  p[0] = 1 / tp.fx * (x_L - tp.cx) * depth_meters;
  p[1] = 1 / tp.fy * ((y_L + y_R) / 2 - tp.cy) * depth_meters;
  return true;
}

__device__ __forceinline__ uint32_t bin_of(float coordinate, float width) {
  return (uint32_t) roundf(coordinate / width);  // merger_projective_impl.cpp:84-85
}

// order-preserving map of a float onto unsigned integers
__device__ __forceinline__ uint32_t float_key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// EST / DIM are compile-time so that an instantiation only carries the registers of its own estimator
// (the double-precision EKF would otherwise set the occupancy of the float estimators too)
// PHASE 0: the whole merger in one kernel.  The pose-based smoother runs as PHASE 1 (everything up to the queued
// smoother work items), smoother_kernel, PHASE 2 (additions + result): its optimisation loop is a long serial
// per-lane computation, and as part of this 256-thread kernel (bin tables in LDS, four workgroups per CU) only
// four of them are in flight per CU -- on its own it runs sixteen.
template <int EST, int DIM, int PHASE>
__global__ __launch_bounds__(kMergeThreads) void merge_kernel(const MergeArgs a) {
  constexpr bool kFront = PHASE != 2;  // correspondences, _updatePoint (PHASE 2: _addPoints + result only)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* owner   = reinterpret_cast<uint32_t*>(smem + a.off_owner);  // first correspondence reaching the bin
  uint32_t* first   = reinterpret_cast<uint32_t*>(smem + a.off_first);  // first addition candidate of the bin
  unsigned long long* best = reinterpret_cast<unsigned long long*>(smem + a.off_best);  // (quality, ~index) of the best candidate
  uint32_t* seen    = reinterpret_cast<uint32_t*>(smem + a.off_seen);   // scene indices already referenced
  MergeShared& sh   = *reinterpret_cast<MergeShared*>(smem + a.off_sh);
  int* wave_tot     = reinterpret_cast<int*>(smem + a.off_wave);
  float* pose_cache = reinterpret_cast<float*>(smem + a.off_pose_cache);  // smoother: [max_frames][21]
  const int tid     = threadIdx.x;
  const int lane    = tid & 63;
  const int wave    = tid >> 6;
  const int map     = blockIdx.x;
  const prs_merge_batch& B = a.b;
  const prs_merger_params& P = a.p;
  const int nbins   = a.nbr * a.nbc;
  int n_points      = B.n_points[map];
  const int n_meas  = B.n_measured[map] < 0 ? 0 : (B.n_measured[map] > B.measurement_stride ? B.measurement_stride : B.n_measured[map]);
  const int n_corr  = B.n_corr ? (B.n_corr[map] < 0 ? 0 : (B.n_corr[map] > B.corr_stride ? B.corr_stride : B.n_corr[map])) : 0;
  const int frame   = B.frame[map];
  const float4* __restrict__ zs = reinterpret_cast<const float4*>(B.measurement) + (size_t) map * B.measurement_stride;
  const uint8_t* __restrict__ zdesc = B.measurement_desc + (size_t) map * B.measurement_stride * 32;
  const prs_corr* __restrict__ corr = B.corr ? B.corr + (size_t) map * B.corr_stride : nullptr;
  const int32_t* __restrict__ imap  = B.scene_index_map ? B.scene_index_map + (size_t) map * B.capacity : nullptr;
  prs_frame_pose* poses = B.poses + (size_t) map * B.max_frames;
  SmootherItem* work    = EST == PRS_EST_SMOOTHER ? static_cast<SmootherItem*>(a.work) + (size_t) map * 2 * B.corr_stride : nullptr;
#define MG_STAMP(i)                                                          \
  do {                                                                       \
    if (a.stamps && tid == 0) {                                              \
      a.stamps[(size_t) map * 16 + (i)] = (unsigned long long) clock64();    \
    }                                                                        \
  } while (0)
  MG_STAMP(0);

  // ---- setTransforms (landmark_estimator_base.hpp:47-56) + this frame's row of the pose table ---------
  if (tid == 0) {
    float Tw[16], Ts[16], Wi[16], Wl[16];
    for (int i = 0; i < 16; ++i) {
      Tw[i] = B.measurement_in_world[(size_t) map * 16 + i];
      Ts[i] = B.measurement_in_scene[(size_t) map * 16 + i];
    }
    se3_inverse(Tw, Wi);
    se3_mul(Ts, Wi, Wl);
    for (int i = 0; i < 16; ++i) {
      sh.sensor_in_world[i]      = Tw[i];
      sh.world_in_sensor[i]      = Wi[i];
      sh.world_in_local_map[i]   = Wl[i];
      sh.measurement_in_scene[i] = Ts[i];
    }
    sh.error    = (frame < 0 || frame >= B.max_frames || n_points < 0 || n_points > B.capacity) ? PRS_ERR_RANGE : 0;
    sh.n_merged = 0;
    sh.base     = 0;
    sh.n_work   = 0;
    sh.n_next   = 0;
    if (!sh.error) {
      for (int i = 0; i < 12; ++i) {
        poses[frame].sensor_in_world[i] = Tw[i];
        poses[frame].world_in_sensor[i] = Wi[i];
      }
    }
  }
  for (int i = tid; i < nbins; i += kMergeThreads) {
    owner[i] = kNoBin;
    first[i] = kNoBin;
    best[i]  = 0ull;
  }
  for (int i = tid; i < (B.capacity + 31) / 32; i += kMergeThreads) {
    seen[i] = 0u;
  }
  __threadfence_block();
  __syncthreads();
  if (PHASE == 2 && a.carry[map].returned) {
    return;  // (block-uniform) the front kernel reported this frame
  }
  if (sh.error) {
    if (tid == 0 && kFront) {
      B.result[map].n_merged = 0;
      B.result[map].n_added  = 0;
      B.result[map].status   = sh.error;
      if (PHASE == 1) {
        a.carry[map] = MergeCarry{0, 0, sh.error, 1};
      }
    }
    return;
  }

  if (EST == PRS_EST_SMOOTHER && PHASE == 0) {
    // every frame of the pose table (this frame's row was written above): world_in_sensor and camera_matrix * R
    for (int f = tid; f < B.max_frames; f += kMergeThreads) {
      const float* W  = f == frame ? sh.world_in_sensor : poses[f].world_in_sensor;
      const float* Km = P.estimator.camera_matrix;
      float* c        = pose_cache + 21 * f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        c[i] = W[i];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {  // landmark_estimator_pose_based_smoother_impl.cpp:89
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          c[12 + 3 * i + j] = (Km[3 * i + 0] * W[0 + j] + Km[3 * i + 1] * W[4 + j]) + Km[3 * i + 2] * W[8 + j];
        }
      }
    }
    __syncthreads();
  }

  MG_STAMP(1);
  // ---- correspondences: inlier reset, appearance gate, first-come bin blocking (:59-122) --------------
  for (int c = tid; c < n_corr; c += kMergeThreads) {
    const prs_corr cr = corr[c];
    int s             = B.corr_from_aligner ? cr.moving_idx : cr.fixed_idx;
    if (s >= 0 && s < B.capacity && imap) {
      s = imap[s];
    }
    const int m = B.corr_from_aligner ? cr.fixed_idx : cr.moving_idx;
    if (s < 0 || s >= n_points || m < 0 || m >= n_meas) {
      sh.error = PRS_ERR_RANGE;
      continue;
    }
    if (atomicOr(&seen[s >> 5], 1u << (s & 31)) & (1u << (s & 31))) {
      sh.error = PRS_ERR_DUPLICATE;
      continue;
    }
    if (kFront) {
      B.inlier[(size_t) map * B.capacity + s] = 0;  // :64
    }
    if (cr.response > P.maximum_distance_appearance) {  // :70-73
      continue;
    }
    if (P.enable_binning) {
      const float4 z    = zs[m];
      const uint32_t br = bin_of(z.y, a.row_w), bc = bin_of(z.x, a.col_w);
      if (br >= (uint32_t) a.nbr || bc >= (uint32_t) a.nbc) {
        sh.error = PRS_ERR_RANGE;
        continue;
      }
      atomicMin(&owner[br * a.nbc + bc], (uint32_t) c);
    }
  }
  __syncthreads();
  if (sh.error) {
    if (tid == 0 && kFront) {
      B.result[map].n_merged = 0;
      B.result[map].n_added  = 0;
      B.result[map].status   = sh.error;
      if (PHASE == 1) {
        a.carry[map] = MergeCarry{0, 0, sh.error, 1};
      }
    }
    return;
  }

  MG_STAMP(2);
  // ---- _updatePoint for every correspondence that owns its bin (:124-128, :192-208) -----------------------
  for (int c = tid; kFront && c < n_corr; c += kMergeThreads) {
    const prs_corr cr = corr[c];
    if (cr.response > P.maximum_distance_appearance) {
      continue;
    }
    const int sc   = B.corr_from_aligner ? cr.moving_idx : cr.fixed_idx;
    const int s    = imap ? imap[sc] : sc;
    const int m    = B.corr_from_aligner ? cr.fixed_idx : cr.moving_idx;
    const float4 z = zs[m];
    if (P.enable_binning) {
      const uint32_t br = bin_of(z.y, a.row_w), bc = bin_of(z.x, a.col_w);
      if (owner[br * a.nbc + bc] != (uint32_t) c) {
        continue;  // :104-119 a correspondence visited earlier blocks the bin
      }
    }
    float lis[3] = {0.0f, 0.0f, 0.0f};
    if (P.variant == PRS_MERGER_STEREO_TRIANGULATION) {  // merger_projective_rigid_stereo_triangulation_impl.cpp:15-35
      if (!triangulate_one(P.triangulator, z, lis)) {
        continue;
      }
    }
    const Landmark l   = landmark_at(B, map, s);
    const float zv[4]  = {z.x, z.y, z.z, z.w};
    int ok;
    if (EST == PRS_EST_WEIGHTED_MEAN) {
      ok = estimate_weighted_mean(P.estimator, sh, l, lis);
    } else if (EST == PRS_EST_EKF) {
      ok = estimate_ekf<DIM>(P.estimator, sh, l, zv);
    } else {
      SmootherItem item;
      item.s = s;
      item.m = m;
      ok     = smoother_begin(P.estimator, sh, poses, frame, B.max_measurements, l, zv, lis, item);
      if (ok == 2) {
        work[atomicAdd(&sh.n_work, 1)] = item;  // iterated below, in rounds
        continue;
      }
    }
    if (ok < 0) {
      sh.error = ok;
    } else if (ok) {  // merger_projective_impl.cpp:203-207
      const uint4* src = reinterpret_cast<const uint4*>(zdesc + 32 * (size_t) m);
      uint4* dst       = reinterpret_cast<uint4*>(l.desc);
      dst[0]           = src[0];
      dst[1]           = src[1];
      atomicAdd(&sh.n_merged, 1);
    }
  }
  __syncthreads();
  MG_STAMP(3);
  if (PHASE == 1) {
    if (tid == 0) {
      a.carry[map] = MergeCarry{sh.n_merged, sh.n_work, sh.error, 0};  // continued by smoother_kernel and PHASE 2
    }
    return;
  }
  if (PHASE == 2) {
    if (tid == 0) {
      sh.n_merged = a.carry[map].n_merged;
      sh.error    = a.carry[map].error;
    }
    __syncthreads();
  }
  if (EST == PRS_EST_SMOOTHER && PHASE == 0) {
    constexpr int kRoundIterations = 8;
    SmootherItem* cur = work;
    SmootherItem* nxt = work + B.corr_stride;
    int n_work        = sh.n_work;
    while (n_work > 0) {  // block-uniform
      for (int i = tid; i < n_work; i += kMergeThreads) {
        SmootherItem item = cur[i];
        const Landmark l  = landmark_at(B, map, item.s);
        if (smoother_iterate(P.estimator, pose_cache, l, item, kRoundIterations)) {
          if (smoother_finish(sh.world_in_local_map, poses, l, item)) {  // merger_projective_impl.cpp:203-207
            const uint4* src = reinterpret_cast<const uint4*>(zdesc + 32 * (size_t) item.m);
            uint4* dst       = reinterpret_cast<uint4*>(l.desc);
            dst[0]           = src[0];
            dst[1]           = src[1];
            atomicAdd(&sh.n_merged, 1);
          }
        } else {
          nxt[atomicAdd(&sh.n_next, 1)] = item;
        }
      }
      __syncthreads();
      n_work = sh.n_next;
      __syncthreads();
      if (tid == 0) {
        sh.n_next = 0;
      }
      SmootherItem* t = cur;
      cur             = nxt;
      nxt             = t;
      __syncthreads();
    }
  }
  MG_STAMP(4);
  const int n_merged = sh.n_merged;
  int status         = 0;
  if (n_corr > 0) {  // :137-150
    if (n_merged == 0) {
      status |= PRS_WARN_NO_MATCHES;
    } else if ((float) n_merged / (float) n_corr < P.target_merge_ratio) {
      status |= PRS_WARN_LOW_RATIO;
    }
  }

  // ---- _addPoints (:55-57, :154-161, :210-305) -------------------------------------------------------------
  int n_added = 0;
  if (!sh.error && (n_corr == 0 || ((uint32_t) n_merged < P.target_number_of_merges && n_merged < n_meas))) {
    if (P.enable_binning) {
      // per free bin: the first measurement (it fixes the bin's place in the output) and the best one
      for (int i = tid; i < n_meas; i += kMergeThreads) {
        const float4 z    = zs[i];
        const uint32_t br = bin_of(z.y, a.row_w), bc = bin_of(z.x, a.col_w);
        if (br >= (uint32_t) a.nbr || bc >= (uint32_t) a.nbc) {
          sh.error = PRS_ERR_RANGE;
          continue;
        }
        const uint32_t bin = br * a.nbc + bc;
        if (owner[bin] != kNoBin) {
          continue;  // :236-241 occupied by a tracked point
        }
        atomicMin(&first[bin], (uint32_t) i);
        // _isBetterForAddition: strictly larger disparity (merger_projective_rigid_stereo_impl.cpp:45-57) /
        // strictly smaller depth (merger_projective_depth_ekf_impl.cpp:50-57) replaces the occupant, so the
        // bin ends with the earliest measurement among the best
        const float q = P.variant == PRS_MERGER_DEPTH_EKF ? -z.z : z.x - z.z;
        const unsigned long long key = ((unsigned long long) float_key(q) << 32) | (unsigned long long) (0xffffffffu - (uint32_t) i);
        atomicMax(&best[bin], key);
      }
      __syncthreads();
    }
    MG_STAMP(5);
    // candidates in the reference's order, triangulated / unprojected, valid ones appended (:267-299)
    for (int i0 = 0; i0 < n_meas && !sh.error; i0 += kMergeThreads) {
      const int i = i0 + tid;
      bool valid  = false;
      int cand    = -1;
      float p[3]  = {0.0f, 0.0f, 0.0f};
      if (i < n_meas) {
        if (!P.enable_binning) {
          cand = i;  // :262-264
        } else {
          const float4 z    = zs[i];
          const uint32_t br = bin_of(z.y, a.row_w), bc = bin_of(z.x, a.col_w);
          const uint32_t bin = br * a.nbc + bc;
          if (br < (uint32_t) a.nbr && bc < (uint32_t) a.nbc && owner[bin] == kNoBin && first[bin] == (uint32_t) i) {
            cand = (int) (0xffffffffu - (uint32_t) (best[bin] & 0xffffffffull));
          }
        }
        if (cand >= 0) {
          const float4 z = zs[cand];
          if (P.variant == PRS_MERGER_DEPTH_EKF) {
            const float d = z.z;
            valid         = d > 0.0f;
            p[0]          = (z.x - P.cx) / P.fx * d;
            p[1]          = (z.y - P.cy) / P.fy * d;
            p[2]          = d;
          } else {
            valid = triangulate_one(P.triangulator, z, p);
          }
        }
      }
      const unsigned long long bal = __ballot(valid);
      if (lane == 0) {
        wave_tot[wave] = __popcll(bal);
      }
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kMergeWaves; ++w) {
        before += w < wave ? wave_tot[w] : 0;
        total += wave_tot[w];
      }
      if (valid) {
        const int idx = n_points + n_added + before + __popcll(bal & ((1ull << lane) - 1ull));
        if (idx >= B.capacity) {
          sh.error = PRS_ERR_SCENE_FULL;
        } else {
          // _initializeLandmark (:308-326) + move into the scene frame (:282-286)
          const Landmark l = landmark_at(B, map, idx);
          float w3[3], loc[3];
          apply_rows(sh.sensor_in_world, p, w3);
          l.state[0] = w3[0];
          l.state[1] = w3[1];
          l.state[2] = w3[2];
          l.state[3] = 0.0f;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            l.cov[k] = (k % 4 == 0) ? 1.0f : 0.0f;
          }
          *l.inlier = 1;
          *l.n_opt  = 0;
          *l.n_meas = 0;
          const float4 z = zs[cand];
          if (l.meas && B.max_measurements > 0) {
            prs_camera_measurement nm;
            nm.point_in_image[0]  = z.x;
            nm.point_in_image[1]  = z.y;
            nm.point_in_image[2]  = z.z;
            nm.point_in_camera[0] = p[0];
            nm.point_in_camera[1] = p[1];
            nm.point_in_camera[2] = p[2];
            nm.frame              = frame;
            l.meas[0]             = nm;
            *l.n_meas             = 1;
          }
          apply_rows(sh.measurement_in_scene, p, loc);
          l.coords[0] = loc[0];
          l.coords[1] = loc[1];
          l.coords[2] = loc[2];
          l.coords[3] = 0.0f;
          const uint4* src = reinterpret_cast<const uint4*>(zdesc + 32 * (size_t) cand);
          uint4* dst       = reinterpret_cast<uint4*>(l.desc);
          dst[0]           = src[0];
          dst[1]           = src[1];
        }
      }
      n_added += total;
      __syncthreads();
    }
  }
  __syncthreads();
  MG_STAMP(6);
#undef MG_STAMP
  if (tid == 0) {
    if (sh.error) {
      B.result[map].n_merged = n_merged;
      B.result[map].n_added  = 0;
      B.result[map].status   = sh.error;
    } else {
      B.n_points[map]        = n_points + n_added;
      B.result[map].n_merged = n_merged;
      B.result[map].n_added  = n_added;
      B.result[map].status   = status;
    }
  }
}

// The optimisation loops of one frame's queued landmarks (PHASE 1 of merge_kernel left them in `work`): ONE wave per
// frame, 1.5 kB of LDS, so that sixteen frames are in flight per CU.  Rounds of eight iterations; landmarks whose
// loop has not ended are compacted into the other list (same scheme as the fused kernel).
constexpr int kSmootherThreads = 64;
__global__ __launch_bounds__(kSmootherThreads) void smoother_kernel(const MergeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MergeShared& sh   = *reinterpret_cast<MergeShared*>(smem);
  float* pose_cache = reinterpret_cast<float*>(smem + ((sizeof(MergeShared) + 15) & ~(size_t) 15));  // [max_frames][21]
  float* ring       = pose_cache + (((size_t) a.b.max_frames * 21 + 3) & ~(size_t) 3);                // [kSmootherRing][5][64]
  const int lane    = threadIdx.x;
  const int map     = blockIdx.x;
  const prs_merge_batch& B   = a.b;
  const prs_merger_params& P = a.p;
  MergeCarry* carry          = a.carry + map;
  int n_work                 = carry->n_work;
  if (carry->returned || n_work <= 0) {
    return;  // (uniform)
  }
  const int frame       = B.frame[map];
  prs_frame_pose* poses = B.poses + (size_t) map * B.max_frames;
  const uint8_t* __restrict__ zdesc = B.measurement_desc + (size_t) map * B.measurement_stride * 32;
  SmootherItem* cur = static_cast<SmootherItem*>(a.work) + (size_t) map * 2 * B.corr_stride;
  SmootherItem* nxt = cur + B.corr_stride;
  if (lane == 0) {  // setTransforms (landmark_estimator_base.hpp:47-56), as in merge_kernel
    float Tw[16], Ts[16], Wi[16], Wl[16];
    for (int i = 0; i < 16; ++i) {
      Tw[i] = B.measurement_in_world[(size_t) map * 16 + i];
      Ts[i] = B.measurement_in_scene[(size_t) map * 16 + i];
    }
    se3_inverse(Tw, Wi);
    se3_mul(Ts, Wi, Wl);
    for (int i = 0; i < 16; ++i) {
      sh.sensor_in_world[i]      = Tw[i];
      sh.world_in_sensor[i]      = Wi[i];
      sh.world_in_local_map[i]   = Wl[i];
      sh.measurement_in_scene[i] = Ts[i];
    }
    sh.n_next = 0;
  }
  __syncthreads();
  for (int f = lane; f < B.max_frames; f += kSmootherThreads) {
    const float* W  = f == frame ? sh.world_in_sensor : poses[f].world_in_sensor;
    const float* Km = P.estimator.camera_matrix;
    float* c        = pose_cache + 21 * f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      c[i] = W[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // landmark_estimator_pose_based_smoother_impl.cpp:89
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        c[12 + 3 * i + j] = (Km[3 * i + 0] * W[0 + j] + Km[3 * i + 1] * W[4 + j]) + Km[3 * i + 2] * W[8 + j];
      }
    }
  }
  __syncthreads();
  constexpr int kRoundIterations = 8;
  int merged = 0;
  int round  = 0;
  while (n_work > 0) {  // uniform
    if (a.tail && round >= kTailRounds && n_work <= kTailPerFrame) {
      // the rest goes to the tail kernel (capacity: kTailPerFrame entries per frame, so the list cannot overflow)
      int base = 0;
      if (lane == 0) {
        base = atomicAdd(a.tail_count, n_work);
      }
      base = __shfl(base, 0, 64);
      if (lane < n_work) {
        TailItem t;
        t.map              = map;
        t.item             = cur[lane];
        a.tail[base + lane] = t;
      }
      break;
    }
    ++round;
    for (int i = lane; i < n_work; i += kSmootherThreads) {
      SmootherItem item = cur[i];
      const Landmark l  = landmark_at(B, map, item.s);
      if (smoother_iterate(P.estimator, pose_cache, l, item, kRoundIterations, ring, lane)) {
        if (smoother_finish(sh.world_in_local_map, poses, l, item)) {  // merger_projective_impl.cpp:203-207
          const uint4* src = reinterpret_cast<const uint4*>(zdesc + 32 * (size_t) item.m);
          uint4* dst       = reinterpret_cast<uint4*>(l.desc);
          dst[0]           = src[0];
          dst[1]           = src[1];
          ++merged;
        }
      } else {
        nxt[atomicAdd(&sh.n_next, 1)] = item;
      }
    }
    __threadfence_block();
    __syncthreads();
    n_work = sh.n_next;
    __syncthreads();
    if (lane == 0) {
      sh.n_next = 0;
    }
    SmootherItem* t = cur;
    cur             = nxt;
    nxt             = t;
    __syncthreads();
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    merged += __shfl_xor(merged, d, 64);
  }
  if (lane == 0) {
    carry->n_merged += merged;
  }
}

// The stragglers of all frames (landmarks whose loop neither converged nor entered a short cycle within the per-frame
// kernel's rounds): 64 per wave regardless of their frame, every lane runs its landmark to the end.  Pose rows come
// from the maps' pose tables in global memory; the result (merged count) goes to the frame's carry.
__global__ __launch_bounds__(kSmootherThreads) void smoother_tail_kernel(const MergeArgs a) {
  __shared__ float ring[kSmootherRing * 5 * 64];
  const int lane = threadIdx.x;
  const prs_merge_batch& B   = a.b;
  const prs_merger_params& P = a.p;
  const int count = *a.tail_count;
  for (int idx = blockIdx.x * kSmootherThreads + lane; idx < count; idx += gridDim.x * kSmootherThreads) {
    const TailItem t  = a.tail[idx];
    const int map     = t.map;
    SmootherItem item = t.item;
    const Landmark l  = landmark_at(B, map, item.s);
    prs_frame_pose* poses = B.poses + (size_t) map * B.max_frames;
    while (!smoother_iterate<false>(P.estimator, reinterpret_cast<const float*>(poses), l, item, kSmootherRing, ring, lane)) {
    }
    // setTransforms (landmark_estimator_base.hpp:47-56): world_in_local_map = measurement_in_scene * sensor_in_world^-1
    float Tw[16], Ts[16], Wi[16], Wl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      Tw[i] = B.measurement_in_world[(size_t) map * 16 + i];
      Ts[i] = B.measurement_in_scene[(size_t) map * 16 + i];
    }
    se3_inverse(Tw, Wi);
    se3_mul(Ts, Wi, Wl);
    if (smoother_finish(Wl, poses, l, item)) {  // merger_projective_impl.cpp:203-207
      const uint8_t* zdesc = B.measurement_desc + (size_t) map * B.measurement_stride * 32;
      const uint4* src     = reinterpret_cast<const uint4*>(zdesc + 32 * (size_t) item.m);
      uint4* dst           = reinterpret_cast<uint4*>(l.desc);
      dst[0]               = src[0];
      dst[1]               = src[1];
      atomicAdd(&a.carry[map].n_merged, 1);
    }
  }
}

static inline uint32_t mg_align16(uint32_t v) {
  return (v + 15u) & ~15u;
}

int merge_batch_launch(prs_context* ctx, const prs_merger_params* params, const prs_merge_batch* batch) {
  if (!params || !batch) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_merge_batch_run: parameters not set");
  }
  const prs_merge_batch& b = *batch;
  if (!b.coords || !b.desc || !b.state || !b.covariance || !b.n_opt || !b.inlier || !b.n_meas || !b.poses || !b.n_points) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_merge_batch_run: scene not set");
  }
  if (!b.measurement || !b.measurement_desc || !b.n_measured || !b.measurement_in_world || !b.measurement_in_scene || !b.frame || !b.result) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_merge_batch_run: measurement not set");
  }
  if (b.batch <= 0) {
    return PRS_OK;
  }
  const prs_estimator_params& e = params->estimator;
  if (params->variant < PRS_MERGER_STEREO_TRIANGULATION || params->variant > PRS_MERGER_DEPTH_EKF || e.type < PRS_EST_WEIGHTED_MEAN ||
      e.type > PRS_EST_SMOOTHER || e.measurement_dim < 2 || e.measurement_dim > 4 || b.capacity <= 0 || b.max_frames <= 0 || b.measurement_stride <= 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_merge_batch_run: unknown merger / estimator or empty strides");
  }
  // Which estimator a merger may own (mapping/instances.cpp:29-40 registers more types than any merger of mapping/mergers drives):
  //   weighted mean / pose-based smoother: the landmark's position in the sensor comes from the MERGER, and only
  //     MergerRigidStereoTriangulation computes one (merger_projective_rigid_stereo_triangulation_impl.cpp:15-35): the 4D3D forms.
  //     The 2D3D / 3D3D forms have no caller in the reference that sets it: refused, not guessed;
  //   stereo / depth filter: measurement layout and merger must agree;
  //   mono filter (ProjectivePointEKF3D, 2-D measurements): no merger of the reference adds points from (u, v) alone: updates only.
  if ((e.type == PRS_EST_WEIGHTED_MEAN || e.type == PRS_EST_SMOOTHER) && (params->variant != PRS_MERGER_STEREO_TRIANGULATION || e.measurement_dim != 4)) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED,
                    "prs_merge_batch_run: the weighted-mean / smoother estimators are served in their 4D3D form under PRS_MERGER_STEREO_TRIANGULATION only "
                    "(no merger of the reference provides landmark_in_sensor for 2-D / 3-D measurements)");
  }
  if (e.type == PRS_EST_EKF && ((e.measurement_dim == 4 && params->variant == PRS_MERGER_DEPTH_EKF) ||
                                (e.measurement_dim == 3 && params->variant != PRS_MERGER_DEPTH_EKF))) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_merge_batch_run: 4-D measurements need a stereo merger, 3-D measurements PRS_MERGER_DEPTH_EKF");
  }
  if (e.type == PRS_EST_EKF && e.measurement_dim == 2 && params->target_number_of_merges != 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED,
                    "prs_merge_batch_run: the mono filter updates landmarks only (target_number_of_merges must be 0: a point cannot be added from (u, v))");
  }
  if (e.type == PRS_EST_SMOOTHER && (!b.meas || b.max_measurements <= 0 || e.measurement_dim < 3)) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_merge_batch_run: the pose-based smoother needs the measurement history");
  }
  if (params->number_of_row_bins == 0 || params->number_of_col_bins == 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_merge_batch_run: bin counts must be positive");
  }
  MergeArgs a;
  a.p     = *params;
  a.b     = b;
  a.row_w = (float) params->canvas_rows / (float) params->number_of_row_bins;  // merger_projective_impl.cpp:30-33
  a.col_w = (float) params->canvas_cols / (float) params->number_of_col_bins;
  if (a.row_w < 1.0f || a.col_w < 1.0f) {  // :35-47 throws
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_merge_batch_run: bin width must be at least 1 pixel");
  }
  a.nbr = (int) params->number_of_row_bins + 2;
  a.nbc = (int) params->number_of_col_bins + 2;
  const uint32_t nbins = (uint32_t) a.nbr * (uint32_t) a.nbc;
  uint32_t off = 0;
  a.off_owner = off; off = mg_align16(off + nbins * 4);
  a.off_first = off; off = mg_align16(off + nbins * 4);
  a.off_best  = off; off = mg_align16(off + nbins * 8);
  a.off_seen  = off; off = mg_align16(off + ((uint32_t) b.capacity + 31) / 32 * 4);
  a.off_sh    = off; off = mg_align16(off + (uint32_t) sizeof(MergeShared));
  a.off_wave  = off; off = mg_align16(off + kMergeWaves * 4);
  a.off_pose_cache = off; off = mg_align16(off + (e.type == PRS_EST_SMOOTHER ? (uint32_t) b.max_frames * 21 * 4 : 0));
  if (off > 160u * 1024u) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_merge_batch_run: bin table / scene do not fit the 160 KiB LDS");
  }
  a.stamps = ctx_stamps(ctx, (size_t) b.batch * 16 * sizeof(unsigned long long));
  a.work   = nullptr;
  if (e.type == PRS_EST_SMOOTHER) {
    if (!b.corr || b.corr_stride <= 0) {
      // no correspondences at all: nothing to iterate, but the pointer must be valid
      a.work = ctx_device_scratch_slot(ctx, 0, 64);
    } else {
      a.work = ctx_device_scratch_slot(ctx, 0, (size_t) b.batch * 2 * (size_t) b.corr_stride * sizeof(SmootherItem));
    }
    if (!a.work) {
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_merge_batch_run: smoother work list allocation failed");
    }
  }
  hipError_t e2 = hipSuccess;
  auto launch = [&](auto kernel) {
    if (off > 64u * 1024u) {
      e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) off);
    }
    if (e2 == hipSuccess) {
      hipLaunchKernelGGL(kernel, dim3(b.batch), dim3(kMergeThreads), off, ctx_stream(ctx), a);
      e2 = hipGetLastError();
    }
  };
  a.carry         = nullptr;
  a.tail          = nullptr;
  a.tail_count    = nullptr;
  a.tail_capacity = 0;
  if (e.type == PRS_EST_WEIGHTED_MEAN) {
    launch(merge_kernel<PRS_EST_WEIGHTED_MEAN, 4, 0>);
  } else if (e.type == PRS_EST_SMOOTHER && (ctx->merge_fused || a.stamps)) {
    launch(merge_kernel<PRS_EST_SMOOTHER, 4, 0>);
  } else if (e.type == PRS_EST_SMOOTHER) {
    a.carry = static_cast<MergeCarry*>(ctx_device_scratch_slot(ctx, 1, (size_t) b.batch * sizeof(MergeCarry)));
    if (!a.carry) {
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_merge_batch_run: carry allocation failed");
    }
    // tail list: [batch * kTailPerFrame] items + the counter in front of them
    a.tail_capacity    = b.batch * kTailPerFrame;
    unsigned char* tmem = static_cast<unsigned char*>(ctx_device_scratch_slot(ctx, 2, 256 + (size_t) a.tail_capacity * sizeof(TailItem)));
    if (!tmem) {
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_merge_batch_run: tail list allocation failed");
    }
    a.tail_count = reinterpret_cast<int*>(tmem);
    a.tail       = reinterpret_cast<TailItem*>(tmem + 256);
    (void) hipMemsetAsync(a.tail_count, 0, sizeof(int), ctx_stream(ctx));
    launch(merge_kernel<PRS_EST_SMOOTHER, 4, 1>);
    if (e2 == hipSuccess) {
      const size_t lds_s = ((sizeof(MergeShared) + 15) & ~(size_t) 15) + (((size_t) b.max_frames * 21 + 3) & ~(size_t) 3) * sizeof(float) +
                           (size_t) kSmootherRing * 5 * 64 * sizeof(float);
      hipLaunchKernelGGL(smoother_kernel, dim3(b.batch), dim3(kSmootherThreads), lds_s, ctx_stream(ctx), a);
      const int tail_waves = b.batch < 2048 ? b.batch : 2048;  // grid-stride over however many stragglers there are
      hipLaunchKernelGGL(smoother_tail_kernel, dim3(tail_waves), dim3(kSmootherThreads), 0, ctx_stream(ctx), a);
      e2 = hipGetLastError();
    }
    launch(merge_kernel<PRS_EST_SMOOTHER, 4, 2>);
  } else if (e.measurement_dim == 4) {
    launch(merge_kernel<PRS_EST_EKF, 4, 0>);
  } else if (e.measurement_dim == 3) {
    launch(merge_kernel<PRS_EST_EKF, 3, 0>);
  } else {
    launch(merge_kernel<PRS_EST_EKF, 2, 0>);
  }
  if (e2 != hipSuccess) {
    return ctx_fail_hip(ctx, e2, "prs_merge_batch_run launch");
  }
  if (a.stamps) {
    ctx_report_stamps(ctx, b.batch, 7, "merge: setup + pose cache | correspondences (gate, bins) | update / smoother start | smoother rounds | addition bins | additions");
  }
  return PRS_OK;
}

// pose_out = prediction * X^-1 (tracker pose update between aligner and merger)
__global__ __launch_bounds__(256) void pose_compose_kernel(int batch, const float* __restrict__ prediction, const float* __restrict__ X,
                                                           float* __restrict__ pose_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) {
    return;
  }
  float P[16], Xm[16], Xi[16], R[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    P[i]  = prediction[(size_t) b * 16 + i];
    Xm[i] = X[(size_t) b * 16 + i];
  }
  se3_inverse(Xm, Xi);
  se3_mul(P, Xi, R);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pose_out[(size_t) b * 16 + i] = R[i];
  }
}

// MotionModelConstantVelocity3D: pose_pred = pose_prev1 * (pose_prev2^-1 * pose_prev1)
__global__ __launch_bounds__(256) void motion_predict_kernel(int batch, const float* __restrict__ prev2, const float* __restrict__ prev1,
                                                             float* __restrict__ pred) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) {
    return;
  }
  float P2[16], P1[16], I2[16], M[16], raw[16], R[16], v[6];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    P2[i] = prev2[(size_t) b * 16 + i];
    P1[i] = prev1[(size_t) b * 16 + i];
  }
  se3_inverse(P2, I2);
  se3_mul(I2, P1, M);
  se3_mul(P1, M, raw);
  // through the unit quaternion: the recursion would otherwise amplify the rotation block's drift from orthonormality
  // by ~2.4x per frame (se3_inverse transposes)
  t2tnq(raw, v);
  tnq2t(v, R);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    pred[(size_t) b * 16 + i] = R[i];
  }
}

int motion_predict_launch(prs_context* ctx, int batch, const float* prev2, const float* prev1, float* pred) {
  if (!prev2 || !prev1 || !pred) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_motion_predict_batch: pose arrays not set");
  }
  if (batch <= 0) {
    return PRS_OK;
  }
  hipLaunchKernelGGL(motion_predict_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx_stream(ctx), batch, prev2, prev1, pred);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_motion_predict_batch launch");
  }
  return PRS_OK;
}

int pose_compose_launch(prs_context* ctx, int batch, const float* prediction, const float* X, float* pose_out) {
  if (!prediction || !X || !pose_out) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_pose_compose_batch: pose arrays not set");
  }
  if (batch <= 0) {
    return PRS_OK;
  }
  hipLaunchKernelGGL(pose_compose_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx_stream(ctx), batch, prediction, X, pose_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_pose_compose_batch launch");
  }
  return PRS_OK;
}

}  // namespace prs
