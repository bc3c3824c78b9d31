/* proslam_oracle_mapping.c -- see proslam_oracle_mapping.h.  Plain sequential C99, compiled with
 * -ffp-contract=off: every product / sum below is evaluated exactly in the order written. */
#include "proslam_oracle_mapping.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- small helpers ------------------------------------------------------------------------- */
static void apply44(const float* T /* 4x4 row-major */, const float* p, float* out) {
  for (int i = 0; i < 3; ++i) {
    out[i] = ((T[4 * i + 0] * p[0] + T[4 * i + 1] * p[1]) + T[4 * i + 2] * p[2]) + T[4 * i + 3];
  }
}
static void apply34(const float* T /* 3x4 row-major */, const float* p, float* out) {
  apply44(T, p, out);
}
static float sqnorm3(const float* a, const float* b) {
  const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return (d0 * d0 + d1 * d1) + d2 * d2;
}

/* setTransforms (landmark_estimator_base.hpp:47-56) */
typedef struct {
  float sensor_in_world[16];
  float world_in_sensor[16];
  float world_in_local_map[16];
} estimator_transforms;

static void set_transforms(const float* measurement_in_world, const float* measurement_in_scene, estimator_transforms* t) {
  memcpy(t->sensor_in_world, measurement_in_world, sizeof(float) * 16);
  orc_se3_inverse(t->sensor_in_world, t->world_in_sensor);
  orc_se3_mul(measurement_in_scene, t->world_in_sensor, t->world_in_local_map); /* :55 */
}

static void fill_pose(const estimator_transforms* t, orc_frame_pose* pose) {
  memcpy(pose->sensor_in_world, t->sensor_in_world, sizeof(float) * 12);
  memcpy(pose->world_in_sensor, t->world_in_sensor, sizeof(float) * 12);
}

/* PointStatisticsField3D::addOptimizationResult (BUILD-DEFINED) */
static void add_optimization_result(orc_map* map, int idx, const float* coords_world, const float* cov9) {
  memcpy(map->state + 4 * (size_t) idx, coords_world, sizeof(float) * 3);
  if (cov9) {
    memcpy(map->covariance + 9 * (size_t) idx, cov9, sizeof(float) * 9);
  }
  ++map->n_opt[idx];
}

/* ---- LandmarkEstimatorWeightedMean_::compute (landmark_estimator_weighted_mean_impl.cpp:7-41) -- */
static int estimate_weighted_mean(const orc_estimator_params* P, const estimator_transforms* t, orc_map* map, int idx,
                                  const float* landmark_in_sensor) {
  map->inlier[idx]   = 0; /* :13 */
  const float* init  = map->state + 4 * (size_t) idx;
  float upd[3];
  apply44(t->sensor_in_world, landmark_in_sensor, upd);            /* :20-21 */
  const float npo = (float) (map->n_opt[idx] + 1u);                /* :23 */
  float cw[3];
  for (int i = 0; i < 3; ++i) {
    cw[i] = (npo * init[i] + upd[i]) / (npo + 1.0f);               /* :25-27 */
  }
  if (sqnorm3(cw, init) > P->maximum_distance_geometry_meters_squared) { /* :30-34 */
    return 0;
  }
  add_optimization_result(map, idx, cw, NULL);                     /* :37 */
  map->inlier[idx] = 1;
  apply44(t->world_in_local_map, cw, map->coords + 4 * (size_t) idx); /* :41 */
  return 1;
}

/* ---- PointEKFBase + the three measurement models, in double ---------------------------------- */
/* S^-1 of a symmetric positive definite dim x dim matrix through LDL^T (BUILD-DEFINED) */
static void spd_inverse(const double* S, int n, double* Sinv) {
  double L[16], D[4];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < n; ++j) {
    double d = S[n * j + j];
    for (int k = 0; k < j; ++k) {
      d -= (L[4 * j + k] * L[4 * j + k]) * D[k];
    }
    D[j]         = d;
    L[4 * j + j] = 1.0;
    for (int i = j + 1; i < n; ++i) {
      double v = S[n * i + j];
      for (int k = 0; k < j; ++k) {
        v -= (L[4 * i + k] * L[4 * j + k]) * D[k];
      }
      L[4 * i + j] = v / d;
    }
  }
  for (int c = 0; c < n; ++c) {
    double y[4], x[4];
    for (int i = 0; i < n; ++i) { /* L y = e_c */
      double v = i == c ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) {
        v -= L[4 * i + k] * y[k];
      }
      y[i] = v;
    }
    for (int i = 0; i < n; ++i) {
      y[i] = y[i] / D[i];
    }
    for (int i = n - 1; i >= 0; --i) { /* L^T x = y */
      double v = y[i];
      for (int k = i + 1; k < n; ++k) {
        v -= L[4 * k + i] * x[k];
      }
      x[i] = v;
    }
    for (int i = 0; i < n; ++i) {
      Sinv[n * i + c] = x[i];
    }
  }
}

/* _computeMeasurementPrediction: stereo_projective_point_ekf_impl.cpp:13-48 (dim 4),
 * projective_depth_point_ekf_impl.cpp:7-36 (dim 3), projective_point_ekf_impl.cpp:16-43 (dim 2) */
static void ekf_prediction(const orc_estimator_params* P, int dim, const double* st, double* h, double* J /* dim x 3 */) {
  const double x = st[0], y = st[1], z = st[2];
  const double z_2     = z * z;
  const double fx_x    = P->fx * x;
  const double fy_y    = P->fy * y;
  const double fx_by_z = P->fx / z;
  const double fy_by_z = P->fy / z;
  if (dim == 4) {
    const double x_h = fx_x + P->cx * z;
    const double y_h = fy_y + P->cy * z;
    h[0] = x_h / z;
    h[1] = y_h / z;
    h[2] = (x_h - P->b_x) / z;
    h[3] = (y_h - P->b_y) / z;
    const double Jd[12] = {fx_by_z, 0.0, -fx_x / z_2, 0.0, fy_by_z, -fy_y / z_2,
                           fx_by_z, 0.0, -(fx_x - P->b_x) / z_2, 0.0, fy_by_z, -(fy_y - P->b_y) / z_2};
    memcpy(J, Jd, sizeof(Jd));
  } else {
    h[0] = fx_by_z * x + P->cx;
    h[1] = fy_by_z * y + P->cy;
    const double Jd[9] = {fx_by_z, 0.0, -fx_x / z_2, 0.0, fy_by_z, -fy_y / z_2, 0.0, 0.0, 1.0};
    memcpy(J, Jd, sizeof(double) * 3 * (size_t) dim);
    if (dim == 3) {
      h[2] = z;
    }
  }
}

/* LandmarkEstimatorEKF_::compute (landmark_estimator_ekf_impl.cpp:17-82) */
static int estimate_ekf(const orc_estimator_params* P, const estimator_transforms* t, orc_map* map, int idx, const float* measurement) {
  const int dim    = P->measurement_dim;
  map->inlier[idx] = 0; /* :24 */
  const float* init = map->state + 4 * (size_t) idx;
  double st[3], cov[9];
  for (int i = 0; i < 3; ++i) {
    st[i] = (double) init[i];
  }
  for (int i = 0; i < 9; ++i) {
    cov[i] = (double) map->covariance[9 * (size_t) idx + i];
  }
  for (int i = 0; i < 3; ++i) { /* :48-50 */
    cov[4 * i] = cov[4 * i] > P->minimum_state_element_covariance ? cov[4 * i] : P->minimum_state_element_covariance;
  }
  /* _predict (point_ekf_base.hpp:63-78): covariance = F cov F^T (+ 0), state = world_in_sensor * state */
  double F[9], tr[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      F[3 * i + j] = (double) t->world_in_sensor[4 * i + j];
    }
    tr[i] = (double) t->world_in_sensor[4 * i + 3];
  }
  double M[9], covp[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      M[3 * i + j] = (F[3 * i + 0] * cov[0 + j] + F[3 * i + 1] * cov[3 + j]) + F[3 * i + 2] * cov[6 + j];
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      covp[3 * i + j] = (M[3 * i + 0] * F[3 * j + 0] + M[3 * i + 1] * F[3 * j + 1]) + M[3 * i + 2] * F[3 * j + 2];
    }
  }
  double sp[3];
  for (int i = 0; i < 3; ++i) {
    sp[i] = ((F[3 * i + 0] * st[0] + F[3 * i + 1] * st[1]) + F[3 * i + 2] * st[2]) + tr[i];
  }
  /* _correct (:81-123) */
  double h[4], J[12];
  ekf_prediction(P, dim, sp, h, J);
  double A[12]; /* J cov: dim x 3 */
  for (int i = 0; i < dim; ++i) {
    for (int j = 0; j < 3; ++j) {
      A[3 * i + j] = (J[3 * i + 0] * covp[0 + j] + J[3 * i + 1] * covp[3 + j]) + J[3 * i + 2] * covp[6 + j];
    }
  }
  double S[16], Sinv[16];
  for (int i = 0; i < dim; ++i) {
    for (int j = 0; j < dim; ++j) {
      const double a = (A[3 * i + 0] * J[3 * j + 0] + A[3 * i + 1] * J[3 * j + 1]) + A[3 * i + 2] * J[3 * j + 2];
      S[dim * i + j] = (i == j ? P->minimum_state_element_covariance : 0.0) + a; /* :28-29 measurement covariance */
    }
  }
  spd_inverse(S, dim, Sinv);
  double B[12]; /* cov J^T: 3 x dim */
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < dim; ++j) {
      B[dim * i + j] = (covp[3 * i + 0] * J[3 * j + 0] + covp[3 * i + 1] * J[3 * j + 1]) + covp[3 * i + 2] * J[3 * j + 2];
    }
  }
  double K[12]; /* 3 x dim */
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < dim; ++j) {
      double s = 0.0;
      for (int k = 0; k < dim; ++k) {
        s += B[dim * i + k] * Sinv[dim * k + j];
      }
      K[dim * i + j] = s;
    }
  }
  double r[4];
  for (int k = 0; k < dim; ++k) {
    r[k] = (double) measurement[k] - h[k];
  }
  for (int i = 0; i < 3; ++i) {
    double s = 0.0;
    for (int k = 0; k < dim; ++k) {
      s += K[dim * i + k] * r[k];
    }
    sp[i] += s;
  }
  double C[9], covc[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < dim; ++k) {
        s += K[dim * i + k] * J[3 * k + j];
      }
      C[3 * i + j] = (i == j ? 1.0 : 0.0) - s;
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      covc[3 * i + j] = (C[3 * i + 0] * covp[0 + j] + C[3 * i + 1] * covp[3 + j]) + C[3 * i + 2] * covp[6 + j];
    }
  }
  /* drop invalid state optimizations (landmark_estimator_ekf_impl.cpp:60-64) */
  double n2 = 0.0;
  for (int i = 0; i < 9; ++i) {
    n2 += covc[i] * covc[i];
  }
  if (sp[2] <= 0.0 || n2 > P->maximum_covariance_norm_squared) {
    return 0;
  }
  const float sf[3] = {(float) sp[0], (float) sp[1], (float) sp[2]};
  float cw[3];
  apply44(t->sensor_in_world, sf, cw); /* :68-69 */
  if (sqnorm3(cw, init) > P->maximum_distance_geometry_meters_squared) { /* :70-74 */
    return 0;
  }
  float covf[9];
  for (int i = 0; i < 9; ++i) {
    covf[i] = (float) covc[i];
  }
  add_optimization_result(map, idx, cw, covf); /* :77-78 */
  map->inlier[idx] = 1;
  apply44(t->world_in_local_map, cw, map->coords + 4 * (size_t) idx); /* :82-83 */
  return 1;
}

/* ---- LandmarkEstimatorPoseBasedSmoother_::compute (landmark_estimator_pose_based_smoother_impl.cpp) -- */
/* _setMeanCoordinatesInWorld (:138-147) */
static void mean_in_world(const orc_camera_measurement* m, uint32_t n, const orc_frame_pose* poses, float* out) {
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (uint32_t k = 0; k < n; ++k) {
    float w[3];
    apply34(poses[m[k].frame].sensor_in_world, m[k].point_in_camera, w);
    acc[0] += w[0];
    acc[1] += w[1];
    acc[2] += w[2];
  }
  const float fn = (float) n;
  out[0] = acc[0] / fn;
  out[1] = acc[1] / fn;
  out[2] = acc[2] / fn;
}

/* x = A^-1 rhs for a 3x3 system by elimination with full pivoting (stand-in for Eigen's fullPivLu) */
static void solve3_full_pivot(const float* A_in, const float* rhs, float* x) {
  float A[9], b[3];
  int col_of[3] = {0, 1, 2};
  memcpy(A, A_in, sizeof(A));
  memcpy(b, rhs, sizeof(b));
  for (int k = 0; k < 3; ++k) {
    int pr = k, pc = k;
    float best = fabsf(A[3 * k + k]);
    for (int i = k; i < 3; ++i) {
      for (int j = k; j < 3; ++j) {
        const float v = fabsf(A[3 * i + j]);
        if (v > best) {
          best = v;
          pr   = i;
          pc   = j;
        }
      }
    }
    if (pr != k) {
      for (int j = 0; j < 3; ++j) {
        const float tmp = A[3 * k + j];
        A[3 * k + j]    = A[3 * pr + j];
        A[3 * pr + j]   = tmp;
      }
      const float tb = b[k];
      b[k]           = b[pr];
      b[pr]          = tb;
    }
    if (pc != k) {
      for (int i = 0; i < 3; ++i) {
        const float tmp = A[3 * i + k];
        A[3 * i + k]    = A[3 * i + pc];
        A[3 * i + pc]   = tmp;
      }
      const int tc = col_of[k];
      col_of[k]    = col_of[pc];
      col_of[pc]   = tc;
    }
    const float piv = A[3 * k + k];
    for (int i = k + 1; i < 3; ++i) {
      const float f = A[3 * i + k] / piv;
      for (int j = k + 1; j < 3; ++j) {
        A[3 * i + j] -= f * A[3 * k + j];
      }
      b[i] -= f * b[k];
    }
  }
  float y[3];
  for (int i = 2; i >= 0; --i) {
    float v = b[i];
    for (int j = i + 1; j < 3; ++j) {
      v -= A[3 * i + j] * y[j];
    }
    y[i] = v / A[3 * i + i];
  }
  for (int i = 0; i < 3; ++i) {
    x[col_of[i]] = y[i];
  }
}

static int estimate_smoother(const orc_estimator_params* P, const estimator_transforms* t, const orc_frame_pose* poses, int32_t frame,
                             orc_map* map, int idx, const float* measurement, const float* landmark_in_sensor) {
  map->inlier[idx] = 0; /* :13 */
  /* addMeasurement (:16-20) */
  if (map->n_meas[idx] >= (uint32_t) map->max_measurements) {
    return ORC_ERR_HISTORY;
  }
  orc_camera_measurement* M = map->meas + (size_t) idx * (size_t) map->max_measurements;
  {
    orc_camera_measurement* nm = M + map->n_meas[idx];
    memcpy(nm->point_in_image, measurement, sizeof(float) * 3);
    memcpy(nm->point_in_camera, landmark_in_sensor, sizeof(float) * 3);
    nm->frame = frame;
    ++map->n_meas[idx];
  }
  const uint32_t n = map->n_meas[idx];
  float* state     = map->state + 4 * (size_t) idx;
  float init[3]    = {state[0], state[1], state[2]};
  float world[3]   = {state[0], state[1], state[2]};
  if (n < P->minimum_number_of_measurements_for_optimization) { /* :29-43 */
    mean_in_world(M, n, poses, world);
    if (sqnorm3(world, init) < P->maximum_distance_geometry_meters_squared) {
      apply44(t->world_in_local_map, world, map->coords + 4 * (size_t) idx);
      memcpy(state, world, sizeof(float) * 3);
      map->inlier[idx] = 1;
      map->n_opt[idx]  = n;
    }
    return map->inlier[idx];
  }
  const float* Km          = P->camera_matrix;
  const float max_kernel   = P->maximum_reprojection_error_pixels_squared;
  float total_previous     = 0.0f; /* :46 */
  uint32_t number_of_inliers = 0;
  for (uint32_t it = 0; it < P->maximum_number_of_iterations; ++it) {
    float H[9], b[3];
    memset(H, 0, sizeof(H));
    memset(b, 0, sizeof(b));
    float total_error_squared    = 0.0f;
    uint32_t number_of_outliers = 0;
    for (uint32_t k = 0; k < n; ++k) {
      float omega[3] = {1.0f, 1.0f, 10.0f}; /* :59-60 */
      const float* W = poses[M[k].frame].world_in_sensor;
      float pc[3];
      apply34(W, world, pc); /* :63 */
      if (pc[2] <= 0.0f) {
        ++number_of_outliers;
        continue;
      }
      float ph[3];
      for (int i = 0; i < 3; ++i) {
        ph[i] = (Km[3 * i + 0] * pc[0] + Km[3 * i + 1] * pc[1]) + Km[3 * i + 2] * pc[2];
      }
      const float c      = ph[2];
      const float inv_c  = 1.0f / c;
      const float inv_c2 = inv_c * inv_c;
      const float pi0 = ph[0] / c, pi1 = ph[1] / c; /* :71 */
      const float e[3] = {pi0 - M[k].point_in_image[0], pi1 - M[k].point_in_image[1], c - M[k].point_in_camera[2]}; /* :74-76 */
      const float error_squared = (e[0] * (omega[0] * e[0]) + e[1] * (omega[1] * e[1])) + e[2] * (omega[2] * e[2]);
      total_error_squared += error_squared;
      if (error_squared > max_kernel) { /* :83-86 saturated kernel */
        const float s = max_kernel / error_squared;
        omega[0] *= s;
        omega[1] *= s;
        omega[2] *= s;
        ++number_of_outliers;
      }
      float Jl[9], Jh[9], J[9];
      for (int i = 0; i < 3; ++i) { /* K * R (:89) */
        for (int j = 0; j < 3; ++j) {
          Jl[3 * i + j] = (Km[3 * i + 0] * W[0 + j] + Km[3 * i + 1] * W[4 + j]) + Km[3 * i + 2] * W[8 + j];
        }
      }
      Jh[0] = inv_c; Jh[1] = 0.0f;  Jh[2] = -ph[0] * inv_c2; /* :94-98 */
      Jh[3] = 0.0f;  Jh[4] = inv_c; Jh[5] = -ph[1] * inv_c2;
      Jh[6] = 0.0f;  Jh[7] = 0.0f;  Jh[8] = 1.0f;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
          J[3 * i + j] = (Jh[3 * i + 0] * Jl[0 + j] + Jh[3 * i + 1] * Jl[3 + j]) + Jh[3 * i + 2] * Jl[6 + j];
        }
      }
      for (int a = 0; a < 3; ++a) { /* :103-104 */
        for (int c2 = 0; c2 < 3; ++c2) {
          H[3 * a + c2] += (J[0 + a] * (omega[0] * J[0 + c2]) + J[3 + a] * (omega[1] * J[3 + c2])) + J[6 + a] * (omega[2] * J[6 + c2]);
        }
        b[a] += (J[0 + a] * (omega[0] * e[0]) + J[3 + a] * (omega[1] * e[1])) + J[6 + a] * (omega[2] * e[2]);
      }
    }
    float nb[3] = {-b[0], -b[1], -b[2]}, dx[3];
    solve3_full_pivot(H, nb, dx); /* :108 */
    world[0] += dx[0];
    world[1] += dx[1];
    world[2] += dx[2];
    number_of_inliers = n - number_of_outliers;
    if (fabsf(total_error_squared - total_previous) < P->convergence_criterion_minimum_chi2_delta) { /* :113-117 */
      break;
    }
    total_previous = total_error_squared;
  }
  if (number_of_inliers > map->n_opt[idx]) { /* :122-127 */
    add_optimization_result(map, idx, world, NULL);
    map->inlier[idx] = 1;
  } else { /* :130-135 */
    mean_in_world(M, n, poses, world);
    memcpy(state, world, sizeof(float) * 3);
  }
  apply44(t->world_in_local_map, world, map->coords + 4 * (size_t) idx); /* :138 */
  return map->inlier[idx];
}

int orc_landmark_estimate(const orc_estimator_params* P,
                          const float* measurement_in_world,
                          const float* measurement_in_scene,
                          const orc_frame_pose* poses,
                          int32_t frame,
                          orc_map* map,
                          int32_t index,
                          const float* measurement,
                          const float* landmark_in_sensor) {
  if (!P || !map || index < 0 || index >= map->n_points || !measurement) {
    return ORC_ERR_NULL;
  }
  estimator_transforms t;
  set_transforms(measurement_in_world, measurement_in_scene, &t);
  if (P->type == ORC_EST_WEIGHTED_MEAN) {
    return estimate_weighted_mean(P, &t, map, index, landmark_in_sensor);
  }
  if (P->type == ORC_EST_EKF) {
    return estimate_ekf(P, &t, map, index, measurement);
  }
  return estimate_smoother(P, &t, poses, frame, map, index, measurement, landmark_in_sensor);
}

/* ---- mergers ------------------------------------------------------------------------------------ */
/* triangulateRectifiedMidpoint (triangulator_rigid_stereo.cpp:60-85) for one measurement */
static int triangulate_one(const orc_triangulator_params* tp, const float* m /* uL vL uR vR */, float* p) {
  uint8_t valid = 0;
  orc_triangulate(m, 1, tp, p, &valid);
  return valid;
}

static uint32_t bin_of(float coordinate, float width) {
  return (uint32_t) roundf(coordinate / width); /* merger_projective_impl.cpp:84-85 */
}

/* MergerProjective_::_initializeLandmark (:308-326) + append in the scene frame (:282-286) */
static int add_landmark(orc_map* map, const estimator_transforms* t, const float* measurement_in_scene, const float* meas3,
                        const uint8_t* desc, const float* p_cam, int32_t frame) {
  if (map->n_points >= map->capacity) {
    return ORC_ERR_SCENE_FULL;
  }
  const int idx = map->n_points;
  apply44(t->sensor_in_world, p_cam, map->state + 4 * (size_t) idx);
  map->state[4 * (size_t) idx + 3] = 0.0f;
  float* cov = map->covariance + 9 * (size_t) idx;
  for (int i = 0; i < 9; ++i) {
    cov[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  }
  map->inlier[idx] = 1;
  map->n_opt[idx]  = 0;
  map->n_meas[idx] = 0;
  if (map->max_measurements > 0) {
    orc_camera_measurement* nm = map->meas + (size_t) idx * (size_t) map->max_measurements;
    memcpy(nm->point_in_image, meas3, sizeof(float) * 3);
    memcpy(nm->point_in_camera, p_cam, sizeof(float) * 3);
    nm->frame        = frame;
    map->n_meas[idx] = 1;
  }
  apply44(measurement_in_scene, p_cam, map->coords + 4 * (size_t) idx);
  map->coords[4 * (size_t) idx + 3] = 0.0f;
  memcpy(map->desc + 32 * (size_t) idx, desc, 32);
  ++map->n_points;
  return 0;
}

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
              orc_merge_result* result) {
  if (!P || !map || !poses || !result || (n_measured > 0 && (!measurement || !measurement_desc)) || (n_corr > 0 && !corr)) {
    return ORC_ERR_NULL;
  }
  memset(result, 0, sizeof(*result));
  const int dim = P->estimator.measurement_dim;
  estimator_transforms t;
  set_transforms(measurement_in_world, measurement_in_scene, &t);
  fill_pose(&t, poses + frame);
  const float row_w = (float) P->canvas_rows / (float) P->number_of_row_bins; /* :30-33 */
  const float col_w = (float) P->canvas_cols / (float) P->number_of_col_bins;
  if (row_w < 1.0f || col_w < 1.0f) {
    return ORC_ERR_NULL; /* :35-47 throws */
  }
  const uint32_t nbr = P->number_of_row_bins + 2, nbc = P->number_of_col_bins + 2;
  uint8_t* occupied = (uint8_t*) calloc((size_t) nbr * nbc, 1);
  uint8_t* seen     = (uint8_t*) calloc((size_t) (map->n_points > 0 ? map->n_points : 1), 1);
  int rc            = 0;
  int n_merged      = 0;

  if (n_corr > 0) {
    for (int ic = 0; ic < n_corr && rc == 0; ++ic) { /* :59-129 */
      int s = corr[ic].fixed_idx;
      if (scene_index_map) {
        s = scene_index_map[s];
      }
      const int m = corr[ic].moving_idx;
      if (s < 0 || s >= map->n_points || m < 0 || m >= n_measured) {
        rc = ORC_ERR_NULL;
        break;
      }
      if (seen[s]) {
        rc = ORC_ERR_DUPLICATE;
        break;
      }
      seen[s]        = 1;
      map->inlier[s] = 0;                                        /* :64 */
      if (corr[ic].response > P->maximum_distance_appearance) { /* :70-73 */
        continue;
      }
      const float* z = measurement + (size_t) dim * (size_t) m;
      if (P->enable_binning) { /* :89-122 */
        const uint32_t br = bin_of(z[1], row_w), bc = bin_of(z[0], col_w);
        if (br >= nbr || bc >= nbc) {
          rc = ORC_ERR_NULL;
          break;
        }
        if (occupied[br * nbc + bc]) {
          continue;
        }
        occupied[br * nbc + bc] = 1;
      }
      /* _updatePoint */
      float lis[3] = {0.0f, 0.0f, 0.0f};
      if (P->variant == ORC_MERGER_STEREO_TRIANGULATION) { /* merger_projective_rigid_stereo_triangulation_impl.cpp:15-35 */
        if (z[0] - z[2] < P->triangulator.minimum_disparity_pixels) {
          continue;
        }
        triangulate_one(&P->triangulator, z, lis);
      }
      int ok;
      if (P->estimator.type == ORC_EST_WEIGHTED_MEAN) {
        ok = estimate_weighted_mean(&P->estimator, &t, map, s, lis);
      } else if (P->estimator.type == ORC_EST_EKF) {
        ok = estimate_ekf(&P->estimator, &t, map, s, z);
      } else {
        ok = estimate_smoother(&P->estimator, &t, poses, frame, map, s, z, lis);
      }
      if (ok < 0) {
        rc = ok;
        break;
      }
      if (ok) { /* merger_projective_impl.cpp:203-207 */
        memcpy(map->desc + 32 * (size_t) s, measurement_desc + 32 * (size_t) m, 32);
        ++n_merged;
      }
    }
    if (rc == 0) {
      const float merge_ratio = (float) n_merged / (float) n_corr; /* :137-150 */
      if (n_merged == 0) {
        result->flags |= ORC_WARN_NO_MATCHES;
      } else if (merge_ratio < P->target_merge_ratio) {
        result->flags |= ORC_WARN_LOW_RATIO;
      }
    }
  }
  /* _addPoints (:56-57, :154-161, :210-305) */
  const int initial = map->n_points;
  if (rc == 0 && (n_corr == 0 || ((uint32_t) n_merged < P->target_number_of_merges && n_merged < n_measured))) {
    int32_t* slot_of_bin = (int32_t*) malloc(sizeof(int32_t) * (size_t) nbr * nbc);
    int32_t* cand        = (int32_t*) malloc(sizeof(int32_t) * (size_t) (n_measured > 0 ? n_measured : 1));
    int n_cand           = 0;
    for (uint32_t i = 0; i < nbr * nbc; ++i) {
      slot_of_bin[i] = -1;
    }
    for (int i = 0; i < n_measured && rc == 0; ++i) {
      const float* z = measurement + (size_t) dim * (size_t) i;
      if (!P->enable_binning) {
        cand[n_cand++] = i; /* :262-264 */
        continue;
      }
      const uint32_t br = bin_of(z[1], row_w), bc = bin_of(z[0], col_w);
      if (br >= nbr || bc >= nbc) {
        rc = ORC_ERR_NULL;
        break;
      }
      if (occupied[br * nbc + bc]) {
        continue; /* :236-241 */
      }
      int32_t* slot = &slot_of_bin[br * nbc + bc];
      if (*slot < 0) {
        *slot          = n_cand; /* :252-259 */
        cand[n_cand++] = i;
      } else {
        const float* o = measurement + (size_t) dim * (size_t) cand[*slot];
        int better;
        if (P->variant == ORC_MERGER_DEPTH_EKF) {
          better = z[2] < o[2]; /* merger_projective_depth_ekf_impl.cpp:50-57 */
        } else {
          better = (z[0] - z[2]) > (o[0] - o[2]); /* merger_projective_rigid_stereo_impl.cpp:45-57 */
        }
        if (better) {
          cand[*slot] = i; /* :247-250 */
        }
      }
    }
    for (int k = 0; k < n_cand && rc == 0; ++k) { /* :267-299 */
      const int i    = cand[k];
      const float* z = measurement + (size_t) dim * (size_t) i;
      float p[3];
      int valid;
      if (P->variant == ORC_MERGER_DEPTH_EKF) {
        const float d = z[2];
        valid         = d > 0.0f;
        p[0]          = (z[0] - P->cx) / P->fx * d;
        p[1]          = (z[1] - P->cy) / P->fy * d;
        p[2]          = d;
      } else {
        valid = triangulate_one(&P->triangulator, z, p);
      }
      if (!valid) {
        continue;
      }
      rc = add_landmark(map, &t, measurement_in_scene, z, measurement_desc + 32 * (size_t) i, p, frame);
    }
    free(slot_of_bin);
    free(cand);
  }
  result->n_merged = n_merged;
  result->n_added  = map->n_points - initial;
  free(occupied);
  free(seen);
  return rc;
}
