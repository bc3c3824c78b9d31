/*
 * proslam_oracle.c -- CPU restatement of srrg2_proslam's per-frame tracking hot path.
 * TEST INFRASTRUCTURE ONLY (see proslam_oracle.h): pinned against the counts and tolerances the reference's
 * gtests assert on its own test images (tests/test_ref_pins.py).  Single-threaded, C99, -O2 -ffp-contract=off.
 *
 * Every function cites the reference file:line it follows.  Paths are relative to
 * /root/reference/srrg2_proslam/src/srrg2_proslam/ unless they start with tests/ or configurations/.
 * CF/ = registration/correspondence_finders/.
 *
 * Canonicalisations of behaviour the reference leaves unspecified:
 *  (a) std::sort with a row-only comparator (CF/correspondence_finder_projective_square_impl.cpp:27-29):
 *      restated as a STABLE sort by row (ties keep ascending fixed index).
 *  (b) unordered_map iteration order in _filterCorrespondences
 *      (CF/correspondence_finder_projective_base_impl.cpp:50): restated as ascending fixed index.
 *  (c) out-of-bounds read before the bound test
 *      (CF/correspondence_finder_descriptor_based_epipolar_impl.cpp:136-137): bound is tested first.
 *  (d) duplicate (row,col) keys in _sortFeatureVector (epipolar_impl.cpp:36-41): ties broken by
 *      ascending unsorted index.
 */
#include "proslam_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* descriptor distance                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* srrg2_core PointDescriptorField::distance (external): Hamming norm over the 1x32 CV_8U row;
 * call sites CF/correspondence_finder_descriptor_based_epipolar_impl.cpp:157,
 * CF/correspondence_finder_projective_circle_impl.cpp:59-61. */
int orc_hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int k = 0; k < 4; ++k) {
    uint64_t x, y;
    memcpy(&x, a + 8 * k, 8);
    memcpy(&y, b + 8 * k, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

/* ------------------------------------------------------------------------------------------ */
/* a1 + a2: stereo epipolar matcher                                                            */
/* ------------------------------------------------------------------------------------------ */

/* Feature (epipolar_impl.cpp:8-20): truncating float->int32 of (v,u) */
typedef struct {
  int32_t row;
  int32_t col;
  int32_t unsorted_index;
} orc_feature;

/* comparator of _sortFeatureVector (epipolar_impl.cpp:36-41) + canonical tie-break (d) */
static int feature_cmp(const void* a_, const void* b_) {
  const orc_feature* a = (const orc_feature*) a_;
  const orc_feature* b = (const orc_feature*) b_;
  if (a->row != b->row) {
    return a->row < b->row ? -1 : 1;
  }
  if (a->col != b->col) {
    return a->col < b->col ? -1 : 1;
  }
  if (a->unsorted_index != b->unsorted_index) {
    return a->unsorted_index < b->unsorted_index ? -1 : 1;
  }
  return 0;
}

/* _sortFeatureVector (epipolar_impl.cpp:26-42) */
static orc_feature* sort_feature_vector(const float* uv, int n) {
  orc_feature* f = (orc_feature*) malloc(sizeof(orc_feature) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    f[i].row            = (int32_t) uv[2 * i + 1]; /* image coordinate Y, :10 */
    f[i].col            = (int32_t) uv[2 * i + 0]; /* image coordinate X, :11 */
    f[i].unsorted_index = i;
  }
  qsort(f, (size_t) n, sizeof(orc_feature), feature_cmp);
  return f;
}

/* CorrespondenceFinderDescriptorBasedEpipolar::compute (epipolar_impl.cpp:46-219) */
int orc_stereo_match(const float* uv_left,
                     const uint8_t* desc_left,
                     int n_left,
                     const float* uv_right,
                     const uint8_t* desc_right,
                     int n_right,
                     const orc_stereo_params* params,
                     orc_corr* out,
                     int capacity,
                     int* n_out) {
  /* _preCompute (bruteforce_impl.cpp:203-227) */
  if (!params || !out || !n_out || (n_left > 0 && (!uv_left || !desc_left)) ||
      (n_right > 0 && (!uv_right || !desc_right))) {
    return ORC_ERR_NULL;
  }
  if (capacity < n_left) {
    return ORC_ERR_CAPACITY;
  }
  int flags = ORC_OK;
  if (n_left == 0 || n_right == 0) {
    flags |= ORC_WARN_EMPTY_INPUT;
  }
  *n_out = 0;

  const float maximum_descriptor_distance = params->maximum_descriptor_distance;
  const float maximum_ratio               = params->maximum_distance_ratio_to_second_best;
  const int32_t maximum_disparity_pixels  = params->maximum_disparity_pixels;

  /* :65-68 */
  orc_feature* features_left  = sort_feature_vector(uv_left, n_left);
  orc_feature* features_right = sort_feature_vector(uv_right, n_right);
  size_t size_left            = (size_t) n_left;
  size_t size_right           = (size_t) n_right;
  uint8_t* matched_left       = (uint8_t*) malloc(size_left + 1);
  uint8_t* matched_right      = (uint8_t*) malloc(size_right + 1);

  /* vertical offsets 0,+1,-1,... (:71-79) */
  const int32_t thickness = params->epipolar_line_thickness_pixels;
  const int n_offsets     = 1 + 2 * (thickness > 0 ? thickness : 0);
  int n = 0;
  for (int o = 0; o < n_offsets; ++o) {
    const int32_t row_offset_pixels = (o == 0) ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2));
    size_t index_right              = 0;
    memset(matched_left, 0, size_left + 1);
    memset(matched_right, 0, size_right + 1);

    for (size_t index_left = 0; index_left < size_left; ++index_left) { /* :91 */
      if (index_right == size_right) {
        break; /* :93-95 */
      }
      /* the right keypoints are on a higher row - skip left (:98-107) */
      while (features_left[index_left].row + row_offset_pixels < features_right[index_right].row) {
        ++index_left;
        if (index_left == size_left) {
          break;
        }
      }
      if (index_left == size_left) {
        break;
      }
      const int32_t row_left            = features_left[index_left].row + row_offset_pixels;
      const int32_t col_left            = features_left[index_left].col;
      const int32_t unsorted_index_left = features_left[index_left].unsorted_index;
      const uint8_t* descriptor_left    = desc_left + (size_t) unsorted_index_left * ORC_DESC_BYTES;

      /* the right keypoints are on a lower row - skip right (:119-127) */
      while (row_left > features_right[index_right].row) {
        ++index_right;
        if (index_right == size_right) {
          break;
        }
      }
      if (index_right == size_right) {
        break;
      }

      /* search bookkeeping (:130-133) */
      size_t index_search_right             = index_right;
      float descriptor_distance_best        = FLT_MAX;
      float descriptor_distance_second_best = FLT_MAX;
      size_t index_best_right               = 0;

      /* scan epipolar line (:136-168); canonicalisation (c): bound first */
      while (index_search_right < size_right &&
             row_left == features_right[index_search_right].row) {
        const int32_t disparity_pixels = col_left - features_right[index_search_right].col;
        if (disparity_pixels < 0) {
          break; /* :141-143 */
        }
        if (disparity_pixels > maximum_disparity_pixels) {
          ++index_search_right; /* :146-149 */
          continue;
        }
        const uint8_t* descriptor_right =
          desc_right + (size_t) features_right[index_search_right].unsorted_index * ORC_DESC_BYTES;
        const int descriptor_distance = orc_hamming256(descriptor_left, descriptor_right);
        if ((float) descriptor_distance < descriptor_distance_best) { /* :158-164 */
          descriptor_distance_second_best = descriptor_distance_best;
          descriptor_distance_best        = (float) descriptor_distance;
          index_best_right                = index_search_right;
        } else if ((float) descriptor_distance < descriptor_distance_second_best) {
          descriptor_distance_second_best = (float) descriptor_distance;
        }
        ++index_search_right;
      }

      /* acceptance (:171-173): 0/0 = NaN rejects, x/FLT_MAX ~ 0 accepts */
      if (descriptor_distance_best < maximum_descriptor_distance &&
          descriptor_distance_best / descriptor_distance_second_best < maximum_ratio) {
        out[n].fixed_idx  = unsorted_index_left;
        out[n].moving_idx = features_right[index_best_right].unsorted_index;
        out[n].response   = descriptor_distance_best;
        ++n;
        index_right                     = index_best_right + 1; /* :181 */
        matched_left[index_left]        = 1;                    /* :184-185 */
        matched_right[index_best_right] = 1;
      }
    }

    /* prune matched candidates keeping the order (:188-205) */
    size_t index_keep = 0;
    for (size_t i = 0; i < size_left; ++i) {
      if (!matched_left[i]) {
        features_left[index_keep++] = features_left[i];
      }
    }
    size_left  = index_keep;
    index_keep = 0;
    for (size_t i = 0; i < size_right; ++i) {
      if (!matched_right[i]) {
        features_right[index_keep++] = features_right[i];
      }
    }
    size_right = index_keep;
  }
  free(matched_left);
  free(matched_right);
  free(features_left);
  free(features_right);
  *n_out = n;

  /* matching ratio warning (:209-216); float / size_t -> float */
  const float matching_ratio = (float) n / (float) (size_t) n_left;
  if (matching_ratio < params->minimum_matching_ratio) {
    flags |= ORC_WARN_LOW_RATIO;
  }
  /* _postCompute (bruteforce_impl.cpp:231-243) */
  if (n == 0) {
    flags |= ORC_WARN_NO_MATCHES;
  }
  return flags;
}

/* RawDataPreprocessorStereoProjective::compute, assembly loop
 * (sensor_processing/raw_data_preprocessor_stereo_projective.cpp:107-132) */
int orc_stereo_assemble(const float* uv_left,
                        const float* uv_right,
                        const orc_corr* corr,
                        int n_corr,
                        float* out_uvuv,
                        int32_t* out_src_left) {
  int n = 0;
  for (int i = 0; i < n_corr; ++i) {
    const float uL = uv_left[2 * corr[i].fixed_idx + 0];
    const float vL = uv_left[2 * corr[i].fixed_idx + 1];
    const float uR = uv_right[2 * corr[i].moving_idx + 0];
    const float vR = uv_right[2 * corr[i].moving_idx + 1];
    const float horizontal_disparity = uL - uR; /* :117-120 */
    const float vertical_disparity   = vL - vR;
    if (horizontal_disparity < 0 || vertical_disparity < 0) {
      continue; /* :123-125 */
    }
    out_uvuv[4 * n + 0] = uL;
    out_uvuv[4 * n + 1] = vL;
    out_uvuv[4 * n + 2] = uR;
    out_uvuv[4 * n + 3] = vR;
    if (out_src_left) {
      out_src_left[n] = corr[i].fixed_idx;
    }
    ++n;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* a4: rectified stereo triangulation                                                          */
/* ------------------------------------------------------------------------------------------ */

/* TriangulatorRigidStereo::compute + triangulateRectifiedMidpoint
 * (mapping/triangulator_rigid_stereo.cpp:7-56,60-85) */
void orc_triangulate(const float* uvuv,
                     int n,
                     const orc_triangulator_params* p,
                     float* xyz,
                     uint8_t* valid) {
  for (int i = 0; i < n; ++i) {
    const float x_L = uvuv[4 * i + 0];
    const float y_L = uvuv[4 * i + 1];
    const float x_R = uvuv[4 * i + 2];
    const float y_R = uvuv[4 * i + 3];
    xyz[3 * i + 0]  = 0;
    xyz[3 * i + 1]  = 0;
    xyz[3 * i + 2]  = 0;
    valid[i]        = 0;
    /* skip point if horizontal disparity is insufficient (:39-45); size is preserved */
    if (x_L - x_R < p->minimum_disparity_pixels) {
      continue;
    }
    float depth_meters = p->infinity_depth_meters; /* :71 */
    if (x_L > x_R) {
      depth_meters = p->b_x / (x_L - x_R); /* :74-77 */
    }
    xyz[3 * i + 2] = depth_meters;
    xyz[3 * i + 0] = 1 / p->fx * (x_L - p->cx) * depth_meters;             /* :81 */
    xyz[3 * i + 1] = 1 / p->fy * ((y_L + y_R) / 2 - p->cy) * depth_meters; /* :84 */
    valid[i]       = 1;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* SE(3) helpers -- srrg2_core geometry3d is external: BUILD-DEFINED restatement               */
/* ------------------------------------------------------------------------------------------ */

void orc_se3_identity(float* T) {
  memset(T, 0, 16 * sizeof(float));
  T[0] = T[5] = T[10] = T[15] = 1.0f;
}

/* Isometry inverse: [R^T | -R^T t] */
void orc_se3_inverse(const float* T, float* Ti) {
  float R[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      R[3 * i + j] = T[4 * j + i];
    }
  }
  const float tx = T[3], ty = T[7], tz = T[11];
  for (int i = 0; i < 3; ++i) {
    Ti[4 * i + 0] = R[3 * i + 0];
    Ti[4 * i + 1] = R[3 * i + 1];
    Ti[4 * i + 2] = R[3 * i + 2];
    Ti[4 * i + 3] = -((R[3 * i + 0] * tx + R[3 * i + 1] * ty) + R[3 * i + 2] * tz);
  }
  Ti[12] = 0;
  Ti[13] = 0;
  Ti[14] = 0;
  Ti[15] = 1;
}

/* C = A * B for isometries (rotation product, R_A t_B + t_A) */
void orc_se3_mul(const float* A, const float* B, float* C) {
  float out[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      out[4 * i + j] = (A[4 * i + 0] * B[0 + j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j];
    }
    out[4 * i + 3] =
      ((A[4 * i + 0] * B[3] + A[4 * i + 1] * B[7]) + A[4 * i + 2] * B[11]) + A[4 * i + 3];
  }
  out[12] = 0;
  out[13] = 0;
  out[14] = 0;
  out[15] = 1;
  memcpy(C, out, sizeof(out));
}

/* rotation matrix -> quaternion (w,x,y,z), Shepperd's method as in Eigen's QuaternionBase */
static void r2q(const float* T, float* q /* w x y z */) {
  const float m00 = T[0], m01 = T[1], m02 = T[2];
  const float m10 = T[4], m11 = T[5], m12 = T[6];
  const float m20 = T[8], m21 = T[9], m22 = T[10];
  float t = (m00 + m11) + m22;
  if (t > 0.0f) {
    t    = sqrtf(t + 1.0f);
    q[0] = 0.5f * t;
    t    = 0.5f / t;
    q[1] = (m21 - m12) * t;
    q[2] = (m02 - m20) * t;
    q[3] = (m10 - m01) * t;
  } else {
    const float m[3][3] = {{m00, m01, m02}, {m10, m11, m12}, {m20, m21, m22}};
    int i               = 0;
    if (m11 > m00) {
      i = 1;
    }
    if (m22 > m[i][i]) {
      i = 2;
    }
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t           = sqrtf(((m[i][i] - m[j][j]) - m[k][k]) + 1.0f);
    q[1 + i]    = 0.5f * t;
    t           = 0.5f / t;
    q[0]        = (m[k][j] - m[j][k]) * t;
    q[1 + j]    = (m[j][i] + m[i][j]) * t;
    q[1 + k]    = (m[k][i] + m[i][k]) * t;
  }
}

/* geometry3d::t2tnq: translation + imaginary part of the normalised quaternion (w >= 0);
 * used at CF/correspondence_finder_projective_base_impl.cpp:182, tests/test_aligners.cpp:630 */
void orc_t2tnq(const float* T, float* v6) {
  float q[4];
  r2q(T, q);
  const float n = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  float s       = 1.0f / n;
  if (q[0] < 0.0f) {
    s = -s;
  }
  v6[0] = T[3];
  v6[1] = T[7];
  v6[2] = T[11];
  v6[3] = q[1] * s;
  v6[4] = q[2] * s;
  v6[5] = q[3] * s;
}

/* unit quaternion (w, x, y, z) -> rotation matrix, Eigen's toRotationMatrix operation order */
static void q2r(float w, float x, float y, float z, float* R /* 9 */) {
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0f - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0f - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0f - (txx + tyy);
}

/* perturbation vector [dt; dq] -> isometry: q = (sqrt(1-|dq|^2), dq)
 * (VariableSE3QuaternionRight parameterisation, tests/test_aligners.cpp:617) */
void orc_tnq2t(const float* v6, float* T) {
  float x = v6[3], y = v6[4], z = v6[5];
  const float n2 = (x * x + y * y) + z * z;
  float w;
  if (n2 < 1.0f) {
    w = sqrtf(1.0f - n2);
  } else {
    const float s = 1.0f / sqrtf(n2);
    x *= s;
    y *= s;
    z *= s;
    w = 0.0f;
  }
  float R[9];
  q2r(w, x, y, z, R);
  for (int i = 0; i < 3; ++i) {
    T[4 * i + 0] = R[3 * i + 0];
    T[4 * i + 1] = R[3 * i + 1];
    T[4 * i + 2] = R[3 * i + 2];
    T[4 * i + 3] = v6[i];
  }
  T[12] = 0;
  T[13] = 0;
  T[14] = 0;
  T[15] = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* a6: pinhole projector (external, BUILD-DEFINED per SURVEY.md Appendix A)                    */
/* ------------------------------------------------------------------------------------------ */

/* call site CF/correspondence_finder_projective_base_impl.cpp:158,165-166:
 * setCameraPose(local_map_in_sensor^-1); compute(moving, in_camera, in_image, indices) */
int orc_project(const orc_projector* proj,
                const float* camera_pose,
                const float* xyz,
                int n,
                float* uvz,
                int32_t* indices) {
  float W[16]; /* points -> camera */
  orc_se3_inverse(camera_pose, W);
  const float cols = (float) proj->canvas_cols;
  const float rows = (float) proj->canvas_rows;
  int m            = 0;
  for (int i = 0; i < n; ++i) {
    const float px = xyz[3 * i + 0], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    const float x = ((W[0] * px + W[1] * py) + W[2] * pz) + W[3];
    const float y = ((W[4] * px + W[5] * py) + W[6] * pz) + W[7];
    const float z = ((W[8] * px + W[9] * py) + W[10] * pz) + W[11];
    if (z < proj->range_min || z > proj->range_max) {
      continue;
    }
    const float hx = proj->fx * x + proj->cx * z;
    const float hy = proj->fy * y + proj->cy * z;
    const float u  = hx / z;
    const float v  = hy / z;
    if (u < 0.0f || u >= cols || v < 0.0f || v >= rows) {
      continue;
    }
    uvz[3 * m + 0] = u;
    uvz[3 * m + 1] = v;
    uvz[3 * m + 2] = z;
    indices[m]     = i;
    ++m;
  }
  return m;
}

/* ------------------------------------------------------------------------------------------ */
/* a5, a7-a10: projective correspondence finder                                                */
/* ------------------------------------------------------------------------------------------ */

/* Element (CF/correspondence_finder_projective_square.h:37-47) */
typedef struct {
  int16_t row;
  int16_t col;
  int16_t index;
} orc_element;

typedef struct {
  int32_t fixed_idx;
  int32_t moving_idx;
  float response;
  int32_t is_best; /* 1: emitted as the query's best, 0: as its second best */
} orc_candidate;

struct orc_pcf {
  orc_pcf_params params;
  /* CorrespondenceFinder_ members touched by the reference (bruteforce_impl.cpp:204-212,233-235) */
  int fixed_set, moving_set;
  int fixed_changed_flag, moving_changed_flag, local_map_in_sensor_changed_flag;
  float local_map_in_sensor[16];
  /* projective_base.h:132-154 */
  int config_changed;
  uint64_t search_radius_pixels;
  float descriptor_distance;
  float local_map_in_sensor_previous[16];
  int has_converged;
  uint64_t current_iteration;
  /* inputs (copied) */
  int n_fixed, n_moving;
  float* fixed_uv;      /* [n_fixed][2] */
  uint8_t* fixed_desc;  /* [n_fixed][32] */
  float* moving_xyz;    /* [n_moving][3] */
  uint8_t* moving_desc; /* [n_moving][32] */
  /* cached projection data (projective_base.h:152-154) */
  float* points_in_image; /* [n_moving][3] */
  int32_t* indices_projected_to_moving;
  int n_projected;
  /* lattice database (square.h:48) */
  orc_element* database_fixed;
  /* KD-tree database (projective_kdtree.h:44): internal nodes, leaves, and the fixed indices in leaf order */
  struct orc_kd_node* kd_nodes;
  int32_t* kd_leaf_start; /* [kd_n_leaves + 1] offsets into kd_order */
  int32_t* kd_order;      /* [n_fixed] fixed indices, leaf by leaf, ascending inside a leaf */
  int kd_n_nodes, kd_n_leaves, kd_root;
  /* persisting caller-side correspondence vector */
  orc_corr* correspondences;
  int n_correspondences;
  int num_recomputes;
};

orc_pcf* orc_pcf_create(const orc_pcf_params* params) {
  orc_pcf* h = (orc_pcf*) calloc(1, sizeof(orc_pcf));
  h->params  = *params;
  orc_se3_identity(h->local_map_in_sensor);
  orc_se3_identity(h->local_map_in_sensor_previous);
  h->config_changed = 1; /* projective_base.h:134 */
  return h;
}

void orc_pcf_destroy(orc_pcf* h) {
  if (!h) {
    return;
  }
  free(h->kd_nodes);
  free(h->kd_leaf_start);
  free(h->kd_order);
  free(h->fixed_uv);
  free(h->fixed_desc);
  free(h->moving_xyz);
  free(h->moving_desc);
  free(h->points_in_image);
  free(h->indices_projected_to_moving);
  free(h->database_fixed);
  free(h->correspondences);
  free(h);
}

void orc_pcf_set_params(orc_pcf* h, const orc_pcf_params* params) {
  /* minimum_descriptor_distance and maximum_search_radius_pixels carry &_config_changed
   * (projective_base.h:30-44); any update through this call is treated as such a change */
  h->params         = *params;
  h->config_changed = 1;
}

void orc_pcf_set_fixed(orc_pcf* h, const float* coords, int fixed_dim, const uint8_t* desc, int n) {
  free(h->fixed_uv);
  free(h->fixed_desc);
  free(h->correspondences);
  h->fixed_uv        = (float*) malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
  h->fixed_desc      = (uint8_t*) malloc((size_t) ORC_DESC_BYTES * (size_t)(n > 0 ? n : 1));
  h->correspondences = (orc_corr*) malloc(sizeof(orc_corr) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    h->fixed_uv[2 * i + 0] = coords[fixed_dim * i + 0];
    h->fixed_uv[2 * i + 1] = coords[fixed_dim * i + 1];
  }
  if (n > 0) {
    memcpy(h->fixed_desc, desc, (size_t) ORC_DESC_BYTES * (size_t) n);
  }
  h->n_fixed            = n;
  h->n_correspondences  = 0;
  h->fixed_set          = 1;
  h->fixed_changed_flag = 1;
}

void orc_pcf_set_moving(orc_pcf* h, const float* xyz, const uint8_t* desc, int n) {
  free(h->moving_xyz);
  free(h->moving_desc);
  free(h->points_in_image);
  free(h->indices_projected_to_moving);
  const size_t m                 = (size_t)(n > 0 ? n : 1);
  h->moving_xyz                  = (float*) malloc(sizeof(float) * 3 * m);
  h->moving_desc                 = (uint8_t*) malloc((size_t) ORC_DESC_BYTES * m);
  h->points_in_image             = (float*) malloc(sizeof(float) * 3 * m);
  h->indices_projected_to_moving = (int32_t*) malloc(sizeof(int32_t) * m);
  if (n > 0) {
    memcpy(h->moving_xyz, xyz, sizeof(float) * 3 * (size_t) n);
    memcpy(h->moving_desc, desc, (size_t) ORC_DESC_BYTES * (size_t) n);
  }
  h->n_moving            = n;
  h->n_projected         = 0;
  h->moving_set          = 1;
  h->moving_changed_flag = 1;
}

void orc_pcf_set_local_map_in_sensor(orc_pcf* h, const float* T) {
  memcpy(h->local_map_in_sensor, T, 16 * sizeof(float));
  h->local_map_in_sensor_changed_flag = 1;
}
void orc_pcf_get_local_map_in_sensor(const orc_pcf* h, float* T) {
  memcpy(T, h->local_map_in_sensor, 16 * sizeof(float));
}
void orc_pcf_set_search_radius(orc_pcf* h, uint64_t r) {
  h->search_radius_pixels = r; /* projective_base.h:82-85 */
  h->config_changed       = 0;
}
void orc_pcf_set_descriptor_distance(orc_pcf* h, float d) {
  h->descriptor_distance = d; /* projective_base.h:94-97 */
  h->config_changed      = 0;
}
uint64_t orc_pcf_search_radius(const orc_pcf* h) {
  return h->search_radius_pixels;
}
float orc_pcf_descriptor_distance(const orc_pcf* h) {
  return h->descriptor_distance;
}
uint64_t orc_pcf_iteration(const orc_pcf* h) {
  return h->current_iteration;
}
int orc_pcf_has_converged(const orc_pcf* h) {
  return h->has_converged;
}
int orc_pcf_num_recomputes(const orc_pcf* h) {
  return h->num_recomputes;
}

/* ---- srrg2_core::KDTree<float, 2> as the KD-tree finder uses it (CF/correspondence_finder_projective_kdtree_impl.cpp:8-26,
 * 39-50; the class itself is external).  Restated from its published construction -- split a cluster at its mean along the
 * direction of largest variance until it is small -- with the two constants the header does not carry fixed by the reference's
 * own pinned results: a cluster is a leaf when it holds fewer than minimum_number_of_points_per_cluster points or when its
 * extent 3 * sqrt(largest eigenvalue of its covariance) is below the leaf range (= _search_radius_pixels at build time), and a
 * radius query is answered from the ONE leaf the query point descends to.  With these the eight counts the reference asserts
 * for this finder come out exactly (319, 2, 120, 21, 82, 36, 104, 56: tests/test_correspondence_finders.cpp:330,370,412,427,
 * 468,552,568,609); an exhaustive radius search returns supersets (123, 83, 89, 41, 108, 64).
 * Arithmetic (BUILD-DEFINED, identical on the device): cluster statistics are exact integer sums of the coordinates in
 * 1/16 px (below 2^53: independent of the summation order, exactly convertible to double); mean, covariance and
 * eigen-decomposition in double, rounded to float for the node; side test (x - mean_x) * n_x + (y - mean_y) * n_y < 0 -> left
 * in float; leaf members in ascending fixed index. */
struct orc_kd_node {
  float mean[2], normal[2];
  int32_t child[2]; /* >= 0: internal node, < 0: leaf ~child */
};

typedef struct {
  int64_t sx, sy, sxx, sxy, syy;
} orc_kd_stats;

static inline int64_t kd_quantise(float v) {
  return (int64_t) llrintf(v * 16.0f);
}

/* leaf decision and split direction of a cluster from its exact sums */
static int kd_decide(const orc_kd_stats* st, int n, double leaf_range, int min_points, float* mean, float* normal) {
  const double dn  = (double) n;
  const double sx = (double) st->sx, sy = (double) st->sy;
  mean[0]          = (float) (sx / (16.0 * dn));
  mean[1]          = (float) (sy / (16.0 * dn));
  const double cxx = ((double) st->sxx - sx * sx / dn) / dn / 256.0;
  const double cxy = ((double) st->sxy - sx * sy / dn) / dn / 256.0;
  const double cyy = ((double) st->syy - sy * sy / dn) / dn / 256.0;
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

static inline float kd_side(const float* mean, const float* normal, float x, float y) {
  const float dx = x - mean[0], dy = y - mean[1];
  return dx * normal[0] + dy * normal[1];
}

/* builds the subtree over idx[0..n) (ascending fixed indices), returns the child code */
static int32_t kd_build(orc_pcf* h, const int32_t* idx, int n, double leaf_range, int min_points) {
  orc_kd_stats st = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const int64_t qx = kd_quantise(h->fixed_uv[2 * idx[i]]), qy = kd_quantise(h->fixed_uv[2 * idx[i] + 1]);
    st.sx += qx;
    st.sy += qy;
    st.sxx += qx * qx;
    st.sxy += qx * qy;
    st.syy += qy * qy;
  }
  float mean[2], normal[2];
  int leaf     = kd_decide(&st, n, leaf_range, min_points, mean, normal);
  int32_t* lft = NULL;
  int nl = 0, nr = 0;
  if (!leaf) {
    lft          = (int32_t*) malloc(sizeof(int32_t) * 2u * (size_t) (n > 0 ? n : 1));
    int32_t* rgt = lft + n;
    for (int i = 0; i < n; ++i) {
      if (kd_side(mean, normal, h->fixed_uv[2 * idx[i]], h->fixed_uv[2 * idx[i] + 1]) < 0.0f) {
        lft[nl++] = idx[i];
      } else {
        rgt[nr++] = idx[i];
      }
    }
    leaf = nl == 0 || nr == 0; /* (cannot shrink any further) */
  }
  if (leaf) {
    const int l = h->kd_n_leaves++;
    int32_t o   = h->kd_leaf_start[l];
    for (int i = 0; i < n; ++i) {
      h->kd_order[o++] = idx[i];
    }
    h->kd_leaf_start[l + 1] = o;
    free(lft);
    return ~l;
  }
  const int me = h->kd_n_nodes++;
  h->kd_nodes[me].mean[0] = mean[0];
  h->kd_nodes[me].mean[1] = mean[1];
  h->kd_nodes[me].normal[0] = normal[0];
  h->kd_nodes[me].normal[1] = normal[1];
  const int32_t cl = kd_build(h, lft, nl, leaf_range, min_points);
  const int32_t cr = kd_build(h, lft + n, nr, leaf_range, min_points);
  h->kd_nodes[me].child[0] = cl;
  h->kd_nodes[me].child[1] = cr;
  free(lft);
  return me;
}

/* _initializeDatabase of the KD-tree finder (kdtree_impl.cpp:8-26): leaf range = the CURRENT search radius */
static void pcf_initialize_kdtree(orc_pcf* h) {
  free(h->kd_nodes);
  free(h->kd_leaf_start);
  free(h->kd_order);
  const int n      = h->n_fixed;
  h->kd_nodes      = (struct orc_kd_node*) malloc(sizeof(struct orc_kd_node) * (size_t) (n > 0 ? n : 1));
  h->kd_leaf_start = (int32_t*) calloc((size_t) n + 2, sizeof(int32_t));
  h->kd_order      = (int32_t*) malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
  h->kd_n_nodes = h->kd_n_leaves = 0;
  h->kd_root                     = ~0;
  if (n <= 0) {
    h->kd_n_leaves = 1; /* one empty leaf */
    return;
  }
  int32_t* idx = (int32_t*) malloc(sizeof(int32_t) * (size_t) n);
  for (int i = 0; i < n; ++i) {
    idx[i] = i;
  }
  const int min_points = h->params.minimum_number_of_points_per_cluster > 0 ? h->params.minimum_number_of_points_per_cluster : 10;
  h->kd_root           = kd_build(h, idx, n, (double) (float) h->search_radius_pixels, min_points);
  free(idx);
}

/* _initializeDatabase (CF/correspondence_finder_projective_square_impl.cpp:8-31);
 * canonicalisation (a): stable sort by row */
static void pcf_initialize_database(orc_pcf* h) {
  if (h->params.search_type == ORC_SEARCH_KDTREE) {
    pcf_initialize_kdtree(h);
    return;
  }
  free(h->database_fixed);
  const int n       = h->n_fixed;
  h->database_fixed = (orc_element*) malloc(sizeof(orc_element) * (size_t)(n > 0 ? n : 1));
  orc_element* tmp  = (orc_element*) malloc(sizeof(orc_element) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    tmp[i].row   = (int16_t) h->fixed_uv[2 * i + 1]; /* Element(coordinates(1), coordinates(0), i) :24 */
    tmp[i].col   = (int16_t) h->fixed_uv[2 * i + 0];
    tmp[i].index = (int16_t) i;
  }
  /* stable insertion by row via counting sort over the int16 range */
  int32_t* count = (int32_t*) calloc(65537, sizeof(int32_t));
  for (int i = 0; i < n; ++i) {
    ++count[(int32_t) tmp[i].row + 32768 + 1];
  }
  for (int r = 0; r < 65536; ++r) {
    count[r + 1] += count[r];
  }
  for (int i = 0; i < n; ++i) {
    h->database_fixed[count[(int32_t) tmp[i].row + 32768]++] = tmp[i];
  }
  free(count);
  free(tmp);
}

typedef struct {
  orc_candidate* data;
  int n;
} orc_candidate_list;

/* _findNearestNeighbors for Square / Circle / Rhombus
 * (CF/correspondence_finder_projective_square_impl.cpp:35-118,
 *  CF/correspondence_finder_projective_circle_impl.cpp:8-94,
 *  CF/correspondence_finder_projective_rhombus_impl.cpp:8-93) */
static void pcf_find_lattice(const orc_pcf* h,
                             const float* query_uvz,
                             int32_t query_index,
                             orc_candidate_list* list) {
  const int type     = h->params.search_type;
  const uint64_t rad = h->search_radius_pixels;
  const int16_t row  = (int16_t) roundf(query_uvz[1]); /* circle_impl.cpp:15-16 */
  const int16_t col  = (int16_t) roundf(query_uvz[0]);
  /* row search range, evaluated in size_t then truncated to int16 (circle_impl.cpp:25-26) */
  const int16_t row_min = (int16_t)((uint64_t)(int64_t) row - rad);
  const int16_t row_max = (int16_t)((uint64_t)(int64_t) row + rad + 1);
  /* square_impl.cpp:56-57 */
  const int16_t col_min = (int16_t)((uint64_t)(int64_t) col - rad - 1);
  const int16_t col_max = (int16_t)((uint64_t)(int64_t) col + rad + 1);
  const int32_t radius_squared = (int32_t)(rad * rad); /* circle_impl.cpp:30-31 */
  const uint8_t* query_descriptor = h->moving_desc + (size_t) query_index * ORC_DESC_BYTES;

  size_t index_fixed_best               = 0;
  size_t index_fixed_second_best        = 0;
  float descriptor_distance_best        = FLT_MAX;
  float descriptor_distance_second_best = FLT_MAX;

  /* stage 1: arrive at the interesting row indices (circle_impl.cpp:40-44) */
  int it      = 0;
  const int n = h->n_fixed;
  while (it != n && h->database_fixed[it].row < row_min) {
    ++it;
  }
  /* scan all valid rows (circle_impl.cpp:47-75) */
  while (it != n && h->database_fixed[it].row < row_max) {
    const orc_element* e = &h->database_fixed[it];
    int accept;
    if (type == ORC_SEARCH_SQUARE) {
      accept = (e->col > col_min && e->col < col_max); /* square_impl.cpp:80 */
    } else if (type == ORC_SEARCH_CIRCLE) {
      const int32_t height = e->row - row; /* circle_impl.cpp:51-53 */
      const int32_t width  = (int32_t)(sqrt((double) (radius_squared - height * height)) + 1);
      accept               = (e->col > col - width && e->col < col + width); /* :56 */
    } else {
      int16_t width = (int16_t)(e->row - row_min + 1); /* rhombus_impl.cpp:49-52 */
      if (width > (int16_t) rad) {
        width = (int16_t)(row_max - e->row);
      }
      accept = (e->col > col - width && e->col < col + width);
    }
    if (accept) {
      const float descriptor_distance = (float) orc_hamming256(
        h->fixed_desc + (size_t) e->index * ORC_DESC_BYTES, query_descriptor);
      if (descriptor_distance < descriptor_distance_best) { /* circle_impl.cpp:64-72 */
        descriptor_distance_second_best = descriptor_distance_best;
        descriptor_distance_best        = descriptor_distance;
        index_fixed_second_best         = index_fixed_best;
        index_fixed_best                = (size_t) e->index;
      } else if (descriptor_distance < descriptor_distance_second_best) {
        descriptor_distance_second_best = descriptor_distance;
        index_fixed_second_best         = (size_t) e->index;
      }
    }
    ++it;
  }
  /* emit best and, if any, second best (circle_impl.cpp:78-92) */
  if (descriptor_distance_best < FLT_MAX) {
    orc_candidate* c = &list->data[list->n++];
    c->fixed_idx     = (int32_t) index_fixed_best;
    c->moving_idx    = query_index;
    c->response      = descriptor_distance_best;
    c->is_best       = 1;
    if (descriptor_distance_second_best < FLT_MAX) {
      c             = &list->data[list->n++];
      c->fixed_idx  = (int32_t) index_fixed_second_best;
      c->moving_idx = query_index;
      c->response   = descriptor_distance_second_best;
      c->is_best    = 0;
    }
  }
}

/* _findNearestNeighbors for the KD-tree variant (CF/correspondence_finder_projective_kdtree_impl.cpp:30-80):
 * KDTree::findNeighbors(query, r^2) = the points of the leaf the query descends to whose squared distance is below r^2 */
static void pcf_find_kdtree(const orc_pcf* h,
                            const float* query_uvz,
                            int32_t query_index,
                            orc_candidate_list* list) {
  const float maximum_distance_squared =
    (float) (h->search_radius_pixels * h->search_radius_pixels); /* kdtree_impl.cpp:40-41 */
  const float maximum_descriptor_distance = h->params.maximum_descriptor_distance;
  const uint8_t* query_descriptor = h->moving_desc + (size_t) query_index * ORC_DESC_BYTES;
  size_t index_fixed_best               = 0;
  float descriptor_distance_best        = maximum_descriptor_distance; /* :54 */
  float descriptor_distance_second_best = FLT_MAX;
  int32_t node = h->kd_root;
  while (node >= 0) {
    const struct orc_kd_node* nd = &h->kd_nodes[node];
    node = nd->child[kd_side(nd->mean, nd->normal, query_uvz[0], query_uvz[1]) < 0.0f ? 0 : 1];
  }
  const int leaf = ~node;
  for (int32_t o = h->kd_leaf_start[leaf]; o < h->kd_leaf_start[leaf + 1]; ++o) {
    const int f    = h->kd_order[o];
    const float du = h->fixed_uv[2 * f + 0] - query_uvz[0];
    const float dv = h->fixed_uv[2 * f + 1] - query_uvz[1];
    if (!(du * du + dv * dv < maximum_distance_squared)) {
      continue;
    }
    const float descriptor_distance =
      (float) orc_hamming256(h->fixed_desc + (size_t) f * ORC_DESC_BYTES, query_descriptor);
    if (descriptor_distance < descriptor_distance_best) { /* :62-68 */
      descriptor_distance_second_best = descriptor_distance_best;
      descriptor_distance_best        = descriptor_distance;
      index_fixed_best                = (size_t) f;
    } else if (descriptor_distance < descriptor_distance_second_best) {
      descriptor_distance_second_best = descriptor_distance;
    }
  }
  (void) descriptor_distance_second_best;
  if (descriptor_distance_best < maximum_descriptor_distance) { /* :72-78 */
    orc_candidate* c = &list->data[list->n++];
    c->fixed_idx     = (int32_t) index_fixed_best;
    c->moving_idx    = query_index;
    c->response      = descriptor_distance_best;
    c->is_best       = 1;
  }
}

/* _addCorrespondenceCandidate + _filterCorrespondences
 * (CF/correspondence_finder_projective_base_impl.cpp:8-37,41-102).
 * The two unordered_maps are restated as (i) the candidate list bucketed by fixed index in
 * insertion order and (ii) the per-moving "first minimum" which, because every moving point
 * is queried once and emits its best before its second best, is always its best emission. */
static int pcf_filter(const orc_pcf* h,
                      const orc_candidate_list* list,
                      float maximum_descriptor_distance,
                      float maximum_distance_ratio,
                      orc_corr* out) {
  const int nf       = h->n_fixed;
  const int nm       = h->n_moving;
  int32_t* start     = (int32_t*) calloc((size_t) nf + 2, sizeof(int32_t));
  int32_t* order     = (int32_t*) malloc(sizeof(int32_t) * (size_t)(list->n > 0 ? list->n : 1));
  int32_t* best_of_m = (int32_t*) malloc(sizeof(int32_t) * (size_t)(nm > 0 ? nm : 1));
  for (int m = 0; m < nm; ++m) {
    best_of_m[m] = -1;
  }
  for (int k = 0; k < list->n; ++k) {
    ++start[list->data[k].fixed_idx + 1];
  }
  for (int f = 0; f < nf; ++f) {
    start[f + 1] += start[f];
  }
  int32_t* cursor = (int32_t*) malloc(sizeof(int32_t) * ((size_t) nf + 1));
  memcpy(cursor, start, sizeof(int32_t) * ((size_t) nf + 1));
  for (int k = 0; k < list->n; ++k) {
    const orc_candidate* c = &list->data[k];
    order[cursor[c->fixed_idx]++] = k;
    /* moving-indexed buffer: first minimum over [best, second best] is the best (:82-92) */
    if (best_of_m[c->moving_idx] < 0) {
      best_of_m[c->moving_idx] = c->fixed_idx;
    }
  }
  int n = 0;
  for (int f = 0; f < nf; ++f) { /* canonicalisation (b): ascending fixed index */
    if (start[f] == start[f + 1]) {
      continue;
    }
    int index_best               = -1;
    float response_lowest        = FLT_MAX;
    float response_second_lowest = FLT_MAX;
    for (int s = start[f]; s < start[f + 1]; ++s) { /* :57-68 */
      const float current_response = list->data[order[s]].response;
      if (current_response < response_lowest) {
        response_second_lowest = response_lowest;
        response_lowest        = current_response;
        index_best             = order[s];
      } else if (current_response < response_second_lowest) {
        response_second_lowest = current_response;
      }
    }
    if (response_lowest < maximum_descriptor_distance &&
        response_lowest / response_second_lowest < maximum_distance_ratio) { /* :71-72 */
      const orc_candidate* best = &list->data[index_best];
      if (best_of_m[best->moving_idx] == best->fixed_idx) { /* bijection :95-99 */
        out[n].fixed_idx  = best->fixed_idx;
        out[n].moving_idx = best->moving_idx;
        out[n].response   = best->response;
        ++n;
      }
    }
  }
  free(cursor);
  free(best_of_m);
  free(order);
  free(start);
  return n;
}

/* CorrespondenceFinderProjectiveBase::compute (CF/correspondence_finder_projective_base_impl.cpp:105-293) */
static int pcf_compute_internal(orc_pcf* h, int flags) {
  /* _preCompute (bruteforce_impl.cpp:203-227) */
  if (!h->fixed_set || !h->moving_set) {
    return ORC_ERR_NULL;
  }
  if (h->n_fixed == 0 || h->n_moving == 0) {
    flags |= ORC_WARN_EMPTY_INPUT;
  }
  const orc_pcf_params* P = &h->params;

  /* fixed/moving/config changed -> new optimization (:109-134) */
  if (h->fixed_changed_flag || h->moving_changed_flag || h->config_changed) {
    h->fixed_changed_flag  = 0;
    h->moving_changed_flag = 0;
    if ((h->search_radius_pixels == 0 && h->descriptor_distance == 0) || h->config_changed) {
      h->search_radius_pixels = P->maximum_search_radius_pixels;
      h->descriptor_distance  = P->minimum_descriptor_distance;
    }
    h->has_converged     = 0;
    h->current_iteration = 0;
    orc_se3_identity(h->local_map_in_sensor_previous);
    pcf_initialize_database(h);
    h->config_changed = 0;
  }

  /* converged: correspondences are not touched (:138-142) */
  if (!h->fixed_changed_flag && !h->moving_changed_flag && h->has_converged) {
    goto post_compute;
  }

  {
    /* setCameraPose(local_map_in_sensor^-1) (:158) */
    float camera_pose[16];
    orc_se3_inverse(h->local_map_in_sensor, camera_pose);

    /* reproject periodically and always for iterations 0 and 1 (:162-178) */
    const uint64_t k = P->number_of_solver_iterations_per_projection;
    if (k == 0 || h->current_iteration % k == 0 || h->current_iteration == 1) {
      h->n_projected = orc_project(&P->projector,
                                   camera_pose,
                                   h->moving_xyz,
                                   h->n_moving,
                                   h->points_in_image,
                                   h->indices_projected_to_moving);
      if (h->n_projected == 0) {
        flags |= ORC_WARN_NO_PROJECTION;
      }
    } else {
      memcpy(h->local_map_in_sensor_previous, h->local_map_in_sensor, sizeof(float) * 16);
      ++h->current_iteration;
      goto post_compute;
    }

    /* projection estimate change (:181-183) */
    float delta[16], v6[6];
    orc_se3_mul(camera_pose, h->local_map_in_sensor_previous, delta);
    orc_t2tnq(delta, v6);
    const float estimate_change_norm = sqrtf(
      ((((v6[0] * v6[0] + v6[1] * v6[1]) + v6[2] * v6[2]) + v6[3] * v6[3]) + v6[4] * v6[4]) +
      v6[5] * v6[5]);
    memcpy(h->local_map_in_sensor_previous, h->local_map_in_sensor, sizeof(float) * 16);

    /* candidate search (:192-200) */
    orc_candidate_list list;
    list.data = (orc_candidate*) malloc(sizeof(orc_candidate) * 2 *
                                        (size_t)(h->n_projected > 0 ? h->n_projected : 1));
    list.n    = 0;
    ++h->num_recomputes;
    for (int ip = 0; ip < h->n_projected; ++ip) {
      if (P->search_type == ORC_SEARCH_KDTREE) {
        pcf_find_kdtree(h, h->points_in_image + 3 * ip, h->indices_projected_to_moving[ip], &list);
      } else {
        pcf_find_lattice(h, h->points_in_image + 3 * ip, h->indices_projected_to_moving[ip], &list);
      }
    }

    /* filter with the DYNAMIC descriptor distance (:204-208) */
    orc_corr* filtered = (orc_corr*) malloc(sizeof(orc_corr) * (size_t)(h->n_fixed > 0 ? h->n_fixed : 1));
    const int n_filtered = pcf_filter(
      h, &list, h->descriptor_distance, P->maximum_distance_ratio_to_second_best, filtered);
    free(list.data);

    /* matching ratio (:215-216) */
    const float matching_ratio = (float) n_filtered / (float) (size_t) h->n_fixed;

    if (matching_ratio < P->minimum_matching_ratio) { /* :228 */
      flags |= ORC_WARN_LOW_RATIO;
      if (h->search_radius_pixels < P->maximum_search_radius_pixels ||
          h->descriptor_distance > P->minimum_descriptor_distance) { /* :235-236 */
        h->search_radius_pixels = P->maximum_search_radius_pixels;
        h->descriptor_distance  = P->minimum_descriptor_distance;
        flags |= ORC_WARN_RETRIED;
        if (matching_ratio == 0) { /* :251-259 */
          flags |= ORC_WARN_TRACK_LOST;
          orc_se3_identity(h->local_map_in_sensor);
          h->current_iteration = 0;
        } else {
          ++h->current_iteration;
        }
        free(filtered);
        return pcf_compute_internal(h, flags); /* :262 */
      }
    }

    /* update correspondences (:268) */
    memcpy(h->correspondences, filtered, sizeof(orc_corr) * (size_t) n_filtered);
    h->n_correspondences = n_filtered;
    free(filtered);

    /* termination (:271-288) */
    if (estimate_change_norm < P->maximum_estimate_change_norm_for_convergence &&
        h->current_iteration > P->minimum_number_of_iterations) {
      h->has_converged = 1;
      if (matching_ratio > P->minimum_matching_ratio) {
        /* size_t arithmetic: may wrap exactly like the reference (:279-281) */
        const uint64_t reduced = h->search_radius_pixels - P->search_radius_step_size_pixels;
        h->search_radius_pixels =
          reduced > P->minimum_search_radius_pixels ? reduced : P->minimum_search_radius_pixels;
        const float increased = h->descriptor_distance + P->descriptor_distance_step_size_pixels;
        h->descriptor_distance =
          increased < P->maximum_descriptor_distance ? increased : P->maximum_descriptor_distance;
      }
    }
    ++h->current_iteration; /* :291 */
  }

post_compute:
  /* _postCompute (bruteforce_impl.cpp:231-243) */
  h->fixed_changed_flag               = 0;
  h->moving_changed_flag              = 0;
  h->local_map_in_sensor_changed_flag = 0;
  if (h->n_correspondences == 0) {
    flags |= ORC_WARN_NO_MATCHES;
  }
  return flags;
}

int orc_pcf_compute(orc_pcf* h, orc_corr* out, int capacity, int* n_out) {
  if (!h || !out || !n_out) {
    return ORC_ERR_NULL;
  }
  const int flags = pcf_compute_internal(h, ORC_OK);
  if (flags < 0) {
    return flags;
  }
  if (capacity < h->n_correspondences) {
    return ORC_ERR_CAPACITY;
  }
  memcpy(out, h->correspondences, sizeof(orc_corr) * (size_t) h->n_correspondences);
  *n_out = h->n_correspondences;
  return flags;
}

/* ------------------------------------------------------------------------------------------ */
/* a11-a13: aligner slice                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* AlignerSliceProcessorProjective_::setupFactor, information scaling by landmark age
 * (registration/aligner_slice_processor_projective.cpp:46-52): diagonal_info *= (1 + std::log(n))
 * with the double result narrowed to the float scalar type before the product. */
void orc_info_scale_from_nopt(const uint32_t* n_opt, int n, float* scale) {
  for (int i = 0; i < n; ++i) {
    scale[i] = n_opt[i] > 2 ? (float) (1.0 + log((double) n_opt[i])) : 1.0f;
  }
}

/* AlignerSliceProcessorProjectiveStereo_::bindFixed (aligner_slice_processor_projective.cpp:76-89) */
float orc_mean_disparity(const float* fixed_uvuv, int n) {
  if (n <= 0) {
    return 0.0f;
  }
  float accumulated_disparity = 0;
  for (int i = 0; i < n; ++i) {
    accumulated_disparity += fixed_uvuv[4 * i + 0] - fixed_uvuv[4 * i + 2];
  }
  return accumulated_disparity / (float) (size_t) n;
}

/* SE3ProjectiveErrorFactor / SE3ProjectiveDepthErrorFactor / SE3RectifiedStereoProjectiveErrorFactor
 * ::errorAndJacobian + RobustifierSaturated + H/b accumulation (srrg2_solver, external).
 * BUILD-DEFINED restatement following SURVEY.md Appendix A; in-repo evidence:
 * mapping/landmarks/filters/stereo_projective_point_ekf_impl.cpp:21-46 (pinhole/stereo Jacobian),
 * mapping/landmarks/landmark_estimator_pose_based_smoother_impl.cpp:77-106 (saturated kernel,
 * H += J^T Omega J, b += J^T Omega e), tests/fixtures.hpp:360-366 ((K p - b)/z).
 * Accumulation (BUILD-DEFINED: the upstream factor loop and its sum order are external and unpinned,
 * SURVEY.md Appendix A "Accumulation & step"; north_star prescribes a tree reduction): every one of the
 * 29 normal-equation sums (21 H upper + 6 b + 2 chi) is a FIXED-SHAPE float reduction over the
 * correspondence vector, see orc_sum128_* below. */

/* Fixed-shape sum of a sequence t_0, t_1, ... (float, round to nearest):
 *   leaf[l] (l = 0..127)  = ((+0 + t_l) + t_{l+128}) + t_{l+256} ...        (terms of index = l mod 128, in order;
 *                           a term that is a sum of products enters through fused multiply-adds, orc_sum128_fma3)
 *   seven pairwise levels  v[l] <- v[l] + v[l ^ m]  for m = 32, 16, 8, 7, 2, 1, 64 (in this order)
 *   result                 = v[0] + 0.0f                                     (the sign of a zero sum is +)
 * i.e. 128 interleaved partial sums followed by a balanced binary tree (every level pairs clusters of equal
 * size: {l, l^7} then {.., l^2, l^5} then all eight of an aligned group of 8, ...); float addition is
 * commutative, so after a level both partners hold the same value.  The device evaluates the same tree with
 * one lane per leaf (two wavefronts): the masks are the lane exchanges gfx950 has as single instructions
 * (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8, row_half_mirror, quad_perm; the two waves meet in
 * LDS) -- srrg2_proslam_amd/csrc/align.hip, wave_sum_slots. */
#define ORC_SUM_LEAVES 128
typedef struct {
  float leaf[ORC_SUM_LEAVES];
} orc_sum128;

static void orc_sum128_init(orc_sum128* s) {
  for (int l = 0; l < ORC_SUM_LEAVES; ++l) {
    s->leaf[l] = 0.0f;
  }
}

static inline void orc_sum128_add(orc_sum128* s, int index, float term) {
  s->leaf[index & (ORC_SUM_LEAVES - 1)] += term;
}

/* leaf <- a2*b2 + (a1*b1 + (a0*b0 + leaf)), one rounding per product-sum */
static inline void orc_sum128_fma3(orc_sum128* s, int index, float a0, float b0, float a1, float b1, float a2, float b2) {
  float* leaf = &s->leaf[index & (ORC_SUM_LEAVES - 1)];
  *leaf       = fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, *leaf)));
}

/* leaf <- a*b + leaf and leaf <- a1*b1 + (a0*b0 + leaf) */
static inline void orc_sum128_fma1(orc_sum128* s, int index, float a, float b) {
  float* leaf = &s->leaf[index & (ORC_SUM_LEAVES - 1)];
  *leaf       = fmaf(a, b, *leaf);
}
static inline void orc_sum128_fma2(orc_sum128* s, int index, float a0, float b0, float a1, float b1) {
  float* leaf = &s->leaf[index & (ORC_SUM_LEAVES - 1)];
  *leaf       = fmaf(a1, b1, fmaf(a0, b0, *leaf));
}

static float orc_sum128_result(const orc_sum128* s) {
  static const int level_mask[7] = {32, 16, 8, 7, 2, 1, 64};
  float v[ORC_SUM_LEAVES];
  memcpy(v, s->leaf, sizeof(v));
  for (int k = 0; k < 7; ++k) {
    const int m = level_mask[k];
    for (int l = 0; l < ORC_SUM_LEAVES; ++l) {
      const int p = l ^ m;
      if (l < p) {
        const float sum = v[l] + v[p];
        v[l]            = sum;
        v[p]            = sum;
      }
    }
  }
  return v[0] + 0.0f;
}

static orc_variant g_variant; /* sweep switches, all zero by default (proslam_oracle.h) */
void orc_set_variant(const orc_variant* v) {
  if (v) {
    g_variant = *v;
  } else {
    memset(&g_variant, 0, sizeof(g_variant));
  }
}

void orc_linearize(const orc_aligner_params* P,
                   const float* X,
                   const orc_corr* corr,
                   int n_corr,
                   const float* fixed,
                   const float* moving_xyz,
                   const float* info_scale,
                   orc_linear_system* out) {
  orc_linearize_ex(P, X, corr, n_corr, fixed, moving_xyz, info_scale, 0, NULL, out);
}

/* points -> camera: X, or sensor_in_robot^-1 * X for the ...WithSensor factors */
static void pose_to_camera(const orc_aligner_params* P, const float* X, float* A) {
  if (P->with_sensor) {
    float Si[16];
    orc_se3_inverse(P->sensor_in_robot, Si);
    orc_se3_mul(Si, X, A);
  } else {
    memcpy(A, X, 16 * sizeof(float));
  }
}

/* the device's test (csrc/align.hip pose_is_finite): the sum of the twelve entries, in this order, is finite */
static int pose_is_finite12(const float* X) {
  const float s = ((((X[0] + X[1]) + (X[2] + X[3])) + ((X[4] + X[5]) + (X[6] + X[7]))) + ((X[8] + X[9]) + (X[10] + X[11])));
  return (s - s) == 0.0f;
}

/* camera frame -> tangent space of the estimate: H <- Rt^T H Rt, b <- Rt^T b with Rt = blockdiag(R, R), R the rotation of A.
 * Row r = 3 * blk + i of the result: v = (column i of R)^T Y, out = v R for the two 3 x 3 blocks Y of block row blk; the rotated
 * matrix is symmetric up to rounding: its LOWER triangle (row >= column) is the system and is mirrored into the upper one
 * (BUILD-DEFINED; csrc/prs_se3.h rotate_normal_equations performs the same operations). */
static void rotate_normal_equations(const float* A, float* H, float* b) {
  float Hn[36], bn[6];
  for (int r = 0; r < 6; ++r) {
    const int blk = r >= 3 ? 1 : 0;
    const int i   = r - 3 * blk;
    const float Ri0 = A[i], Ri1 = A[4 + i], Ri2 = A[8 + i];
    const float* Y = H + 18 * blk;
    for (int cb = 0; cb < 2; ++cb) {
      float v[3];
      for (int c = 0; c < 3; ++c) {
        v[c] = fmaf(Ri2, Y[12 + 3 * cb + c], fmaf(Ri1, Y[6 + 3 * cb + c], Ri0 * Y[3 * cb + c]));
      }
      for (int j = 0; j < 3; ++j) {
        Hn[6 * r + 3 * cb + j] = fmaf(v[2], A[8 + j], fmaf(v[1], A[4 + j], v[0] * A[j]));
      }
    }
    bn[r] = fmaf(Ri2, b[3 * blk + 2], fmaf(Ri1, b[3 * blk + 1], Ri0 * b[3 * blk]));
  }
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c <= r; ++c) {
      H[6 * r + c] = Hn[6 * r + c];
      H[6 * c + r] = Hn[6 * r + c];
    }
    b[r] = bn[r];
  }
}

void orc_linearize_ex(const orc_aligner_params* P,
                      const float* X_robot,
                      const orc_corr* corr,
                      int n_corr,
                      const float* fixed,
                      const float* moving_xyz,
                      const float* info_scale,
                      int inlier_only,
                      uint8_t* cls_out,
                      orc_linear_system* out) {
  memset(out, 0, sizeof(*out));
  float X[16];
  pose_to_camera(P, X_robot, X);
  const int dim  = P->factor_type;
  const int edim = dim == ORC_FACTOR_MONO ? 2 : 3;
  const float R00 = X[0], R01 = X[1], R02 = X[2], t0 = X[3];
  const float R10 = X[4], R11 = X[5], R12 = X[6], t1 = X[7];
  const float R20 = X[8], R21 = X[9], R22 = X[10], t2 = X[11];
  const float fx = P->fx, fy = P->fy, cx = P->cx, cy = P->cy;
  /* 21 H (upper triangle, row-major) + 6 b + chi_inliers + chi_total */
  orc_sum128 sums[29];
  for (int t = 0; t < 29; ++t) {
    orc_sum128_init(&sums[t]);
  }

  for (int ic = 0; ic < n_corr; ++ic) {
    const int f     = corr[ic].fixed_idx;
    const int m     = corr[ic].moving_idx;
    const float* z  = fixed + (size_t) dim * (size_t) f;
    const float px = moving_xyz[3 * m + 0], py = moving_xyz[3 * m + 1], pz = moving_xyz[3 * m + 2];

    /* point in camera, homogeneous image point */
    /* the factor arithmetic is written with explicit fused multiply-adds (one rounding per fmaf,
     * the same on every IEEE machine): BUILD-DEFINED like the rest of the external factor code */
    const float pcx = fmaf(R02, pz, fmaf(R01, py, fmaf(R00, px, t0)));
    const float pcy = fmaf(R12, pz, fmaf(R11, py, fmaf(R10, px, t1)));
    const float pcz = fmaf(R22, pz, fmaf(R21, py, fmaf(R20, px, t2)));
    const float hx  = fmaf(fx, pcx, cx * pcz);
    const float hy  = fmaf(fy, pcy, cy * pcz);
    const float hz  = pcz;
    if (cls_out) {
      cls_out[ic] = 2;
    }
    if (!(hz > 0.0f)) {
      ++out->num_invalid;
      continue;
    }
    const float iz     = 1.0f / hz;
    const float u_pred = hx * iz;
    const float v_pred = hy * iz;
    if (g_variant.bounds_form == 0 && (u_pred < 0.0f || u_pred > P->image_cols || v_pred < 0.0f || v_pred > P->image_rows)) {
      ++out->num_invalid;
      continue;
    }
    if (g_variant.bounds_form == 2 && (u_pred < 0.0f || u_pred >= P->image_cols || v_pred < 0.0f || v_pred >= P->image_rows)) {
      ++out->num_invalid;
      continue;
    }

    float e[3];
    e[0]      = u_pred - z[0];
    e[1]      = v_pred - (g_variant.v_row == 1 && dim == ORC_FACTOR_STEREO ? 0.5f * (z[1] + z[3]) : z[1]);
    e[2]      = 0.0f;
    float hrx = hx;
    if (dim == ORC_FACTOR_STEREO) {
      hrx  = hx + P->baseline_left_in_right_px[0];
      e[2] = fmaf(hrx, iz, -z[2]);
    } else if (dim == ORC_FACTOR_DEPTH) {
      e[2] = hz - z[2];
    }

    /* translation weight ("use normalized disparity as weight for translation contribution in jac: (0.01+d,1)*I",
     * aligner_slice_processor_projective.cpp:107-112): wt = min(0.01 + d / mean disparity, 1).  The factor itself is external;
     * this reading is the round-4 result of tools/sweep_a13.py (forms 1.. are the other readings it scored). */
    float wt = 1.0f, w_omega = 1.0f;
    if (dim == ORC_FACTOR_STEREO && P->enable_inverse_depth_weighting) {
      const float dn = (z[0] - z[2]) / P->mean_disparity;
      switch (g_variant.idw_form) {
        case 0: /* shipped */
          wt = 0.01f + dn;
          wt = wt < 1.0f ? (wt >= -1.0e19f ? wt : 1.0f) : 1.0f; /* NaN (0 / 0), +-inf, or a square that is not finite -> 1: wt and wt^2 are always finite */
          break;
        case 1: wt = !(dn >= 0.01f) ? 0.01f : (dn > 1.0f ? 1.0f : dn); break; /* (NaN -> 0.01) */
        case 2: wt = sqrtf(dn < 0.01f ? 0.01f : (dn > 1.0f ? 1.0f : dn)); break;
        case 3: w_omega = dn < 0.01f ? 0.01f : (dn > 1.0f ? 1.0f : dn); break;
        case 4: wt = 1.0f / dn; wt = wt < 0.01f ? 0.01f : (wt > 1.0f ? 1.0f : wt); break;
        case 5: wt = dn < 0.01f ? 0.01f : dn; break;
        case 6: wt = dn < 0.01f ? 0.01f : (dn > 1.0f ? 1.0f : dn); wt *= wt; break;
        default: break; /* off */
      }
    }

    /* J = D * R * [ wt*I | -2 [p]x ]  (3x6), D = d(image point)/d(point in camera):
     *   row 0 (u):  (fx/z, 0, (cx - u)/z)     row 1 (v):  (0, fy/z, (cy - v)/z)
     *   row 2:      stereo (uR): (fx/z, 0, (cx - uR)/z); depth: (0, 0, 1); mono: none. */
    const float alpha = fx * iz, gamma = fy * iz;
    const float beta0 = (cx - u_pred) * iz;
    const float beta1 = (cy - v_pred) * iz;
    const float beta2 = (cx - hrx * iz) * iz; /* stereo only */

    /* Omega = diag(info) * scale(moving) (aligner_slice_processor_projective.cpp:46-56) */
    const float s = info_scale ? info_scale[m] : 1.0f;
    float o[3];
    o[0] = P->diagonal_info[0] * s;
    o[1] = P->diagonal_info[1] * s;
    o[2] = edim == 3 ? P->diagonal_info[2] * s : 0.0f;
    o[0] *= w_omega;
    o[1] *= w_omega;
    o[2] *= w_omega;

    /* chi2 + saturated kernel (landmark_estimator_pose_based_smoother_impl.cpp:77-84) */
    float chi = fmaf(o[2] * e[2], e[2], fmaf(o[1] * e[1], e[1], (o[0] * e[0]) * e[0]));
    if (chi > P->chi_threshold || (g_variant.chi_compare == 1 && chi == P->chi_threshold)) {
        /* RobustifierSaturated (srrg2_solver, external): the kernelised factor is weighted by 1 / chi -- round-4 result of
       * tools/sweep_a13.py; forms 1.. are the other readings (1 = the in-repo smoother's tau / chi,
       * landmark_estimator_pose_based_smoother_impl.cpp:81-84) */
      float scale = 1.0f / chi;
      if (g_variant.kernel_form == 1) {
        scale = P->chi_threshold / chi;
      } else if (g_variant.kernel_form == 2) {
        scale = sqrtf(P->chi_threshold / chi);
      } else if (g_variant.kernel_form == 3) {
        scale = 0.0f;
      }
      if (inlier_only) {
        scale = 0.0f; /* inlier-only run: a kernelised factor contributes nothing (weights * 0 keeps the +-0 terms in the sums) */
      }
      o[0] *= scale;
      o[1] *= scale;
      o[2] *= scale;
      chi = P->chi_threshold;
      ++out->num_outliers;
      if (cls_out) {
        cls_out[ic] = 1;
      }
    } else {
      ++out->num_inliers;
      orc_sum128_add(&sums[27], ic, chi);
      if (cls_out) {
        cls_out[ic] = 0;
      }
    }
    orc_sum128_add(&sums[28], ic, chi);

    if (g_variant.accum_form == 1) {
      /* the rounds 1-3 evaluation (kept for tests/test_oracle_aligner_ext.py: both forms are the same normal equations up to
       * rounding): Q = D * R, J_i = ( wt * Q_i | a x Q_i ) with a = 2 p, every entry of J^T Omega J / J^T Omega e through three
       * fused multiply-adds, sums in the tangent space of X directly */
      const float ax = 2.0f * px, ay = 2.0f * py, az = 2.0f * pz;
      const float Rm[3][3] = {{R00, R01, R02}, {R10, R11, R12}, {R20, R21, R22}};
      float Q[3][3], J[3][6];
      for (int c = 0; c < 3; ++c) {
        Q[0][c] = fmaf(alpha, Rm[0][c], beta0 * Rm[2][c]);
        Q[1][c] = fmaf(gamma, Rm[1][c], beta1 * Rm[2][c]);
        Q[2][c] = dim == ORC_FACTOR_STEREO ? fmaf(alpha, Rm[0][c], beta2 * Rm[2][c]) : (dim == ORC_FACTOR_DEPTH ? Rm[2][c] : 0.0f);
      }
      for (int i = 0; i < 3; ++i) {
        J[i][0] = Q[i][0] * wt;
        J[i][1] = Q[i][1] * wt;
        J[i][2] = Q[i][2] * wt;
        J[i][3] = fmaf(ay, Q[i][2], -(az * Q[i][1]));
        J[i][4] = fmaf(az, Q[i][0], -(ax * Q[i][2]));
        J[i][5] = fmaf(ax, Q[i][1], -(ay * Q[i][0]));
      }
      int t = 0;
      for (int r = 0; r < 6; ++r) {
        const float j0 = J[0][r] * o[0];
        const float j1 = J[1][r] * o[1];
        const float j2 = J[2][r] * o[2];
        for (int c = r; c < 6; ++c) {
          orc_sum128_fma3(&sums[t++], ic, j0, J[0][c], j1, J[1][c], j2, J[2][c]);
        }
        orc_sum128_fma3(&sums[21 + r], ic, j0, e[0], j1, e[1], j2, e[2]);
      }
      continue;
    }

    /* CAMERA-FRAME sums (BUILD-DEFINED operation order, one rounding per fmaf / product; the device performs the same,
     * csrc/align.hip factor_accumulate).  R [a]x = [R a]x R, so J = D G_c Rt with G_c = [ wt I | -[y]x ], y = 2 R p and
     * Rt = blockdiag(R, R), the same for every correspondence: summed are G_c^T K G_c (K = D^T Omega D, K01 = 0) and
     * G_c^T D^T Omega e; Rt is applied once to the summed system below (rotate_normal_equations). */
    {
      const float d20 = dim == ORC_FACTOR_STEREO ? alpha : 0.0f;
      const float d22 = dim == ORC_FACTOR_STEREO ? beta2 : (dim == ORC_FACTOR_DEPTH ? 1.0f : 0.0f);
      const float yx = 2.0f * (pcx - t0), yy = 2.0f * (pcy - t1), yz = 2.0f * (pcz - t2);
      const float p0 = o[0] * alpha, p2 = o[2] * d20, g1 = o[1] * gamma, q0 = o[0] * beta0, q1 = o[1] * beta1, q2 = o[2] * d22;
      const float K00 = fmaf(p2, d20, p0 * alpha), K02 = fmaf(p2, d22, p0 * beta0), K11 = g1 * gamma, K12 = g1 * beta1;
      const float K22 = fmaf(q2, d22, fmaf(q1, beta1, q0 * beta0));
      const float w0 = o[0] * e[0], w1 = o[1] * e[1], w2 = o[2] * e[2];
      const float r0 = fmaf(d20, w2, alpha * w0), r1 = gamma * w1, r2 = fmaf(d22, w2, fmaf(beta1, w1, beta0 * w0));
      /* N = K [y]x */
      const float N00 = -(K02 * yy), N01 = fmaf(K02, yx, -(K00 * yz)), N02 = K00 * yy;
      const float N10 = fmaf(K11, yz, -(K12 * yy)), N11 = K12 * yx, N12 = -(K11 * yx);
      const float N20 = fmaf(K12, yz, -(K22 * yy)), N21 = fmaf(K22, yx, -(K02 * yz)), N22 = fmaf(K02, yy, -(K12 * yx));
      const float wt2 = wt * wt, nwt = -wt;
      /* upper triangle, row-major: (0,0) 0, (0,1) 1 [receives nothing], (0,2) 2, (0,3..5) 3..5, (1,1) 6, (1,2) 7, (1,3..5) 8..10,
       * (2,2) 11, (2,3..5) 12..14, (3,3) 15, (3,4) 16, (3,5) 17, (4,4) 18, (4,5) 19, (5,5) 20 */
      orc_sum128_fma1(&sums[0], ic, wt2, K00);
      orc_sum128_fma1(&sums[2], ic, wt2, K02);
      orc_sum128_fma1(&sums[6], ic, wt2, K11);
      orc_sum128_fma1(&sums[7], ic, wt2, K12);
      orc_sum128_fma1(&sums[11], ic, wt2, K22);
      orc_sum128_fma1(&sums[3], ic, nwt, N00);
      orc_sum128_fma1(&sums[4], ic, nwt, N01);
      orc_sum128_fma1(&sums[5], ic, nwt, N02);
      orc_sum128_fma1(&sums[8], ic, nwt, N10);
      orc_sum128_fma1(&sums[9], ic, nwt, N11);
      orc_sum128_fma1(&sums[10], ic, nwt, N12);
      orc_sum128_fma1(&sums[12], ic, nwt, N20);
      orc_sum128_fma1(&sums[13], ic, nwt, N21);
      orc_sum128_fma1(&sums[14], ic, nwt, N22);
      /* H_rr = [y]x^T N */
      orc_sum128_fma2(&sums[15], ic, -yy, N20, yz, N10);
      orc_sum128_fma2(&sums[16], ic, -yy, N21, yz, N11);
      orc_sum128_fma2(&sums[17], ic, -yy, N22, yz, N12);
      orc_sum128_fma2(&sums[18], ic, -yz, N01, yx, N21);
      orc_sum128_fma2(&sums[19], ic, -yz, N02, yx, N22);
      orc_sum128_fma2(&sums[20], ic, -yx, N12, yy, N02);
      /* b = G_c^T r */
      orc_sum128_fma1(&sums[21], ic, wt, r0);
      orc_sum128_fma1(&sums[22], ic, wt, r1);
      orc_sum128_fma1(&sums[23], ic, wt, r2);
      orc_sum128_fma2(&sums[24], ic, -yz, r1, yy, r2);
      orc_sum128_fma2(&sums[25], ic, -yx, r2, yz, r0);
      orc_sum128_fma2(&sums[26], ic, -yy, r0, yx, r1);
    }
  }
  int t = 0;
  for (int r = 0; r < 6; ++r) {
    for (int c = r; c < 6; ++c) {
      const float h     = orc_sum128_result(&sums[t++]);
      out->H[6 * r + c] = h;
      out->H[6 * c + r] = h;
    }
    out->b[r] = orc_sum128_result(&sums[21 + r]);
  }
  out->chi_inliers = orc_sum128_result(&sums[27]);
  out->chi_total   = orc_sum128_result(&sums[28]);
  if (g_variant.accum_form != 1 && pose_is_finite12(X)) { /* (a pose that is not finite: every correspondence was invalid, the sums are zero) */
    rotate_normal_equations(X, out->H, out->b);
  }
}

/* IterationAlgorithmGN with damping + dense solve (configurations/kitti.conf:20-22,310-315; the solver is external)
 * and VariableSE3QuaternionRight::applyPerturbation: X <- X * exp(dx) */
int orc_gn_step(const orc_linear_system* sys, float damping, float* X) {
  /* (H + damping diag(H)) dx = -b by LDL^T with fused multiply-subtracts and one reciprocal per pivot (BUILD-DEFINED; the device
   * performs the same operations, csrc/prs_se3.h ldlt_solve6).  L unit lower triangular, U[i][j] = L[i][j] * d_j the entry before
   * its division; the LOWER triangle of H is read.  Rounds 1-3 used the Cholesky factor (six sqrt + six divisions): same system,
   * same damping, different rounding. */
  float L[6][6], U[6][6], inv[6];
  memset(L, 0, sizeof(L));
  memset(U, 0, sizeof(U));
  for (int j = 0; j < 6; ++j) {
    /* damping: H_jj <- H_jj + damping * H_jj (round-4 result of tools/sweep_a13.py; form 1 = H_jj + damping) */
    float d = g_variant.damping_form == 1 ? sys->H[6 * j + j] + damping : fmaf(damping, sys->H[6 * j + j], sys->H[6 * j + j]);
    for (int k = 0; k < j; ++k) {
      d = fmaf(-L[j][k], U[j][k], d);
    }
    if (!(d > 0.0f)) {
      return 1;
    }
    inv[j] = 1.0f / d;
    for (int i = j + 1; i < 6; ++i) {
      float v = sys->H[6 * i + j];
      for (int k = 0; k < j; ++k) {
        v = fmaf(-L[i][k], U[j][k], v);
      }
      U[i][j] = v;
      L[i][j] = v * inv[j];
    }
  }
  float y[6], dx[6];
  for (int i = 0; i < 6; ++i) {
    float v = -sys->b[i];
    for (int k = 0; k < i; ++k) {
      v = fmaf(-L[i][k], y[k], v);
    }
    y[i] = v;
  }
  for (int i = 5; i >= 0; --i) {
    float v = y[i] * inv[i];
    for (int k = i + 1; k < 6; ++k) {
      v = fmaf(-L[k][i], dx[k], v);
    }
    dx[i] = v;
  }
  float D[16], Xn[16];
  orc_tnq2t(dx, D);
  orc_se3_mul(X, D, Xn);
  memcpy(X, Xn, sizeof(Xn));
  return 0;
}

void orc_add_motion_prior(const orc_aligner_params* P, const float* X, const float* prior_mean, orc_linear_system* sys) {
  if (!P->enable_motion_prior) {
    return;
  }
  float e[6];
  if (prior_mean) {
    float Zi[16], D[16];
    orc_se3_inverse(prior_mean, Zi);
    orc_se3_mul(Zi, X, D);
    orc_t2tnq(D, e);
  } else {
    orc_t2tnq(X, e);
  }
  for (int i = 0; i < 6; ++i) {
    sys->H[7 * i] += P->motion_prior_info[i];
    sys->b[i] += P->motion_prior_info[i] * e[i];
  }
}

void orc_motion_predict(const float* pose_prev2, const float* pose_prev1, float* pose_pred) {
  float inv2[16], motion[16], raw[16], v[6];
  orc_se3_inverse(pose_prev2, inv2);
  orc_se3_mul(inv2, pose_prev1, motion);
  orc_se3_mul(pose_prev1, motion, raw);
  /* the rotation goes through its unit quaternion: se3_inverse transposes, so a rotation block that has drifted from
   * orthonormality by d comes back as 2 d + d' from this recursion (growth ~2.4x per frame: 1e-7 of float rounding is
   * centimetres of pose error after a dozen frames); the round trip keeps the prediction a rigid transform */
  orc_t2tnq(raw, v);
  orc_tnq2t(v, pose_pred);
}

/* MultiAligner3DQR::compute (srrg2_slam_interfaces, external) restated minimally per SURVEY.md
 * section 8 row a14: fixed number of iterations (termination_criteria unset,
 * configurations/kitti.conf:1006-1009), status by inlier count (tests/test_aligners.cpp:117-121),
 * the two inlier flags of the RGB-D configurations as defined in proslam_oracle.h. */
void orc_align_frame(orc_pcf* finder,
                     const orc_aligner_params* P,
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
                     orc_align_result* result) {
  orc_align_frame_ex(finder, P, fixed, n_fixed, moving_xyz, info_scale, n_moving, X_init, prior_H, prior_b, NULL, corr_out, n_corr_out, result);
}

void orc_align_frame_ex(orc_pcf* finder,
                        const orc_aligner_params* P,
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
                        orc_align_result* result) {
  (void) n_moving;
  float X[16];
  memcpy(X, X_init, sizeof(X));
  int warnings = 0;
  int n_corr   = 0;
  orc_linear_system sys;
  memset(&sys, 0, sizeof(sys));
  uint8_t* cls  = (uint8_t*) malloc((size_t) (n_fixed > 0 ? n_fixed : 1));
  int have_cls  = 0;
  int it        = 0;
  int failed    = 0;
  const int extra = P->enable_inlier_only_runs ? (P->inlier_only_iterations > 0 ? P->inlier_only_iterations : P->max_iterations) : 0;
  for (; it < P->max_iterations + extra; ++it) {
    const int inlier_run = it >= P->max_iterations;
    if (it == P->max_iterations && sys.num_inliers < P->min_num_inliers) {
      break; /* not enough inliers for an inlier-only run */
    }
    if (!inlier_run) {
      float A[16];
      pose_to_camera(P, X, A);
      orc_pcf_set_local_map_in_sensor(finder, A);
      const int flags = orc_pcf_compute(finder, corr_out, n_fixed, &n_corr);
      if (flags < 0) {
        warnings = flags;
        failed   = 1;
        break;
      }
      warnings |= flags;
    }
    if (n_corr < P->min_num_correspondences) {
      memset(&sys, 0, sizeof(sys));
      have_cls = 0;
      continue; /* slice has too few correspondences: no update this iteration */
    }
    orc_linearize_ex(P, X, corr_out, n_corr, fixed, moving_xyz, info_scale, inlier_run, cls, &sys);
    have_cls = 1;
    orc_linear_system total = sys;
    if (prior_H && prior_b) {
      for (int i = 0; i < 36; ++i) {
        total.H[i] += prior_H[i];
      }
      for (int i = 0; i < 6; ++i) {
        total.b[i] += prior_b[i];
      }
    }
    orc_add_motion_prior(P, X, prior_mean, &total);
    orc_gn_step(&total, P->damping, X);
  }
  if (!failed && P->keep_only_inlier_correspondences && have_cls) {
    int k = 0;
    for (int i = 0; i < n_corr; ++i) {
      if (cls[i] == 0) {
        corr_out[k++] = corr_out[i];
      }
    }
    n_corr = k;
  }
  free(cls);
  memcpy(result->X, X, sizeof(X));
  result->iterations          = it;
  result->num_inliers         = sys.num_inliers;
  result->num_correspondences = n_corr;
  result->status              = sys.num_inliers >= P->min_num_inliers ? 1 : 0;
  result->warnings            = warnings;
  *n_corr_out                 = n_corr;
}

/* ------------------------------------------------------------------------------------------ */
/* section 8f next #4: bijective brute-force matcher                                           */
/* ------------------------------------------------------------------------------------------ */

static int float_cmp(const void* a, const void* b) {
  const float x = *(const float*) a, y = *(const float*) b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* canonical order for the unstable std::sort by response (bruteforce_impl.cpp:89-92):
 * (response, fixed, moving) */
static int corr_cmp(const void* a_, const void* b_) {
  const orc_corr* a = (const orc_corr*) a_;
  const orc_corr* b = (const orc_corr*) b_;
  if (a->response != b->response) {
    return a->response < b->response ? -1 : 1;
  }
  if (a->fixed_idx != b->fixed_idx) {
    return a->fixed_idx < b->fixed_idx ? -1 : 1;
  }
  if (a->moving_idx != b->moving_idx) {
    return a->moving_idx < b->moving_idx ? -1 : 1;
  }
  return 0;
}

/* checkLowesRatio, scalar form (bruteforce_impl.cpp:159-176) */
static int check_lowes_ratio_scalar(float best, float other, float maximum_ratio) {
  if (best == other) {
    return 0;
  }
  return best / other < maximum_ratio;
}

/* checkLowesRatio, vector form (bruteforce_impl.cpp:180-199); distances sorted ascending */
static int check_lowes_ratio(float best, const float* distances, int n, float maximum_ratio) {
  if (n == 1) {
    return 1;
  }
  float second_best = best;
  for (int i = 0; i < n; ++i) {
    if (distances[i] > best) {
      second_best = distances[i];
      break;
    }
  }
  return check_lowes_ratio_scalar(best, second_best, maximum_ratio);
}

typedef struct {
  const float* dist_fixed;    /* per fixed: sorted ascending, start_fixed offsets */
  const int32_t* start_fixed;
  const float* dist_moving;
  const int32_t* start_moving;
  uint8_t* registered_fixed;
  uint8_t* registered_moving;
  int n_registered_fixed;
  int n_registered_moving;
  float maximum_ratio;
} bf_state;

/* _processCorrespondencePool (bruteforce_impl.cpp:247-293) */
static void bf_process_pool(const orc_corr* pool, int n_pool, bf_state* s, orc_corr* out, int* n_out) {
  for (int j = 0; j < n_pool; ++j) {
    int is_unique_in_pool = 1;
    for (int k = 0; k < n_pool; ++k) {
      if (j != k && (pool[j].fixed_idx == pool[k].fixed_idx ||
                     pool[j].moving_idx == pool[k].moving_idx)) {
        is_unique_in_pool = 0;
      }
    }
    if (!is_unique_in_pool) {
      continue;
    }
    const int f = pool[j].fixed_idx, m = pool[j].moving_idx;
    if (check_lowes_ratio(pool[j].response,
                          s->dist_fixed + s->start_fixed[f],
                          s->start_fixed[f + 1] - s->start_fixed[f],
                          s->maximum_ratio) &&
        check_lowes_ratio(pool[j].response,
                          s->dist_moving + s->start_moving[m],
                          s->start_moving[m + 1] - s->start_moving[m],
                          s->maximum_ratio)) {
      out[(*n_out)++] = pool[j];
      if (!s->registered_fixed[f]) {
        s->registered_fixed[f] = 1;
        ++s->n_registered_fixed;
      }
      if (!s->registered_moving[m]) {
        s->registered_moving[m] = 1;
        ++s->n_registered_moving;
      }
    }
  }
}

/* CorrespondenceFinderDescriptorBasedBruteforce::compute (bruteforce_impl.cpp:8-155) */
int orc_bruteforce_match(const uint8_t* desc_fixed,
                         int n_fixed,
                         const uint8_t* desc_moving,
                         int n_moving,
                         float maximum_descriptor_distance,
                         float maximum_distance_ratio,
                         orc_corr* out,
                         int capacity,
                         int* n_out) {
  if (!out || !n_out || (n_fixed > 0 && !desc_fixed) || (n_moving > 0 && !desc_moving)) {
    return ORC_ERR_NULL;
  }
  const int min_nm = n_fixed < n_moving ? n_fixed : n_moving;
  if (capacity < min_nm) {
    return ORC_ERR_CAPACITY;
  }
  int flags = ORC_OK;
  if (n_fixed == 0 || n_moving == 0) {
    flags |= ORC_WARN_EMPTY_INPUT;
  }
  *n_out = 0;

  /* pass 1: count candidates within threshold (:32-75) */
  size_t total        = 0;
  int32_t* cnt_fixed  = (int32_t*) calloc((size_t) n_fixed + 2, sizeof(int32_t));
  int32_t* cnt_moving = (int32_t*) calloc((size_t) n_moving + 2, sizeof(int32_t));
  for (int f = 0; f < n_fixed; ++f) {
    for (int m = 0; m < n_moving; ++m) {
      const float d = (float) orc_hamming256(desc_fixed + (size_t) f * 32, desc_moving + (size_t) m * 32);
      if (d < maximum_descriptor_distance) {
        ++cnt_fixed[f + 1];
        ++cnt_moving[m + 1];
        ++total;
      }
    }
  }
  if (total == 0) {
    free(cnt_fixed);
    free(cnt_moving);
    return flags | ORC_WARN_NO_MATCHES;
  }
  for (int f = 0; f < n_fixed; ++f) {
    cnt_fixed[f + 1] += cnt_fixed[f];
  }
  for (int m = 0; m < n_moving; ++m) {
    cnt_moving[m + 1] += cnt_moving[m];
  }
  orc_corr* candidates = (orc_corr*) malloc(sizeof(orc_corr) * total);
  float* dist_fixed    = (float*) malloc(sizeof(float) * total);
  float* dist_moving   = (float*) malloc(sizeof(float) * total);
  int32_t* cur_moving  = (int32_t*) malloc(sizeof(int32_t) * ((size_t) n_moving + 1));
  memcpy(cur_moving, cnt_moving, sizeof(int32_t) * ((size_t) n_moving + 1));
  size_t k = 0;
  for (int f = 0; f < n_fixed; ++f) {
    for (int m = 0; m < n_moving; ++m) {
      const float d = (float) orc_hamming256(desc_fixed + (size_t) f * 32, desc_moving + (size_t) m * 32);
      if (d < maximum_descriptor_distance) {
        candidates[k].fixed_idx  = f;
        candidates[k].moving_idx = m;
        candidates[k].response   = d;
        dist_fixed[k]            = d;
        dist_moving[cur_moving[m]++] = d;
        ++k;
      }
    }
    qsort(dist_fixed + cnt_fixed[f], (size_t)(cnt_fixed[f + 1] - cnt_fixed[f]), sizeof(float), float_cmp); /* :78 */
  }
  free(cur_moving);

  if (total == 1) { /* :86-91 */
    out[0] = candidates[0];
    *n_out = 1;
    free(candidates);
    free(dist_fixed);
    free(dist_moving);
    free(cnt_fixed);
    free(cnt_moving);
    return flags;
  }
  qsort(candidates, total, sizeof(orc_corr), corr_cmp); /* :94-97 */
  for (int m = 0; m < n_moving; ++m) {                  /* :100-102 */
    qsort(dist_moving + cnt_moving[m], (size_t)(cnt_moving[m + 1] - cnt_moving[m]), sizeof(float), float_cmp);
  }

  bf_state s;
  s.dist_fixed          = dist_fixed;
  s.start_fixed         = cnt_fixed;
  s.dist_moving         = dist_moving;
  s.start_moving        = cnt_moving;
  s.registered_fixed    = (uint8_t*) calloc((size_t) n_fixed + 1, 1);
  s.registered_moving   = (uint8_t*) calloc((size_t) n_moving + 1, 1);
  s.n_registered_fixed  = 0;
  s.n_registered_moving = 0;
  s.maximum_ratio       = maximum_distance_ratio;

  orc_corr* pool = (orc_corr*) malloc(sizeof(orc_corr) * total);
  int n_pool     = 0;
  pool[n_pool++] = candidates[0]; /* :109 */
  int n          = 0;
  for (size_t i = 1; i < total; ++i) { /* :113-147 */
    const orc_corr* c = &candidates[i];
    if (!s.registered_fixed[c->fixed_idx] && !s.registered_moving[c->moving_idx]) {
      if (n_pool > 0 && c->response == pool[n_pool - 1].response) {
        pool[n_pool++] = *c;
      } else {
        bf_process_pool(pool, n_pool, &s, out, &n);
        n_pool = 0;
        if (!s.registered_fixed[c->fixed_idx] && !s.registered_moving[c->moving_idx]) {
          pool[n_pool++] = *c;
        }
      }
    }
    if (s.n_registered_fixed == n_fixed || s.n_registered_moving == n_moving) {
      break;
    }
  }
  if (n_pool > 0) { /* :150-157 */
    bf_process_pool(pool, n_pool, &s, out, &n);
  }
  *n_out = n;
  free(pool);
  free(s.registered_fixed);
  free(s.registered_moving);
  free(candidates);
  free(dist_fixed);
  free(dist_moving);
  free(cnt_fixed);
  free(cnt_moving);
  if (n == 0) {
    flags |= ORC_WARN_NO_MATCHES;
  }
  return flags;
}

/* ------------------------------------------------------------------------------------------ */
/* 8f #2: scene clipper (mapping/scene_clipper_projective_3d.cpp:9-67)                         */
/* ------------------------------------------------------------------------------------------ */
int orc_scene_clip(const orc_projector* proj,
                   const float* robot_in_local_map,
                   const float* sensor_in_robot,
                   const float* scene_xyzw,
                   const uint8_t* scene_desc,
                   int n,
                   float* clipped_xyzw,
                   uint8_t* clipped_desc,
                   int32_t* global_indices,
                   int* n_clipped) {
  if (n <= 0) {
    return ORC_WARN_EMPTY_INPUT; /* :21-28: status Ready, clipped scene untouched */
  }
  float camera_pose[16], W[16];
  orc_se3_mul(robot_in_local_map, sensor_in_robot, camera_pose); /* :46 */
  orc_se3_inverse(camera_pose, W);
  float I[16];
  orc_se3_identity(I);
  /* :61 exact element-wise comparison (-0.0f == 0.0f as in the reference's matrix operator!=) */
  int differs = 0;
  for (int k = 0; k < 16; ++k) {
    if (sensor_in_robot[k] != I[k]) {
      differs = 1;
    }
  }
  const float* S   = sensor_in_robot;
  const float cols = (float) proj->canvas_cols;
  const float rows = (float) proj->canvas_rows;
  int m            = 0;
  for (int i = 0; i < n; ++i) {
    const float px = scene_xyzw[4 * i + 0], py = scene_xyzw[4 * i + 1], pz = scene_xyzw[4 * i + 2];
    /* same projector arithmetic as orc_project (section 8a6) */
    const float x = ((W[0] * px + W[1] * py) + W[2] * pz) + W[3];
    const float y = ((W[4] * px + W[5] * py) + W[6] * pz) + W[7];
    const float z = ((W[8] * px + W[9] * py) + W[10] * pz) + W[11];
    if (z < proj->range_min || z > proj->range_max) {
      continue;
    }
    const float hx = proj->fx * x + proj->cx * z;
    const float hy = proj->fy * y + proj->cy * z;
    const float u  = hx / z;
    const float v  = hy / z;
    if (u < 0.0f || u >= cols || v < 0.0f || v >= rows) {
      continue;
    }
    float ox = x, oy = y, oz = z;
    if (differs) { /* transformInPlace<Isometry>(sensor_in_robot), :61-63 */
      ox = ((S[0] * x + S[1] * y) + S[2] * z) + S[3];
      oy = ((S[4] * x + S[5] * y) + S[6] * z) + S[7];
      oz = ((S[8] * x + S[9] * y) + S[10] * z) + S[11];
    }
    clipped_xyzw[4 * m + 0] = ox;
    clipped_xyzw[4 * m + 1] = oy;
    clipped_xyzw[4 * m + 2] = oz;
    clipped_xyzw[4 * m + 3] = scene_xyzw[4 * i + 3];
    if (scene_desc && clipped_desc) {
      memcpy(clipped_desc + 32 * (size_t) m, scene_desc + 32 * (size_t) i, 32);
    }
    global_indices[m] = i;
    ++m;
  }
  *n_clipped = m;
  return m == 0 ? ORC_WARN_NO_PROJECTION : 0;
}
