/* proslam_oracle_features.c -- see proslam_oracle_features.h */
#include "proslam_oracle_features.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* 16-pixel Bresenham circle of radius 3, clockwise from 12 o'clock */
static const int kCircle[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0},  {3, 1},   {2, 2},   {1, 3},
                                   {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

void orc_brief_pattern(int8_t* pattern) {
  uint32_t x = 0x12345678u;
  int n      = 0;
  while (n < 256) {
    int v[4];
    for (int k = 0; k < 4; ++k) {
      x    = x * 1664525u + 1013904223u;
      v[k] = (int) ((x >> 8) % 9u) - 4 + (int) ((x >> 16) % 9u) - 4 + (int) ((x >> 24) % 11u) - 5;
    }
    if (v[0] == v[2] && v[1] == v[3]) {
      continue; /* a pair of identical points carries no information */
    }
    for (int k = 0; k < 4; ++k) {
      pattern[4 * n + k] = (int8_t) v[k];
    }
    ++n;
  }
}

/* largest threshold for which (r, c) still passes the FAST-9 segment test, 0 = not even at `threshold` */
static int fast_score_at(const uint8_t* image, int cols, int r, int c, int threshold) {
  const int v = image[(size_t) r * cols + c];
  int d[16];
  for (int i = 0; i < 16; ++i) {
    d[i] = (int) image[(size_t) (r + kCircle[i][1]) * cols + (c + kCircle[i][0])] - v;
  }
  int best = -256;
  for (int k = 0; k < 16; ++k) {
    int lo = 255, hi = 255; /* min of (c_i - v) and of (v - c_i) over the arc k .. k+8 */
    for (int j = 0; j < 9; ++j) {
      const int x = d[(k + j) & 15];
      lo          = x < lo ? x : lo;
      hi          = -x < hi ? -x : hi;
    }
    best = lo > best ? lo : best;
    best = hi > best ? hi : best;
  }
  /* all nine exceed v by more than t'  <=>  t' < best */
  return best > threshold ? best - 1 : 0;
}

void orc_fast_scores(const uint8_t* image, int rows, int cols, int threshold, uint8_t* score) {
  memset(score, 0, (size_t) rows * cols);
  for (int r = 3; r < rows - 3; ++r) {
    for (int c = 3; c < cols - 3; ++c) {
      score[(size_t) r * cols + c] = (uint8_t) fast_score_at(image, cols, r, c, threshold);
    }
  }
}

typedef struct {
  int32_t r, c, response, order;
} keypoint;

static int by_response(const void* a, const void* b) {
  const keypoint* x = (const keypoint*) a;
  const keypoint* y = (const keypoint*) b;
  if (x->response != y->response) {
    return y->response - x->response; /* decreasing response (intensity_feature_extractor_binned.cpp:182-186) */
  }
  return x->order - y->order; /* canonical tie-break */
}

int orc_extract_features(const orc_extractor_params* P,
                         const uint8_t* image,
                         int rows,
                         int cols,
                         float* uv,
                         float* intensity,
                         uint8_t* desc,
                         int capacity) {
  if (!P || !image || rows < 7 || cols < 7 || P->detector_threshold < 1 || P->number_of_detectors_vertical <= 0 ||
      P->number_of_detectors_horizontal <= 0) {
    return -1;
  }
  uint8_t* score = (uint8_t*) malloc((size_t) rows * cols);
  orc_fast_scores(image, rows, cols, P->detector_threshold, score);
  /* keypoints in raster order, optionally non-maximum suppressed */
  keypoint* kp = (keypoint*) malloc(sizeof(keypoint) * ((size_t) rows * cols + 1));
  int n        = 0;
  for (int r = 3; r < rows - 3; ++r) {
    for (int c = 3; c < cols - 3; ++c) {
      const int s = score[(size_t) r * cols + c];
      if (s == 0) {
        continue;
      }
      int keep = 1;
      if (P->enable_non_maximum_suppression) {
        for (int dr = -1; dr <= 1 && keep; ++dr) {
          for (int dc = -1; dc <= 1; ++dc) {
            if ((dr || dc) && score[(size_t) (r + dr) * cols + (c + dc)] >= s) {
              keep = 0;
              break;
            }
          }
        }
      }
      if (keep) {
        kp[n].r        = r;
        kp[n].c        = c;
        kp[n].response = s;
        kp[n].order    = n;
        ++n;
      }
    }
  }
  /* region grid (intensity_feature_extractor_binned.cpp:47-92) */
  const int nv = P->number_of_detectors_vertical, nh = P->number_of_detectors_horizontal;
  const int regions          = nv * nh;
  const float rows_per       = (float) rows / (float) nv;
  const float cols_per       = (float) cols / (float) nh;
  const int target_per       = (int) ((float) P->target_number_of_keypoints / (float) regions); /* :72-76 */
  keypoint* sel              = (keypoint*) malloc(sizeof(keypoint) * (size_t) (n > 0 ? n : 1));
  keypoint* bucket           = (keypoint*) malloc(sizeof(keypoint) * (size_t) (n > 0 ? n : 1));
  int n_sel                  = 0;
  for (int g = 0; g < regions; ++g) {
    int nb = 0;
    for (int i = 0; i < n; ++i) {
      const int region = (int) floorf((float) kp[i].r / rows_per) * nh + (int) ((float) kp[i].c / cols_per); /* :85-92 */
      if (region == g) {
        bucket[nb++] = kp[i];
      }
    }
    if (nb < target_per) { /* :174-178 */
      memcpy(sel + n_sel, bucket, sizeof(keypoint) * (size_t) nb);
      n_sel += nb;
    } else { /* :179-195 */
      qsort(bucket, (size_t) nb, sizeof(keypoint), by_response);
      memcpy(sel + n_sel, bucket, sizeof(keypoint) * (size_t) target_per);
      n_sel += target_per;
    }
  }
  /* descriptors; keypoints too close to the border are removed (OpenCV's runByImageBorder inside compute) */
  int8_t pattern[1024];
  orc_brief_pattern(pattern);
  int m = 0, rc = 0;
  for (int i = 0; i < n_sel; ++i) {
    const int r = sel[i].r, c = sel[i].c;
    if (r < ORC_FEATURE_BORDER || r >= rows - ORC_FEATURE_BORDER || c < ORC_FEATURE_BORDER || c >= cols - ORC_FEATURE_BORDER) {
      continue;
    }
    if (m >= capacity) {
      rc = ORC_ERR_KEYPOINTS;
      break;
    }
    uint8_t* d = desc + 32 * (size_t) m;
    memset(d, 0, 32);
    for (int t = 0; t < 256; ++t) {
      int s[2];
      for (int q = 0; q < 2; ++q) {
        const int x = c + pattern[4 * t + 2 * q], y = r + pattern[4 * t + 2 * q + 1];
        int acc     = 0;
        for (int dy = -2; dy <= 2; ++dy) {
          for (int dx = -2; dx <= 2; ++dx) {
            acc += image[(size_t) (y + dy) * cols + (x + dx)];
          }
        }
        s[q] = acc;
      }
      if (s[0] < s[1]) {
        d[t >> 3] |= (uint8_t) (1u << (t & 7));
      }
    }
    uv[2 * m + 0] = (float) c;
    uv[2 * m + 1] = (float) r;
    intensity[m]  = (float) image[(size_t) r * cols + c]; /* intensity_feature_extractor_base.cpp:80 */
    ++m;
  }
  free(score);
  free(kp);
  free(sel);
  free(bucket);
  return rc < 0 ? rc : m;
}
