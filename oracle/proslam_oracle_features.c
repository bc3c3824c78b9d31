/* proslam_oracle_features.c -- see proslam_oracle_features.h */
#include "proslam_oracle_features.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* 16-pixel Bresenham circle of radius 3, clockwise from 12 o'clock */
static const int kCircle[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0},  {3, 1},   {2, 2},   {1, 3},
                                   {0, 3},  {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

static const int8_t kOrbPattern[1024] = {
#include "orb_pattern.inc"
};

void orc_orb_pattern(int8_t* pattern) {
  memcpy(pattern, kOrbPattern, sizeof(kOrbPattern));
}

/* cv::GaussianBlur 7x7, sigma 2 on CV_8U: getGaussianKernel(7, 2) = exp(-x^2 / 8) normalised, converted to 8 fractional
 * bits (createSeparableLinearFilter: bits = 8 for 8-bit smoothing kernels), row pass in int, column pass
 * (sum + 2^15) >> 16 with saturation; BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba). */
static int reflect101(int i, int n) {
  if (n == 1) {
    return 0;
  }
  while (i < 0 || i >= n) {
    i = i < 0 ? -i : 2 * (n - 1) - i;
  }
  return i;
}

void orc_gaussian_blur7(const uint8_t* image, int rows, int cols, uint8_t* blurred) {
  int k[7];
  {
    double g[7], sum = 0.0;
    for (int i = 0; i < 7; ++i) {
      const double x = (double) i - 3.0;
      g[i]           = exp(-0.5 / (2.0 * 2.0) * x * x);
      sum += g[i];
    }
    for (int i = 0; i < 7; ++i) {
      const float f = (float) (g[i] * (1.0 / sum)); /* the kernel is a CV_32F matrix */
      k[i]          = (int) lrint((double) f * 256.0);
    }
  }
  int32_t* h = (int32_t*) malloc(sizeof(int32_t) * (size_t) rows * cols);
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      int32_t s = 0;
      for (int i = 0; i < 7; ++i) {
        s += k[i] * (int32_t) image[(size_t) r * cols + reflect101(c + i - 3, cols)];
      }
      h[(size_t) r * cols + c] = s;
    }
  }
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      int32_t s = 0;
      for (int i = 0; i < 7; ++i) {
        s += k[i] * h[(size_t) reflect101(r + i - 3, rows) * cols + c];
      }
      s                              = (s + (1 << 15)) >> 16;
      blurred[(size_t) r * cols + c] = (uint8_t) (s > 255 ? 255 : s);
    }
  }
  free(h);
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

/* ---- GNU libstdc++ std::sort (bits/stl_algo.h, bits/stl_heap.h), comparator comp(a, b) = a.response > b.response.
 * Element moves are value copies; only the comparator's verdicts steer the algorithm. ---- */
typedef struct {
  int32_t response, order;
} sitem;
#define SCOMP(a, b) ((a).response > (b).response)

static void s_swap(sitem* a, sitem* b) {
  const sitem t = *a;
  *a            = *b;
  *b            = t;
}

static void s_push_heap(sitem* first, long hole, long top, sitem value) {
  long parent = (hole - 1) / 2;
  while (hole > top && SCOMP(first[parent], value)) {
    first[hole] = first[parent];
    hole        = parent;
    parent      = (hole - 1) / 2;
  }
  first[hole] = value;
}

static void s_adjust_heap(sitem* first, long hole, long len, sitem value) {
  const long top = hole;
  long child     = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (SCOMP(first[child], first[child - 1])) {
      --child;
    }
    first[hole] = first[child];
    hole        = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child       = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole        = child - 1;
  }
  s_push_heap(first, hole, top, value);
}

static void s_pop_heap(sitem* first, sitem* last, sitem* result) {
  const sitem value = *result;
  *result           = *first;
  s_adjust_heap(first, 0, last - first, value);
}

static void s_heapsort(sitem* first, sitem* last) { /* __partial_sort(first, last, last) */
  const long len = last - first;
  if (len >= 2) { /* __make_heap */
    long parent = (len - 2) / 2;
    for (;;) {
      const sitem value = first[parent];
      s_adjust_heap(first, parent, len, value);
      if (parent == 0) {
        break;
      }
      --parent;
    }
  }
  /* __heap_select's loop over [middle, last) is empty for middle == last; __sort_heap: */
  while (last - first > 1) {
    --last;
    s_pop_heap(first, last, last);
  }
}

static void s_move_median_to_first(sitem* result, sitem* a, sitem* b, sitem* c) {
  if (SCOMP(*a, *b)) {
    if (SCOMP(*b, *c)) {
      s_swap(result, b);
    } else if (SCOMP(*a, *c)) {
      s_swap(result, c);
    } else {
      s_swap(result, a);
    }
  } else if (SCOMP(*a, *c)) {
    s_swap(result, a);
  } else if (SCOMP(*b, *c)) {
    s_swap(result, c);
  } else {
    s_swap(result, b);
  }
}

static sitem* s_unguarded_partition(sitem* first, sitem* last, const sitem* pivot) {
  for (;;) {
    while (SCOMP(*first, *pivot)) {
      ++first;
    }
    --last;
    while (SCOMP(*pivot, *last)) {
      --last;
    }
    if (!(first < last)) {
      return first;
    }
    s_swap(first, last);
    ++first;
  }
}

static void s_introsort_loop(sitem* first, sitem* last, long depth_limit) {
  while (last - first > 16) {
    if (depth_limit == 0) {
      s_heapsort(first, last);
      return;
    }
    --depth_limit;
    sitem* mid = first + (last - first) / 2;
    s_move_median_to_first(first, first + 1, mid, last - 1);
    sitem* cut = s_unguarded_partition(first + 1, last, first);
    s_introsort_loop(cut, last, depth_limit);
    last = cut;
  }
}

static void s_unguarded_linear_insert(sitem* last) {
  const sitem val = *last;
  sitem* next     = last - 1;
  while (SCOMP(val, *next)) {
    *last = *next;
    last  = next;
    --next;
  }
  *last = val;
}

static void s_insertion_sort(sitem* first, sitem* last) {
  if (first == last) {
    return;
  }
  for (sitem* i = first + 1; i != last; ++i) {
    if (SCOMP(*i, *first)) {
      const sitem val = *i;
      memmove(first + 1, first, sizeof(sitem) * (size_t) (i - first));
      *first = val;
    } else {
      s_unguarded_linear_insert(i);
    }
  }
}

static void s_std_sort(sitem* first, sitem* last) {
  if (first == last) {
    return;
  }
  long lg = 0; /* std::__lg(n) = floor(log2(n)) */
  for (long n = last - first; n > 1; n >>= 1) {
    ++lg;
  }
  s_introsort_loop(first, last, 2 * lg);
  if (last - first > 16) {
    s_insertion_sort(first, first + 16);
    for (sitem* i = first + 16; i != last; ++i) {
      s_unguarded_linear_insert(i);
    }
  } else {
    s_insertion_sort(first, last);
  }
}

void orc_std_sort_desc(const int32_t* response, int n, int32_t* order_out) {
  sitem* v = (sitem*) malloc(sizeof(sitem) * (size_t) (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    v[i].response = response[i];
    v[i].order    = i;
  }
  s_std_sort(v, v + n);
  for (int i = 0; i < n; ++i) {
    order_out[i] = v[i].order;
  }
  free(v);
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
  const int regions    = nv * nh;
  const float rows_per = (float) rows / (float) nv;
  const float cols_per = (float) cols / (float) nh;
  const int target_per = (int) ((float) P->target_number_of_keypoints / (float) regions); /* :72-76, a size_t member */
  keypoint* sel        = (keypoint*) malloc(sizeof(keypoint) * (size_t) (n > 0 ? n : 1));
  keypoint* bucket     = (keypoint*) malloc(sizeof(keypoint) * (size_t) (n > 0 ? n : 1));
  int32_t* resp        = (int32_t*) malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
  int32_t* perm        = (int32_t*) malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
  int n_sel            = 0;
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
      const int take = target_per;
      if (P->selection_order == ORC_SELECT_LIBSTDCXX) {
        for (int i = 0; i < nb; ++i) {
          resp[i] = bucket[i].response;
        }
        orc_std_sort_desc(resp, nb, perm);
        for (int i = 0; i < take; ++i) {
          sel[n_sel + i] = bucket[perm[i]];
        }
      } else {
        qsort(bucket, (size_t) nb, sizeof(keypoint), by_response);
        memcpy(sel + n_sel, bucket, sizeof(keypoint) * (size_t) take);
      }
      n_sel += take;
    }
  }
  /* cv::ORB::compute: runByImageBorder(edgeThreshold = 31), blur, 256 comparisons */
  uint8_t* blurred = (uint8_t*) malloc((size_t) rows * cols);
  orc_gaussian_blur7(image, rows, cols, blurred);
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
      const int8_t* p = kOrbPattern + 4 * t;
      const int t0    = blurred[(size_t) (r + p[1]) * cols + (c + p[0])];
      const int t1    = blurred[(size_t) (r + p[3]) * cols + (c + p[2])];
      if (t0 < t1) {
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
  free(resp);
  free(perm);
  free(blurred);
  return rc < 0 ? rc : m;
}
