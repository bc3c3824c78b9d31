/* proslam_oracle_features.h -- CPU oracle, SURVEY.md section 8f row 3: intensity feature extraction
 * (FAST keypoints, region-binned selection, ORB-256 descriptors).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY PINNED on reference-held golden numbers (tests/test_ref_pins.py, fixtures tests/golden/ref_*.npz made by
 * tools/make_ref_fixtures.py from the reference's own test images): with selection_order = ORC_SELECT_LIBSTDCXX this
 * restatement reproduces every feature count the reference's tests assert on those images
 *   (test_feature_extractors.cpp:22,111,115,131,135,151-165; test_correspondence_finders.cpp:25,62,116,166,204,458,599)
 * and, through its descriptors, the match counts of the matchers
 *   (test_correspondence_finders.cpp:37,72,126,214,274,290,509; test_measurement_adaptors.cpp:26-56,110,130).
 *
 * The reference's extractor is
 *   sensor_processing/feature_extractors/intensity_feature_extractor_binned.cpp:7-208 (detection regions,
 *   per-region selection by response) around two OpenCV calls: cv::FastFeatureDetector::detect
 *   (intensity_feature_extractor_base.cpp:121-123, threshold + non-maximum suppression) and cv::ORB::compute
 *   (:139-170, descriptor_type "ORB-256" is the default and what every shipped .conf uses).  OpenCV is not part of the
 *   reference tree (its version is not pinned either); restated from its published algorithms:
 *   - FAST-9 segment test on the 16-pixel Bresenham circle (corner iff 9 contiguous circle pixels are all brighter than
 *     v + t or all darker than v - t), response = largest threshold for which the pixel is still a corner (OpenCV's
 *     cornerScore), non-maximum suppression keeps a corner whose response is strictly greater than that of its 8
 *     neighbours, the outermost 3 pixels are not examined, keypoints come out in raster order;
 *   - cv::ORB::compute on PROVIDED keypoints: keypoints closer than edgeThreshold = 31 px to the border are removed
 *     (KeyPointsFilter::runByImageBorder), no orientation is computed for provided keypoints (FAST leaves angle = -1
 *     degree: the pattern rotated by -1 degree rounds back to itself for offsets <= 13), the image is smoothed with
 *     GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) in OpenCV's 8-bit fixed-point separable filter (kernel
 *     round(256 g) = 18 34 49 55 49 34 18, result (sum + 2^15) >> 16, saturated) and bit i of the descriptor is
 *     blurred(p + a_i) < blurred(p + b_i) for the 256 learned pairs of bit_pattern_31_ (orb_pattern.inc).
 * What IS in-repo and restated exactly: the region grid (:47-92), the coordinate -> region table (:85-92),
 * target per region = float(target) / regions truncated (:72-76), "fewer than target: keep all in detection order, else
 * std::sort by decreasing response and keep the best" (:171-196).  std::sort is not stable and the comparator only
 * looks at the response, so WHICH of several equal responses survive the cut is implementation defined:
 *   ORC_SELECT_CANONICAL  ties broken by detection order (what the GPU's parallel sort does by default);
 *   ORC_SELECT_LIBSTDCXX  the permutation GNU libstdc++'s std::sort produces (introsort: median-of-3 quicksort to
 *                         runs of 16, heapsort below the depth limit, final insertion sort), restated in
 *                         orc_std_sort_desc(); tests/test_oracle_features.py checks it against g++'s std::sort.
 *                         This is the mode that reproduces the reference's pinned counts. */
#ifndef PROSLAM_ORACLE_FEATURES_H
#define PROSLAM_ORACLE_FEATURES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SELECT_CANONICAL = 0, ORC_SELECT_LIBSTDCXX = 1 };

typedef struct {
  int32_t detector_threshold;                 /* intensity_feature_extractor_base.h:36-40 (15 KITTI) */
  int32_t enable_non_maximum_suppression;     /* :48-52 */
  int32_t target_number_of_keypoints;         /* :54-58 (1000 KITTI) */
  int32_t number_of_detectors_vertical;       /* intensity_feature_extractor_binned.h:17-22 (3 KITTI) */
  int32_t number_of_detectors_horizontal;     /* :23-28 */
  int32_t selection_order;                    /* ORC_SELECT_* */
} orc_extractor_params;

enum { ORC_FEATURE_BORDER = 31, ORC_ERR_KEYPOINTS = -10 };

/* cv::ORB's 256 point pairs (x1, y1, x2, y2), each coordinate in [-13, 13] */
void orc_orb_pattern(int8_t* pattern1024);

/* cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on an 8-bit image (fixed-point path) */
void orc_gaussian_blur7(const uint8_t* image, int rows, int cols, uint8_t* blurred);

/* FAST response map: score[r][c] = corner response (0 = no corner), borders 0 */
void orc_fast_scores(const uint8_t* image, int rows, int cols, int threshold, uint8_t* score);

/* permutation GNU libstdc++'s std::sort(first, last, [](a, b){ return a.response > b.response; }) leaves:
 * order_out[i] = original position of the element that ends at position i */
void orc_std_sort_desc(const int32_t* response, int n, int32_t* order_out);

/* the whole extractor: image -> keypoints (u, v as float), intensity, 32-byte descriptors.
 * Returns the number of features or ORC_ERR_KEYPOINTS when more than `capacity` keypoints survive. */
int orc_extract_features(const orc_extractor_params* P,
                         const uint8_t* image,
                         int rows,
                         int cols,
                         float* uv,       /* [capacity][2] */
                         float* intensity,/* [capacity] */
                         uint8_t* desc,   /* [capacity][32] */
                         int capacity);

#ifdef __cplusplus
}
#endif
#endif
