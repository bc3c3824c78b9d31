/* proslam_oracle_features.h -- CPU oracle, SURVEY.md section 8f row 3: intensity feature extraction
 * (FAST keypoints, region-binned selection, 256-bit binary descriptors).  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference's extractor is
 *   sensor_processing/feature_extractors/intensity_feature_extractor_binned.cpp:7-208 (detection regions,
 *   per-region selection by response) around two OpenCV calls: cv::FastFeatureDetector::detect
 *   (intensity_feature_extractor_base.cpp:121-123, threshold + non-maximum suppression) and
 *   cv::ORB / BRIEF ::compute (:139-170).  OpenCV is not part of the reference tree, so
 *   - the detector is restated from the published FAST-9 segment test on the 16-pixel Bresenham circle
 *     (corner iff 9 contiguous circle pixels are all brighter than v + t or all darker than v - t), the
 *     response is the largest threshold for which the pixel is still a corner, non-maximum suppression
 *     keeps a corner whose response is strictly greater than that of its 8 neighbours, the outermost 3
 *     pixels are not examined, keypoints come out in raster order;
 *   - the descriptor is BUILD-DEFINED: BRIEF-style, 256 intensity comparisons of 5x5 box sums at point
 *     pairs inside a 31x31 patch; the pair table comes from orc_brief_pattern() (a fixed linear
 *     congruential sequence, roughly Gaussian offsets); keypoints closer than 17 px to the border are
 *     dropped, like OpenCV's runByImageBorder.  It is NOT bit-compatible with cv::ORB.
 * What IS in-repo and restated exactly: the region grid (:47-92), the coordinate -> region table (:85-92),
 * target per region = float(target) / regions truncated (:72-76), "fewer than target: keep all in
 * detection order, else sort by decreasing response and keep the best" (:171-196; std::sort is unstable,
 * ties are broken by detection order here). */
#ifndef PROSLAM_ORACLE_FEATURES_H
#define PROSLAM_ORACLE_FEATURES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t detector_threshold;                 /* intensity_feature_extractor_base.h:36-40 (15 KITTI) */
  int32_t enable_non_maximum_suppression;     /* :48-52 */
  int32_t target_number_of_keypoints;         /* :54-58 (1000 KITTI) */
  int32_t number_of_detectors_vertical;       /* intensity_feature_extractor_binned.h:17-22 (3 KITTI) */
  int32_t number_of_detectors_horizontal;     /* :23-28 */
} orc_extractor_params;

enum { ORC_FEATURE_BORDER = 17, ORC_ERR_KEYPOINTS = -10 };

/* 256 point pairs (x1, y1, x2, y2), each coordinate in [-13, 13] */
void orc_brief_pattern(int8_t* pattern1024);

/* FAST response map: score[r][c] = corner response (0 = no corner), borders 0 */
void orc_fast_scores(const uint8_t* image, int rows, int cols, int threshold, uint8_t* score);

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
