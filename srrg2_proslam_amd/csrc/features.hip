// features.hip -- intensity feature extraction on the device:
// IntensityFeatureExtractorBinned_::computeKeypoints + compute
// (sensor_processing/feature_extractors/intensity_feature_extractor_binned.cpp:7-208,
//  intensity_feature_extractor_base.cpp:56-95) around the two OpenCV calls the reference makes:
// cv::FastFeatureDetector::detect with non-maximum suppression (:121-123) and cv::ORB::compute (:139-170).
// OpenCV is not part of the reference tree; both are restated from their published algorithms (see
// include/proslam_hip.h) and the restatement is pinned by the counts the reference's own tests assert on its own
// test images (tests/test_ref_pins_gpu.py).  The region grid and the per-region selection by response follow the
// in-repo code line by line; PRS_SELECT_LIBSTDCXX reproduces the tie order of GNU std::sort there.
//
// Four launches per batch of images:
//   fast_blur_kernel   one 64x64 pixel tile per workgroup (tile + halo in LDS): compass test on packed 16-bit lanes, four
//                      pixels per lane, survivors in dense per-wave lists; arc minima on dense lanes; the 7x7 Gaussian
//                      (fixed point, what ORB samples) of the tile on the matrix cores (v_mfma_i32_16x16x32_i8, exact),
//                      written in 128-byte blocks; non-maximum suppression as a second walk over the lists; the tile's
//                      detections appended to the image's list (one atomic per tile).  Bound by vector issue
//                      (one vector instruction per cycle and CU: profiles/r05/features_pmc.txt)
//   raster_order_kernel one workgroup per image: the appended detections sorted by pixel index = the order
//                      cv::FAST reports them in (bucket sort; rounds 2-4 compacted a dense response map)
//   select_describe_kernel one workgroup per image: region histogram, one bitonic sort of
//                      (region, response, order) keys in LDS (or the replay of std::sort), per-region selection, border filter
//   describe_kernel    one wave per 24 kept keypoints: 256 comparisons of smoothed pixels per keypoint
#include "prs_device.h"
#include "prs_host.h"

namespace prs {

constexpr int kTileW = 64, kTileH = 64, kFastThreads = 256;
constexpr int kTilePitch = kTileW + 8;  // 4-px halo on both sides
constexpr int kNmsThreads = 1024, kSelThreads = 512;
constexpr int kDefaultRaw = 8192;       // raw detections per image the selection sort holds by default
constexpr int kMaxRawLimit = 32768;     // ... at most (prs_extractor_params.max_raw_detections; 128 KB of LDS keys)
constexpr int kFeatureBorder = 31;      // cv::ORB edgeThreshold: keypoints closer to the border are removed (runByImageBorder)
constexpr int kMaxRegions = 256;
constexpr int kSelScratch = 512;        // 16-bit words of LDS scratch per wave of the selection kernel
constexpr int kWinRadius = 13, kWinWords = 8;  // bit_pattern_31_ stays within +-13 px: 27 rows (32 are fetched) of 8 aligned words
constexpr int kDescThreads = 256, kDescPerWave = 24;  // describe_kernel: four waves, 24 keypoints each (at most 64: one per lane of the wave's slot list); 4 .. 64 measured, flat within 3 %
// cv::GaussianBlur(7x7, sigma 2) on 8-bit data: round(256 * exp(-x^2 / 8) / sum) = 18 34 49 55 49 34 18, sum of the taps = 257

struct FeatureArgs {
  prs_extractor_params p;
  prs_extract_batch b;
  // smoothed image (defined 3 px inside the border; ORB reads >= 18 px inside) in 128-byte blocks of 16 columns x 8 rows: the
  // 27 x 28 window of a keypoint touches ~12 cache lines instead of ~33 rows of a row-major image (the describe pass is bound by
  // the lines it pulls from L2).  Pixel (y, x) of image i: blur + i * blur_stride + blur_offset(y, x, blur_ncb)
  uint8_t* blur;
  int blur_ncb;        // blocks per block row = ceil(cols / 16)
  size_t blur_stride;  // bytes per image
  uint32_t* raw;      // [batch][max_raw] response << 24 | pixel index; appended tile by tile, then put in raster order
  int32_t* n_raw;     // [batch] zeroed before the tile kernel; -1 after raster_order_kernel = more than max_raw
  int max_raw;        // capacity of `raw` and of the selection sort (power of two)
  unsigned long long* stamps;  // PRS_STAMPS=1: phase clocks of thread 0 of select_describe_kernel, 16 words per image (diagnostic)
  uint32_t* kept;     // [batch][stride] pixel index of every selected keypoint, in output order (select -> describe)
  // the descriptor's pair table as byte offsets into the 27 x 28 window of smoothed pixels around a keypoint (fill_window_offsets)
  uint16_t pair_first[256], pair_second[256];
  float rows_per, cols_per;
  int target_per, regions;
};

__device__ __forceinline__ uint32_t blur_offset(const int y, const int x, const int ncb) {
  return (uint32_t) (((y >> 3) * ncb + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15));
}

// ---- word-parallel helpers: four pixels per 32-bit LDS word, two 16-bit lanes per VALU operation ----
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (us2) (__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {  // per lane a - b (sign bit = "a < b" for small values)
  return __builtin_bit_cast(uint32_t, (ss2) (__builtin_bit_cast(ss2, a) - __builtin_bit_cast(ss2, b)));
}
__device__ __forceinline__ uint32_t pk_sign_fill(uint32_t a) {  // per lane 0xffff where negative, else 0
  return __builtin_bit_cast(uint32_t, (ss2) (__builtin_bit_cast(ss2, a) >> (ss2) (15)));
}
__device__ __forceinline__ uint32_t bytes_even(uint32_t x) { return x & 0x00ff00ffu; }                          // bytes 0, 2 -> lanes
__device__ __forceinline__ uint32_t bytes_odd(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c01u); }  // bytes 1, 3 -> lanes
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) {  // bit j <=> byte j of the word is non-zero
  return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u);
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // LDS written by one lane is read by other lanes of the wave
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lanes_below(uint64_t m, int base) {  // base + number of set bits of m below this lane
  return (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, (uint32_t) base));
}

constexpr int kTileWords  = kTilePitch / 4;   // 18 words per tile row
constexpr int kTileRows   = kTileH + 8;       // 4-px halo above and below
// Survivor lists are sized for what images hold, not for the worst case (round 4): a wave collects at most 608 compass survivors
// (an image tile has ~150-400 per wave; a wave examines at most 1216 pixels) and re-examines at most 128 pixels with the other polarity;
// a survivor that finds its list full is scored on the spot by its own lane instead (same arithmetic, same result, divergent but
// rare).  With the Gaussian's horizontal sums in the same LDS bytes once the lists are dead the workgroup needs 19.4 KB instead of
// 38.3 KB: EIGHT workgroups per CU instead of four = eight waves per SIMD, which retire ~36 % more vector instructions per cycle
// than four (profiles/r04/valu_issue_rates.txt; the kernel is bound by vector issue).
constexpr int kListCap    = 608;
constexpr int kSecondCap  = 128;
constexpr uint32_t kBrightFlag = 0x4000u, kDarkFlag = 0x8000u, kIdMask = 0x1fffu;

// FAST-9 score of one polarity at the pixel `c` (LDS tile pointer): the largest arc minimum of s * (circle - centre)
// over the sixteen arcs of nine pixels; the pixel is a corner of that polarity at threshold t iff the value exceeds t,
// and its response is then the value - 1 (the largest threshold that still detects).  A bright and a dark arc of nine
// cannot both fit on sixteen pixels, so at most one polarity of a pixel ever exceeds t.
__device__ __forceinline__ int arc_best(const uint8_t* c, bool dark) {
  constexpr int P = kTilePitch;
  constexpr int off[16] = {-3 * P, -3 * P + 1, -2 * P + 2, -P + 3, 3, P + 3, 2 * P + 2, 3 * P + 1, 3 * P, 3 * P - 1, 2 * P - 2, P - 3, -3, -P - 3, -2 * P - 2, -3 * P - 1};
  const int s = dark ? -1 : 1, bias = -s * (int) c[0];
  int e[16], m2[16], m4[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(e[i]) : "v"((int) c[off[i]]), "v"(s), "v"(bias));  // s * (circle - centre), one instruction
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    m2[i] = min(e[i], e[(i + 1) & 15]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    m4[i] = min(m2[i], m2[(i + 2) & 15]);
  }
  int best = -256;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    best = max(best, min(min(m4[i], m4[(i + 4) & 15]), e[(i + 8) & 15]));  // minimum over the arc i .. i+8
  }
  return best;
}

// the response of one survivor that found its list full, scored by its own lane: bright first, dark only if bright failed.
// Out of line: the compass pass has twenty places where a list can overflow, and it never does on an image (inlined, these
// copies made the tile kernel 139 KB of code for a 64 KB instruction cache)
__device__ __attribute__((noinline)) void score_overflowed(const uint8_t* tile8, uint8_t* resp8, const uint32_t entry, const int t) {
  constexpr uint32_t kBright = 0x4000u, kDark = 0x8000u;
  const int id = (int) (entry & 0x1fffu);
  int best     = (entry & kBright) ? arc_best(tile8 + id, false) : arc_best(tile8 + id, true);
  if (best <= t && (entry & (kBright | kDark)) == (kBright | kDark)) {
    best = arc_best(tile8 + id, true);
  }
  if (best > t) {
    resp8[id] = (uint8_t) (best - 1);
  }
}

// image tile (+ 4-px halo) in LDS -> compass test on four pixels per lane -> survivors (dense lists, one per wave) ->
// arc minima on dense lanes -> responses of the tile and its 1-px ring -> non-maximum suppression, four pixels per
// lane; the separable 7x7 Gaussian of the tile on the side (packed 16-bit horizontal pass, 32-bit vertical pass).
template <bool BORDER>
__device__ __forceinline__ void fast_blur_tile(const FeatureArgs& a, uint32_t* tile32, uint32_t* resp32, uint16_t (*list)[kListCap],
                                              uint16_t (*second)[kSecondCap], int* list_n) {
  const int rows = a.b.rows, cols = a.b.cols, pitch = a.b.pitch;
  const int img = blockIdx.z;
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  const uint8_t* __restrict__ src = a.b.images + (size_t) img * rows * pitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(tile32);
  uint8_t* resp8       = reinterpret_cast<uint8_t*>(resp32);
  if (tid == 0) {
    list_n[4] = 0;  // detections of the tile (suppression pass; after a list overflow: words with a detection / detections)
    list_n[5] = 0;
    list_n[6] = 0;  // set when a compass list overflows
  }
  auto stamp = [&](const int i) {  // PRS_STAMPS=1: one interior tile of every image
    if (a.stamps && tid == 0 && blockIdx.x == 5 && blockIdx.y == 2) {
      a.stamps[((size_t) a.b.batch + img) * 16 + i] = (unsigned long long) clock64();
    }
  };
  stamp(0);
  // ---- tile + halo: all loads of a lane in flight before the first LDS store.  252 lanes take 14 rows of 18 words at a
  //      time (one division per lane; the six loads differ by a uniform 14 rows) ----
  {
    constexpr int kLoadRows = kFastThreads / kTileWords, kLoaders = kLoadRows * kTileWords, kLoads = (kTileRows + kLoadRows - 1) / kLoadRows;  // 14, 252, 6
    const int ty0 = tid / kTileWords, tw = tid - ty0 * kTileWords;
    const int gx = x0 - 4 + 4 * tw;
    uint32_t w[kLoads];
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int ty = ty0 + kLoadRows * u;
      w[u]         = 0;
      if (tid < kLoaders && ty < kTileRows) {
        int gy = y0 + ty - 4;
        if (!BORDER) {
          const uint8_t* rows_u = src + (size_t) (y0 - 4 + kLoadRows * u) * pitch;  // (uniform)
          __builtin_memcpy(&w[u], rows_u + ((size_t) ty0 * pitch + gx), 4);  // global loads may be unaligned, the LDS stores are not
        } else {
          gy = gy < 0 ? 0 : (gy >= rows ? rows - 1 : gy);  // clamped pixels never reach an output
          const uint8_t* row = src + (size_t) gy * pitch;
          if (gx >= 0 && gx + 3 < cols) {
            __builtin_memcpy(&w[u], row + gx, 4);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int x = gx + j < 0 ? 0 : (gx + j >= cols ? cols - 1 : gx + j);
              w[u] |= (uint32_t) row[x] << (8 * j);
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      if (tid < kLoaders && ty0 + kLoadRows * u < kTileRows) {
        tile32[tid + kLoaders * u] = w[u];
        resp32[tid + kLoaders * u] = 0;
      }
    }
  }
  __syncthreads();
  stamp(1);
  const int t       = a.p.detector_threshold;  // 1 .. 254 (checked by the host)
  const uint32_t T2 = (uint32_t) t * 0x00010001u;
  const int w1 = tid & 15, r0 = tid >> 4;
  // ---- compass test.  Nine contiguous circle pixels always contain two ADJACENT compass points (N, E, S, W), and
  // (N&E)|(E&S)|(S&W)|(W&N) = (N|S)&(E|W): a pixel can only be a bright corner if min(max(N,S), max(E,W)) > v + t,
  // a dark one if max(min(N,S), min(E,W)) < v - t.  Response-tile rows 0..65 x words 1..16 (response columns 1..64);
  // the two ring columns follow pixel by pixel.  Entry = tile byte offset | polarity flags.
  int n_mine         = 0;  // wave-uniform
  uint16_t* my_list  = list[wave];
  auto push = [&](uint32_t entry, bool ok) {
    const uint64_t m = __ballot(ok);
    if (ok) {
      const int pos = lanes_below(m, n_mine);
      if (pos < kListCap) {
        my_list[pos] = (uint16_t) entry;
      } else {
        score_overflowed(tile8, resp8, entry, t);
        list_n[6] = 1;
      }
    }
    n_mine += __popcll(m);
  };
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int row = r0 + 16 * k;  // response-tile row; tile row = row + 3
    bool act      = row < kTileH + 2;
    if (BORDER) {
      act = act && (unsigned) (y0 + row - 1 - 3) <= (unsigned) (rows - 7);  // the outermost 3 pixels are not examined
    }
    if (k == 4 && wave != 0) {
      break;  // rows 64, 65 belong to the first 32 lanes
    }
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;  // no flags: not a candidate
    if (act) {
      const int wi = (row + 3) * kTileWords + 1 + w1;
      const uint32_t C = tile32[wi], N = tile32[wi - 3 * kTileWords], S = tile32[wi + 3 * kTileWords];
      const uint32_t Wst = __builtin_amdgcn_alignbyte(C, tile32[wi - 1], 1);  // the four pixels 3 to the left
      const uint32_t Est = __builtin_amdgcn_alignbyte(tile32[wi + 1], C, 3);  // 3 to the right
      uint32_t code[2];
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const uint32_t c = par ? bytes_odd(C) : bytes_even(C), n = par ? bytes_odd(N) : bytes_even(N), s = par ? bytes_odd(S) : bytes_even(S);
        const uint32_t e = par ? bytes_odd(Est) : bytes_even(Est), w = par ? bytes_odd(Wst) : bytes_even(Wst);
        const uint32_t hi = pk_min(pk_max(n, s), pk_max(e, w));
        const uint32_t lo = pk_max(pk_min(n, s), pk_min(e, w));
        const uint32_t fb = pk_sub(c + T2, hi);  // lane negative <=> hi > v + t
        const uint32_t fd = pk_sub(lo + T2, c);  // lane negative <=> lo < v - t
        code[par]         = ((fb >> 1) & 0x40004000u) | (fd & 0x80008000u);
      }
      const uint32_t id0 = 4u * (uint32_t) wi;
      e0 = (code[0] & 0xc000u) | id0;
      e1 = (code[1] & 0xc000u) | id0 | 1u;
      e2 = (code[0] >> 16) | id0 | 2u;
      e3 = (code[1] >> 16) | id0 | 3u;
    }
    bool ok0 = e0 >= kBrightFlag, ok1 = e1 >= kBrightFlag, ok2 = e2 >= kBrightFlag, ok3 = e3 >= kBrightFlag;
    if (BORDER) {
      const unsigned gxm3 = (unsigned) (x0 + 4 * w1 - 3), lim = (unsigned) (cols - 7);
      ok0 = ok0 && gxm3 <= lim;
      ok1 = ok1 && gxm3 + 1u <= lim;
      ok2 = ok2 && gxm3 + 2u <= lim;
      ok3 = ok3 && gxm3 + 3u <= lim;
    }
    // every lane of the wave takes part: the list position is a wave-wide count.  As long as the list has room for all of the
    // iteration's candidates (a scalar test; always, on an image) a lane's entries go side by side behind those of the lanes
    // below it: one chain of eight v_mbcnt instead of four position computations with a capacity test each
    const uint64_t m0 = __ballot(ok0), m1 = __ballot(ok1), m2 = __ballot(ok2), m3 = __ballot(ok3);
    const int coming = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
    if (n_mine + coming <= kListCap) {
      uint16_t* slot = my_list + lanes_below(m0, lanes_below(m1, lanes_below(m2, lanes_below(m3, n_mine))));
      if (ok0) {
        *slot++ = (uint16_t) e0;
      }
      if (ok1) {
        *slot++ = (uint16_t) e1;
      }
      if (ok2) {
        *slot++ = (uint16_t) e2;
      }
      if (ok3) {
        *slot = (uint16_t) e3;
      }
      n_mine += coming;
    } else {
      push(e0, ok0);
      push(e1, ok1);
      push(e2, ok2);
      push(e3, ok3);
    }
  }
  if (wave < 3) {  // ring columns: response column 0 (tile x = 3) and 65 (tile x = 68), one pixel per lane
    const int row = min(tid >> 1, kTileH + 1), tx = (tid & 1) ? kTileW + 4 : 3;
    bool ok = tid < 2 * (kTileH + 2);
    if (BORDER) {
      ok = ok && (unsigned) (y0 + row - 1 - 3) <= (unsigned) (rows - 7) && (unsigned) (x0 + tx - 4 - 3) <= (unsigned) (cols - 7);
    }
    const int id     = (row + 3) * kTilePitch + tx;
    const uint8_t* c = tile8 + id;
    const int v = c[0], n = c[-3 * kTilePitch], e = c[3], so = c[3 * kTilePitch], w = c[-3];
    const bool bright = min(max(n, so), max(e, w)) > v + t, dark = max(min(n, so), min(e, w)) < v - t;
    push((uint32_t) id | (bright ? kBrightFlag : 0u) | (dark ? kDarkFlag : 0u), ok && (bright || dark));
  }
  if (lane == 0) {
    list_n[wave] = n_mine < kListCap ? n_mine : kListCap;
  }
  __syncthreads();
  stamp(2);
  // ---- arc minima on dense lanes: entry j of the concatenated lists ----
  const int c0 = list_n[0], c1 = c0 + list_n[1], c2 = c1 + list_n[2], total = c2 + list_n[3];
  int n_second          = 0;  // wave-uniform
  uint16_t* my_second   = second[wave];
  for (int base = 0; base < total; base += kFastThreads) {
    const int j = base + tid;
    bool again   = false;
    int again_id = 0;
    if (j < total) {
      const int seg = (j >= c0) + (j >= c1) + (j >= c2);
      const int off = j - (seg == 0 ? 0 : (seg == 1 ? c0 : (seg == 2 ? c1 : c2)));
      const uint32_t en = list[seg][off];
      const int id      = (int) (en & kIdMask);
      const bool dark   = !(en & kBrightFlag);
      const int best    = arc_best(tile8 + id, dark);
      if (best > t) {
        resp8[id] = (uint8_t) (best - 1);
      }
      again = best <= t && (en & (kBrightFlag | kDarkFlag)) == (kBrightFlag | kDarkFlag);  // bright failed, dark still open
      again_id = id;
    }
    const uint64_t m = __ballot(again);  // every lane of the wave takes part
    if (again) {
      const int pos = lanes_below(m, n_second);
      if (pos < kSecondCap) {
        my_second[pos] = (uint16_t) again_id;
      } else {  // list full: the lane looks at the dark arcs itself (the pixel is on a first list: nothing to flag)
        score_overflowed(tile8, resp8, (uint32_t) again_id | kDarkFlag, t);
      }
    }
    n_second += __popcll(m);
  }
  n_second = n_second < kSecondCap ? n_second : kSecondCap;
  for (int base = 0; base < n_second; base += 64) {  // this wave's own list: LDS operations of a wave complete in order
    const int j = base + lane;
    if (j < n_second) {
      const int id   = my_second[j];
      const int best = arc_best(tile8 + id, true);
      if (best > t) {
        resp8[id] = (uint8_t) (best - 1);
      }
    }
  }
  // ---- 7x7 Gaussian on the matrix cores (round 5; rounds 2-4: packed 16-bit / v_dot vector passes through LDS, a fifth of the
  // kernel's vector instructions -- and the kernel is bound by exactly those).  Separable, fixed point, both passes as
  // v_mfma_i32_16x16x32_i8 against a banded matrix of the taps; integer products and sums: exact.  Wave w owns the tile's
  // columns 16 w .. 16 w + 15 and needs nobody else's data, so nothing goes through LDS and there is no barrier:
  //   rows:    H[80 x 16] = P[80 x 32] * Bh[32 x 16], five MFMAs; P = the tile's rows -3 .. 76 (beyond the halo: whatever
  //            the clamped row holds, those sums feed no output), columns 16 w - 4 .. 16 w + 27, as signed bytes p - 128;
  //            the accumulator starts at 128 * 257 (the taps sum to 257), so H is the true sum 0 .. 65535.  An MFMA result has
  //            its column on the lane and four consecutive rows in its registers = the A operand of a product that sums over rows;
  //   columns: out^T[16 x 16] = H^T[16 x 32] * Bv[32 x 16] per 16 output rows, H split into its two bytes (each as b - 128, two
  //            MFMAs, recombined with a shift); the K order inside the operand is the one the first result arrives in (rows
  //            4 g + r of two row blocks), Bv is built for that order.  Result: four consecutive pixels of one row per lane.
  // Lane maps checked with exact integer data: tools/probes/mfma_i8_probe.hip.
  {
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr unsigned long long kTaps = 0x0012223137312212ull;  // bytes 0 .. 6 = 18 34 49 55 49 34 18
    auto taps_from = [](const int d) -> unsigned long long {       // byte j = tap j + d (0 outside 0 .. 6)
      return d >= 0 ? (d < 8 ? kTaps >> (8 * d) : 0ull) : (d > -8 ? kTaps << (-8 * d) : 0ull);
    };
    const int li = lane & 15, g = lane >> 4;
    // Bh[k][n]: input column k (tile byte 16 w + k) feeds output column n (tile byte 16 w + 4 + n) with tap k - n - 1
    const long bh = (long) taps_from(8 * g - li - 1);
    // Bv[k][q]: byte t < 4 of lane group g is H row 4 g + t of the first row block, t >= 4 row 16 + 4 g + t - 4 (second block);
    // H row h feeds output row q with tap h - q
    const long bv = (long) ((taps_from(4 * g - li) & 0xffffffffull) | (taps_from(16 + 4 * g - li) << 32));
    const uint8_t* strip = tile8 + 16 * wave + 8 * min(g, 2);  // (the fourth lane group's columns carry no tap: any bytes will do)
    uint32_t hlo[5], hhi[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int ty = min(16 * b + li + 1, kTileRows - 1);
      const uint64_t p = *reinterpret_cast<const uint64_t*>(strip + ty * kTilePitch);  // (8-byte aligned: the pitch is 72)
      const v4i h = __builtin_amdgcn_mfma_i32_16x16x32_i8((long) (p ^ 0x8080808080808080ull), bh, v4i{32896, 32896, 32896, 32896}, 0, 0, 0);
      const uint32_t p01 = __builtin_amdgcn_perm((uint32_t) h[1], (uint32_t) h[0], 0x05010400u);  // h0.b0 h1.b0 h0.b1 h1.b1
      const uint32_t p23 = __builtin_amdgcn_perm((uint32_t) h[3], (uint32_t) h[2], 0x05010400u);
      hlo[b] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;
      hhi[b] = __builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u;
    }
    uint8_t* __restrict__ blur = a.blur + (size_t) img * a.blur_stride;
#pragma unroll
    for (int j = 0; j < kTileH / 16; ++j) {
      const long alo = (long) (((uint64_t) hlo[j + 1] << 32) | hlo[j]), ahi = (long) (((uint64_t) hhi[j + 1] << 32) | hhi[j]);
      // sum g H = sum g (lo - 128) + 256 sum g (hi - 128) + 257 * 128 * 257; + 2^15 for the rounding
      const v4i slo = __builtin_amdgcn_mfma_i32_16x16x32_i8(alo, bv, v4i{65664, 65664, 65664, 65664}, 0, 0, 0);
      const v4i shi = __builtin_amdgcn_mfma_i32_16x16x32_i8(ahi, bv, v4i{32896, 32896, 32896, 32896}, 0, 0, 0);
      uint32_t v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = min((uint32_t) slo[r] + ((uint32_t) shi[r] << 8), 0x00ffffffu);  // (sum + 2^15) >> 16, saturated (the taps sum to 257 / 256)
      }
      const uint32_t out = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u) | (__builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u) << 16);
      const int gy = y0 + 16 * j + li, gx = x0 + 16 * wave + 4 * g;
      if (!BORDER || (gy < rows && gx < 16 * a.blur_ncb)) {  // (the padding columns of the last block take whatever the word holds)
        *reinterpret_cast<uint32_t*>(blur + blur_offset(gy, gx, a.blur_ncb)) = out;
      }
    }
  }
  stamp(3);
  __syncthreads();  // the lists are dead: their bytes hold the tile's words with a detection from here on
  stamp(4);
  // ---- non-maximum suppression (strictly greater than the 8 neighbours).  A non-zero response never sits on the outermost
  // 3 pixels, so the ring values are real responses of real pixels, and every non-zero byte of the tile proper is a pixel of
  // the image.  Every response belongs to a pixel of the compass lists (unless a list overflowed), so the lists are walked once
  // more on dense lanes: a pixel with a response looks at its eight neighbours (round 5; the sweep over all 4096 pixels below,
  // four per lane, was a sixth of the kernel's vector instructions).  Survivors go to a list of the workgroup (in the tile's
  // pixels, dead since the last barrier) in no particular order; raster_order_kernel sorts.
  const bool nms = a.p.enable_non_maximum_suppression != 0;
  if (list_n[6] == 0) {  // (uniform)
    uint32_t* found = tile32;  // at most 32 x 32 survivors of a 64 x 64 tile
    for (int base = 0; base < total; base += kFastThreads) {
      const int j = base + tid;
      if (j < total) {
        const int seg = (j >= c0) + (j >= c1) + (j >= c2);
        const int off = j - (seg == 0 ? 0 : (seg == 1 ? c0 : (seg == 2 ? c1 : c2)));
        const int id  = (int) (list[seg][off] & kIdMask);
        const uint32_t sc = resp8[id];
        const int ty = (id * 3641) >> 18, tx = id - ty * kTilePitch;  // id / 72 (exact below 5184)
        if (sc != 0u && (unsigned) (ty - 4) < (unsigned) kTileH && (unsigned) (tx - 4) < (unsigned) kTileW) {
          bool keep = true;
          if (nms) {
            constexpr int P = kTilePitch;
            const uint8_t* c = resp8 + id;
            const uint32_t m = max(max(max((uint32_t) c[-P - 1], (uint32_t) c[-P]), max((uint32_t) c[-P + 1], (uint32_t) c[-1])),
                                   max(max((uint32_t) c[1], (uint32_t) c[P - 1]), max((uint32_t) c[P], (uint32_t) c[P + 1])));
            keep = sc > m;
          }
          if (keep) {
            const uint32_t entry = (sc << 24) | ((uint32_t) (y0 + ty - 4) * (uint32_t) cols + (uint32_t) (x0 + tx - 4));
            if (nms) {
              found[atomicAdd(&list_n[4], 1)] = entry;
            } else {  // without suppression a tile may hold up to 4096 detections: every lane appends its own
              const int pos = atomicAdd(a.n_raw + img, 1);
              if (pos < a.max_raw) {
                a.raw[(size_t) img * a.max_raw + pos] = entry;
              }
            }
          }
        }
      }
    }
    __syncthreads();
    stamp(5);
    const int n_found = list_n[4];  // (uniform)
    if (n_found > 0) {
      if (tid == 0) {
        list_n[1] = atomicAdd(a.n_raw + img, n_found);  // the tile's span of the image's list: ONE atomic
      }
      __syncthreads();
      const int base = list_n[1];
      uint32_t* __restrict__ raw = a.raw + (size_t) img * a.max_raw;
      for (int i = tid; i < n_found; i += kFastThreads) {
        if (base + i < a.max_raw) {  // beyond: the image fails with PRS_ERR_CAPACITY (raster_order_kernel sees the count)
          raw[base + i] = found[i];
        }
      }
    }
    return;
  }
  // ---- a list overflowed (more than half of a wave's pixels passed the compass test: noise, not an image): the sweep over
  // all pixels, four per lane.  Words with a survivor go to a list of the workgroup (in the bytes of the compass lists).
  uint2* tile_words = reinterpret_cast<uint2*>(list);  // (response word, row << 4 | word of the tile): at most 1024 of 8 bytes
#pragma unroll
  for (int k = 0; k < kTileH / 16; ++k) {
    const int ly = r0 + 16 * k;
    const int wi = (ly + 4) * kTileWords + 1 + w1;
    const uint32_t Cm = resp32[wi];
    uint32_t out = Cm;
    if (nms) {
      // per row: the even pixels (0, 2) and the odd ones (1, 3) as 16-bit lanes, plus the left neighbours of the even pixels
      // (previous word's pixel 3, pixel 1) and the right neighbours of the odd ones (pixel 2, next word's pixel 0): one v_and and
      // three v_perm per row instead of two byte alignments and six unpacks; the right neighbours of the even pixels are the odd
      // lanes themselves, the left neighbours of the odd pixels the even lanes
      auto row_lanes = [&](const int w, uint32_t& E, uint32_t& O, uint32_t& OL, uint32_t& ER) {
        const uint32_t W = resp32[w], Pw = resp32[w - 1], Nw = resp32[w + 1];
        E  = bytes_even(W);
        O  = bytes_odd(W);
        OL = __builtin_amdgcn_perm(Pw, W, 0x0c010c07u);  // (Pw.b3, W.b1)
        ER = __builtin_amdgcn_perm(Nw, W, 0x0c040c02u);  // (W.b2, Nw.b0)
      };
      uint32_t Et, Ot, OLt, ERt, Em, Om, OLm, ERm, Eb, Ob, OLb, ERb;
      row_lanes(wi - kTileWords, Et, Ot, OLt, ERt);
      row_lanes(wi, Em, Om, OLm, ERm);
      row_lanes(wi + kTileWords, Eb, Ob, OLb, ERb);
      const uint32_t around_even = pk_max(pk_max(pk_max(OLt, Et), pk_max(Ot, OLm)), pk_max(pk_max(Om, OLb), pk_max(Eb, Ob)));
      const uint32_t around_odd  = pk_max(pk_max(pk_max(Et, Ot), pk_max(ERt, Em)), pk_max(pk_max(ERm, Eb), pk_max(Ob, ERb)));
      uint32_t kept[2];
      kept[0] = Em & pk_sign_fill(pk_sub(around_even, Em));  // lane survives <=> largest neighbour < s
      kept[1] = Om & pk_sign_fill(pk_sub(around_odd, Om));
      out = kept[0] | (kept[1] << 8);
    }
    if (out != 0u) {
      // ~25 of a KITTI tile's 4096 pixels survive, but four out of five waves have SOME lane in here: the wave-wide cost of this
      // block is what counts, so it only files the word (PMC: the per-byte version was a fifth of the kernel's vector instructions)
      const int at = atomicAdd(&list_n[4], 1);
      tile_words[at] = make_uint2(out, (uint32_t) ((ly << 4) | w1));
    }
  }
  __syncthreads();
  stamp(5);
  // ---- the listed words -> the image's list: every entry counts its survivors and takes its place in the tile's span, the
  //      tile reserves the span with ONE atomic, the entries are unpacked there
  const int n_words = list_n[4];  // (uniform)
  if (n_words > 0) {
    constexpr int kPerThread = (kTileW / 4) * kTileH / kFastThreads;  // 4: a tile has 1024 words
    uint2 e[kPerThread];
    int off[kPerThread];
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      const int i = tid + kFastThreads * u;
      e[u]        = make_uint2(0u, 0u);
      off[u]      = 0;
      if (i < n_words) {
        e[u]   = tile_words[i];
        off[u] = atomicAdd(&list_n[5], __popc(nonzero_bytes(e[u].x)));
      }
    }
    __syncthreads();
    if (tid == 0) {
      list_n[1] = atomicAdd(a.n_raw + img, list_n[5]);
    }
    __syncthreads();
    const int base = list_n[1];
    uint32_t* __restrict__ raw = a.raw + (size_t) img * a.max_raw;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
      if (tid + kFastThreads * u < n_words) {
        const uint32_t pix0 = (uint32_t) (y0 + (int) (e[u].y >> 4)) * (uint32_t) cols + (uint32_t) (x0 + 4 * (int) (e[u].y & 15u));
        int pos             = base + off[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t sj = (e[u].x >> (8 * j)) & 0xffu;
          if (sj) {
            if (pos < a.max_raw) {  // beyond: the image fails with PRS_ERR_CAPACITY (raster_order_kernel sees the count)
              raw[pos] = (sj << 24) | (pix0 + (uint32_t) j);
            }
            ++pos;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(kFastThreads, 8) void fast_blur_kernel(const FeatureArgs a) {  // 8 workgroups of 4 waves per CU
  __shared__ __attribute__((aligned(16))) uint32_t tile32[kTileRows * kTileWords];
  __shared__ __attribute__((aligned(16))) uint32_t resp32[kTileRows * kTileWords];  // same geometry as the tile
  // the survivor lists (compass -> arc passes) and the words with a detection (suppression pass) share their bytes
  constexpr size_t kListBytes = sizeof(uint16_t) * (kFastThreads / 64) * (kListCap + kSecondCap);
  constexpr size_t kWordBytes = sizeof(uint2) * (kTileW / 4) * kTileH;
  __shared__ __attribute__((aligned(16))) unsigned char shared_bytes[kListBytes > kWordBytes ? kListBytes : kWordBytes];
  uint16_t (*list)[kListCap]     = reinterpret_cast<uint16_t (*)[kListCap]>(shared_bytes);
  uint16_t (*second)[kSecondCap] = reinterpret_cast<uint16_t (*)[kSecondCap]>(shared_bytes + sizeof(uint16_t) * (kFastThreads / 64) * kListCap);
  __shared__ int list_n[kFastThreads / 64 + 3];
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  // a tile whose halo lies inside the image needs no coordinate checks at all (block-uniform)
  if (x0 >= 4 && x0 + kTileW + 4 <= a.b.cols && y0 >= 4 && y0 + kTileH + 4 <= a.b.rows) {
    fast_blur_tile<false>(a, tile32, resp32, list, second, list_n);
  } else {
    fast_blur_tile<true>(a, tile32, resp32, list, second, list_n);
  }
}

// The detections of an image in the order cv::FAST reports them (row by row): the tiles appended theirs in the order
// they finished.  Pixel indices are distinct, so raster order is a bucket sort: 1024 equal spans of the image (a monotone
// function of the pixel index), one counting pass, one scan (a bucket per thread), a scatter into the buckets (LDS) and,
// per entry, its rank among the handful of entries of its bucket.  One workgroup per image, three barriers; an image with
// more than max_raw detections is marked (-1) and fails in select_describe_kernel.
constexpr int kRasterBuckets = kNmsThreads;
__global__ __launch_bounds__(kNmsThreads) void raster_order_kernel(const FeatureArgs a, const int bucket_shift) {
  extern __shared__ __attribute__((aligned(16))) uint32_t staged[];  // [max_raw] entries grouped by bucket
  __shared__ uint32_t count[kRasterBuckets], first[kRasterBuckets], fill[kRasterBuckets];
  __shared__ uint32_t wave_tot[kNmsThreads / 64];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* __restrict__ raw = a.raw + (size_t) img * a.max_raw;
  const int n = a.n_raw[img];
  if (n > a.max_raw) {
    if (tid == 0) {
      a.n_raw[img] = -1;
    }
    return;
  }
  if (n <= 1) {
    return;
  }
  count[tid] = 0;
  fill[tid]  = 0;
  __syncthreads();
  for (int i = tid; i < n; i += kNmsThreads) {
    atomicAdd(&count[(raw[i] & 0xffffffu) >> bucket_shift], 1u);
  }
  __syncthreads();
  {  // exclusive scan of the bucket sizes, one bucket per thread
    const uint32_t c = count[tid];
    uint32_t incl    = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t) __shfl_up((int) incl, o, 64);
      if (lane >= o) {
        incl += v;
      }
    }
    if (lane == 63) {
      wave_tot[wave] = incl;
    }
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < kNmsThreads / 64; ++w) {
      before += w < wave ? wave_tot[w] : 0u;
    }
    first[tid] = before + incl - c;
  }
  __syncthreads();
  for (int i = tid; i < n; i += kNmsThreads) {
    const uint32_t e = raw[i];
    const uint32_t b = (e & 0xffffffu) >> bucket_shift;
    staged[first[b] + atomicAdd(&fill[b], 1u)] = e;
  }
  __syncthreads();  // (every entry of `raw` has been read: the sorted entries go back to the same place)
  for (int i = tid; i < n; i += kNmsThreads) {
    const uint32_t e = staged[i], pix = e & 0xffffffu;
    const uint32_t b = pix >> bucket_shift;
    const uint32_t f = first[b], c = count[b];
    uint32_t rank    = 0;
    for (uint32_t j = 0; j < c; ++j) {
      rank += (staged[f + j] & 0xffffffu) < pix ? 1u : 0u;
    }
    raw[f + rank] = e;
  }
}

// ---- GNU libstdc++ std::sort (bits/stl_algo.h, bits/stl_heap.h) on packed items, comparator a.response > b.response
// (intensity_feature_extractor_binned.cpp:182-186): the response is the low byte of an item, the rest rides along.
// One wave runs it for one region (tests compare the permutation with what g++ itself produces).
__device__ __forceinline__ bool scomp(uint32_t a, uint32_t b) {
  return (a & 0xffu) > (b & 0xffu);
}

__device__ void std_adjust_heap(uint32_t* first, int hole, int len, uint32_t value) {
  const int top = hole;
  int child     = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (scomp(first[child], first[child - 1])) {
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
  int parent = (hole - 1) / 2;  // __push_heap
  while (hole > top && scomp(first[parent], value)) {
    first[hole] = first[parent];
    hole        = parent;
    parent      = (hole - 1) / 2;
  }
  first[hole] = value;
}

__device__ void std_heapsort(uint32_t* first, int len) {  // __partial_sort(first, last, last)
  if (len >= 2) {
    for (int parent = (len - 2) / 2;; --parent) {
      std_adjust_heap(first, parent, len, first[parent]);
      if (parent == 0) {
        break;
      }
    }
  }
  for (int last = len - 1; last >= 1; --last) {  // __sort_heap: __pop_heap(first, last, last)
    const uint32_t value = first[last];
    first[last]          = first[0];
    std_adjust_heap(first, 0, last, value);
  }
}

// __unguarded_partition_pivot on [first, last) (more than 16 items) by ONE WAVE, same permutation and same cut as the serial
// loop.  The serial scan pairs the k-th "left stop" (an item that does not compare before the pivot, from the left) with the
// k-th "right stop" (an item the pivot does not compare before, from the right) and swaps them until they cross; where a scan
// stops only depends on values that no earlier swap of the same partition has touched, or on swapped items, which always
// stop it.  So 64 positions from each side are classified at once (ballot), the stops are ranked into tl[] / tr[], pair j is
// swapped by lane j, and the first pair that has crossed gives the cut; memory is re-read after every round, which makes
// swapped items act as the sentinels they are in the serial loop.  tl, tr: 64 ints of LDS each.
__device__ __forceinline__ int std_partition_wave(uint32_t* v, const int first, const int last, const int lane, int* tl, int* tr) {
  if (lane == 0) {  // __move_median_to_first(first, first + 1, mid, last - 1)
    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
    int pick;
    if (scomp(v[a], v[b])) {
      pick = scomp(v[b], v[c]) ? b : (scomp(v[a], v[c]) ? c : a);
    } else {
      pick = scomp(v[a], v[c]) ? a : (scomp(v[b], v[c]) ? c : b);
    }
    const uint32_t t = v[first];
    v[first]         = v[pick];
    v[pick]          = t;
  }
  wave_sync_lds();
  const uint32_t pr              = v[first] & 0xffu;
  const unsigned long long below = (1ull << lane) - 1ull;
  int lo = first + 1, hi = last;
  for (;;) {
    const int pl  = lo + lane;
    const int ph  = hi - 1 - lane;
    const bool fl = pl < last && !((v[pl < last ? pl : first] & 0xffu) > pr);
    const bool fr = ph >= first && !(pr > (v[ph >= first ? ph : first] & 0xffu));
    const unsigned long long bl = __ballot(fl), br = __ballot(fr);
    if (fl) {
      tl[__popcll(bl & below)] = pl;
    }
    if (fr) {
      tr[__popcll(br & below)] = ph;
    }
    wave_sync_lds();
    const int nl = __popcll(bl), nr = __popcll(br);
    const int m  = nl < nr ? nl : nr;
    int pa = 0, pb = 0;
    if (lane < m) {
      pa = tl[lane];
      pb = tr[lane];
    }
    const unsigned long long crossed = __ballot(lane < m && !(pa < pb));
    const int jstar                  = crossed ? (int) __ffsll((long long) crossed) - 1 : m;
    if (lane < jstar) {
      const uint32_t va = v[pa], vb = v[pb];
      v[pa]             = vb;
      v[pb]             = va;
    }
    wave_sync_lds();
    if (crossed) {
      const int l = tl[jstar];
      if (jstar == 0) {
        return l;
      }
      const int r = tr[jstar - 1];
      return l < r ? l : r;
    }
    const int lo_next = jstar > 0 ? tl[jstar - 1] + 1 : (nl == 0 ? lo + 64 : lo);
    const int hi_next = jstar > 0 ? tr[jstar - 1] : (nr == 0 ? hi - 64 : hi);
    lo                = lo_next;
    hi                = hi_next;
    wave_sync_lds();  // tl / tr are rewritten by the next round
  }
}

// The whole subtree of __introsort_loop below a range of <= 64 items, in registers: lane i holds item i, every lane carries the
// bounds and the depth budget of the range ("segment") its item is in, and all segments of a level are partitioned at once.
// One __unguarded_partition_pivot in closed form: with l_0 < l_1 < .. the left stops (items that do not compare before the
// pivot) and r_0 > r_1 > .. the right stops (items the pivot does not compare before, the pivot's own position included), the
// serial loop swaps the pairs k < K, K = number of k with l_k < r_k, and returns min(l_K, r_(K-1)) (the scans stop at swapped
// items; see std_partition_wave).  Lane first + k collects pair k (ds_permute), a stop fetches its partner's position from
// there (ds_bpermute) and then its partner's item.  Segments that run out of depth go through LDS for the heapsort on one
// lane; segments of <= 16 items end with their share of the final insertion sort (stable, decreasing response).
__device__ __forceinline__ void std_sort_small_wave(uint32_t* v, const int first, const int n, const int depth, const int lane) {
  uint32_t x = lane < n ? v[first + lane] : 0u;
  int sf = 0, sl = n, sd = depth;
  bool heaped = false;
  const unsigned long long bit = 1ull << lane, below = bit - 1ull, above = ~(below | bit);
  for (;;) {
    const bool open = lane < n && !heaped && sl - sf > 16;
    if (!__ballot(open)) {
      break;
    }
    const bool dry = open && sd == 0;
    if (__ballot(dry)) {
      unsigned long long heads = __ballot(dry && lane == sf);
      if (dry) {
        v[first + lane] = x;
      }
      wave_sync_lds();
      while (heads) {
        const int h = (int) __ffsll((long long) heads) - 1;
        heads &= heads - 1ull;
        const int l = __builtin_amdgcn_readlane(sl, h);
        if (lane == 0) {
          std_heapsort(v + first + h, l - h);
        }
      }
      wave_sync_lds();
      if (dry) {
        x      = v[first + lane];
        heaped = true;  // heap-sorted: the final insertion pass finds nothing to move
      }
    }
    const bool act = open && !dry;
    if (!__ballot(act)) {
      continue;
    }
    // __move_median_to_first(first, first + 1, mid, last - 1)
    const int a = act ? sf + 1 : lane, b = act ? sf + ((sl - sf) >> 1) : lane, c = act ? sl - 1 : lane;
    const uint32_t xa = (uint32_t) __shfl((int) x, a, 64), xb = (uint32_t) __shfl((int) x, b, 64), xc = (uint32_t) __shfl((int) x, c, 64);
    const uint32_t x0 = (uint32_t) __shfl((int) x, act ? sf : lane, 64);
    int pick;
    uint32_t xp;
    if (scomp(xa, xb)) {
      const bool bc = scomp(xb, xc), ac = scomp(xa, xc);
      pick = bc ? b : (ac ? c : a);
      xp   = bc ? xb : (ac ? xc : xa);
    } else {
      const bool ac = scomp(xa, xc), bc = scomp(xb, xc);
      pick = ac ? a : (bc ? c : b);
      xp   = ac ? xa : (bc ? xc : xb);
    }
    if (act) {
      x = lane == sf ? xp : (lane == pick ? x0 : x);
    }
    const uint32_t pr = xp & 0xffu, r = x & 0xffu;
    const bool fl = act && lane > sf && !(r > pr);
    const bool fr = act && !(pr > r);
    const unsigned long long upto = sl >= 64 ? ~0ull : ((1ull << sl) - 1ull);
    const unsigned long long seg  = act ? (upto & ~((1ull << sf) - 1ull)) : 0ull;
    const unsigned long long SL = __ballot(fl) & seg, SR = __ballot(fr) & seg;
    const int kl = __popcll(SL & below), kr = __popcll(SR & above);
    const int nl = __popcll(SL), nr = __popcll(SR);
    const int m  = nl < nr ? nl : nr;
    // lane sf + k receives l_k and r_k; everybody else sends to a lane nobody reads (the segment's last lane can only be
    // the destination of a stop when every item of the segment is one; a lane outside the open segments sends to itself)
    const int idle = act ? sl - 1 : lane;
    const int TL   = __builtin_amdgcn_ds_permute((fl ? sf + kl : idle) << 2, lane);
    const int TR   = __builtin_amdgcn_ds_permute((fr ? sf + kr : idle) << 2, lane);
    const bool uncrossed = act && lane - sf < m && TL < TR;
    const int K          = __popcll(__ballot(uncrossed) & seg);  // pairs the serial loop swaps
    const int partner_l  = __shfl(TR, act ? sf + kl : lane, 64);
    const int partner_r  = __shfl(TL, act ? sf + kr : lane, 64);
    const int lK         = __shfl(TL, act && K < nl ? sf + K : lane, 64);
    const int rK1        = __shfl(TR, act && K > 0 ? sf + K - 1 : lane, 64);
    int src = lane;
    if (fl && kl < K) {
      src = partner_l;
    } else if (fr && kr < K) {
      src = partner_r;
    }
    x = (uint32_t) __shfl((int) x, src, 64);
    if (act) {
      int cut = 1 << 20;
      cut     = K < nl && lK < cut ? lK : cut;
      cut     = K > 0 && rK1 < cut ? rK1 : cut;
      if (lane < cut) {
        sl = cut;
      } else {
        sf = cut;
      }
      --sd;
    }
  }
  // __final_insertion_sort inside every segment (nothing enters or leaves one): stable sort by decreasing response
  const uint32_t r = x & 0xffu;
  int rank         = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j       = sf + t;
    const uint32_t rj = (uint32_t) __shfl((int) r, j < 64 ? j : 63, 64);
    rank += (j < sl && (rj > r || (rj == r && j < lane))) ? 1 : 0;
  }
  wave_sync_lds();
  if (lane < n) {
    v[first + (heaped ? lane : sf + rank)] = x;
  }
  wave_sync_lds();
}

// GNU libstdc++'s std::sort of every region, replayed by ALL waves of the workgroup: the ranges __introsort_loop recurses into
// are disjoint, so the order in which they are finished does not change the result.  Ranges of more than 64 items wait in one
// LIFO of the workgroup (a private stack of the wave takes over when it is full); a wave takes a range, partitions it (whole
// wave, std_partition_wave), hands the right part over and continues on the left one; heapsort (depth limit) runs on one lane;
// a range of <= 64 items is finished in registers (std_sort_small_wave).  `remaining` counts the items that are not in a
// finished range yet: the waves leave when it reaches zero.
constexpr int kSortQueueCap  = 1024;
constexpr int kSortSpinLimit = 1 << 20;  // polls of an empty queue before a wave gives up (a legitimate wait is a few hundred)
struct SortQueue {
  int lock, top, remaining, failed;
  uint2 e[kSortQueueCap];  // first | depth << 16, last
};

__device__ __forceinline__ void sort_queue_seed(SortQueue& sq, const int first, const int n) {  // one thread, before the barrier
  if (n <= 1) {
    return;
  }
  int lg = 0;
  for (int m = n; m > 1; m >>= 1) {
    ++lg;
  }
  sq.e[sq.top++] = make_uint2((uint32_t) first | ((uint32_t) (2 * lg) << 16), (uint32_t) (first + n));
  sq.remaining += n;
}

// the queue's spin lock, taken by one lane of a wave; every access to top / e[] happens between lock and unlock
__device__ __forceinline__ bool sort_queue_lock(SortQueue& sq) {
  for (int tries = 0; __hip_atomic_exchange(&sq.lock, 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; ++tries) {
    if (tries > kSortSpinLimit) {
      __hip_atomic_store(&sq.failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
__device__ __forceinline__ void sort_queue_unlock(SortQueue& sq) {
  __hip_atomic_store(&sq.lock, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// `region_start` / `keep` (round 5): the caller only reads the first `keep` positions of every region (start[key >> 23] is where the
// region of a key begins).  The ranges __introsort_loop recurses into are disjoint and the final insertion sort never moves an item
// across a cut, so a range that begins at or behind the region's `keep`-th position cannot change what is read: it is dropped
// unsorted (a region of ~330 detections with 111 kept: about half of the partitions and register passes).  nullptr: sort everything.
__device__ __forceinline__ void std_sort_worker(uint32_t* v, SortQueue& sq, const int lane, int* q, const uint32_t* region_start = nullptr,
                                                const int keep = 0) {
  int* tl    = q;
  int* tr    = q + 64;
  int* stack = q + 128;  // private (first, last, depth) triples: the right siblings along this wave's path, at most 2 lg + 1 <= 31
  int sp     = 0;
  for (;;) {
    int first, last, depth;
    if (sp > 0) {
      --sp;
      first = stack[3 * sp];
      last  = stack[3 * sp + 1];
      depth = stack[3 * sp + 2];
      wave_sync_lds();
    } else {
      int got = 0;
      uint2 e = make_uint2(0u, 0u);
      if (lane == 0) {
        for (int polls = 0;; ++polls) {
          if (__hip_atomic_load(&sq.remaining, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= 0 ||
              __hip_atomic_load(&sq.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
            got = -1;
            break;
          }
          if (polls > kSortSpinLimit) {  // cannot happen; a loud per-image error instead of a hung device if it ever does
            __hip_atomic_store(&sq.failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            got = -1;
            break;
          }
          if (__hip_atomic_load(&sq.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > 0) {
            if (sort_queue_lock(sq)) {
              const int t = __hip_atomic_load(&sq.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (t > 0) {
                e = sq.e[t - 1];
                __hip_atomic_store(&sq.top, t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                got = 1;
              }
              sort_queue_unlock(sq);
            }
            if (got) {
              break;
            }
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      got = __builtin_amdgcn_readfirstlane(got);
      if (got < 0) {
        return;
      }
      const uint32_t ex = (uint32_t) __builtin_amdgcn_readfirstlane((int) e.x);
      first             = (int) (ex & 0xffffu);
      depth             = (int) (ex >> 16);
      last              = __builtin_amdgcn_readfirstlane((int) e.y);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // the items of the range were written by another wave
    }
    const int limit = region_start ? (int) region_start[v[first] >> 23] + keep : 0x7fffffff;  // (one region per range)
    int finished = 0;  // items of ranges this chain has finished
    for (;;) {
      const int size = last - first;
      if (size <= 64) {
        if (size > 1) {
          std_sort_small_wave(v, first, size, depth, lane);
        }
        finished += size;
        break;
      }
      if (depth == 0) {
        if (lane == 0) {
          std_heapsort(v + first, size);
        }
        wave_sync_lds();
        finished += size;  // heap-sorted: the final insertion pass finds nothing to move
        break;
      }
      --depth;
      const int cut   = std_partition_wave(v, first, last, lane, tl, tr);
      const int rsize = last - cut;
      if (cut >= limit) {
        finished += rsize;  // nobody reads these positions
      } else if (rsize <= 64) {
        if (rsize > 1) {
          std_sort_small_wave(v, cut, rsize, depth, lane);
        }
        finished += rsize;
      } else {
        int pushed = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the items of the right part, for the wave that takes it
        if (lane == 0 && sort_queue_lock(sq)) {
          const int t = __hip_atomic_load(&sq.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (t < kSortQueueCap) {
            sq.e[t] = make_uint2((uint32_t) cut | ((uint32_t) depth << 16), (uint32_t) last);
            __hip_atomic_store(&sq.top, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pushed = 1;
          }
          sort_queue_unlock(sq);
        }
        pushed = __builtin_amdgcn_readfirstlane(pushed);
        if (!pushed) {
          if (lane == 0) {
            stack[3 * sp]     = cut;
            stack[3 * sp + 1] = last;
            stack[3 * sp + 2] = depth;
          }
          ++sp;
          wave_sync_lds();
        }
      }
      last = cut;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
      __hip_atomic_fetch_add(&sq.remaining, -finished, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

__global__ __launch_bounds__(kSelThreads) void select_describe_kernel(const FeatureArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t keys[];  // [max_raw]
  __shared__ uint32_t count[kMaxRegions + 1];
  __shared__ uint32_t start[kMaxRegions + 1];
  __shared__ int wave_tot[kSelThreads / 64];
  __shared__ SortQueue sortq;
  __shared__ uint16_t patch[(kSelThreads / 64) * kSelScratch];  // scratch: per-wave region counts (scatter), partition lists and stack (sort)
  const int rows = a.b.rows, cols = a.b.cols;
  const int img  = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* __restrict__ raw = a.raw + (size_t) img * a.max_raw;
  const bool std_order = a.p.selection_order == PRS_SELECT_LIBSTDCXX;
  const int n = a.n_raw[img];
  auto stamp = [&](const int i) {
    if (a.stamps && tid == 0) {
      a.stamps[(size_t) img * 16 + i] = (unsigned long long) clock64();
    }
  };
  stamp(0);
  if (n < 0) {  // more raw detections than the selection can hold: loud per-image error
    if (tid == 0) {
      a.b.n_features[img] = 0;
      a.b.status[img]     = PRS_ERR_CAPACITY;
    }
    return;
  }
  for (int i = tid; i <= a.regions; i += kSelThreads) {
    count[i] = 0;
  }
  int n_sort = 1024;
  while (n_sort < n) {
    n_sort <<= 1;
  }
  __syncthreads();
  auto region_of = [&](const uint32_t pix) -> uint32_t {  // intensity_feature_extractor_binned.cpp:85-92
    const int r = (int) (pix / (uint32_t) cols), c = (int) (pix - (uint32_t) r * (uint32_t) cols);
    const int g = (int) floorf((float) r / a.rows_per) * a.p.number_of_detectors_horizontal + (int) ((float) c / a.cols_per);
    return (uint32_t) (g < a.regions ? g : a.regions - 1);
  };
  if (std_order) {
    // ---- libstdc++ order: every region's std::vector before its std::sort = its keypoints in detection order.  A stable
    //      scatter by region: every wave owns a contiguous span of the detections, counts its share of every region, the counts
    //      become offsets (regions x waves), and the wave writes its items behind those of the waves before it.
    uint32_t* wcnt      = reinterpret_cast<uint32_t*>(patch) + wave * kMaxRegions;
    const int span      = (((n + kSelThreads / 64 - 1) / (kSelThreads / 64)) + 63) & ~63;
    const int span_from = wave * span;
    const int span_to   = span_from + span < n ? span_from + span : n;
    for (int g = lane; g < a.regions; g += 64) {
      wcnt[g] = 0;
    }
    wave_sync_lds();
    for (int i0 = span_from; i0 < span_to; i0 += 64) {
      const int i       = i0 + lane;
      const bool valid  = i < span_to;
      const uint32_t g  = valid ? region_of(raw[i] & 0xffffffu) : 0xffffffffu;
      unsigned long long todo = __ballot(valid);
      while (todo) {  // one trip per region present among these 64 detections (raster order: a handful)
        const int leader           = (int) __ffsll((long long) todo) - 1;
        const uint32_t gl          = (uint32_t) __shfl((int) g, leader, 64);
        const unsigned long long b = __ballot(g == gl);
        if (lane == leader) {
          wcnt[gl] += (uint32_t) __popcll(b);
        }
        todo &= ~b;
        wave_sync_lds();
      }
    }
    __syncthreads();
    for (int g = tid; g < a.regions; g += kSelThreads) {
      uint32_t* col = reinterpret_cast<uint32_t*>(patch) + g;
      uint32_t run  = 0;
      for (int w = 0; w < kSelThreads / 64; ++w) {
        const uint32_t t      = col[w * kMaxRegions];
        col[w * kMaxRegions] = run;
        run += t;
      }
      count[g] = run;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (int g = 0; g < a.regions; ++g) {
        start[g] = run;
        run += count[g];
      }
      start[a.regions] = run;
    }
    __syncthreads();
    for (int i0 = span_from; i0 < span_to; i0 += 64) {
      const int i       = i0 + lane;
      const bool valid  = i < span_to;
      const uint32_t rw = valid ? raw[i] : 0u;
      const uint32_t g  = valid ? region_of(rw & 0xffffffu) : 0xffffffffu;
      unsigned long long todo = __ballot(valid);
      while (todo) {
        const int leader           = (int) __ffsll((long long) todo) - 1;
        const uint32_t gl          = (uint32_t) __shfl((int) g, leader, 64);
        const unsigned long long b = __ballot(g == gl);
        if (g == gl) {
          const uint32_t rank = (uint32_t) __popcll(b & ((1ull << lane) - 1ull));
          keys[start[gl] + wcnt[gl] + rank] = (gl << 23) | ((uint32_t) i << 8) | (rw >> 24);
        }
        wave_sync_lds();
        if (lane == leader) {
          wcnt[gl] += (uint32_t) __popcll(b);
        }
        todo &= ~b;
        wave_sync_lds();
      }
    }
    __syncthreads();
  } else {
    // ---- region of every keypoint (intensity_feature_extractor_binned.cpp:85-92) + region sizes ------------
    for (int i = tid; i < n_sort; i += kSelThreads) {
      uint32_t region = 0xffffffffu;
      if (i < n) {
        region = region_of(raw[i] & 0xffffffu);
        atomicAdd(&count[region], 1u);
      }
      keys[i] = region;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t run = 0;
      for (int g = 0; g < a.regions; ++g) {
        start[g] = run;
        run += count[g];
      }
      start[a.regions] = run;
    }
    // ---- keys.  Canonical order: (region, sort field, detection order); a region below its target keeps detection order
    //      (:174-178), the others are ordered by decreasing response (:179-195), ties by detection order.
    for (int i = tid; i < n; i += kSelThreads) {
      const uint32_t region = keys[i];
      const uint32_t s      = raw[i] >> 24;
      const uint32_t field  = count[region] < (uint32_t) a.target_per ? 0u : 255u - s;
      keys[i]               = (region << 23) | (field << 15) | (uint32_t) i;
    }
    __syncthreads();
    // ---- bitonic sort of the keys in LDS -------------------------------------------------------------------
    for (int size = 2; size <= n_sort; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (n_sort >> 1); t += kSelThreads) {
          const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
          const int hi = lo + stride;
          const bool up = (lo & size) == 0;
          const uint32_t x = keys[lo], y = keys[hi];
          if ((x > y) == up) {
            keys[lo] = y;
            keys[hi] = x;
          }
        }
        __syncthreads();
      }
    }
  }
  stamp(1);
  if (std_order) {  // the waves of the workgroup replay the reference's std::sort of every region that is sorted at all
    if (tid == 0) {
      sortq.lock = sortq.top = sortq.remaining = sortq.failed = 0;
      for (int g = 0; g < a.regions; ++g) {
        if (count[g] >= (uint32_t) a.target_per) {
          sort_queue_seed(sortq, (int) start[g], (int) count[g]);
        }
      }
    }
    __syncthreads();
    std_sort_worker(keys, sortq, lane, reinterpret_cast<int*>(patch + wave * kSelScratch), start, a.target_per);
    __syncthreads();
    if (sortq.failed) {  // a wave gave up waiting (std_sort_worker): loud per-image error
      if (tid == 0) {
        a.b.n_features[img] = 0;
        a.b.status[img]     = PRS_ERR_HIP;
      }
      return;
    }
  }
  stamp(2);
  // ---- selection + border filter + ordered output slots, 1024 sorted positions at a time ------------------------
  uint32_t* __restrict__ kept = a.kept + (size_t) img * a.b.stride;
  int running   = 0;
  bool overflow = false;
  for (int p0 = 0; p0 < n; p0 += kSelThreads) {
    const int p = p0 + tid;
    bool keep   = false;
    uint32_t pix = 0;
    if (p < n) {
      const uint32_t key    = keys[p];
      const uint32_t region = key >> 23;
      const uint32_t rank   = (uint32_t) p - start[region];
      pix                   = raw[std_order ? (key >> 8) & 0x7fffu : key & 0x7fffu] & 0xffffffu;
      const int r           = (int) (pix / (uint32_t) cols);
      const int c           = (int) (pix - (uint32_t) r * (uint32_t) cols);
      keep = (count[region] < (uint32_t) a.target_per || rank < (uint32_t) a.target_per) && r >= kFeatureBorder &&
             r < rows - kFeatureBorder && c >= kFeatureBorder && c < cols - kFeatureBorder;
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) {
      wave_tot[wave] = __popcll(bal);
    }
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSelThreads / 64; ++w) {
      before += w < wave ? wave_tot[w] : 0;
      total += wave_tot[w];
    }
    if (keep) {
      const int slot = running + before + __popcll(bal & ((1ull << lane) - 1ull));
      if (slot < a.b.stride) {
        kept[slot] = pix;
      } else {
        overflow = true;
      }
    }
    running += total;
    __syncthreads();
  }
  stamp(3);
  const int any_overflow = __syncthreads_or(overflow ? 1 : 0);
  stamp(4);
  if (any_overflow) {
    if (tid == 0) {
      a.b.n_features[img] = 0;
      a.b.status[img]     = PRS_ERR_CAPACITY;
    }
    return;
  }
  if (tid == 0) {
    a.b.n_features[img] = running;
    a.b.status[img]     = running == 0 ? PRS_WARN_NO_MATCHES : PRS_OK;  // :126-131 "no keypoints detected"
  }
}

// ---- descriptors (cv::ORB on provided keypoints: no orientation, unrotated pattern): one wave per keypoint, sixteen
//      keypoints per wave.  The smoothed pixels around the keypoint (27 rows, 32 bytes from the word that holds column
//      c - 13) come in as 216 aligned words = four loads per lane out of ~12 cache lines of the blocked image; the loads of
//      the next keypoint are in flight while this one is compared; the window sits in LDS, every lane evaluates four
//      comparisons and a ballot IS eight bytes of the descriptor: bit t lands in byte t / 8, bit t % 8.
//      (Rounds 2-4 described inside the selection kernel, one byte load per distinct cell of the pair table from a row-major
//      image: ~33 cache lines per keypoint.)
__global__ __launch_bounds__(kDescThreads) void describe_kernel(const FeatureArgs a, const int chunks, const int per_wave) {
  __shared__ uint32_t window[kDescThreads / 64][2][4 * 64];
  const int cols = a.b.cols;
  // consecutive workgroups go to consecutive XCDs (eight L2s): the chunks of one image stay on one of them
  const int group  = (int) blockIdx.x >> 3;
  const int img    = (group / chunks) * 8 + ((int) blockIdx.x & 7), chunk = group % chunks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (img >= a.b.batch) {
    return;
  }
  const int n_kept = min(a.b.n_features[img], a.b.stride);  // (0 for an image that failed)
  const int s0     = (chunk * (kDescThreads / 64) + wave) * per_wave;
  if (s0 >= n_kept) {
    return;
  }
  const int cnt = min(per_wave, n_kept - s0);
  const uint8_t* __restrict__ blur = a.blur + (size_t) img * a.blur_stride;
  const uint8_t* __restrict__ src  = a.b.images + (size_t) img * a.b.rows * a.b.pitch;
  prs_kp2* __restrict__ out_kp     = a.b.keypoints + (size_t) img * a.b.stride;
  float* __restrict__ out_int      = a.b.intensity ? a.b.intensity + (size_t) img * a.b.stride : nullptr;
  uint8_t* __restrict__ out_desc   = a.b.descriptors + (size_t) img * a.b.stride * PRS_DESC_BYTES;
  const uint32_t my_pix = a.kept[(size_t) img * a.b.stride + s0 + min(lane, cnt - 1)];
  const int my_r = (int) (my_pix / (uint32_t) cols), my_c = (int) (my_pix - (uint32_t) my_r * (uint32_t) cols);
  if (lane < cnt) {  // keypoint and intensity of the wave's sixteen slots at once
    out_kp[s0 + lane] = prs_kp2{(float) my_c, (float) my_r};
    if (out_int) {
      out_int[s0 + lane] = (float) src[(size_t) my_r * a.b.pitch + my_c];  // intensity_feature_extractor_base.cpp:80
    }
  }
  int o1[4], o2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o1[j] = a.pair_first[64 * j + lane];
    o2[j] = a.pair_second[64 * j + lane];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asm volatile("" : "+v"(o1[j]), "+v"(o2[j]));  // the offsets have arrived before the loop (its waits then only count the loop's own loads)
  }
  const int ncb = a.blur_ncb, block_row = 128 * ncb;
  // word e = lane + 64 u of the window: row e / 8 (rows 27 .. 31 ride along: they are inside the image too), bytes 4 (e % 8) .. + 3
  // from the aligned column; eight rows further down is the next row of blocks
  auto fetch = [&](const int i, uint32_t (&v)[4]) {
    const int r = __builtin_amdgcn_readlane(my_r, i), c = __builtin_amdgcn_readlane(my_c, i);
    const uint32_t off = blur_offset(r - kWinRadius + (lane >> 3), ((c - kWinRadius) & ~3) + 4 * (lane & 7), ncb);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = *reinterpret_cast<const uint32_t*>(blur + (off + (uint32_t) (u * block_row)));  // a kept keypoint is >= 31 px inside the image
    }
  };
  // dword k of the descriptor = half (k & 1) of the ballot of comparisons 64 (k >> 1) ..: lane k stores it
  uint32_t* my_word = reinterpret_cast<uint32_t*>(out_desc + (size_t) s0 * PRS_DESC_BYTES) + (lane & 7);
  auto describe = [&](const int i, const uint32_t (&v)[4], uint32_t* win) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      win[lane + 64 * u] = v[u];
    }
    wave_sync_lds();
    const int c = __builtin_amdgcn_readlane(my_c, i);
    const uint8_t* win8 = reinterpret_cast<const uint8_t*>(win) + ((c - kWinRadius) & 3);
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned long long bits = __ballot(win8[o1[j]] < win8[o2[j]]);
      word = lane == 2 * j ? (uint32_t) bits : (lane == 2 * j + 1 ? (uint32_t) (bits >> 32) : word);
    }
    if (lane < 8) {
      my_word[i * (PRS_DESC_BYTES / 4)] = word;
    }
  };
  uint32_t va[4], vb[4];
  fetch(0, va);
  // the window of keypoint i + 2 reuses the LDS of keypoint i: LDS operations of a wave complete in order.  The fetches are
  // unconditional (the last keypoint is fetched again past the end): a branch around them would make the compiler wait for
  // ALL loads before the compare, the one just issued included
  for (int i = 0; i < cnt; i += 2) {
    fetch(min(i + 1, cnt - 1), vb);
    describe(i, va, window[wave][0]);
    fetch(min(i + 2, cnt - 1), va);
    if (i + 1 < cnt) {
      describe(i + 1, vb, window[wave][1]);
    }
  }
}

static const int8_t kOrbPattern[1024] = {
#include "orb_pattern.inc"
};

// every comparison's two cells as byte offsets into the 27-row x 28-byte window around the keypoint
static void fill_window_offsets(const int8_t* pattern, FeatureArgs* a) {
  auto off = [](int x, int y) { return (uint16_t) ((y + kWinRadius) * 4 * kWinWords + (x + kWinRadius)); };  // + (c - 13) & 3 on the device
  for (int t = 0; t < 256; ++t) {
    a->pair_first[t]  = off(pattern[4 * t], pattern[4 * t + 1]);
    a->pair_second[t] = off(pattern[4 * t + 2], pattern[4 * t + 3]);
  }
}

// ---- prs_selection_order: the permutation the reference's std::sort leaves for one region's responses ----------------
__global__ __launch_bounds__(kSelThreads) void selection_order_kernel(const uint8_t* __restrict__ response, const int n, int32_t* __restrict__ order,
                                                                      int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint32_t keys[];  // [n]
  __shared__ SortQueue sortq;
  __shared__ int scratch[(kSelThreads / 64) * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += kSelThreads) {
    keys[i] = ((uint32_t) i << 8) | (uint32_t) response[i];
  }
  if (tid == 0) {
    sortq.lock = sortq.top = sortq.remaining = sortq.failed = 0;
    sort_queue_seed(sortq, 0, n);
  }
  __syncthreads();
  std_sort_worker(keys, sortq, lane, scratch + wave * 256);
  __syncthreads();
  for (int i = tid; i < n; i += kSelThreads) {
    order[i] = (int32_t) (keys[i] >> 8);
  }
  if (tid == 0) {
    *status = sortq.failed ? PRS_ERR_HIP : PRS_OK;
  }
}

int selection_order_launch(prs_context* ctx, const uint8_t* response_dev, int n, int32_t* order_dev, int32_t* status_dev) {
  const size_t lds = (size_t) (n > 0 ? n : 1) * sizeof(uint32_t);
  hipError_t e     = hipFuncSetAttribute(reinterpret_cast<const void*>(selection_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_selection_order: LDS for the keys");
  }
  hipLaunchKernelGGL(selection_order_kernel, dim3(1), dim3(kSelThreads), lds, ctx->stream, response_dev, n, order_dev, status_dev);
  e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_selection_order launch");
  }
  return PRS_OK;
}

int extract_features_launch(prs_context* ctx, const prs_extractor_params* params, const prs_extract_batch* batch) {
  if (!params || !batch || !batch->images) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_extract_features_batch: image not set");
  }
  if (!batch->keypoints || !batch->descriptors || !batch->n_features || !batch->status) {
    // intensity_feature_extractor_base.cpp:59-64: "target feature buffer not set"
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_extract_features_batch: target feature buffer not set");
  }
  if (batch->batch <= 0) {
    return PRS_OK;
  }
  const int regions = params->number_of_detectors_vertical * params->number_of_detectors_horizontal;
  if (params->number_of_detectors_vertical <= 0 || params->number_of_detectors_horizontal <= 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_extract_features_batch: invalid number of detectors");  // binned.cpp:13-20
  }
  if (batch->rows < 7 || batch->cols < 7 || batch->pitch < batch->cols || (size_t) batch->rows * batch->cols >= (1u << 24) ||
      regions > kMaxRegions || params->detector_threshold < 1 || params->detector_threshold > 254 || batch->stride <= 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED,
                    "prs_extract_features_batch: image below 7x7 or above 2^24 pixels, more than 256 regions, or threshold outside [1,254]");
  }
  int max_raw = params->max_raw_detections > 0 ? params->max_raw_detections : kDefaultRaw;
  if (max_raw > kMaxRawLimit || (params->selection_order != PRS_SELECT_CANONICAL && params->selection_order != PRS_SELECT_LIBSTDCXX)) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_extract_features_batch: max_raw_detections above 32768 or unknown selection_order");
  }
  {
    int pow2 = 1024;
    while (pow2 < max_raw) {
      pow2 <<= 1;
    }
    max_raw = pow2;
  }
  FeatureArgs a;
  a.p = *params;
  a.b = *batch;
  const size_t npix = (size_t) batch->rows * batch->cols;
  a.kept  = static_cast<uint32_t*>(ctx_device_scratch_slot(ctx, 0, (size_t) batch->batch * (size_t) batch->stride * sizeof(uint32_t)));
  a.blur_ncb    = (batch->cols + 15) / 16;
  a.blur_stride = (size_t) ((batch->rows + 7) / 8) * a.blur_ncb * 128;
  a.blur        = static_cast<uint8_t*>(ctx_device_scratch_slot(ctx, 1, (size_t) batch->batch * a.blur_stride));
  uint32_t* rawbuf = static_cast<uint32_t*>(ctx_device_scratch_slot(ctx, 2, (size_t) batch->batch * ((size_t) max_raw + 1) * 4));
  if (!a.kept || !a.blur || !rawbuf) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_extract_features_batch: scratch allocation failed");
  }
  a.raw   = rawbuf;
  a.n_raw   = reinterpret_cast<int32_t*>(rawbuf + (size_t) batch->batch * max_raw);
  a.max_raw = max_raw;
  a.stamps  = ctx_stamps(ctx, (size_t) batch->batch * 2 * 16 * sizeof(unsigned long long));
  fill_window_offsets(kOrbPattern, &a);
  a.rows_per   = (float) batch->rows / (float) params->number_of_detectors_vertical;   // binned.cpp:52-55
  a.cols_per   = (float) batch->cols / (float) params->number_of_detectors_horizontal;
  a.regions    = regions;
  a.target_per = (int) ((float) params->target_number_of_keypoints / (float) regions);  // :72-76
  hipStream_t stream = ctx_stream(ctx);
  const size_t lds_keys = (size_t) max_raw * sizeof(uint32_t);
  hipError_t e = hipMemsetAsync(a.n_raw, 0, (size_t) batch->batch * sizeof(int32_t), stream);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features_batch: detection counters");
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(raster_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_keys);
  if (e == hipSuccess) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_describe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_keys);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features_batch: LDS for the selection sort");
  }
  int bucket_shift = 0;  // 1024 spans of 2^shift pixels cover the image
  while (((npix - 1) >> bucket_shift) >= (size_t) kRasterBuckets) {
    ++bucket_shift;
  }
  const int per_wave  = kDescPerWave;
  const int per_block = (kDescThreads / 64) * per_wave;
  const int chunks    = (batch->stride + per_block - 1) / per_block;
  const dim3 tiles((batch->cols + kTileW - 1) / kTileW, (batch->rows + kTileH - 1) / kTileH, batch->batch);
  hipLaunchKernelGGL(fast_blur_kernel, tiles, dim3(kFastThreads), 0, stream, a);
  hipLaunchKernelGGL(raster_order_kernel, dim3(batch->batch), dim3(kNmsThreads), lds_keys, stream, a, bucket_shift);
  hipLaunchKernelGGL(select_describe_kernel, dim3(batch->batch), dim3(kSelThreads), lds_keys, stream, a);
  hipLaunchKernelGGL(describe_kernel, dim3((unsigned) (((batch->batch + 7) / 8) * 8 * chunks), 1, 1), dim3(kDescThreads), 0, stream, a, chunks, per_wave);
  e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features_batch launch");
  }
  if (a.stamps) {
    ctx_report_stamps(ctx, batch->batch, 5, "select_describe_kernel: keys | sort | selection | tail");
    if (batch->cols > 5 * kTileW && batch->rows > 2 * kTileH) {
      ctx_report_stamps(ctx, batch->batch, 6, "fast_blur_kernel, tile (5, 2): load | compass | arcs + Gaussian | barrier | suppression", false, (size_t) batch->batch);
    }
  }
  return PRS_OK;
}

}  // namespace prs
