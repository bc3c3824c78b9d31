// features.hip -- intensity feature extraction on the device:
// IntensityFeatureExtractorBinned_::computeKeypoints + compute
// (sensor_processing/feature_extractors/intensity_feature_extractor_binned.cpp:7-208,
//  intensity_feature_extractor_base.cpp:56-95) around the two OpenCV calls the reference makes:
// FAST detection with non-maximum suppression (:121-123) and a 256-bit binary descriptor (:139-170).
// OpenCV is not part of the reference tree: the detector is the published FAST-9 segment test with the
// arc-minimum response, the descriptor is BUILD-DEFINED (BRIEF-style comparisons of 5x5 box sums, pair
// table from a fixed linear congruential sequence) -- see include/proslam_hip.h.  The region grid and the
// per-region selection by response follow the in-repo code line by line.
//
// Three launches per batch of images:
//   fast_box_kernel    one 64x16 pixel tile per workgroup (tile + halo in LDS): FAST response map (u8)
//                      and 5x5 box-sum map (u16); HBM-bound apart from the segment test itself
//   nms_compact_kernel one workgroup per image: non-maximum suppression + ordered (raster) compaction
//   select_describe_kernel one workgroup per image: region histogram, one bitonic sort of
//                      (region, response, order) keys in LDS, per-region selection, border filter,
//                      256 box-sum comparisons per kept keypoint
#include "prs_device.h"
#include "prs_host.h"

namespace prs {

constexpr int kTileW = 64, kTileH = 64, kFastThreads = 256;
constexpr int kTilePitch = kTileW + 8;  // 4-px halo on both sides
constexpr int kNmsThreads = 1024, kSelThreads = 1024;
constexpr int kMaxRaw = 8192;           // raw detections per image the selection sort can hold
constexpr int kFeatureBorder = 17;      // keypoints closer to the border get no descriptor
constexpr int kMaxRegions = 256;

struct FeatureArgs {
  prs_extractor_params p;
  prs_extract_batch b;
  uint8_t* score;     // [batch][rows][cols]
  uint16_t* box;      // [batch][rows][cols] 5x5 sums
  uint32_t* raw;      // [batch][kMaxRaw] response << 24 | pixel index, raster order
  int32_t* n_raw;     // [batch]
  int8_t pattern[1024];
  float rows_per, cols_per;
  int target_per, regions;
};

// FAST-9: does a 16-bit circular mask contain 9 contiguous set bits?
__device__ __forceinline__ bool has_arc9(uint32_t m) {
  m |= m << 16;  // unroll the circle
  uint32_t x = m & (m >> 1);
  x &= x >> 2;   // runs of 4
  x &= x >> 4;   // runs of 8
  x &= m >> 8;   // runs of 9
  return (x & 0xffffu) != 0u;
}

// FAST-9 response of the pixel at `c` (LDS tile pointer, row pitch kTilePitch): largest threshold that still
// detects, 0 = no corner at threshold t
__device__ __forceinline__ int fast_response(const uint8_t* c, int t) {
  const int v = c[0];
  int d[16];
  d[0]  = c[-3 * kTilePitch + 0];
  d[1]  = c[-3 * kTilePitch + 1];
  d[2]  = c[-2 * kTilePitch + 2];
  d[3]  = c[-1 * kTilePitch + 3];
  d[4]  = c[3];
  d[5]  = c[1 * kTilePitch + 3];
  d[6]  = c[2 * kTilePitch + 2];
  d[7]  = c[3 * kTilePitch + 1];
  d[8]  = c[3 * kTilePitch + 0];
  d[9]  = c[3 * kTilePitch - 1];
  d[10] = c[2 * kTilePitch - 2];
  d[11] = c[1 * kTilePitch - 3];
  d[12] = c[-3];
  d[13] = c[-1 * kTilePitch - 3];
  d[14] = c[-2 * kTilePitch - 2];
  d[15] = c[-3 * kTilePitch - 1];
  uint32_t brighter = 0, darker = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    d[i] -= v;
    brighter |= (d[i] > t ? 1u : 0u) << i;
    darker |= (d[i] < -t ? 1u : 0u) << i;
  }
  const bool cb = has_arc9(brighter), cd = has_arc9(darker);
  if (!cb && !cd) {
    return 0;
  }
  // response = (max over arcs of the arc minimum of |difference|) - 1.  An arc can only be all brighter or
  // all darker than the centre; when just one polarity fires the other cannot hold the maximum above t
  int best = -256;
#pragma unroll
  for (int sign = 0; sign < 2; ++sign) {
    if (sign == 0 ? !cb : !cd) {
      continue;
    }
    int e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      e[i] = sign ? -d[i] : d[i];
    }
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m2[i] = min(e[i], e[(i + 1) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m4[i] = min(m2[i], m2[(i + 2) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      m8[i] = min(m4[i], m4[(i + 4) & 15]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      best = max(best, min(m8[i], e[(i + 8) & 15]));  // minimum over the arc i .. i+8
    }
  }
  return best > t ? best - 1 : 0;
}

// image tile (+ 4-px halo) -> responses of the tile + 1-px ring -> non-maximum suppression in LDS -> the response
// map holds a non-zero value only at surviving keypoints; 5x5 box sums of the tile on the side
__global__ __launch_bounds__(kFastThreads) void fast_box_kernel(const FeatureArgs a) {
  constexpr int kHalo = 4;
  constexpr int kSW = kTileW + 2, kSH = kTileH + 2;  // response tile with its 1-px ring
  __shared__ __attribute__((aligned(4))) uint8_t tile[(kTileH + 2 * kHalo) * kTilePitch];
  __shared__ uint8_t resp[kSH * kSW];
  __shared__ uint16_t hsum[(kTileH + 4) * kTileW];  // horizontal 5-sums of the tile rows -2 .. kTileH+1
  __shared__ uint16_t cand[kSH * kSW];              // response-tile positions that pass the compass test
  __shared__ int n_cand;
  if (threadIdx.x == 0) {
    n_cand = 0;
  }
  const int rows = a.b.rows, cols = a.b.cols, pitch = a.b.pitch;
  const int img  = blockIdx.z;
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  const uint8_t* __restrict__ src = a.b.images + (size_t) img * rows * pitch;
  const int tid = threadIdx.x;
  // tile + halo.  A tile whose halo lies inside the image takes it in 4-byte pieces (18 per row; global loads may
  // be unaligned, the LDS stores are not: the tile array and its pitch are multiples of four); tiles on the image
  // border go byte by byte with clamped coordinates (clamped pixels never reach an output).
  if (x0 >= kHalo && x0 + kTileW + kHalo <= cols && y0 >= kHalo && y0 + kTileH + kHalo <= rows) {  // (block-uniform)
    constexpr int kWords = kTilePitch / 4;
    for (int i = tid; i < (kTileH + 2 * kHalo) * kWords; i += kFastThreads) {
      const int ty = i / kWords, tw = i - ty * kWords;
      const uint8_t* g = src + (size_t) (y0 + ty - kHalo) * pitch + (x0 - kHalo) + 4 * tw;
      uint32_t w;
      __builtin_memcpy(&w, g, 4);
      *reinterpret_cast<uint32_t*>(tile + ty * kTilePitch + 4 * tw) = w;
    }
  } else {
    for (int i = tid; i < (kTileH + 2 * kHalo) * (kTileW + 2 * kHalo); i += kFastThreads) {
      const int ty = i / (kTileW + 2 * kHalo), tx = i - ty * (kTileW + 2 * kHalo);
      int gy = y0 + ty - kHalo, gx = x0 + tx - kHalo;
      gy     = gy < 0 ? 0 : (gy >= rows ? rows - 1 : gy);
      gx     = gx < 0 ? 0 : (gx >= cols ? cols - 1 : gx);
      tile[ty * kTilePitch + tx] = src[(size_t) gy * pitch + gx];
    }
  }
  __syncthreads();
  const int t = a.p.detector_threshold;
  // Nine contiguous circle pixels always contain two ADJACENT compass points (circle indices 0, 4, 8, 12), so a
  // pixel can only be a corner if two adjacent compass points are both brighter than v + t or both darker than
  // v - t.  That four-read test runs for every pixel; the pixels that pass it are collected (order irrelevant) and
  // the full segment test then runs on dense lanes.
  for (int i = tid; i < kSH * kSW; i += kFastThreads) {
    const int sy = i / kSW, sx = i - sy * kSW;
    const int gx = x0 + sx - 1, gy = y0 + sy - 1;
    bool candidate = false;
    if (gx >= 3 && gx < cols - 3 && gy >= 3 && gy < rows - 3) {  // the outermost 3 pixels are not examined
      const uint8_t* c = tile + (sy - 1 + kHalo) * kTilePitch + (sx - 1 + kHalo);
      const int v = c[0];
      const int n = c[-3 * kTilePitch], e = c[3], so = c[3 * kTilePitch], w = c[-3];
      // adjacent pairs (N,E) (E,S) (S,W) (W,N): both brighter than v + t  <=>  the largest pair-minimum is; both
      // darker than v - t  <=>  the smallest pair-maximum is (min / max instructions, no compare-and-select chains)
      const int bright = max(max(min(n, e), min(e, so)), max(min(so, w), min(w, n)));
      const int dark   = min(min(max(n, e), max(e, so)), min(max(so, w), max(w, n)));
      candidate        = bright > v + t || dark < v - t;
    }
    resp[i] = 0;
    if (candidate) {
      cand[atomicAdd(&n_cand, 1)] = (uint16_t) i;
    }
  }
  __syncthreads();
  for (int j = tid; j < n_cand; j += kFastThreads) {
    const int i  = cand[j];
    const int sy = i / kSW, sx = i - sy * kSW;
    resp[i] = (uint8_t) fast_response(tile + (sy - 1 + kHalo) * kTilePitch + (sx - 1 + kHalo), t);
  }
  for (int i = tid; i < (kTileH + 4) * kTileW; i += kFastThreads) {  // separable 5x5 box sum, horizontal pass
    const int hy = i >> 6, hx = i & 63;
    const uint8_t* c = tile + (hy - 2 + kHalo) * kTilePitch + (hx + kHalo);
    hsum[i]          = (uint16_t) (((int) c[-2] + (int) c[-1]) + ((int) c[0] + (int) c[1]) + (int) c[2]);
  }
  __syncthreads();
  const int lx = tid & 63;
  uint8_t* __restrict__ score = a.score + (size_t) img * rows * cols;
  uint16_t* __restrict__ box  = a.box + (size_t) img * rows * cols;
#pragma unroll
  for (int k = 0; k < kTileH / 4; ++k) {
    const int ly = (tid >> 6) + 4 * k;
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx >= cols || gy >= rows) {
      continue;
    }
    int sum = 0;
    if (gx >= 2 && gx < cols - 2 && gy >= 2 && gy < rows - 2) {
      const uint16_t* h = hsum + (ly + 2) * kTileW + lx;  // vertical pass (integer sums: any order gives the same value)
      sum = ((int) h[-2 * kTileW] + (int) h[-kTileW]) + ((int) h[0] + (int) h[kTileW]) + (int) h[2 * kTileW];
    }
    box[(size_t) gy * cols + gx] = (uint16_t) sum;
    const uint8_t* q = resp + (ly + 1) * kSW + (lx + 1);
    int s            = q[0];
    if (s && a.p.enable_non_maximum_suppression) {
      // strictly greater than the 8 neighbours; a non-zero response never sits on the outermost 3 pixels, so the
      // ring values are real responses of real pixels
      const bool keep = q[-kSW - 1] < s && q[-kSW] < s && q[-kSW + 1] < s && q[-1] < s && q[1] < s && q[kSW - 1] < s && q[kSW] < s && q[kSW + 1] < s;
      s               = keep ? s : 0;
    }
    score[(size_t) gy * cols + gx] = (uint8_t) s;
  }
}

// raster-order compaction of the (already suppressed) response map.
// Every wave owns a contiguous range of the image; it counts its survivors, the 16 counts are scanned once,
// then the wave rescans its range (response map still in L2) and writes at its offset: two barriers per image.
__device__ __forceinline__ uint32_t nms_mask4(const FeatureArgs&, const uint8_t* __restrict__, int, int, uint32_t word) {
  // the response map is already suppressed: survivors are its non-zero bytes
  return ((word & 0xffu) ? 1u : 0u) | ((word & 0xff00u) ? 2u : 0u) | ((word & 0xff0000u) ? 4u : 0u) | ((word & 0xff000000u) ? 8u : 0u);
}

__global__ __launch_bounds__(kNmsThreads) void nms_compact_kernel(const FeatureArgs a) {
  __shared__ int wave_tot[kNmsThreads / 64];
  const int rows = a.b.rows, cols = a.b.cols;
  const int img  = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint8_t* __restrict__ score = a.score + (size_t) img * rows * cols;
  uint32_t* __restrict__ raw        = a.raw + (size_t) img * kMaxRaw;
  const int n_pix = rows * cols;
  // ranges in units of 256 pixels (64 lanes x 4), so that every lane reads one aligned 32-bit word per step
  const int n_chunks = (n_pix + 255) / 256;
  const int per_wave = (n_chunks + kNmsThreads / 64 - 1) / (kNmsThreads / 64);
  const int c_begin = wave * per_wave, c_end = min(c_begin + per_wave, n_chunks);
  const bool aligned = (((size_t) score) & 3) == 0;
  auto load4 = [&](int i) -> uint32_t {
    if (aligned && i + 3 < n_pix) {
      return *reinterpret_cast<const uint32_t*>(score + i);
    }
    uint32_t w = 0;
    for (int j = 0; j < 4; ++j) {
      w |= (i + j < n_pix ? (uint32_t) score[i + j] : 0u) << (8 * j);
    }
    return w;
  };
  int mine = 0;
  for (int ch = c_begin; ch < c_end; ch += 8) {  // eight independent loads in flight per lane
    uint32_t word[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = (ch + u) * 256 + 4 * lane;
      word[u]     = (ch + u < c_end && i < n_pix) ? load4(i) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (word[u]) {
        mine += __popc(nms_mask4(a, score, cols, (ch + u) * 256 + 4 * lane, word[u]));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mine += __shfl_xor(mine, o, 64);
  }
  if (lane == 0) {
    wave_tot[wave] = mine;
  }
  __syncthreads();
  int offset = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kNmsThreads / 64; ++w) {
    offset += w < wave ? wave_tot[w] : 0;
    total += wave_tot[w];
  }
  if (total <= kMaxRaw) {
    for (int ch0 = c_begin; ch0 < c_end; ch0 += 8) {
      uint32_t words[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = (ch0 + u) * 256 + 4 * lane;
        words[u]    = (ch0 + u < c_end && i < n_pix) ? load4(i) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i         = (ch0 + u) * 256 + 4 * lane;
        const uint32_t word = words[u];
        const uint32_t keep = word ? nms_mask4(a, score, cols, i, word) : 0u;
        if (__ballot(keep != 0u) == 0ull) {
          continue;  // nothing in these 256 pixels (wave-uniform)
        }
        const int cnt = __popc(keep);
        int incl      = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o, 64);
          if (lane >= o) {
            incl += v;
          }
        }
        int slot = offset + incl - cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((keep >> j) & 1u) {
            raw[slot++] = (((word >> (8 * j)) & 0xffu) << 24) | (uint32_t) (i + j);
          }
        }
        offset += __shfl(incl, 63, 64);
      }
    }
  }
  if (tid == 0) {
    a.n_raw[img] = total > kMaxRaw ? -1 : total;
  }
}

__global__ __launch_bounds__(kSelThreads) void select_describe_kernel(const FeatureArgs a) {
  __shared__ uint32_t keys[kMaxRaw];
  __shared__ uint32_t count[kMaxRegions + 1];
  __shared__ uint32_t start[kMaxRegions + 1];
  __shared__ int8_t pattern[1024];
  __shared__ int wave_tot[kSelThreads / 64];
  __shared__ uint16_t patch[(kSelThreads / 64) * 27 * 27];  // per wave: box sums around the keypoint being described
  const int rows = a.b.rows, cols = a.b.cols;
  const int img  = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* __restrict__ raw = a.raw + (size_t) img * kMaxRaw;
  const uint16_t* __restrict__ box = a.box + (size_t) img * rows * cols;
  const uint8_t* __restrict__ src  = a.b.images + (size_t) img * rows * a.b.pitch;
  const int n = a.n_raw[img];
  if (n < 0) {  // more raw detections than the selection can hold: loud per-image error
    if (tid == 0) {
      a.b.n_features[img] = 0;
      a.b.status[img]     = PRS_ERR_CAPACITY;
    }
    return;
  }
  for (int i = tid; i < 1024; i += kSelThreads) {
    pattern[i] = a.pattern[i];
  }
  for (int i = tid; i <= a.regions; i += kSelThreads) {
    count[i] = 0;
  }
  __syncthreads();
  // ---- region of every keypoint (intensity_feature_extractor_binned.cpp:85-92) + region sizes ------------
  uint32_t my_region[kMaxRaw / kSelThreads];
#pragma unroll
  for (int k = 0; k < kMaxRaw / kSelThreads; ++k) {
    const int i  = k * kSelThreads + tid;
    my_region[k] = 0;
    if (i < n) {
      const uint32_t pix = raw[i] & 0xffffffu;
      const int r = (int) (pix / (uint32_t) cols), c = (int) (pix - (uint32_t) r * (uint32_t) cols);
      const int region = (int) floorf((float) r / a.rows_per) * a.p.number_of_detectors_horizontal + (int) ((float) c / a.cols_per);
      my_region[k]     = (uint32_t) (region < a.regions ? region : a.regions - 1);
      atomicAdd(&count[my_region[k]], 1u);
    }
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
  // ---- keys: (region, sort field, detection order); a region below its target keeps detection order (:174-178),
  //      the others are ordered by decreasing response (:179-195), ties by detection order
#pragma unroll
  for (int k = 0; k < kMaxRaw / kSelThreads; ++k) {
    const int i = k * kSelThreads + tid;
    uint32_t key = 0xffffffffu;
    if (i < n) {
      const uint32_t s     = raw[i] >> 24;
      const uint32_t field = count[my_region[k]] < (uint32_t) a.target_per ? 0u : 255u - s;
      key                  = (my_region[k] << 21) | (field << 13) | (uint32_t) i;
    }
    keys[i] = key;
  }
  __syncthreads();
  // ---- bitonic sort of the 8192 keys in LDS -------------------------------------------------------------------
  int n_sort = 1024;
  while (n_sort < n) {
    n_sort <<= 1;
  }
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
  // ---- selection + border filter + ordered output slots, 1024 sorted positions at a time ------------------------
  prs_kp2* __restrict__ out_kp   = a.b.keypoints + (size_t) img * a.b.stride;
  float* __restrict__ out_int    = a.b.intensity ? a.b.intensity + (size_t) img * a.b.stride : nullptr;
  uint8_t* __restrict__ out_desc = a.b.descriptors + (size_t) img * a.b.stride * PRS_DESC_BYTES;
  int running   = 0;
  bool overflow = false;
  for (int p0 = 0; p0 < n; p0 += kSelThreads) {
    const int p = p0 + tid;
    bool keep   = false;
    uint32_t pix = 0;
    if (p < n) {
      const uint32_t key    = keys[p];
      const uint32_t region = key >> 21;
      const uint32_t rank   = (uint32_t) p - start[region];
      pix                   = raw[key & 0x1fffu] & 0xffffffu;
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
    __syncthreads();  // keys[p0 ..] have been read: the front of the array is reused for the kept list
    if (keep) {
      const int slot = running + before + __popcll(bal & ((1ull << lane) - 1ull));
      if (slot < a.b.stride) {
        keys[slot] = pix;  // slot <= p: never overtakes the sorted positions still to be read
      } else {
        overflow = true;
      }
    }
    running += total;
    __syncthreads();
  }
  const int n_kept = running < a.b.stride ? running : a.b.stride;
  // ---- descriptors: one wave per keypoint.  The 27x27 window of box sums the pair table can reach is staged in
  //      LDS with row-contiguous reads (scattered 2-byte reads from global memory would serialise in the
  //      texture addresser), then every lane evaluates four comparisons and a ballot IS eight bytes of the
  //      descriptor: bit t lands in byte t / 8, bit t % 8
  {
    constexpr int kWin = 27;  // offsets -13 .. 13
    int o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int8_t* pp = pattern + 4 * (64 * j + lane);
      o1[j]            = ((int) pp[1] + 13) * kWin + ((int) pp[0] + 13);
      o2[j]            = ((int) pp[3] + 13) * kWin + ((int) pp[2] + 13);
    }
    uint16_t* win = patch + wave * (kWin * kWin);
    for (int slot = wave; slot < n_kept; slot += kSelThreads / 64) {
      const uint32_t pix = keys[slot];
      const int r = (int) (pix / (uint32_t) cols), c = (int) (pix - (uint32_t) r * (uint32_t) cols);
      const uint16_t* __restrict__ corner = box + (size_t) (r - 13) * cols + (c - 13);
      uint16_t v[12];  // 12 x 64 >= 729: all loads are issued before the first one is consumed
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int e  = lane + 64 * u;
        const int ec = e < kWin * kWin ? e : kWin * kWin - 1;
        const int wr = ec / kWin, wc = ec - wr * kWin;
        v[u]         = corner[(size_t) wr * cols + wc];
      }
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int e = lane + 64 * u;
        if (e < kWin * kWin) {
          win[e] = v[u];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      unsigned long long bits[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bits[j] = __ballot(win[o1[j]] < win[o2[j]]);
      }
      if (lane < 4) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(out_desc + (size_t) slot * PRS_DESC_BYTES);
        d[lane]               = lane == 0 ? bits[0] : (lane == 1 ? bits[1] : (lane == 2 ? bits[2] : bits[3]));
      }
      if (lane == 4) {
        out_kp[slot] = prs_kp2{(float) c, (float) r};
        if (out_int) {
          out_int[slot] = (float) src[(size_t) r * a.b.pitch + c];  // intensity_feature_extractor_base.cpp:80
        }
      }
      __builtin_amdgcn_wave_barrier();  // the window is rewritten for the wave's next keypoint
    }
  }
  if (__syncthreads_or(overflow ? 1 : 0)) {
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

// 256 point pairs (x1, y1, x2, y2) in [-13, 13]: fixed linear congruential sequence, roughly bell shaped
void fill_brief_pattern(int8_t* pattern) {
  uint32_t x = 0x12345678u;
  int n      = 0;
  while (n < 256) {
    int v[4];
    for (int k = 0; k < 4; ++k) {
      x    = x * 1664525u + 1013904223u;
      v[k] = (int) ((x >> 8) % 9u) - 4 + (int) ((x >> 16) % 9u) - 4 + (int) ((x >> 24) % 11u) - 5;
    }
    if (v[0] == v[2] && v[1] == v[3]) {
      continue;
    }
    for (int k = 0; k < 4; ++k) {
      pattern[4 * n + k] = (int8_t) v[k];
    }
    ++n;
  }
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
  FeatureArgs a;
  a.p = *params;
  a.b = *batch;
  const size_t npix = (size_t) batch->rows * batch->cols;
  a.score = static_cast<uint8_t*>(ctx_device_scratch_slot(ctx, 0, (size_t) batch->batch * npix));
  a.box   = static_cast<uint16_t*>(ctx_device_scratch_slot(ctx, 1, (size_t) batch->batch * npix * 2));
  uint32_t* rawbuf = static_cast<uint32_t*>(ctx_device_scratch_slot(ctx, 2, (size_t) batch->batch * (kMaxRaw + 1) * 4));
  if (!a.score || !a.box || !rawbuf) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_extract_features_batch: scratch allocation failed");
  }
  a.raw   = rawbuf;
  a.n_raw = reinterpret_cast<int32_t*>(rawbuf + (size_t) batch->batch * kMaxRaw);
  fill_brief_pattern(a.pattern);
  a.rows_per   = (float) batch->rows / (float) params->number_of_detectors_vertical;   // binned.cpp:52-55
  a.cols_per   = (float) batch->cols / (float) params->number_of_detectors_horizontal;
  a.regions    = regions;
  a.target_per = (int) ((float) params->target_number_of_keypoints / (float) regions);  // :72-76
  hipStream_t stream = ctx_stream(ctx);
  const dim3 tiles((batch->cols + kTileW - 1) / kTileW, (batch->rows + kTileH - 1) / kTileH, batch->batch);
  hipLaunchKernelGGL(fast_box_kernel, tiles, dim3(kFastThreads), 0, stream, a);
  hipLaunchKernelGGL(nms_compact_kernel, dim3(batch->batch), dim3(kNmsThreads), 0, stream, a);
  hipLaunchKernelGGL(select_describe_kernel, dim3(batch->batch), dim3(kSelThreads), 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_extract_features_batch launch");
  }
  return PRS_OK;
}

}  // namespace prs
