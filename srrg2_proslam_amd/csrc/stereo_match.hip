// stereo_match.hip -- batched stereo epipolar descriptor matcher for gfx950 (MI355X): the general kernel and the
// dispatch (frames of <= 2048 keypoints are served by stereo_match_v5.hip, same phases with a leaner instruction stream).
//
// Replaces CorrespondenceFinderDescriptorBasedEpipolar<..>::compute
// (registration/correspondence_finders/correspondence_finder_descriptor_based_epipolar_impl.cpp:46-219)
// with an optional fused stereo-adaptor assembly + rectified triangulation epilogue
// (sensor_processing/raw_data_preprocessor_stereo_projective.cpp:107-132,
//  mapping/triangulator_rigid_stereo.cpp:7-85).
//
// One 1024-thread workgroup owns one stereo pair; a launch covers `batch` independent frames.
// Every HBM byte of a frame is read once, coalesced, by loads issued before any compute:
//   A. keypoint coordinates (8 B/lane) and descriptor rows (2 x 16 B/lane) go to registers;
//      thread t owns keypoints t, t+1024, ... of both images for the whole kernel.
//   B. (row, col) truncation, per-row histogram with LDS atomics, single-wave scan, bucket scatter
//      and in-row rank give the reference's (row, col, index)-sorted feature vectors
//      (epipolar_impl.cpp:26-42) without a comparison sort.
//   C. right descriptor rows are parked in LDS; left rows stay in their owner's registers.
//   D. candidate scoring is fully parallel: every left keypoint binary-searches its disparity
//      window in the sorted right row and scores up to four in-window candidates (popcount of
//      its register-resident row against LDS rows), ignoring the reference's moving cursor; the
//      distances are packed (9 bit each) with the window start into one 8-byte record.
//   E. one lane per epipolar row replays the reference's serial chain (index_right = best + 1,
//      epipolar_impl.cpp:181) on those records: masking candidates left of the cursor and taking
//      best / second best needs no descriptor access.  Windows with more than four candidates
//      (never seen on KITTI-shaped input) are re-scored in the chain.
//   F. matches are emitted in sorted-left traversal order via a per-row scan; the epilogue builds
//      the (uL,vL,uR,vR) fixed cloud, copies the left descriptor and triangulates.
// The output is bit-identical to the sequential algorithm (tests/test_stereo_match_gpu.py).
#include "prs_device.h"
#include "prs_host.h"

namespace prs {

struct StereoArgs {
  prs_stereo_params p;
  prs_stereo_batch b;
  prs_triangulator_params tri;
  int epilogue;
  int sort_cap;  // entries of each sorted / bucket array (>= stride, >= image_rows + 1)
  uint32_t off_desc_l, off_desc_r, off_sorted_l, off_sorted_r, off_bucket, off_rowstart_l, off_rowstart_r;
  uint32_t off_rowcnt, off_bits, off_misc, off_tab, off_kp_r;
  // exact integer form of `best < max_distance && best / second < max_ratio` (epipolar_impl.cpp:171-173),
  // evaluated on the host with the reference's float operations (fill_accept_table)
  int best_lim;        // accept iff best < best_lim ...
  int16_t bmax[258];   // ... and best <= bmax[second] (257 = no second candidate)
  unsigned long long* stamps;  // diagnostic: [batch][16] shader-clock stamps of thread 0 (NULL = off)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kStereoThreads = 1024;
constexpr uint32_t kNone16   = 0xffffu;

#define PRS_STAMP(i)                                                        \
  do {                                                                      \
    if (a.stamps && tid == 0) {                                             \
      a.stamps[(size_t) frame * 16 + (i)] = (unsigned long long) clock64(); \
    }                                                                       \
  } while (0)

__device__ __forceinline__ int hamming_u32x4(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
  int d = __popc(a0.x ^ b0.x);
  d += __popc(a0.y ^ b0.y);
  d += __popc(a0.z ^ b0.z);
  d += __popc(a0.w ^ b0.w);
  d += __popc(a1.x ^ b1.x);
  d += __popc(a1.y ^ b1.y);
  d += __popc(a1.z ^ b1.z);
  d += __popc(a1.w ^ b1.w);
  return d;
}

// exclusive scan of n counters by ONE wave (all 64 lanes of the calling wave take part).
// in[] holds counts, out[] receives the exclusive prefix (out may alias in).
// returns the total in every lane.  Up to 8 counters per lane are read with independent loads.
template <typename TIn, typename TOut>
__device__ __forceinline__ uint32_t wave_exclusive_scan(const TIn* in, TOut* out, int n) {
  const int lane  = threadIdx.x & 63;
  const int chunk = (n + 63) >> 6;
  const int start = lane * chunk;
  uint32_t sum    = 0;
  uint32_t v[8];
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = start + j;
      v[j]        = (j < chunk && r < n) ? (uint32_t) in[r] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sum += v[j];
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = start + j;
      sum += r < n ? (uint32_t) in[r] : 0u;
    }
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
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = start + j;
      if (j < chunk && r < n) {
        out[r] = (TOut) run;
        run += v[j];
      }
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = start + j;
      if (r < n) {
        const uint32_t h = (uint32_t) in[r];
        out[r]           = (TOut) run;
        run += h;
      }
    }
  }
  return __shfl(incl, 63, 64);
}

// candidate record written by the scoring phase, one per sorted-left position:
//   x = d0 | d1 << 9 | d2 << 18          (9-bit Hamming distances of candidates lo .. lo+3,
//   y = d3 | lo << 9 | n << 22             511 = candidate pruned by an earlier pass)
// n = number of in-window right features (0 none, 1..4 scored, 7 = more than four: re-score)
constexpr uint32_t kDistPruned  = 511u;
constexpr uint32_t kCntOverflow = 7u;

template <int KPT, bool STAGE>
__global__ __launch_bounds__(kStereoThreads) void stereo_match_kernel(const StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid    = threadIdx.x;
  const int stride = a.b.stride;
  const int rows   = a.p.image_rows;

  u32x4* ldR         = reinterpret_cast<u32x4*>(smem + a.off_desc_r);
  uint32_t* sortedL  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_l);
  uint32_t* sortedR  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_r);
  uint32_t* bucketL  = reinterpret_cast<uint32_t*>(smem + a.off_bucket);
  uint32_t* bucketR  = bucketL + a.sort_cap;
  uint2* res         = reinterpret_cast<uint2*>(smem + a.off_bucket);  // aliases both buckets after the sort
  uint16_t* rsL      = reinterpret_cast<uint16_t*>(smem + a.off_rowstart_l);
  uint16_t* rsR      = reinterpret_cast<uint16_t*>(smem + a.off_rowstart_r);
  uint16_t* rowcnt   = reinterpret_cast<uint16_t*>(smem + a.off_rowcnt);
  const int nwords   = (stride + 31) >> 5;
  uint32_t* bitsL    = reinterpret_cast<uint32_t*>(smem + a.off_bits);  // left sorted position matched in an earlier pass
  uint32_t* bitsR    = bitsL + nwords;                                  // right sorted position matched
  uint32_t* bitsK    = bitsR + nwords;                                  // epilogue: match index kept
  uint16_t* prefK    = reinterpret_cast<uint16_t*>(bitsK + nwords);     // epilogue: kept matches before word w
  int* misc          = reinterpret_cast<int*>(smem + a.off_misc);       // [0] error, [1] pass matches, [2] pass kept
  int16_t* tab       = reinterpret_cast<int16_t*>(smem + a.off_tab);    // Lowe acceptance table
  prs_kp2* ldKR      = reinterpret_cast<prs_kp2*>(smem + a.off_kp_r);   // staged variant: right coordinates for the epilogue
  uint32_t* histL    = sortedL;  // the histograms die before the sorted arrays are born
  uint32_t* histR    = sortedR;

  // The staged variant keeps ONE workgroup per CU, so nothing would cover the latency of a frame's
  // first loads.  It therefore runs persistently (grid = CUs, frames strided over the workgroups):
  // the coordinates of the next frame (the first thing a frame needs) are requested while this frame
  // is scored, chained and emitted; the descriptor rows of a frame are requested when it starts and
  // arrive while its coordinates are sorted (they are first touched by the staging / scoring phases).
  constexpr bool PERSIST = STAGE;
  prs_kp2 cLn[KPT], cRn[KPT];
  int nLn = 0, nRn = 0;
  auto fetch_coords = [&](int f) {
    int fl = a.b.n_left[f];
    int fr = a.b.n_right[f];
    fl     = fl < 0 ? 0 : (fl > stride ? stride : fl);
    fr     = fr < 0 ? 0 : (fr > stride ? stride : fr);
    nLn    = fl;
    nRn    = fr;
    const size_t fbase = (size_t) f * (size_t) stride;
    const prs_kp2* __restrict__ fkL = a.b.left_kp + fbase;
    const prs_kp2* __restrict__ fkR = a.b.right_kp + fbase;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kStereoThreads + tid;
      cLn[k]      = fkL[i < fl ? i : (fl > 0 ? fl - 1 : 0)];
      cRn[k]      = fkR[i < fr ? i : (fr > 0 ? fr - 1 : 0)];
    }
  };
  for (int i = tid; i < 258; i += kStereoThreads) {
    tab[i] = a.bmax[i];  // published by the first barrier of the frame
  }
  if (PERSIST && (int) blockIdx.x < a.b.batch) {
    fetch_coords((int) blockIdx.x);
  }
  for (int frame = blockIdx.x; frame < a.b.batch; frame += gridDim.x) {
  PRS_STAMP(0);
  if (!PERSIST) {
    fetch_coords(frame);
  }
  const int nL      = nLn;
  const int nR      = nRn;
  const size_t base = (size_t) frame * (size_t) stride;
  const prs_kp2* __restrict__ kpR = a.b.right_kp + base;
  const u32x4* __restrict__ gdL   = reinterpret_cast<const u32x4*>(a.b.left_desc + base * PRS_DESC_BYTES);
  const u32x4* __restrict__ gdR   = reinterpret_cast<const u32x4*>(a.b.right_desc + base * PRS_DESC_BYTES);
  prs_kp2 cL[KPT], cR[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    cL[k] = cLn[k];
    cR[k] = cRn[k];
  }
  // ---- A: issue every descriptor read of this frame up front -------------------------------------
  // thread t owns descriptor rows t, t+1024, ..: tail lanes re-read the last valid row (same
  // cache line, no extra HBM traffic) so the loads stay unconditional and in registers
  u32x4 dL[STAGE ? 2 * KPT : 2], dR[STAGE ? 2 * KPT : 2];
  if (STAGE) {
    const int lastL = nL > 0 ? nL - 1 : 0;
    const int lastR = nR > 0 ? nR - 1 : 0;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i  = k * kStereoThreads + tid;
      const int il = i < lastL ? i : lastL;
      const int ir = i < lastR ? i : lastR;
      dL[2 * k]     = gdL[2 * il];
      dL[2 * k + 1] = gdL[2 * il + 1];
      dR[2 * k]     = gdR[2 * ir];
      dR[2 * k + 1] = gdR[2 * ir + 1];
    }
  }
  const int next_frame = frame + (int) gridDim.x;

  // ---- B: Feature{row,col,unsorted_index} + counting sort by row (epipolar_impl.cpp:8-42) ----
  for (int i = tid; i <= rows; i += kStereoThreads) {
    histL[i] = 0;
    histR[i] = 0;
  }
  for (int i = tid; i < 3 * nwords; i += kStereoThreads) {
    bitsL[i] = 0;
  }
  if (tid < 4) {
    misc[tid] = 0;
  }
  __syncthreads();
  PRS_STAMP(1);

  int rowL[KPT], rowR[KPT], posL[KPT];
  uint32_t keyL[KPT], keyR[KPT], slotL[KPT], slotR[KPT];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = k * kStereoThreads + tid;
    rowL[k]     = -1;
    rowR[k]     = -1;
    posL[k]     = 0;
    keyL[k] = keyR[k] = slotL[k] = slotR[k] = 0;
    if (i < nL) {
      const float u = cL[k].u, v = cL[k].v;
      if (u >= 0.0f && u < 32768.0f && v >= 0.0f && v < (float) rows) {
        rowL[k]  = (int) v;  // truncation, epipolar_impl.cpp:10
        keyL[k]  = ((uint32_t) (int) u << 16) | (uint32_t) i;  // (col, unsorted index)
        slotL[k] = atomicAdd(&histL[rowL[k]], 1u);
      } else {
        bad = true;
      }
    }
    if (i < nR) {
      const float u = cR[k].u, v = cR[k].v;
      if (u >= 0.0f && u < 32768.0f && v >= 0.0f && v < (float) rows) {
        rowR[k]  = (int) v;
        keyR[k]  = ((uint32_t) (int) u << 16) | (uint32_t) i;
        slotR[k] = atomicAdd(&histR[rowR[k]], 1u);
      } else {
        bad = true;
      }
    }
  }
  if (bad) {
    misc[0] = 1;
  }
  __syncthreads();
  if (misc[0]) {  // outside the supported domain: loud per-frame error, no partial output
    if (tid == 0) {
      a.b.n_matches[frame] = 0;
      a.b.status[frame]    = PRS_ERR_RANGE;
      if (a.epilogue) {
        a.b.n_fixed[frame] = 0;
      }
    }
    if (PERSIST && next_frame < a.b.batch) {
      fetch_coords(next_frame);
    }
    __syncthreads();  // misc[] is rewritten by the next frame
    continue;
  }
  PRS_STAMP(2);
  if (tid < 64) {
    wave_exclusive_scan(histL, rsL, rows + 1);
  } else if (tid < 128) {
    wave_exclusive_scan(histR, rsR, rows + 1);
  }
  __syncthreads();
  PRS_STAMP(3);

  // scatter into row buckets (arbitrary order inside a row) ...
  int sL[KPT], sR[KPT], lenL[KPT], lenR[KPT];
  int maxlen = 0;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    sL[k] = sR[k] = lenL[k] = lenR[k] = 0;
    if (rowL[k] >= 0) {
      sL[k]   = rsL[rowL[k]];
      lenL[k] = rsL[rowL[k] + 1] - sL[k];
      bucketL[sL[k] + slotL[k]] = keyL[k];
      maxlen  = lenL[k] > maxlen ? lenL[k] : maxlen;
    }
    if (rowR[k] >= 0) {
      sR[k]   = rsR[rowR[k]];
      lenR[k] = rsR[rowR[k] + 1] - sR[k];
      bucketR[sR[k] + slotR[k]] = keyR[k];
      maxlen  = lenR[k] > maxlen ? lenR[k] : maxlen;
    }
  }
  __syncthreads();
  // ... then rank inside the row by (col, unsorted index): epipolar_impl.cpp:36-41 + canonical
  // tie-break.  The 2*KPT rank loops run interleaved so their LDS reads overlap.
  {
    int rankL[KPT], rankR[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      rankL[k] = rankR[k] = 0;
    }
    // the first 8 bucket entries are read with immediate offsets (the arrays are padded by 8
    // entries, reads past the row's end are masked); longer rows take the tail loop
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const uint32_t bl = bucketL[sL[k] + j];
        const uint32_t br = bucketR[sR[k] + j];
        rankL[k] += (j < lenL[k] && bl < keyL[k]) ? 1 : 0;
        rankR[k] += (j < lenR[k] && br < keyR[k]) ? 1 : 0;
      }
    }
    for (int j = 8; j < maxlen; ++j) {
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const uint32_t bl = bucketL[sL[k] + (j < lenL[k] ? j : 0)];
        const uint32_t br = bucketR[sR[k] + (j < lenR[k] ? j : 0)];
        rankL[k] += (j < lenL[k] && bl < keyL[k]) ? 1 : 0;
        rankR[k] += (j < lenR[k] && br < keyR[k]) ? 1 : 0;
      }
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      if (rowL[k] >= 0) {
        posL[k]          = sL[k] + rankL[k];
        sortedL[posL[k]] = keyL[k];
      }
      if (rowR[k] >= 0) {
        sortedR[sR[k] + rankR[k]] = keyR[k];
      }
    }
  }
  PRS_STAMP(4);
  // ---- C: descriptor rows land in LDS ---------------------------------------------------------
  if (STAGE) {
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kStereoThreads + tid;
      if (i < nR) {
        ldR[2 * i]     = dR[2 * k];
        ldR[2 * i + 1] = dR[2 * k + 1];
        ldKR[i]        = cR[k];
      }
    }
  }
  __syncthreads();  // buckets are dead from here on: res[] may be written
  PRS_STAMP(5);
  if (PERSIST && next_frame < a.b.batch) {
    fetch_coords(next_frame);  // in flight while this frame is scored, chained and emitted
  }

  const int best_lim    = a.best_lim;
  const int max_disp    = a.p.maximum_disparity_pixels;
  const int thickness   = a.p.epipolar_line_thickness_pixels > 0 ? a.p.epipolar_line_thickness_pixels : 0;
  const int n_offsets   = 1 + 2 * thickness;
  const bool multipass  = n_offsets > 1;
  prs_corr* __restrict__ out = a.b.matches + base;
  int out_base               = 0;
  int fixed_base             = 0;

  for (int o = 0; o < n_offsets; ++o) {
    const int off = o == 0 ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2));  // 0,+1,-1,+2,-2 (epipolar_impl.cpp:71-79)

    // ---- D: every left keypoint scores its in-window candidates (cursor ignored) -------------
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      if (rowL[k] < 0) {
        continue;
      }
      const int p  = posL[k];
      uint2 r      = make_uint2(0u, 0u);
      const int rr = rowL[k] + off;
      const bool pruned = multipass && ((bitsL[p >> 5] >> (p & 31)) & 1u);
      if (rr >= 0 && rr < rows && !pruned) {
        const int col_l = (int) (keyL[k] >> 16);
        const int rs    = rsR[rr];
        const int re    = rsR[rr + 1];
        // in-window right features are contiguous in the sorted row:
        //   lo = first q with col_r >= col_l - max_disp   (epipolar_impl.cpp:146-149)
        //   hi = first q with col_r >  col_l              (epipolar_impl.cpp:141-143)
        int lo_a = rs, lo_b = re, hi_a = rs, hi_b = re;
        const int col_min = col_l - max_disp;
        while (lo_a < lo_b || hi_a < hi_b) {
          const int ml = (lo_a + lo_b) >> 1, mh = (hi_a + hi_b) >> 1;
          const int cl = (int) (sortedR[lo_a < lo_b ? ml : rs] >> 16);
          const int ch = (int) (sortedR[hi_a < hi_b ? mh : rs] >> 16);
          if (lo_a < lo_b) {
            if (cl < col_min) {
              lo_a = ml + 1;
            } else {
              lo_b = ml;
            }
          }
          if (hi_a < hi_b) {
            if (ch <= col_l) {
              hi_a = mh + 1;
            } else {
              hi_b = mh;
            }
          }
        }
        const int lo = lo_a;
        const int n  = hi_a - lo_a;
        if (n > 4) {
          r = make_uint2(0u, ((uint32_t) lo << 9) | (kCntOverflow << 22));
        } else if (n > 0) {
          u32x4 d0, d1;
          if (STAGE) {
            d0 = dL[2 * k];
            d1 = dL[2 * k + 1];
          } else {
            const int i = k * kStereoThreads + tid;
            d0          = gdL[2 * i];
            d1          = gdL[2 * i + 1];
          }
          uint32_t dist[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dist[j] = 0;
            if (j < n) {
              const int q = lo + j;
              if (multipass && ((bitsR[q >> 5] >> (q & 31)) & 1u)) {
                dist[j] = kDistPruned;  // pruned (epipolar_impl.cpp:197-205)
              } else {
                const int idx_r = (int) (sortedR[q] & 0xffffu);
                u32x4 e0, e1;
                if (STAGE) {
                  e0 = ldR[2 * idx_r];
                  e1 = ldR[2 * idx_r + 1];
                } else {
                  e0 = gdR[2 * idx_r];
                  e1 = gdR[2 * idx_r + 1];
                }
                dist[j] = (uint32_t) hamming_u32x4(d0, d1, e0, e1);
              }
            }
          }
          r = make_uint2(dist[0] | (dist[1] << 9) | (dist[2] << 18), dist[3] | ((uint32_t) lo << 9) | ((uint32_t) n << 22));
        }
      }
      res[p] = r;
    }
    if (a.epilogue) {
      for (int i = tid; i < nwords; i += kStereoThreads) {
        bitsK[i] = 0;
      }
    }
    __syncthreads();
    PRS_STAMP(6);

    // ---- E: one lane per epipolar row replays the serial cursor chain ------------------------
    for (int r = tid; r < rows; r += kStereoThreads) {
      const int rr = r + off;
      const int ls = rsL[r], le = rsL[r + 1];
      int cnt      = 0;
      if (rr >= 0 && rr < rows && ls < le) {
        int c        = rsR[rr];
        const int re = rsR[rr + 1];
        for (int p = ls; p < le; ++p) {
          const uint2 rec  = res[p];
          uint2 outrec     = make_uint2(0u, 0u);
          const uint32_t n = (rec.y >> 22) & 7u;
          if (n != 0u && c < re) {
            const int lo    = (int) ((rec.y >> 9) & 0x1fffu);
            uint32_t best = kNone16, second = kNone16, best_q = 0;
            if (n == kCntOverflow) {
              // more than four in-window candidates: score them here, from the cursor on
              const uint32_t kl = sortedL[p];
              const int col_l   = (int) (kl >> 16);
              const int idx_l   = (int) (kl & 0xffffu);
              const u32x4 d0 = gdL[2 * idx_l], d1 = gdL[2 * idx_l + 1];
              for (int q = c > lo ? c : lo; q < re; ++q) {
                const uint32_t kr = sortedR[q];
                if (col_l - (int) (kr >> 16) < 0) {
                  break;  // epipolar_impl.cpp:141-143
                }
                if (multipass && ((bitsR[q >> 5] >> (q & 31)) & 1u)) {
                  continue;
                }
                const int idx_r = (int) (kr & 0xffffu);
                u32x4 e0, e1;
                if (STAGE) {
                  e0 = ldR[2 * idx_r];
                  e1 = ldR[2 * idx_r + 1];
                } else {
                  e0 = gdR[2 * idx_r];
                  e1 = gdR[2 * idx_r + 1];
                }
                const uint32_t d = (uint32_t) hamming_u32x4(d0, d1, e0, e1);
                if (d < best) {  // epipolar_impl.cpp:158-164
                  second = best;
                  best   = d;
                  best_q = (uint32_t) q;
                } else if (d < second) {
                  second = d;
                }
              }
            } else {
              const uint32_t dist[4] = {rec.x & 511u, (rec.x >> 9) & 511u, (rec.x >> 18) & 511u, rec.y & 511u};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                // candidates left of the cursor were consumed by an earlier match (epipolar_impl.cpp:181)
                if ((uint32_t) j < n && lo + j >= c && dist[j] != kDistPruned) {
                  if (dist[j] < best) {  // epipolar_impl.cpp:158-164 (kNone16 > any distance)
                    second = best;
                    best   = dist[j];
                    best_q = (uint32_t) (lo + j);
                  } else if (dist[j] < second) {
                    second = dist[j];
                  }
                }
              }
            }
            if (best != kNone16) {
              // epipolar_impl.cpp:171-173 through the host-evaluated table (no float division in the chain)
              if ((int) best < best_lim && (int) best <= (int) tab[second == kNone16 ? 257u : second]) {
                outrec = make_uint2((sortedR[best_q] & 0xffffu) | (best << 16), (uint32_t) cnt | ((uint32_t) (o + 1) << 16));
                ++cnt;
                c = (int) best_q + 1;  // epipolar_impl.cpp:181
                if (multipass) {
                  atomicOr(&bitsL[p >> 5], 1u << (p & 31));
                  atomicOr(&bitsR[best_q >> 5], 1u << (best_q & 31));
                }
              }
            }
          }
          res[p] = outrec;
        }
      } else {
        for (int p = ls; p < le; ++p) {
          res[p] = make_uint2(0u, 0u);
        }
      }
      rowcnt[r] = (uint16_t) cnt;
    }
    __syncthreads();
    PRS_STAMP(7);

    // ---- F: emit in sorted-left traversal order ----------------------------------------------
    if (tid < 64) {
      const uint32_t total = wave_exclusive_scan(rowcnt, rowcnt, rows);
      if (tid == 0) {
        misc[1] = (int) total;
      }
    }
    __syncthreads();
    const int pass_matches = misc[1];
    int m_out[KPT];
    uint32_t m_idx_r[KPT];
    bool m_keep[KPT];
    prs_kp2 m_kr[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      m_out[k]   = -1;
      m_keep[k]  = false;
      m_idx_r[k] = 0;
      m_kr[k]    = prs_kp2{0.f, 0.f};
      if (rowL[k] >= 0) {
        const uint2 rec = res[posL[k]];
        if ((rec.y >> 16) == (uint32_t) (o + 1)) {
          m_out[k]   = (int) rowcnt[rowL[k]] + (int) (rec.y & 0xffffu);  // index inside this pass
          m_idx_r[k] = rec.x & 0xffffu;
          prs_corr cr;
          cr.fixed_idx  = k * kStereoThreads + tid;
          cr.moving_idx = (int) m_idx_r[k];
          cr.response   = (float) (rec.x >> 16);
          out[out_base + m_out[k]] = cr;
          if (a.epilogue) {
            m_kr[k] = STAGE ? ldKR[m_idx_r[k]] : kpR[m_idx_r[k]];
            // raw_data_preprocessor_stereo_projective.cpp:117-125
            const float hd = cL[k].u - m_kr[k].u, vd = cL[k].v - m_kr[k].v;
            m_keep[k] = !(hd < 0.0f || vd < 0.0f);
            if (m_keep[k]) {
              atomicOr(&bitsK[m_out[k] >> 5], 1u << (m_out[k] & 31));
            }
          }
        }
      }
    }
    if (a.epilogue) {
      __syncthreads();
      if (tid < 64) {
        // kept matches before each 32-match word (<= 8192 / 32 = 256 words)
        const int chunk = (nwords + 63) >> 6;
        uint32_t sum    = 0;
        for (int j = 0; j < chunk; ++j) {
          const int w = tid * chunk + j;
          sum += w < nwords ? (uint32_t) __popc(bitsK[w]) : 0u;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(incl, d, 64);
          if (tid >= d) {
            incl += t;
          }
        }
        uint32_t run = incl - sum;
        for (int j = 0; j < chunk; ++j) {
          const int w = tid * chunk + j;
          if (w < nwords) {
            prefK[w] = (uint16_t) run;
            run += (uint32_t) __popc(bitsK[w]);
          }
        }
        if (tid == 63) {
          misc[2] = (int) incl;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        if (m_keep[k]) {
          const int w    = m_out[k] >> 5;
          const int slot = fixed_base + (int) prefK[w] + __popc(bitsK[w] & ((1u << (m_out[k] & 31)) - 1u));
          const size_t g = base + (size_t) slot;
          const float x_L = cL[k].u, y_L = cL[k].v, x_R = m_kr[k].u, y_R = m_kr[k].v;
          reinterpret_cast<float4*>(a.b.fixed_uvuv)[g] = make_float4(x_L, y_L, x_R, y_R);
          u32x4* fd = reinterpret_cast<u32x4*>(a.b.fixed_desc) + 2 * g;
          if (STAGE) {
            fd[0] = dL[2 * k];
            fd[1] = dL[2 * k + 1];
          } else {
            const int i = k * kStereoThreads + tid;
            fd[0]       = gdL[2 * i];
            fd[1]       = gdL[2 * i + 1];
          }
          // triangulator_rigid_stereo.cpp:39-45,60-85 (operation order kept)
          float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!(x_L - x_R < a.tri.minimum_disparity_pixels)) {
            float depth = a.tri.infinity_depth_meters;
            if (x_L > x_R) {
              depth = a.tri.b_x / (x_L - x_R);
            }
            pt.z = depth;
            pt.x = 1 / a.tri.fx * (x_L - a.tri.cx) * depth;
            pt.y = 1 / a.tri.fy * ((y_L + y_R) / 2 - a.tri.cy) * depth;
            pt.w = 1.0f;
          }
          reinterpret_cast<float4*>(a.b.fixed_xyz)[g] = pt;
        }
      }
      fixed_base += misc[2];
    }
    out_base += pass_matches;
    if (o + 1 < n_offsets) {
      __syncthreads();  // res[], rowcnt[], misc[] are rewritten by the next pass
    }
  }

  PRS_STAMP(8);
  if (tid == 0) {
    int flags = PRS_OK;
    if (nL == 0 || nR == 0) {
      flags |= PRS_WARN_EMPTY_INPUT;  // bruteforce_impl.cpp:217-226
    }
    if (out_base == 0) {
      flags |= PRS_WARN_NO_MATCHES;  // bruteforce_impl.cpp:237-242
    }
    const float matching_ratio = (float) out_base / (float) nL;  // epipolar_impl.cpp:209-210
    if (matching_ratio < a.p.minimum_matching_ratio) {
      flags |= PRS_WARN_LOW_RATIO;
    }
    a.b.n_matches[frame] = out_base;
    a.b.status[frame]    = flags;
    if (a.epilogue) {
      a.b.n_fixed[frame] = fixed_base;
    }
  }
  if (PERSIST) {
    __syncthreads();  // the LDS arrays are rewritten by the next frame
  }
  }  // frames of this workgroup
}

// TriangulatorRigidStereo::compute on a flat device array (mapping/triangulator_rigid_stereo.cpp:7-85)
__global__ __launch_bounds__(256) void triangulate_kernel(const prs_triangulator_params t,
                                                          const float4* __restrict__ uvuv,
                                                          int64_t n,
                                                          float4* __restrict__ xyz4) {
  for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
    const float4 m  = uvuv[i];
    const float x_L = m.x, y_L = m.y, x_R = m.z, y_R = m.w;
    float4 pt       = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(x_L - x_R < t.minimum_disparity_pixels)) {
      float depth = t.infinity_depth_meters;
      if (x_L > x_R) {
        depth = t.b_x / (x_L - x_R);
      }
      pt.z = depth;
      pt.x = 1 / t.fx * (x_L - t.cx) * depth;
      pt.y = 1 / t.fy * ((y_L + y_R) / 2 - t.cy) * depth;
      pt.w = 1.0f;
    }
    xyz4[i] = pt;
  }
}

static inline uint32_t align_up(uint32_t v, uint32_t a) {
  return (v + a - 1) / a * a;
}

template <int KPT, bool STAGE>
static hipError_t launch_variant(const StereoArgs& a, size_t lds, hipStream_t stream) {
  auto kernel = stereo_match_kernel<KPT, STAGE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  if (e != hipSuccess) {
    return e;
  }
  int grid = a.b.batch;
  if (STAGE) {
    // persistent: one workgroup per CU (the LDS footprint allows no more), frames strided over them
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
        cus < grid) {
      grid = cus;
    }
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kStereoThreads), lds, stream, a);
  return hipGetLastError();
}

void fill_accept_table(const prs_stereo_params* params, int* best_lim, int16_t* bmax) {
  const float max_dist = params->maximum_descriptor_distance, ratio = params->maximum_distance_ratio_to_second_best;
  int lim = 0;
  while (lim <= 256 && (float) lim < max_dist) {
    ++lim;
  }
  *best_lim = lim;
  for (int s = 0; s <= 257; ++s) {
    const float fs = s == 257 ? 3.402823466e+38f : (float) s;
    int bm         = -1;
    for (int b = 0; b <= 256; ++b) {
      if ((float) b / fs < ratio) {
        bm = b;  // monotone in b for fs > 0; for fs == 0 the quotient is NaN or +inf: never accepted
      } else if (s != 0) {
        break;
      }
    }
    bmax[s] = (int16_t) (s == 0 ? -1 : bm);
  }
}

int stereo_match_batch_launch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch) {
  if (!params || !batch || !batch->left_kp || !batch->left_desc || !batch->n_left || !batch->right_kp ||
      !batch->right_desc || !batch->n_right || !batch->matches || !batch->n_matches || !batch->status) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_stereo_match_batch: fixed, moving or correspondences not set");
  }
  if (batch->batch <= 0) {
    return PRS_OK;
  }
  const int stride = batch->stride;
  if (stride <= 0 || stride > 8192 || params->image_rows <= 0 || params->image_rows > 4096) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_stereo_match_batch: stride must be in [1,8192], image_rows in [1,4096]");
  }
  if (params->epipolar_line_thickness_pixels > 120) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_stereo_match_batch: epipolar_line_thickness_pixels > 120");
  }
  {
    // instruction-lean staged kernel for frames of <= 2048 keypoints (stereo_match_v5.hip)
    const int rc5 = stereo_match_v5_launch(ctx, params, batch);
    if (rc5 != 1) {
      return rc5;
    }
  }
  StereoArgs a;
  a.p        = *params;
  a.b        = *batch;
  a.epilogue = (batch->fixed_uvuv && batch->fixed_desc && batch->n_fixed && batch->fixed_xyz && batch->triangulator) ? 1 : 0;
  if (a.epilogue) {
    a.tri = *batch->triangulator;
  } else {
    a.tri = prs_triangulator_params{1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
  fill_accept_table(params, &a.best_lim, a.bmax);
  a.stamps              = ctx_stamps(ctx, (size_t) batch->batch * 16 * sizeof(unsigned long long));
  const int kpt         = stride <= 1024 ? 1 : (stride <= 2048 ? 2 : (stride <= 4096 ? 4 : 8));
  const uint32_t rows1  = (uint32_t) params->image_rows + 1;
  a.sort_cap            = (int) (((uint32_t) stride > rows1 ? (uint32_t) stride : rows1));
  const uint32_t nwords = ((uint32_t) stride + 31) / 32;
  // LDS carve; every offset is a multiple of 16 (cdna_hip_programming.md Guideline 17)
  auto carve = [&](bool stage, StereoArgs& s) -> size_t {
    uint32_t off          = 0;
    const uint32_t dbytes = stage ? (uint32_t) stride * PRS_DESC_BYTES : 0;
    s.off_desc_l     = off;  // left rows are not staged
    s.off_desc_r     = off; off = align_up(off + dbytes, 16);
    s.off_sorted_l   = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_sorted_r   = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_bucket     = off; off = align_up(off + (uint32_t) s.sort_cap * 8 + 32, 16);  // bucketL | bucketR (+8 pad), later res[]
    s.off_rowstart_l = off; off = align_up(off + (rows1 + 1) * 2, 16);
    s.off_rowstart_r = off; off = align_up(off + (rows1 + 1) * 2, 16);
    s.off_rowcnt     = off; off = align_up(off + (rows1 + 1) * 2, 16);
    s.off_bits       = off; off = align_up(off + nwords * (3 * 4 + 2), 16);
    s.off_misc       = off; off = align_up(off + 16, 16);
    s.off_tab        = off; off = align_up(off + 258 * 2, 16);
    s.off_kp_r       = off; off = align_up(off + (stage ? (uint32_t) stride * 8 : 0), 16);
    return off;
  };
  const size_t lds_limit = 160 * 1024;
  bool stage             = true;
  size_t lds             = carve(true, a);
  if (lds > lds_limit || kpt > 2 || ctx_force_unstaged(ctx)) {
    stage = false;
    lds   = carve(false, a);
  }
  if (lds > lds_limit) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_stereo_match_batch: frame does not fit the 160 KiB LDS");
  }
  hipStream_t stream = ctx_stream(ctx);
  hipError_t e       = hipSuccess;
  // descriptor staging only exists for frames that fit the LDS (stride <= 2048)
  if (stage && kpt == 1) {
    e = launch_variant<1, true>(a, lds, stream);
  } else if (stage && kpt == 2) {
    e = launch_variant<2, true>(a, lds, stream);
  } else if (kpt == 1) {
    e = launch_variant<1, false>(a, lds, stream);
  } else if (kpt == 2) {
    e = launch_variant<2, false>(a, lds, stream);
  } else if (kpt == 4) {
    e = launch_variant<4, false>(a, lds, stream);
  } else {
    e = launch_variant<8, false>(a, lds, stream);
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_stereo_match_batch launch");
  }
  if (a.stamps) {
    ctx_report_stamps(ctx, batch->batch, 9,
                      "stereo_match: issue+zero | coords+hist | (err check) | scan | scatter+rank | stage-write | score | chain | emit");
  }
  return PRS_OK;
}

int triangulate_launch(prs_context* ctx, const prs_triangulator_params* params, const float* d_uvuv, int64_t n, float* d_xyz4) {
  if (!params || (n > 0 && (!d_uvuv || !d_xyz4))) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_triangulate: input or result buffer not set");
  }
  if (n <= 0) {
    return PRS_OK;
  }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) {
    blocks = 2048;
  }
  hipLaunchKernelGGL(triangulate_kernel, dim3((unsigned) blocks), dim3(256), 0, ctx_stream(ctx), *params,
                     reinterpret_cast<const float4*>(d_uvuv), n, reinterpret_cast<float4*>(d_xyz4));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_triangulate launch");
  }
  return PRS_OK;
}

} // namespace prs
