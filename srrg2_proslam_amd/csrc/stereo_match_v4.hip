// stereo_match_v4.hip -- second-generation batched stereo epipolar matcher (gfx950), selected when the
// caller states the image width (prs_stereo_params.image_cols > 0) and stride <= 2048.
//
// Same contract and bit-identical output as stereo_match.hip (which stays as the general fallback);
// the restructuring removes instructions, which is what bounds the first generation:
//   * (row, 128-px column block) bins replace the per-row buckets: a block scan over <= 8192 bins gives
//     the (row, col, index) order directly, the in-bin rank loop sees ~1 entry instead of ~5 per row,
//     and a left keypoint's disparity window is the union of <= 3 adjacent right bins;
//   * no LDS descriptor staging: a left row stays in its owner's registers, right rows are gathered
//     (32 B) when scored -> 50 KB of LDS per frame, 512 threads, three frames resident per CU whose
//     load and compute phases overlap;
//   * keypoints whose disparity window cannot be touched by the reference's cursor ("heads": no right
//     candidate is shared with the previous left keypoint of the row) are accepted/rejected by their
//     own thread; the one-lane-per-row replay of `index_right = best + 1`
//     (epipolar_impl.cpp:181) only scores the followers;
//   * Lowe's ratio + distance threshold are evaluated through a host-built integer table that is
//     exactly equivalent to `best < max_distance && best / second < max_ratio` (epipolar_impl.cpp:171-173);
//   * output slots come from popcount prefixes over match bitsets instead of per-row scans.
#include <stdio.h>
#include <string.h>

#include "prs_device.h"
#include "prs_host.h"

namespace prs {

constexpr int kV4Threads   = 512;
constexpr uint32_t kNone9  = 511u;
constexpr int kMaxBins     = 8192;

struct StereoV4Args {
  prs_stereo_params p;
  prs_stereo_batch b;
  prs_triangulator_params tri;
  int epilogue;
  int cbs;      // log2 of the column-block width
  int ncb;      // column blocks per row
  int nb;       // bins = image_rows * ncb
  int pair_cap; // entries of the dense (left, right) pair list
  int best_lim; // accept iff best < best_lim ...
  int16_t bmax[258];  // ... and best <= bmax[second] (257 = no second candidate)
  uint32_t off_bins_l, off_bins_r, off_sorted_l, off_sorted_r, off_res, off_pairs, off_runs, off_bits, off_tab, off_misc;
  unsigned long long* stamps;
};

typedef unsigned int v4u32x4 __attribute__((ext_vector_type(4)));

#define PRS_V4_STAMP(i)                                                     \
  do {                                                                      \
    if (a.stamps && tid == 0) {                                             \
      a.stamps[(size_t) frame * 16 + (i)] = (unsigned long long) clock64(); \
    }                                                                       \
  } while (0)

__device__ __forceinline__ uint32_t v4_hamming(const v4u32x4& a0, const v4u32x4& a1, const v4u32x4& b0, const v4u32x4& b1) {
  uint32_t d = __popc(a0.x ^ b0.x);
  d += __popc(a0.y ^ b0.y);
  d += __popc(a0.z ^ b0.z);
  d += __popc(a0.w ^ b0.w);
  d += __popc(a1.x ^ b1.x);
  d += __popc(a1.y ^ b1.y);
  d += __popc(a1.z ^ b1.z);
  d += __popc(a1.w ^ b1.w);
  return d;
}

// candidate record (one per sorted-left position), written by the scoring phase:
//   x = d0 | d1 << 9 | d2 << 18 | n << 27        n: in-window right keypoints (1..4, 7 = more than four)
//   y = d3 | lo << 9                              lo: sorted-right position of the first one
// final record, written by whoever decides the keypoint (its own thread for heads, the row lane else):
//   x = right unsorted index | distance << 16 | kFinal
//   y = best_q | pass_tag << 16                   pass_tag = pass + 1 when matched, 0 when not
constexpr uint32_t kFinal = 1u << 31;  // in x
constexpr uint32_t kHead  = 1u << 30;  // in x: the reference's cursor cannot cut this keypoint's window

template <int KPT>
__global__ __launch_bounds__(kV4Threads, 4) void stereo_match_v4_kernel(const StereoV4Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid    = threadIdx.x;
  const int lane   = tid & 63;
  const int wave   = tid >> 6;
  const int frame  = blockIdx.x;
  const int stride = a.b.stride;
  const int rows   = a.p.image_rows;
  const int cols   = a.p.image_cols;
  const int ncb    = a.ncb;
  const int cbs    = a.cbs;
  const int nb     = a.nb;
  int nL           = a.b.n_left[frame];
  int nR           = a.b.n_right[frame];
  nL               = nL < 0 ? 0 : (nL > stride ? stride : nL);
  nR               = nR < 0 ? 0 : (nR > stride ? stride : nR);
  const size_t base = (size_t) frame * (size_t) stride;
  const prs_kp2* __restrict__ kpL = a.b.left_kp + base;
  const prs_kp2* __restrict__ kpR = a.b.right_kp + base;
  const v4u32x4* __restrict__ gdL = reinterpret_cast<const v4u32x4*>(a.b.left_desc + base * PRS_DESC_BYTES);
  const v4u32x4* __restrict__ gdR = reinterpret_cast<const v4u32x4*>(a.b.right_desc + base * PRS_DESC_BYTES);

  uint16_t* binsL    = reinterpret_cast<uint16_t*>(smem + a.off_bins_l);  // counters, then bin starts (nb + 1)
  uint16_t* binsR    = reinterpret_cast<uint16_t*>(smem + a.off_bins_r);
  uint32_t* sortedL  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_l);  // (col << 16) | unsorted index
  uint32_t* sortedR  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_r);
  uint32_t* bucketL  = reinterpret_cast<uint32_t*>(smem + a.off_res);
  uint32_t* bucketR  = bucketL + stride + 8;
  uint2* res         = reinterpret_cast<uint2*>(smem + a.off_res);  // aliases the buckets after the sort
  uint32_t* pairs    = reinterpret_cast<uint32_t*>(smem + a.off_pairs);  // p | j << 13 | q << 15
  uint32_t* runs     = reinterpret_cast<uint32_t*>(smem + a.off_runs);   // p | row << 13
  const int nwords   = (stride + 31) >> 5;
  uint32_t* bitsM    = reinterpret_cast<uint32_t*>(smem + a.off_bits);  // sorted-left position matched in this pass
  uint32_t* bitsK    = bitsM + nwords;                                  // match index kept by the adaptor
  uint32_t* bitsPL   = bitsK + nwords;                                  // pruned left positions (multi-pass)
  uint32_t* bitsPR   = bitsPL + nwords;                                 // pruned right positions
  uint16_t* prefM    = reinterpret_cast<uint16_t*>(bitsPR + nwords);
  uint16_t* prefK    = prefM + nwords + 2;
  int16_t* tab       = reinterpret_cast<int16_t*>(smem + a.off_tab);
  int* misc          = reinterpret_cast<int*>(smem + a.off_misc);  // [0] error [1] pass matches [2] pass kept, [8..] wave totals

  PRS_V4_STAMP(0);
  // ---- A: coalesced loads of this thread's keypoints and its left descriptor rows ---------------
  prs_kp2 cL[KPT], cR[KPT];
  {
    const int lastL = nL > 0 ? nL - 1 : 0;
    const int lastR = nR > 0 ? nR - 1 : 0;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kV4Threads + tid;
      cL[k]       = kpL[i < lastL ? i : lastL];
      cR[k]       = kpR[i < lastR ? i : lastR];
    }
    // the 256-bit rows are scored by gathers later on; stream them through the cache hierarchy once,
    // fully coalesced (16 B/lane), so those gathers are L2 hits and HBM sees one sequential read per byte
#pragma unroll
    for (int k = 0; k < 2 * KPT; ++k) {
      const int i      = k * kV4Threads + tid;
      const v4u32x4 wl = gdL[i < 2 * lastL + 1 ? i : 2 * lastL + 1];
      const v4u32x4 wr = gdR[i < 2 * lastR + 1 ? i : 2 * lastR + 1];
      asm volatile("" ::"v"(wl), "v"(wr));
    }
  }
  // ---- B1: clear bins and bitsets, load the acceptance table ------------------------------------
  {
    uint32_t* w = reinterpret_cast<uint32_t*>(smem + a.off_bins_l);
    const int nw = (int) ((a.off_sorted_l - a.off_bins_l) >> 2);  // both bin arrays are contiguous
    for (int i = tid; i < nw; i += kV4Threads) {
      w[i] = 0;
    }
    for (int i = tid; i < 4 * nwords; i += kV4Threads) {
      bitsM[i] = 0;
    }
    if (tid < 258) {
      tab[tid] = a.bmax[tid];
    }
    if (tid < 8) {
      misc[tid] = 0;
    }
  }
  __syncthreads();
  PRS_V4_STAMP(1);

  // ---- B2: Feature{row,col,unsorted_index} (epipolar_impl.cpp:8-20) + bin counting ---------------
  int rowL[KPT], rowR[KPT], binL[KPT], binR[KPT], posL[KPT];
  uint32_t keyL[KPT], keyR[KPT], slotL[KPT], slotR[KPT];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = k * kV4Threads + tid;
    rowL[k] = rowR[k] = -1;
    binL[k] = binR[k] = posL[k] = 0;
    keyL[k] = keyR[k] = slotL[k] = slotR[k] = 0;
    if (i < nL) {
      const float u = cL[k].u, v = cL[k].v;
      if (u >= 0.0f && u < (float) cols && v >= 0.0f && v < (float) rows) {
        rowL[k]          = (int) v;
        const int col    = (int) u;
        keyL[k]          = ((uint32_t) col << 16) | (uint32_t) i;
        binL[k]          = rowL[k] * ncb + (col >> cbs);
        const int sh     = (binL[k] & 1) << 4;
        const uint32_t o = atomicAdd(reinterpret_cast<uint32_t*>(binsL) + (binL[k] >> 1), 1u << sh);
        slotL[k]         = (o >> sh) & 0xffffu;
      } else {
        bad = true;
      }
    }
    if (i < nR) {
      const float u = cR[k].u, v = cR[k].v;
      if (u >= 0.0f && u < (float) cols && v >= 0.0f && v < (float) rows) {
        rowR[k]          = (int) v;
        const int col    = (int) u;
        keyR[k]          = ((uint32_t) col << 16) | (uint32_t) i;
        binR[k]          = rowR[k] * ncb + (col >> cbs);
        const int sh     = (binR[k] & 1) << 4;
        const uint32_t o = atomicAdd(reinterpret_cast<uint32_t*>(binsR) + (binR[k] >> 1), 1u << sh);
        slotR[k]         = (o >> sh) & 0xffffu;
      } else {
        bad = true;
      }
    }
  }
  if (bad) {
    misc[0] = 1;
  }
  __syncthreads();
  if (misc[0]) {  // outside the stated image: loud per-frame error, no partial output
    if (tid == 0) {
      a.b.n_matches[frame] = 0;
      a.b.status[frame]    = PRS_ERR_RANGE;
      if (a.epilogue) {
        a.b.n_fixed[frame] = 0;
      }
    }
    return;
  }
  PRS_V4_STAMP(2);

  // ---- B3: exclusive scan of both bin arrays (left in the low, right in the high half-word) ------
  {
    const int n     = nb + 1;
    const int ipt   = ((n + kV4Threads - 1) / kV4Threads + 1) & ~1;  // even: bins are read as 32-bit pairs
    const int start = tid * ipt;
    uint32_t sum    = 0;
    for (int j = 0; j < ipt; j += 2) {
      const int bin = start + j;
      if (bin < n) {
        const uint32_t wl = reinterpret_cast<const uint32_t*>(binsL)[bin >> 1];
        const uint32_t wr = reinterpret_cast<const uint32_t*>(binsR)[bin >> 1];
        sum += ((wl & 0xffffu) + (wl >> 16)) | (((wr & 0xffffu) + (wr >> 16)) << 16);
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
    if (lane == 63) {
      misc[8 + wave] = (int) incl;
    }
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) {
      run += (uint32_t) misc[8 + w];
    }
    for (int j = 0; j < ipt; j += 2) {
      const int bin = start + j;
      if (bin < n) {
        const uint32_t wl = reinterpret_cast<const uint32_t*>(binsL)[bin >> 1];
        const uint32_t wr = reinterpret_cast<const uint32_t*>(binsR)[bin >> 1];
        const uint32_t l0 = run & 0xffffu, r0 = run >> 16;
        const uint32_t l1 = l0 + (wl & 0xffffu), r1 = r0 + (wr & 0xffffu);
        reinterpret_cast<uint32_t*>(binsL)[bin >> 1] = l0 | (l1 << 16);
        reinterpret_cast<uint32_t*>(binsR)[bin >> 1] = r0 | (r1 << 16);
        run = (l1 + (wl >> 16)) | ((r1 + (wr >> 16)) << 16);
      }
    }
  }
  __syncthreads();
  PRS_V4_STAMP(3);

  // ---- B4: scatter into bins, rank inside the bin by (col, unsorted index) -----------------------
  int sL[KPT], sR[KPT], lenL[KPT], lenR[KPT];
  int maxlen = 0;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    sL[k] = sR[k] = lenL[k] = lenR[k] = 0;
    if (rowL[k] >= 0) {
      sL[k]   = binsL[binL[k]];
      lenL[k] = binsL[binL[k] + 1] - sL[k];
      bucketL[sL[k] + slotL[k]] = keyL[k];
      maxlen  = lenL[k] > maxlen ? lenL[k] : maxlen;
    }
    if (rowR[k] >= 0) {
      sR[k]   = binsR[binR[k]];
      lenR[k] = binsR[binR[k] + 1] - sR[k];
      bucketR[sR[k] + slotR[k]] = keyR[k];
      maxlen  = lenR[k] > maxlen ? lenR[k] : maxlen;
    }
  }
  __syncthreads();
  {
    int rankL[KPT], rankR[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      rankL[k] = rankR[k] = 0;
    }
    // a bin holds ~0.3 keypoints: two unrolled reads cover almost every wave, longer bins loop
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const uint32_t bl = bucketL[sL[k] + j];
        const uint32_t br = bucketR[sR[k] + j];
        rankL[k] += (j < lenL[k] && bl < keyL[k]) ? 1 : 0;
        rankR[k] += (j < lenR[k] && br < keyR[k]) ? 1 : 0;
      }
    }
    for (int j = 2; j < maxlen; ++j) {
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
  __syncthreads();  // buckets are dead from here on: res[] may be written
  PRS_V4_STAMP(4);

  const int max_disp   = a.p.maximum_disparity_pixels;
  const int best_lim   = a.best_lim;
  const int thickness  = a.p.epipolar_line_thickness_pixels > 0 ? a.p.epipolar_line_thickness_pixels : 0;
  const int n_offsets  = 1 + 2 * thickness;
  const bool multipass = n_offsets > 1;
  prs_corr* __restrict__ out = a.b.matches + base;
  int out_base   = 0;
  int fixed_base = 0;

  for (int o = 0; o < n_offsets; ++o) {
    const int off = o == 0 ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2));  // 0,+1,-1,+2,-2 (epipolar_impl.cpp:71-79)
    const uint32_t tag = (uint32_t) (o + 1) << 16;

    // ---- D1: disparity windows.  Every left keypoint finds its in-window right keypoints (a
    //      contiguous range of the sorted right row), decides whether it is a HEAD (no candidate shared
    //      with the previous left keypoint of the row: the reference's cursor cannot cut its window)
    //      and queues one (left, right) pair per candidate in a dense LDS list.
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int p  = posL[k];
      uint2 r      = make_uint2(kFinal | kHead, 0u);  // decided, unmatched
      int n_pairs = 0, lo = 0;
      const int rr = rowL[k] + off;
      const bool pruned = multipass && rowL[k] >= 0 && ((bitsPL[p >> 5] >> (p & 31)) & 1u);
      if (rowL[k] >= 0 && rr >= 0 && rr < rows && !pruned) {
        const int col_l   = (int) (keyL[k] >> 16);
        const int col_min = col_l - max_disp;
        const int cb_lo   = (col_min > 0 ? col_min : 0) >> cbs;
        const int cb_hi   = col_l >> cbs;
        const int seg0    = binsR[rr * ncb + cb_lo];
        const int seg1    = binsR[rr * ncb + cb_hi + 1];
        const int row_first = binsL[rowL[k] * ncb];
        const int col_prev  = p > row_first ? (int) (sortedL[p - 1] >> 16) : -1;
        int n_lt = 0, n_le = 0, n_prev = 0;  // right entries of the segment with col < col_min, <= col_l, <= col_prev
        for (int q = seg0; q < seg1; ++q) {
          const int cr = (int) (sortedR[q] >> 16);
          n_lt += cr < col_min ? 1 : 0;      // epipolar_impl.cpp:146-149
          n_le += cr <= col_l ? 1 : 0;       // epipolar_impl.cpp:141-143
          n_prev += cr <= col_prev ? 1 : 0;
        }
        lo                = seg0 + n_lt;
        const int n       = n_le - n_lt;
        const uint32_t hd = n_prev <= n_lt ? kHead : 0u;
        if (n > 4) {
          r = make_uint2(7u << 27, (uint32_t) lo << 9);  // scored by the run walk (never a head)
        } else if (n > 0) {
          n_pairs = n;
          r       = make_uint2(((uint32_t) n << 27) | hd, (uint32_t) lo << 9);
        } else {
          r = make_uint2(kFinal | hd, 0u);
        }
      }
      // one LDS atomic per wave reserves the pair slots of its 64 keypoints
      int incl = n_pairs;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) {
          incl += t;
        }
      }
      int wave_base = 0;
      if (lane == 63) {
        wave_base = atomicAdd(&misc[3], incl);
      }
      wave_base = __shfl(wave_base, 63, 64);
      if (n_pairs > 0) {
        const int slot = wave_base + incl - n_pairs;
        if (slot + n_pairs <= a.pair_cap) {
          for (int j = 0; j < n_pairs; ++j) {
            pairs[slot + j] = (uint32_t) p | ((uint32_t) j << 13) | ((uint32_t) (lo + j) << 15);
          }
        } else {
          for (int j = 0; j < n_pairs && slot + j < a.pair_cap; ++j) {
            pairs[slot + j] = 0xffffffffu;  // list full: this keypoint is scored by the run walk instead
          }
          r = make_uint2(7u << 27, (uint32_t) lo << 9);
        }
      }
      if (rowL[k] >= 0) {
        res[p] = r;
      }
    }
    for (int i = tid; i < 2 * nwords; i += kV4Threads) {
      bitsM[i] = 0;  // bitsM and bitsK
    }
    __syncthreads();
    PRS_V4_STAMP(5);

    // ---- D2: dense scoring of the queued pairs (both 256-bit rows are gathered, 2 x 16 B each; two
    //      pairs per iteration keep eight gathers in flight per lane) ---------------------------------
    {
      const int np = misc[3] < a.pair_cap ? misc[3] : a.pair_cap;
      for (int t0 = tid; t0 < np; t0 += 2 * kV4Threads) {
        const int t1      = t0 + kV4Threads;
        const uint32_t e0 = pairs[t0];
        const uint32_t e1 = t1 < np ? pairs[t1] : 0xffffffffu;
        const int p0 = (int) (e0 & 0x1fffu), j0 = (int) ((e0 >> 13) & 3u), q0 = (int) (e0 >> 15);
        const int p1 = (int) (e1 & 0x1fffu), j1 = (int) ((e1 >> 13) & 3u), q1 = (int) (e1 >> 15);
        const bool v0 = e0 != 0xffffffffu && !(multipass && ((bitsPR[q0 >> 5] >> (q0 & 31)) & 1u));
        const bool v1 = e1 != 0xffffffffu && !(multipass && ((bitsPR[q1 >> 5] >> (q1 & 31)) & 1u));
        const int il0 = v0 ? (int) (sortedL[p0] & 0xffffu) : 0, ir0 = v0 ? (int) (sortedR[q0] & 0xffffu) : 0;
        const int il1 = v1 ? (int) (sortedL[p1] & 0xffffu) : 0, ir1 = v1 ? (int) (sortedR[q1] & 0xffffu) : 0;
        const v4u32x4 a0 = gdL[2 * il0], a1 = gdL[2 * il0 + 1], b0 = gdR[2 * ir0], b1 = gdR[2 * ir0 + 1];
        const v4u32x4 c0 = gdL[2 * il1], c1 = gdL[2 * il1 + 1], d0 = gdR[2 * ir1], d1 = gdR[2 * ir1 + 1];
        if (e0 != 0xffffffffu) {
          const uint32_t d = v0 ? v4_hamming(a0, a1, b0, b1) : kNone9;  // pruned (epipolar_impl.cpp:197-205)
          uint32_t* w      = reinterpret_cast<uint32_t*>(&res[p0]);
          atomicOr(j0 == 3 ? w + 1 : w, j0 == 3 ? d : d << (9 * j0));
        }
        if (e1 != 0xffffffffu) {
          const uint32_t d = v1 ? v4_hamming(c0, c1, d0, d1) : kNone9;
          uint32_t* w      = reinterpret_cast<uint32_t*>(&res[p1]);
          atomicOr(j1 == 3 ? w + 1 : w, j1 == 3 ? d : d << (9 * j1));
        }
      }
    }
    __syncthreads();
    PRS_V4_STAMP(6);

    // ---- D3: heads are decided by their own thread; followers that start a run are queued -------
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      bool run_start = false;
      if (rowL[k] >= 0) {
        const int p     = posL[k];
        const uint2 rec = res[p];
        if (rec.x & kFinal) {
          // nothing in its window
        } else if (rec.x & kHead) {
          const uint32_t n = (rec.x >> 27) & 7u;
          const int lo     = (int) ((rec.y >> 9) & 0x1fffu);
          const uint32_t dist[4] = {rec.x & 511u, (rec.x >> 9) & 511u, (rec.x >> 18) & 511u, rec.y & 511u};
          uint32_t best = 0xffffu, second = 0xffffu, bj = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if ((uint32_t) j < n && dist[j] != kNone9) {
              if (dist[j] < best) {  // epipolar_impl.cpp:158-164
                second = best;
                best   = dist[j];
                bj     = (uint32_t) j;
              } else if (dist[j] < second) {
                second = dist[j];
              }
            }
          }
          uint2 r = make_uint2(kFinal | kHead, 0u);
          // epipolar_impl.cpp:171-173 through the exact integer table
          if (best != 0xffffu && (int) best < best_lim && (int) best <= (int) tab[second == 0xffffu ? 257 : second]) {
            const uint32_t bq = (uint32_t) lo + bj;
            r = make_uint2((sortedR[bq] & 0xffffu) | (best << 16) | kFinal | kHead, bq | tag);
          }
          res[p] = r;  // readers of this slot only test kHead, which the decision keeps
        } else {
          // follower: it starts a run when the keypoint before it is a head (or it opens the row)
          const int row_first = binsL[rowL[k] * ncb];
          run_start           = p == row_first || (res[p - 1].x & kHead);
        }
      }
      const unsigned long long bal = __ballot(run_start);
      if (bal) {
        int wave_base = 0;
        if (lane == 0) {
          wave_base = atomicAdd(&misc[4], __popcll(bal));
        }
        wave_base = __shfl(wave_base, 0, 64);
        if (run_start) {
          runs[wave_base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t) posL[k] | ((uint32_t) rowL[k] << 13);
        }
      }
    }
    __syncthreads();
    PRS_V4_STAMP(7);

    // ---- E: one lane per RUN of followers replays the cursor (index_right = best + 1) -----------
    {
      const int nruns = misc[4];
      for (int t = tid; t < nruns; t += kV4Threads) {
        const uint32_t e = runs[t];
        const int p0 = (int) (e & 0x1fffu), r = (int) (e >> 13);
        const int rr = r + off;
        const int row_first = binsL[r * ncb], row_end = binsL[(r + 1) * ncb];
        const int re = binsR[(rr + 1) * ncb];
        int c        = binsR[rr * ncb];
        if (p0 > row_first) {
          const uint2 h = res[p0 - 1];  // the head in front of the run
          if ((h.y >> 16) == (uint32_t) (o + 1)) {
            c = (int) (h.y & 0x1fffu) + 1;  // epipolar_impl.cpp:181
          }
        }
        for (int p = p0; p < row_end; ++p) {
          const uint2 rec = res[p];
          if (rec.x & kHead) {
            break;  // the next head is independent of this run
          }
          if (rec.x & kFinal) {
            continue;  // nothing in its window
          }
          uint2 outrec     = make_uint2(kFinal, 0u);
          const uint32_t n = (rec.x >> 27) & 7u;
          if (c < re) {
            const int lo = (int) ((rec.y >> 9) & 0x1fffu);
            uint32_t best = 0xffffu, second = 0xffffu, best_q = 0;
            if (n == 7u) {
              // more than four in-window candidates (or pair list full): score them here, from the cursor on
              const uint32_t kl = sortedL[p];
              const int col_l   = (int) (kl >> 16);
              const int idx_l   = (int) (kl & 0xffffu);
              const v4u32x4 d0 = gdL[2 * idx_l], d1 = gdL[2 * idx_l + 1];
              for (int q = c > lo ? c : lo; q < re; ++q) {
                const uint32_t kr = sortedR[q];
                if (col_l - (int) (kr >> 16) < 0) {
                  break;
                }
                if (multipass && ((bitsPR[q >> 5] >> (q & 31)) & 1u)) {
                  continue;
                }
                const int idx_r  = (int) (kr & 0xffffu);
                const uint32_t d = v4_hamming(d0, d1, gdR[2 * idx_r], gdR[2 * idx_r + 1]);
                if (d < best) {
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
                if ((uint32_t) j < n && lo + j >= c && dist[j] != kNone9) {
                  if (dist[j] < best) {
                    second = best;
                    best   = dist[j];
                    best_q = (uint32_t) (lo + j);
                  } else if (dist[j] < second) {
                    second = dist[j];
                  }
                }
              }
            }
            if (best != 0xffffu && (int) best < best_lim && (int) best <= (int) tab[second == 0xffffu ? 257 : second]) {
              outrec = make_uint2((sortedR[best_q] & 0xffffu) | (best << 16) | kFinal, best_q | tag);
              c      = (int) best_q + 1;
            }
          }
          res[p] = outrec;
        }
      }
    }
    __syncthreads();
    PRS_V4_STAMP(8);
    if (tid == 0) {
      misc[3] = 0;
      misc[4] = 0;
    }

    // ---- F: output slots from popcount prefixes over the match bitset ---------------------------
    uint2 m_rec[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      m_rec[k] = make_uint2(0u, 0u);
      if (rowL[k] >= 0) {
        m_rec[k] = res[posL[k]];
        if ((m_rec[k].y >> 16) == (uint32_t) (o + 1)) {
          atomicOr(&bitsM[posL[k] >> 5], 1u << (posL[k] & 31));
          if (multipass) {
            const uint32_t bq = m_rec[k].y & 0x1fffu;
            atomicOr(&bitsPL[posL[k] >> 5], 1u << (posL[k] & 31));
            atomicOr(&bitsPR[bq >> 5], 1u << (bq & 31));
          }
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      const int chunk = (nwords + 63) >> 6;
      uint32_t sum    = 0;
      for (int j = 0; j < chunk; ++j) {
        const int w = lane * chunk + j;
        sum += w < nwords ? (uint32_t) __popc(bitsM[w]) : 0u;
      }
      uint32_t incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) {
          incl += t;
        }
      }
      uint32_t run = incl - sum;
      for (int j = 0; j < chunk; ++j) {
        const int w = lane * chunk + j;
        if (w < nwords) {
          prefM[w] = (uint16_t) run;
          run += (uint32_t) __popc(bitsM[w]);
        }
      }
      if (lane == 63) {
        misc[1] = (int) incl;
      }
    }
    __syncthreads();
    const int pass_matches = misc[1];
    int m_out[KPT];
    bool m_keep[KPT];
    prs_kp2 m_kr[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      m_out[k]  = -1;
      m_keep[k] = false;
      m_kr[k]   = prs_kp2{0.f, 0.f};
      if (rowL[k] >= 0 && (m_rec[k].y >> 16) == (uint32_t) (o + 1)) {
        const int p = posL[k];
        m_out[k]    = (int) prefM[p >> 5] + __popc(bitsM[p >> 5] & ((1u << (p & 31)) - 1u));
        prs_corr cr;
        cr.fixed_idx  = k * kV4Threads + tid;
        cr.moving_idx = (int) (m_rec[k].x & 0xffffu);
        cr.response   = (float) ((m_rec[k].x >> 16) & 0x1ffu);
        out[out_base + m_out[k]] = cr;
        if (a.epilogue) {
          m_kr[k] = kpR[cr.moving_idx];
          // raw_data_preprocessor_stereo_projective.cpp:117-125
          const float hd = cL[k].u - m_kr[k].u, vd = cL[k].v - m_kr[k].v;
          m_keep[k] = !(hd < 0.0f || vd < 0.0f);
          if (m_keep[k]) {
            atomicOr(&bitsK[m_out[k] >> 5], 1u << (m_out[k] & 31));
          }
        }
      }
    }
    if (a.epilogue) {
      __syncthreads();
      if (wave == 0) {
        const int chunk = (nwords + 63) >> 6;
        uint32_t sum    = 0;
        for (int j = 0; j < chunk; ++j) {
          const int w = lane * chunk + j;
          sum += w < nwords ? (uint32_t) __popc(bitsK[w]) : 0u;
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(incl, d, 64);
          if (lane >= d) {
            incl += t;
          }
        }
        uint32_t run = incl - sum;
        for (int j = 0; j < chunk; ++j) {
          const int w = lane * chunk + j;
          if (w < nwords) {
            prefK[w] = (uint16_t) run;
            run += (uint32_t) __popc(bitsK[w]);
          }
        }
        if (lane == 63) {
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
          v4u32x4* fd = reinterpret_cast<v4u32x4*>(a.b.fixed_desc) + 2 * g;
          const int il = k * kV4Threads + tid;
          fd[0]        = gdL[2 * il];
          fd[1]        = gdL[2 * il + 1];
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
      __syncthreads();  // res[], bitsets, misc[] are rewritten by the next pass
    }
  }

  PRS_V4_STAMP(9);
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

static inline uint32_t v4_align16(uint32_t v) {
  return (v + 15u) / 16u * 16u;
}

// returns PRS_OK when launched, 1 when this generation does not cover the request (caller falls back)
int stereo_match_v4_launch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch) {
  const int stride = batch->stride;
  if (params->image_cols <= 0 || params->image_cols > 32767 || stride > 4 * kV4Threads || ctx_force_unstaged(ctx) || ctx_matcher_v3(ctx)) {
    return 1;
  }
  StereoV4Args a;
  a.p        = *params;
  a.b        = *batch;
  a.epilogue = (batch->fixed_uvuv && batch->fixed_desc && batch->n_fixed && batch->fixed_xyz && batch->triangulator) ? 1 : 0;
  a.tri      = a.epilogue ? *batch->triangulator : prs_triangulator_params{1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // column blocks of 128 px, widened until the bin table fits
  int cbs = 7;
  int ncb = (params->image_cols + (1 << cbs) - 1) >> cbs;
  while ((long long) params->image_rows * ncb + 2 > kMaxBins) {
    ++cbs;
    ncb = (params->image_cols + (1 << cbs) - 1) >> cbs;
  }
  a.cbs = cbs;
  a.ncb = ncb;
  a.nb  = params->image_rows * ncb;
  fill_accept_table(params, &a.best_lim, a.bmax);
  const uint32_t nwords = ((uint32_t) stride + 31) / 32;
  uint32_t off = 0;
  const uint32_t bin_bytes = v4_align16(((uint32_t) a.nb + 2) * 2);
  a.off_bins_l   = off; off += bin_bytes;
  a.off_bins_r   = off; off += bin_bytes;
  a.off_sorted_l = off; off = v4_align16(off + (uint32_t) stride * 4);
  a.off_sorted_r = off; off = v4_align16(off + (uint32_t) stride * 4);
  a.off_res      = off; off = v4_align16(off + ((uint32_t) stride + 8) * 8);
  a.pair_cap     = 2 * stride > 1024 ? 2 * stride : 1024;  // >= stride + 8 entries, also holds the run list
  a.off_pairs    = off; off = v4_align16(off + (uint32_t) a.pair_cap * 4);
  a.off_runs     = a.off_pairs;  // the run list is born after the pair list died
  a.off_bits     = off; off = v4_align16(off + nwords * 16 + (nwords + 2) * 4);
  a.off_tab      = off; off = v4_align16(off + 258 * 2);
  a.off_misc     = off; off = v4_align16(off + 64);
  const size_t lds = off;
  a.stamps = ctx_stamps(ctx, (size_t) batch->batch * 16 * sizeof(unsigned long long));
  hipStream_t stream = ctx_stream(ctx);
  hipError_t e       = hipSuccess;
  const int kpt      = stride <= kV4Threads ? 1 : (stride <= 2 * kV4Threads ? 2 : 4);
#define PRS_V4_LAUNCH(K)                                                                                               \
  do {                                                                                                                 \
    auto kernel = stereo_match_v4_kernel<K>;                                                                           \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
    if (e == hipSuccess) {                                                                                             \
      hipLaunchKernelGGL(kernel, dim3(batch->batch), dim3(kV4Threads), lds, stream, a);                                \
      e = hipGetLastError();                                                                                           \
    }                                                                                                                  \
  } while (0)
  if (kpt == 1) {
    PRS_V4_LAUNCH(1);
  } else if (kpt == 2) {
    PRS_V4_LAUNCH(2);
  } else {
    PRS_V4_LAUNCH(4);
  }
#undef PRS_V4_LAUNCH
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_stereo_match_batch (v4) launch");
  }
  if (a.stamps) {
    int occ = -1;
    if (kpt == 4) {
      (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stereo_match_v4_kernel<4>, kV4Threads, lds);
    }
    hipFuncAttributes fa;
    memset(&fa, 0, sizeof(fa));
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(stereo_match_v4_kernel<4>));
    fprintf(stderr, "[prs stamps] stereo_match_v4: %zu B LDS per workgroup, %d workgroups per CU by the occupancy query (numRegs %d, static LDS %zu, local %zu, maxThreads %d)\n",
            lds, occ, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxThreadsPerBlock);
    ctx_report_stamps(ctx, batch->batch, 10, "stereo_match_v4: loads+clear | bin count | bin scan | scatter+rank | windows | pair scoring | heads+runs | run walk | emit");
  }
  return PRS_OK;
}

}  // namespace prs
