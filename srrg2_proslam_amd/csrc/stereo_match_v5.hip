// stereo_match_v5.hip -- instruction-lean staged stereo matcher for gfx950 (frames of <= 2048 keypoints).
//
// Same algorithm, phases and outputs as stereo_match.hip (CorrespondenceFinderDescriptorBasedEpipolar<..>::compute,
// registration/correspondence_finders/correspondence_finder_descriptor_based_epipolar_impl.cpp:46-219, with the
// fused stereo-adaptor + triangulator epilogue); what changes is the instruction count, which is what
// bounds that kernel (profiles/r01/matcher_pmc_diag.json: the four SIMDs issue ~40 k wave-instructions per
// frame, memory and LDS are far from busy and a second resident workgroup does not shorten a frame):
//   * every epipolar row of the bucket / sorted arrays is padded to a multiple of four entries with
//     0x7fffffff sentinels, so in-row ranks and disparity-window bounds are counted with 16-byte LDS reads
//     and a subtract + funnel shift per entry (no per-entry length test, no binary search, no VCC);
//   * the scoring phase (all lanes busy) also evaluates, for every possible cursor position inside the
//     window, whether the keypoint would match and with which candidate (a 4 x 3 bit table, built by one
//     backwards sweep over the <= 4 candidates); the serial cursor chain (one lane per row, 6 of 16
//     waves) then needs one 4-byte read, one shift and one 2-byte write per left keypoint;
//   * the right coordinates the epilogue needs are staged in LDS next to the descriptor rows.
// A window of more than four candidates is scored into a pool of 32-bit entries (LDS left over by the layout); the scoring lane
// then sweeps its window backwards and leaves, for every position the chain's cursor may stand at, the verdict of the candidates from
// there on (round 5: the chain looks one word up instead of comparing the window again, which on row-skewed real keypoints was most of
// the longest row's chain); rows with a window the pool has no room for are replayed by a second, rarely entered sweep that
// re-scores those windows from global memory.
#include <type_traits>

#include "prs_device.h"
#include "prs_host.h"

namespace prs {
namespace v5 {  // (named: rocprofv3 summaries key on the kernel name up to its first parenthesis)

struct Args5 {
  prs_stereo_params p;
  prs_stereo_batch b;
  prs_triangulator_params tri;
  int epilogue;
  int cap;     // padded sorted positions (multiple of 4); entries [cap, cap + 4) of every key array stay sentinels
  int nwords;  // 32-bit words covering cap positions
  uint32_t off_desc_r, off_kp_r, off_sorted_l, off_sorted_r, off_bucket, off_hist, off_rs, off_len, off_rowcnt, off_out, off_bits, off_misc,
    off_tab, off_pool;
  int pool_cap;  // 32-bit entries of the candidate pool (windows of more than four candidates), 0: none
  int best_lim;       // accept iff best < best_lim ...
  int16_t bmax[258];  // ... and best <= bmax[second] (257 = no second candidate), see fill_accept_table
  unsigned long long* stamps;
};

typedef unsigned int q32 __attribute__((ext_vector_type(4)));
constexpr int kT                = 1024;
constexpr uint32_t kNone        = 0xffffu;
constexpr uint32_t kOverflow    = 1u << 31;
constexpr uint32_t kPooled      = 1u << 30;  // ... and its candidates were scored into the pool: res[p].x = pool offset | candidates << 16
// pool entry of one candidate while its window is being scored: distance (0..256) | kept by the stereo adaptor << 12 | pruned by an
// earlier pass << 13; once the window is swept, entry j holds the verdict of the candidates j .. n-1:
//   accepted << 31 | best candidate kept << 30 | (best candidate - the window's start) << 9 | best distance
constexpr uint32_t kPoolPruned  = 1u << 13;
// candidate record of a sorted-left position (written by the scoring phase):
//   res[p].x  = verdict[0..3], 4 bit each (bits 16..19 stay zero: "cursor beyond the window")
//   res[p].y  = lo (13 bit: sorted position of the first in-window candidate) | more-than-four-candidates << 31
//   dist4[p]  = four 8-bit distances of candidates lo .. lo+3 (255 = none / pruned / >= 255: never accepted,
//               best_lim <= 255); for a more-than-four record: the left keypoint's (col << 16 | index) key
// verdict[m] = accepted << 3 | candidate << 1 | kept: what the chain does once its cursor consumed m candidates
// chain output per sorted-left position: verdict << 28 | rescored << 27 | kept-before-in-row << 12 | matches-before-in-row
constexpr uint32_t kOutRescored = 1u << 27;

#define PRS5_STAMP(i)                                                       \
  do {                                                                      \
    if (a.stamps && tid == 0) {                                             \
      a.stamps[(size_t) frame * 16 + (i)] = (unsigned long long) clock64(); \
    }                                                                       \
  } while (0)

__device__ __forceinline__ uint32_t hamming5(const q32& a0, const q32& a1, const q32& b0, const q32& b1) {
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

// Keys are (col << 16 | index) with col < 32768: 31-bit values, and so are the compare bounds.  For such
// a, b the top bit of a - b says a < b, which needs no compare-to-VCC round trip (every VALU write of
// VCC costs wait states before the next VALU read of it on gfx950): below_bits shifts the four sign bits
// of a quad into a mask whose population count is the number of entries below the bound.  The sentinel
// 0x7fffffff pads the rows; it is above every key and never below a bound.
constexpr uint32_t kSentinel = 0x7fffffffu;
__device__ __forceinline__ uint32_t below_bits(uint32_t mask, const q32& q, uint32_t bound) {
  mask = __builtin_amdgcn_alignbit(mask, q.x - bound, 31);
  mask = __builtin_amdgcn_alignbit(mask, q.y - bound, 31);
  mask = __builtin_amdgcn_alignbit(mask, q.z - bound, 31);
  return __builtin_amdgcn_alignbit(mask, q.w - bound, 31);
}
__device__ __forceinline__ int below3(const q32& q0, const q32& q1, const q32& q2, uint32_t bound) {
  return __popc(below_bits(below_bits(below_bits(0u, q0, bound), q1, bound), q2, bound));
}
__device__ __forceinline__ int below(const q32& q, uint32_t bound) {
  return __popc(below_bits(0u, q, bound));
}

// The scored candidates of a window of more than four, swept backwards: entry j becomes the verdict of the candidates j .. n-1 --
// best / second best (epipolar_impl.cpp:158-164: the first of equal distances wins, the second best counts multiplicity) as the smallest
// and second smallest of the unique keys (distance, position, keep bit), then the acceptance test (:171-173) -- what the chain's cursor
// finds when it stands at j.  Out of line: crowded windows are rare (a few dozen per real frame, none in most synthetic ones) and the
// scoring phase around the call is bound by its registers.
__device__ __attribute__((noinline)) void sweep_pooled_window(uint32_t* w, const int n, const int16_t* tab, const int best_lim) {
  uint32_t bestk = 0xffffffffu, seck = 0xffffffffu;
  for (int j = n - 1; j >= 0; --j) {
    const uint32_t e   = w[j];
    const uint32_t key = (e & kPoolPruned) ? 0xffffffffu : (((e & 0x1ffu) << 17) | ((uint32_t) j << 1) | ((e >> 12) & 1u));
    const uint32_t hk  = key > bestk ? key : bestk;
    seck               = hk < seck ? hk : seck;
    bestk              = key < bestk ? key : bestk;
    const uint32_t best = bestk == 0xffffffffu ? kNone : bestk >> 17, second = seck == 0xffffffffu ? kNone : seck >> 17;
    const bool accept   = best != kNone && (int) best < best_lim && (int) best <= (int) tab[second == kNone ? 257u : second];
    w[j] = accept ? (1u << 31) | ((bestk & 1u) << 30) | (((bestk >> 1) & 0xffffu) << 9) | best : 0u;
  }
}

// inclusive prefix sum over the 64 lanes of a wave on the DPP network (round 4; was six ds_bpermute round trips): Hillis-Steele inside
// every row of 16 lanes (row_shr 1, 2, 4, 8, lanes shifted in from outside a row read 0), then lane 15 of rows 0 / 2 onto rows 1 / 3
// (row_bcast:15) and lane 31 onto the upper half (row_bcast:31)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
  v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
  return v;
}

// ONE wave: start[r] = sum over r' < r of the count rounded up to a multiple of four, len[r] = count
__device__ __forceinline__ void wave_padded_scan(const uint32_t* hist, uint16_t* start, uint16_t* len, int n) {
  const int lane  = threadIdx.x & 63;
  const int chunk = (n + 63) >> 6;
  const int first = lane * chunk;
  uint32_t sum    = 0;
  uint32_t v[8];
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = first + j;
      v[j]        = (j < chunk && r < n) ? hist[r] : 0u;
      sum += (v[j] + 3u) & ~3u;
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = first + j;
      sum += r < n ? ((hist[r] + 3u) & ~3u) : 0u;
    }
  }
  const uint32_t incl = wave_inclusive_scan(sum);
  uint32_t run        = incl - sum;
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = first + j;
      if (j < chunk && r < n) {
        start[r] = (uint16_t) run;
        len[r]   = (uint16_t) v[j];
        run += (v[j] + 3u) & ~3u;
      }
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = first + j;
      if (r < n) {
        const uint32_t h = hist[r];
        start[r]         = (uint16_t) run;
        len[r]           = (uint16_t) h;
        run += (h + 3u) & ~3u;
      }
    }
  }
}

// ONE wave: in-place exclusive scan of n counters, returns the total in every lane
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t* cnt, int n) {
  const int lane  = threadIdx.x & 63;
  const int chunk = (n + 63) >> 6;
  const int first = lane * chunk;
  uint32_t sum    = 0;
  uint32_t v[8];
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = first + j;
      v[j]        = (j < chunk && r < n) ? cnt[r] : 0u;
      sum += v[j];
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = first + j;
      sum += r < n ? cnt[r] : 0u;
    }
  }
  const uint32_t incl = wave_inclusive_scan(sum);
  uint32_t run        = incl - sum;
  if (chunk <= 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = first + j;
      if (j < chunk && r < n) {
        cnt[r] = run;
        run += v[j];
      }
    }
  } else {
    for (int j = 0; j < chunk; ++j) {
      const int r = first + j;
      if (r < n) {
        const uint32_t h = cnt[r];
        cnt[r]           = run;
        run += h;
      }
    }
  }
  return __shfl(incl, 63, 64);
}

// MULTI: epipolar_line_thickness_pixels > 0 (several passes over row offsets 0, +1, -1, .. with the matched keypoints of earlier passes
// pruned: epipolar_impl.cpp:197-205).  kitti.conf / euroc.conf run ONE pass: their instantiation carries neither the bit sets nor the
// per-candidate pruning tests (round 5)
// EPI: the fused stereo-adaptor + triangulator epilogue is wanted (prs_stereo_batch carries its output buffers)
template <int KPT, bool MULTI, bool EPI>
__global__ __launch_bounds__(kT) void stereo_match5_kernel(const Args5 a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid    = threadIdx.x;
  const int stride = a.b.stride;
  const int rows   = a.p.image_rows;
  const int cap    = a.cap;
  const int nwords = a.nwords;

  q32* ldR          = reinterpret_cast<q32*>(smem + a.off_desc_r);          // right descriptor rows by unsorted index
  prs_kp2* ldKR     = reinterpret_cast<prs_kp2*>(smem + a.off_kp_r);        // right coordinates by unsorted index
  uint32_t* dist4   = reinterpret_cast<uint32_t*>(smem + a.off_sorted_l);   // candidate distances per sorted-left position
  uint32_t* sortedR = reinterpret_cast<uint32_t*>(smem + a.off_sorted_r);   // (col << 16 | index), rows padded to 4
  uint32_t* bucketL = reinterpret_cast<uint32_t*>(smem + a.off_bucket);     // same layout, arbitrary order inside a row
  uint32_t* bucketR = bucketL + cap + 4;
  uint2* res        = reinterpret_cast<uint2*>(smem + a.off_bucket);        // candidate records, alias both buckets after the sort
  uint32_t* histL   = reinterpret_cast<uint32_t*>(smem + a.off_hist);
  uint32_t* histR   = histL + rows + 1;
  uint16_t* rsL     = reinterpret_cast<uint16_t*>(smem + a.off_rs);         // first sorted position of a row
  uint16_t* rsR     = rsL + rows + 2;
  uint16_t* lenL    = reinterpret_cast<uint16_t*>(smem + a.off_len);        // keypoints in a row
  uint16_t* lenR    = lenL + rows + 2;
  uint32_t* rowsum  = reinterpret_cast<uint32_t*>(smem + a.off_rowcnt);     // matches | kept << 16 of a row, then their prefix
  uint32_t* outv    = reinterpret_cast<uint32_t*>(smem + a.off_out);        // chain output per sorted-left position
  uint32_t* bitsL   = reinterpret_cast<uint32_t*>(smem + a.off_bits);       // left sorted position matched in an earlier pass
  uint32_t* bitsR   = bitsL + nwords;                                       // right sorted position matched
  int* misc         = reinterpret_cast<int*>(smem + a.off_misc);            // [0] error, [1] pass matches, [2] pass kept
  int16_t* tab      = reinterpret_cast<int16_t*>(smem + a.off_tab);         // Lowe acceptance table
  uint32_t* pool    = reinterpret_cast<uint32_t*>(smem + a.off_pool);       // scored candidates of the windows with more than four

  // persistent: grid = CUs, frames strided over the workgroups; the next frame's coordinates are
  // requested while this frame is scored, its descriptor rows when it starts (see stereo_match.hip)
  prs_kp2 cLn[KPT], cRn[KPT];
  int nLn = 0, nRn = 0;
  auto fetch_coords = [&](int f) {
    int fl = a.b.n_left[f];
    int fr = a.b.n_right[f];
    fl     = fl < 0 ? 0 : (fl > stride ? stride : fl);
    fr     = fr < 0 ? 0 : (fr > stride ? stride : fr);
    nLn    = fl;
    nRn    = fr;
    const size_t fbase              = (size_t) f * (size_t) stride;
    const prs_kp2* __restrict__ fkL = a.b.left_kp + fbase;
    const prs_kp2* __restrict__ fkR = a.b.right_kp + fbase;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kT + tid;
      cLn[k]      = fkL[i < fl ? i : (fl > 0 ? fl - 1 : 0)];
      cRn[k]      = fkR[i < fr ? i : (fr > 0 ? fr - 1 : 0)];
    }
  };
  for (int i = tid; i < 258; i += kT) {
    tab[i] = a.bmax[i];  // published by the first barrier of the frame
  }
  if ((int) blockIdx.x < a.b.batch) {
    fetch_coords((int) blockIdx.x);
  }
  const q32 ones = {kSentinel, kSentinel, kSentinel, kSentinel};

  for (int frame = blockIdx.x; frame < a.b.batch; frame += gridDim.x) {
    PRS5_STAMP(0);
    const int nL      = nLn;
    const int nR      = nRn;
    const size_t base = (size_t) frame * (size_t) stride;
    const q32* __restrict__ gdL = reinterpret_cast<const q32*>(a.b.left_desc + base * PRS_DESC_BYTES);
    const q32* __restrict__ gdR = reinterpret_cast<const q32*>(a.b.right_desc + base * PRS_DESC_BYTES);
    prs_kp2 cL[KPT], cR[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      cL[k] = cLn[k];
      cR[k] = cRn[k];
    }
    // ---- A: the right descriptor rows are requested up front (tail lanes re-read the last row), the left
    // rows (first used by the scoring phase) once the coordinates are binned: issuing a frame's 128 kB at
    // once blocks every wave at the memory pipeline for ~6 k cycles
    q32 dL[2 * KPT], dR[2 * KPT];
    {
      const int lastR = nR > 0 ? nR - 1 : 0;
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const int i  = k * kT + tid;
        const int ir = i < lastR ? i : lastR;
        dR[2 * k]     = gdR[2 * ir];
        dR[2 * k + 1] = gdR[2 * ir + 1];
      }
    }
    const int next_frame = frame + (int) gridDim.x;

    // ---- B: Feature{row,col,unsorted_index} + counting sort by row (epipolar_impl.cpp:8-42) ----------
    for (int i = tid; i < 2 * (rows + 1); i += kT) {
      histL[i] = 0;  // histL | histR
    }
    {
      q32* fill = reinterpret_cast<q32*>(bucketL);  // bucketL | bucketR, then sortedR: sentinels everywhere
      for (int i = tid; i < (cap + 4) / 2; i += kT) {
        fill[i] = ones;
      }
      q32* fill_s = reinterpret_cast<q32*>(sortedR);
      for (int i = tid; i < (cap + 4) / 4; i += kT) {
        fill_s[i] = ones;
      }
    }
    for (int i = tid; i < 2 * nwords; i += kT) {
      bitsL[i] = 0;  // bitsL | bitsR
    }
    if (tid < 8) {
      misc[tid] = 0;  // [4]: entries of the candidate pool in use
    }
    __syncthreads();
    PRS5_STAMP(1);

    int rowL[KPT], rowR[KPT], posL[KPT];
    uint32_t keyL[KPT], keyR[KPT], slotL[KPT], slotR[KPT];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kT + tid;
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
        if (EPI) {
          a.b.n_fixed[frame] = 0;
        }
      }
      if (next_frame < a.b.batch) {
        fetch_coords(next_frame);
      }
      __syncthreads();  // misc[] is rewritten by the next frame
      continue;
    }
    PRS5_STAMP(2);
    {
      const int lastL = nL > 0 ? nL - 1 : 0;
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const int i  = k * kT + tid;
        const int il = i < lastL ? i : lastL;
        dL[2 * k]     = gdL[2 * il];
        dL[2 * k + 1] = gdL[2 * il + 1];
      }
    }
    // ---- C: right descriptor rows and coordinates land in LDS (their registers are free for the sort) ----
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int i = k * kT + tid;
      if (i < nR) {
        ldR[2 * i]     = dR[2 * k];
        ldR[2 * i + 1] = dR[2 * k + 1];
        ldKR[i]        = cR[k];
      }
    }
    if (tid < 64) {
      wave_padded_scan(histL, rsL, lenL, rows);
    } else if (tid < 128) {
      wave_padded_scan(histR, rsR, lenR, rows);
    }
    __syncthreads();
    PRS5_STAMP(3);

    // scatter into the padded row buckets (arbitrary order inside a row) ...
    int sL[KPT], sR[KPT], nrowL[KPT], nrowR[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      sL[k] = sR[k] = cap;  // lanes without a keypoint read the sentinel quad
      nrowL[k] = nrowR[k] = 0;
      if (rowL[k] >= 0) {
        sL[k]    = rsL[rowL[k]];
        nrowL[k] = lenL[rowL[k]];
        bucketL[sL[k] + slotL[k]] = keyL[k];
      }
      if (rowR[k] >= 0) {
        sR[k]    = rsR[rowR[k]];
        nrowR[k] = lenR[rowR[k]];
        bucketR[sR[k] + slotR[k]] = keyR[k];
      }
    }
    __syncthreads();
    // ... then rank inside the row by (col, unsorted index): epipolar_impl.cpp:36-41 + canonical tie-break.
    // Three 16-byte reads cover rows of up to twelve keypoints; the pad entries compare as "not smaller".
    {
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const q32 l0 = *reinterpret_cast<const q32*>(bucketL + sL[k]);
        const q32 l1 = *reinterpret_cast<const q32*>(bucketL + (nrowL[k] > 4 ? sL[k] + 4 : cap));
        const q32 l2 = *reinterpret_cast<const q32*>(bucketL + (nrowL[k] > 8 ? sL[k] + 8 : cap));
        const q32 r0 = *reinterpret_cast<const q32*>(bucketR + sR[k]);
        const q32 r1 = *reinterpret_cast<const q32*>(bucketR + (nrowR[k] > 4 ? sR[k] + 4 : cap));
        const q32 r2 = *reinterpret_cast<const q32*>(bucketR + (nrowR[k] > 8 ? sR[k] + 8 : cap));
        int rankL    = below3(l0, l1, l2, keyL[k]);
        int rankR    = below3(r0, r1, r2, keyR[k]);
        for (int j = 12; j < nrowL[k]; j += 4) {
          rankL += below(*reinterpret_cast<const q32*>(bucketL + sL[k] + j), keyL[k]);
        }
        for (int j = 12; j < nrowR[k]; j += 4) {
          rankR += below(*reinterpret_cast<const q32*>(bucketR + sR[k] + j), keyR[k]);
        }
        if (rowL[k] >= 0) {
          posL[k] = sL[k] + rankL;  // the sorted left keys themselves are never needed
        }
        if (rowR[k] >= 0) {
          sortedR[sR[k] + rankR] = keyR[k];
        }
      }
    }
    PRS5_STAMP(4);
    __syncthreads();  // buckets are dead from here on: res[] may be written
    PRS5_STAMP(5);
    if (next_frame < a.b.batch) {
      fetch_coords(next_frame);  // in flight while this frame is scored, chained and emitted
    }

    const int best_lim   = a.best_lim;
    const int max_disp   = a.p.maximum_disparity_pixels;
    const int thickness  = a.p.epipolar_line_thickness_pixels > 0 ? a.p.epipolar_line_thickness_pixels : 0;
    const int n_offsets  = MULTI ? 1 + 2 * thickness : 1;
    const bool multipass = MULTI && n_offsets > 1;
    prs_corr* __restrict__ out = a.b.matches + base;
    int out_base               = 0;
    int fixed_base             = 0;

    for (int o = 0; o < n_offsets; ++o) {
      const int off = o == 0 ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2));  // 0,+1,-1,+2,-2 (epipolar_impl.cpp:71-79)

      // ---- D: every left keypoint scores its in-window candidates and tabulates the chain's verdicts ----
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        if (rowL[k] < 0) {
          continue;
        }
        const int p       = posL[k];
        uint2 r           = make_uint2(0u, 0u);
        uint32_t d4       = 0;
        const int rr      = rowL[k] + off;
        const bool pruned = multipass && ((bitsL[p >> 5] >> (p & 31)) & 1u);
        if (rr >= 0 && rr < rows && !pruned) {
          const int col_l = (int) (keyL[k] >> 16);
          const int rs    = rsR[rr];
          const int rn    = lenR[rr];
          // in-window right features are contiguous in the sorted row:
          //   lo = first q with col_r >= col_l - max_disp   (epipolar_impl.cpp:146-149)
          //   hi = first q with col_r >  col_l              (epipolar_impl.cpp:141-143)
          const int col_min  = col_l - max_disp;
          const uint32_t klo = (uint32_t) (col_min > 0 ? col_min : 0) << 16;  // keys below it: col < col_min
          const uint32_t kh  = (uint32_t) (col_l + 1) << 16;                   // keys below it: col <= col_l
          const uint32_t khi = kh < kSentinel ? kh : kSentinel;                // (col_l = 32767: every key, no sentinel)
          const q32 w0       = *reinterpret_cast<const q32*>(sortedR + (rn > 0 ? rs : cap));
          const q32 w1       = *reinterpret_cast<const q32*>(sortedR + (rn > 4 ? rs + 4 : cap));
          const q32 w2       = *reinterpret_cast<const q32*>(sortedR + (rn > 8 ? rs + 8 : cap));
          int n_lo           = below3(w0, w1, w2, klo);
          int n_hi           = below3(w0, w1, w2, khi);
          for (int j = 12; j < rn; j += 4) {
            const q32 w = *reinterpret_cast<const q32*>(sortedR + rs + j);
            n_lo += below(w, klo);
            n_hi += below(w, khi);
          }
          const int lo = rs + n_lo;
          const int n  = n_hi - n_lo;
          if (n > 4) {
            // more than four candidates: their distances (and keep bits) go to the pool, so that the chain only compares;
            // a window the pool has no room for is scored by the chain's second sweep from global memory
            r  = make_uint2(0u, (uint32_t) lo | kOverflow);
            d4 = keyL[k];
            int poff = -1;
            if (a.pool_cap > 0) {
              const int got = atomicAdd(&misc[4], n);
              poff          = got + n <= a.pool_cap ? got : -1;
            }
            if (poff >= 0) {
              const q32 d0 = dL[2 * k], d1 = dL[2 * k + 1];
              for (int j = 0; j < n; ++j) {
                const int q    = lo + j;
                uint32_t entry = kPoolPruned;
                if (!(multipass && ((bitsR[q >> 5] >> (q & 31)) & 1u))) {
                  const int idx_r = (int) (sortedR[q] & 0xffffu);
                  entry           = hamming5(d0, d1, ldR[2 * idx_r], ldR[2 * idx_r + 1]);
                  if (EPI) {
                    const prs_kp2 kr = ldKR[idx_r];
                    const float hd = cL[k].u - kr.u, vd = cL[k].v - kr.v;
                    entry |= (hd < 0.0f || vd < 0.0f) ? 0u : (1u << 12);
                  }
                }
                pool[poff + j] = entry;
              }
              sweep_pooled_window(pool + poff, n, tab, best_lim);
              r = make_uint2((uint32_t) poff | ((uint32_t) n << 16), (uint32_t) lo | kOverflow | kPooled);
            }
          } else if (n > 0) {
            const q32 d0 = dL[2 * k], d1 = dL[2 * k + 1];
            // kNone: no such candidate, or pruned by an earlier pass (epipolar_impl.cpp:197-205)
            uint32_t dist[4], code[4];  // code = candidate << 1 | kept by the stereo adaptor
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              dist[j] = kNone;
              code[j] = (uint32_t) j << 1;
              if (j < n) {
                const int q = lo + j;
                if (!(multipass && ((bitsR[q >> 5] >> (q & 31)) & 1u))) {
                  const int idx_r = (int) (sortedR[q] & 0xffffu);
                  dist[j]         = hamming5(d0, d1, ldR[2 * idx_r], ldR[2 * idx_r + 1]);
                  if (EPI) {
                    // raw_data_preprocessor_stereo_projective.cpp:117-125
                    const prs_kp2 kr = ldKR[idx_r];
                    const float hd = cL[k].u - kr.u, vd = cL[k].v - kr.v;
                    code[j] |= (hd < 0.0f || vd < 0.0f) ? 0u : 1u;
                  }
                }
              }
            }
            // verdict[m] = what the chain does when its cursor has consumed the first m candidates:
            // best / second best over candidates m.. (epipolar_impl.cpp:158-164: the first of equal
            // distances wins, the second-best counts multiplicity), then the acceptance test (:171-173).
            // One sweep from the last candidate to the first visits every suffix.  A dead candidate
            // (distance kNone) leaves best / second as they are or replaces kNone by kNone, so the sweep
            // needs no branches; best < best_lim <= 255 rejects an empty suffix.
            uint32_t best = kNone, second = kNone, best_code = 0, verdicts = 0;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
              const uint32_t d  = dist[j];
              const bool closer = d <= best;
              const uint32_t s2 = d < second ? d : second;
              second            = closer ? best : s2;
              best_code         = closer ? code[j] : best_code;
              best              = closer ? d : best;
              const int limit   = (int) tab[second < 257u ? second : 257u];
              const bool accept = (int) best < best_lim && (int) best <= limit;
              verdicts |= accept ? (8u | best_code) << (4 * j) : 0u;
            }
            const uint32_t b0 = dist[0] < 255u ? dist[0] : 255u, b1 = dist[1] < 255u ? dist[1] : 255u;
            const uint32_t b2 = dist[2] < 255u ? dist[2] : 255u, b3 = dist[3] < 255u ? dist[3] : 255u;
            r  = make_uint2(verdicts, (uint32_t) lo);
            d4 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
          }
        }
        res[p]   = r;
        dist4[p] = d4;
      }
      __syncthreads();
      PRS5_STAMP(6);

      // ---- E: one lane per epipolar row replays the serial cursor chain on the verdict tables -------
      // A row that holds a window of more than four candidates needs descriptor reads from global
      // memory; it is left to a second sweep so that the common sweep holds no VMEM instruction (a
      // vmcnt wait inside it would also wait for the coordinate prefetch of the next frame).
      auto chain_row = [&](const int r, auto rescoring) -> uint32_t {
        constexpr bool kRescore = decltype(rescoring)::value;
        const int rr = r + off;
        const int ls = rsL[r], le = ls + lenL[r];
        if (rr < 0 || rr >= rows) {
          for (int p = ls; p < le; ++p) {
            outv[p] = 0;
          }
          return 0;
        }
        int c         = rsR[rr];
        const int re  = c + lenR[rr];
        uint32_t cnt = 0, kept = 0, flags = 0;
        uint2 w = res[ls];  // (the arrays are padded: reading one record past the row is harmless)
        for (int p = ls; p < le; ++p) {
          const uint2 w_next = res[p + 1];
          const int lo       = (int) (w.y & 0x1fffu);
          if (w.y & kPooled) {
            // more than four in-window candidates, scored into the pool: best / second best from the cursor on
            const uint32_t poff = w.x & 0xffffu;
            const int hi        = lo + (int) (w.x >> 16);
            // the verdict of the candidates from the cursor on was left in the pool by the scoring lane
            const int sfrom       = c > lo ? c - lo : 0;
            const uint32_t word   = sfrom < hi - lo ? pool[poff + (uint32_t) sfrom] : 0u;
            const uint32_t best   = word & 0x1ffu;
            const uint32_t best_q = (uint32_t) lo + ((word >> 9) & 0xffffu), best_keep = (word >> 30) & 1u;
            if (word >> 31) {
              dist4[p] = (sortedR[best_q] & 0xffffu) | (best << 16);  // (res[p] keeps the pool reference: the row may be replayed)
              outv[p]  = ((8u | best_keep) << 28) | kOutRescored | (kept << 12) | cnt;
              ++cnt;
              kept += best_keep;
              c = (int) best_q + 1;  // epipolar_impl.cpp:181
              if (multipass && (kRescore || !(flags & kOverflow))) {
                atomicOr(&bitsL[p >> 5], 1u << (p & 31));
                atomicOr(&bitsR[best_q >> 5], 1u << (best_q & 31));
              }
            } else {
              outv[p] = 0;
            }
          } else if (kRescore && (w.y & kOverflow)) {
            // more than four in-window candidates: score them here, from the cursor on
            const uint32_t kl = dist4[p];
            const int col_l   = (int) (kl >> 16);
            const int idx_l   = (int) (kl & 0xffffu);
            const q32 d0 = gdL[2 * idx_l], d1 = gdL[2 * idx_l + 1];
            uint32_t best = kNone, second = kNone, best_q = 0;
            for (int q = c > lo ? c : lo; q < re; ++q) {
              const uint32_t kr = sortedR[q];
              if (col_l - (int) (kr >> 16) < 0) {
                break;  // epipolar_impl.cpp:141-143
              }
              if (multipass && ((bitsR[q >> 5] >> (q & 31)) & 1u)) {
                continue;
              }
              const int idx_r  = (int) (kr & 0xffffu);
              const uint32_t d = hamming5(d0, d1, ldR[2 * idx_r], ldR[2 * idx_r + 1]);
              if (d < best) {  // epipolar_impl.cpp:158-164
                second = best;
                best   = d;
                best_q = (uint32_t) q;
              } else if (d < second) {
                second = d;
              }
            }
            if (best != kNone && (int) best < best_lim && (int) best <= (int) tab[second == kNone ? 257u : second]) {
              const uint32_t idx_r = sortedR[best_q] & 0xffffu;
              res[p].x             = idx_r | (best << 16);
              uint32_t keep        = 0;
              if (EPI) {
                const prs_kp2 kl2 = a.b.left_kp[base + (size_t) idx_l];
                const prs_kp2 kr2 = ldKR[idx_r];
                const float hd = kl2.u - kr2.u, vd = kl2.v - kr2.v;
                keep = (hd < 0.0f || vd < 0.0f) ? 0u : 1u;
              }
              outv[p] = ((8u | keep) << 28) | kOutRescored | (kept << 12) | cnt;
              ++cnt;
              kept += keep;
              c = (int) best_q + 1;  // epipolar_impl.cpp:181
              if (multipass) {
                atomicOr(&bitsL[p >> 5], 1u << (p & 31));
                atomicOr(&bitsR[best_q >> 5], 1u << (best_q & 31));
              }
            } else {
              outv[p] = 0;
            }
          } else {
            flags |= w.y;
            // candidates left of the cursor were consumed by an earlier match (epipolar_impl.cpp:181);
            // the verdicts of candidates that do not exist, and the fifth one, are zero
            int m = c - lo;
            m     = m < 0 ? 0 : (m > 4 ? 4 : m);
            const uint32_t e    = (w.x >> (4 * m)) & 15u;
            const bool accepted = e >= 8u;
            const int q         = lo + (int) ((e >> 1) & 3u);
            outv[p]             = accepted ? (e << 28) | (kept << 12) | cnt : 0u;
            cnt += accepted ? 1u : 0u;
            kept += e & 1u;
            c = accepted ? q + 1 : c;  // epipolar_impl.cpp:181
            // (behind a more-than-four record the cursor of this sweep is not the chain's: the row is
            // replayed by the second sweep and nothing may be marked on its behalf here)
            if (multipass && accepted && !(flags & kOverflow)) {
              atomicOr(&bitsL[p >> 5], 1u << (p & 31));
              atomicOr(&bitsR[q >> 5], 1u << (q & 31));
            }
          }
          w = w_next;
        }
        return (!kRescore && (flags & kOverflow)) ? 0xffffffffu : (cnt | (kept << 16));
      };
      for (int r = tid; r < rows; r += kT) {
        const uint32_t sum = chain_row(r, std::false_type());
        if (sum == 0xffffffffu) {
          misc[3] = 1;
        }
        rowsum[r] = sum;  // 0xffffffff: left to the second sweep
      }
      __syncthreads();
      if (misc[3]) {
        for (int r = tid; r < rows; r += kT) {
          if (rowsum[r] == 0xffffffffu) {
            rowsum[r] = chain_row(r, std::true_type());
          }
        }
        __syncthreads();
        if (tid == 0) {
          misc[3] = 0;  // the next pass starts clean (its scoring barrier publishes this)
        }
      }
      if (tid == 0) {
        misc[4] = 0;  // the pool is free again (published by the barriers of the emit phase)
      }
      PRS5_STAMP(7);

      // ---- F: emit in sorted-left traversal order (one scan carries matches and kept matches) ---------
      if (tid < 64) {
        const uint32_t total = wave_scan_u32(rowsum, rows);
        if (tid == 0) {
          misc[1] = (int) total;
        }
      }
      __syncthreads();
      const int pass_matches = misc[1] & 0xffff;
      const int pass_kept    = (int) ((uint32_t) misc[1] >> 16);
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        if (rowL[k] < 0) {
          continue;
        }
        const uint32_t v = outv[posL[k]];
        if (v >> 31) {
          const uint2 rec = res[posL[k]];
          uint32_t idx_r, best;
          if (v & kOutRescored) {
            const uint32_t rx = (rec.y & kPooled) ? dist4[posL[k]] : rec.x;
            idx_r             = rx & 0xffffu;
            best              = rx >> 16;
          } else {
            const uint32_t j = (v >> 29) & 3u;
            best             = (dist4[posL[k]] >> (8 * j)) & 255u;
            idx_r            = sortedR[(rec.y & 0x1fffu) + j] & 0xffffu;
          }
          const uint32_t before = rowsum[rowL[k]];
          prs_corr cr;
          cr.fixed_idx  = k * kT + tid;
          cr.moving_idx = (int) idx_r;
          cr.response   = (float) best;
          out[out_base + (int) (before & 0xffffu) + (int) (v & 0xfffu)] = cr;  // index inside this pass
          if (EPI && ((v >> 28) & 1u)) {
            const int slot   = fixed_base + (int) (before >> 16) + (int) ((v >> 12) & 0xfffu);
            const size_t g   = base + (size_t) slot;
            const prs_kp2 kr = ldKR[idx_r];
            const float x_L = cL[k].u, y_L = cL[k].v, x_R = kr.u, y_R = kr.v;
            reinterpret_cast<float4*>(a.b.fixed_uvuv)[g] = make_float4(x_L, y_L, x_R, y_R);
            q32* fd = reinterpret_cast<q32*>(a.b.fixed_desc) + 2 * g;
            fd[0]   = dL[2 * k];
            fd[1]   = dL[2 * k + 1];
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
      }
      fixed_base += pass_kept;
      out_base += pass_matches;
      if (o + 1 < n_offsets) {
        __syncthreads();  // res[], outv[], rowsum[], misc[] are rewritten by the next pass
      }
    }

    PRS5_STAMP(8);
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
      if (EPI) {
        a.b.n_fixed[frame] = fixed_base;
      }
    }
    __syncthreads();  // the LDS arrays are rewritten by the next frame
  }
}

inline uint32_t up16(uint32_t v) {
  return (v + 15u) & ~15u;
}

template <int KPT>
hipError_t launch5(const Args5& a, size_t lds, hipStream_t stream) {
  const bool multi = a.p.epipolar_line_thickness_pixels > 0;
  auto kernel      = a.epilogue ? (multi ? stereo_match5_kernel<KPT, true, true> : stereo_match5_kernel<KPT, false, true>)
                                : (multi ? stereo_match5_kernel<KPT, true, false> : stereo_match5_kernel<KPT, false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  if (e != hipSuccess) {
    return e;
  }
  // persistent: one workgroup per CU (the LDS footprint allows no more), frames strided over them
  int grid = a.b.batch, dev = 0, cus = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
      cus < grid) {
    grid = cus;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kT), lds, stream, a);
  return hipGetLastError();
}

}  // namespace v5
using namespace v5;

// returns 1 when the frame shape is outside this kernel's domain (the caller falls back to stereo_match_kernel)
int stereo_match_v5_launch(prs_context* ctx, const prs_stereo_params* params, const prs_stereo_batch* batch) {
  const int stride = batch->stride;
  const int rows   = params->image_rows;
  if (stride > 2 * kT || ctx_force_unstaged(ctx) || ctx_matcher_v3(ctx)) {
    return 1;
  }
  Args5 a;
  // every non-empty row is padded by at most three entries
  const uint32_t cap = ((uint32_t) stride + 3u * (uint32_t) (rows < stride ? rows : stride) + 3u) & ~3u;
  if (cap > 8188u) {
    return 1;  // sorted positions are stored in 13 bits
  }
  a.cap    = (int) cap;
  a.nwords = (int) ((cap + 31u) / 32u);
  uint32_t off = 0;
  const uint32_t rows2 = (uint32_t) rows + 2u;
  a.off_desc_r   = off; off = up16(off + (uint32_t) stride * PRS_DESC_BYTES);
  a.off_kp_r     = off; off = up16(off + (uint32_t) stride * 8u);
  a.off_sorted_l = off; off = up16(off + (cap + 4u) * 4u);  // dist4
  a.off_sorted_r = off; off = up16(off + (cap + 4u) * 4u);
  a.off_bucket   = off; off = up16(off + 2u * (cap + 4u) * 4u);  // bucketL | bucketR, later res[cap]
  a.off_hist     = off; off = up16(off + 2u * rows2 * 4u);
  a.off_rs       = off; off = up16(off + 2u * rows2 * 2u);
  a.off_len      = off; off = up16(off + 2u * rows2 * 2u);
  a.off_rowcnt   = off; off = up16(off + rows2 * 4u);
  a.off_out      = off; off = up16(off + cap * 4u);
  a.off_bits     = off; off = up16(off + (uint32_t) a.nwords * 2u * 4u);
  a.off_misc     = off; off = up16(off + 32u);
  a.off_tab      = off; off = up16(off + 258u * 2u);
  if (off > 160u * 1024u) {
    return 1;
  }
  // what is left of the 160 KB holds the scored candidates of crowded windows (four bytes each, at most 4096)
  a.off_pool = off;
  a.pool_cap = (int) (((160u * 1024u - off) / 4u) & ~3u);
  a.pool_cap = a.pool_cap > 4096 ? 4096 : (a.pool_cap < 64 ? 0 : a.pool_cap);
  off        = up16(off + (uint32_t) a.pool_cap * 4u);
  const size_t lds = off;
  a.p        = *params;
  a.b        = *batch;
  a.epilogue = (batch->fixed_uvuv && batch->fixed_desc && batch->n_fixed && batch->fixed_xyz && batch->triangulator) ? 1 : 0;
  if (a.epilogue) {
    a.tri = *batch->triangulator;
  } else {
    a.tri = prs_triangulator_params{1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
  fill_accept_table(params, &a.best_lim, a.bmax);
  if (a.best_lim > 255) {
    return 1;  // the candidate records keep 8 bits per distance
  }
  a.stamps           = ctx_stamps(ctx, (size_t) batch->batch * 16 * sizeof(unsigned long long));
  hipStream_t stream = ctx_stream(ctx);
  const hipError_t e = stride <= kT ? launch5<1>(a, lds, stream) : launch5<2>(a, lds, stream);
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_stereo_match_batch launch");
  }
  if (a.stamps) {
    ctx_report_stamps(ctx, batch->batch, 9,
                      "stereo_match_v5: issue+fill | coords+hist | (err check) | scan | scatter+rank | stage-write | score | chain | emit");
  }
  return PRS_OK;
}

}  // namespace prs
