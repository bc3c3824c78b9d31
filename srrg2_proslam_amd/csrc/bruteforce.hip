// bruteforce.hip -- CorrespondenceFinderDescriptorBasedBruteforce::compute on the device
// (CF/correspondence_finder_descriptor_based_bruteforce_impl.cpp:8-155, pool processing :247-293,
// Lowe checks :157-199).  SURVEY.md section 8f #4.
//
// The reference enumerates all N_f x N_m Hamming distances, keeps the candidates below the
// threshold, sorts them by distance and registers them pool by pool (one pool per distinct
// distance): a candidate is registered iff neither index is registered yet, no other pool member
// shares its fixed or moving index, and Lowe's ratio holds in the distance list of its fixed AND
// of its moving index.  Registrations only ever happen between pools, so the sequential loop is a
// level-synchronous process over the distinct distances 0, 1, 2, ... -- which is how it runs here:
//
//   phase 1 (dense): all pairs scored, the candidates appended to the pair's list, the per-index distance bitmaps and candidate
//            counts Lowe's check needs updated.  Popcount kernels: one lane per fixed descriptor (rows in registers), the moving
//            rows through scalar loads (uniform address), 2 VALU per 32-bit word, candidates recorded branch-free and drained
//            every 32 rows.  Matrix cores (bf_matrix_phase1, the default for batches that fill the chip; and the split
//            bruteforce_dense_mfma_kernel): v_mfma_i32_16x16x64_i8 on 0 / 1 x +1 / -1 bytes, exact.  Real descriptors put 1.6 % of
//            all pairs below a threshold of 50 bits (uniform random rows: one per point): candidates are NOT rare;
//   phase 2: Lowe flags per candidate (next larger distance = first set bit above d in the
//            bitmap), counting sort of the candidates by distance level (bitmaps and level lists in LDS when they fit);
//   phase 3: the levels in ascending order: pool membership, uniqueness counts, registration (two barriers per level);
//   phase 4: emit ordered by (distance, fixed index) -- the canonical order this build defines for the
//            reference's unstable std::sort by response only (:94-97).
#include <stdlib.h>
#include <type_traits>

#include "prs_device.h"
#include "prs_host.h"

namespace prs {

constexpr int kBfThreads = 1024;
constexpr int kBfWaves   = kBfThreads / 64;
constexpr int kBfLevels  = 256;  // distances 0..255 (a candidate at 256 needs a threshold above 256: unsupported)

struct BfArgs {
  prs_bruteforce_batch b;
  float max_ratio;
  int lim;             // candidate iff d < lim  (== (float) d < maximum_descriptor_distance)
  int nw;              // bitmap words per index = ceil(lim / 32)
  int cap;             // candidate capacity per frame
  uint2* cand;         // [grid][cap] (fixed | moving << 16, distance)
  uint2* by_level;     // [grid][cap] (fixed | moving << 16, lowe ok)
  uint32_t* bitmaps;   // [grid][(fixed_stride + moving_stride) * nw]
  // LDS carve (bytes)
  uint32_t off_cnt_f, off_cnt_m, off_reg_f, off_reg_m, off_acc, off_hist;
  // what of the registration state lives in LDS instead of the scratch rows in global memory (the registration phases are chains of
  // dependent reads: one real cloud pair of 1350 x 1350 points spent 80 of its 135 us on them at L2 latency):
  uint32_t off_bm;     // the distance bitmaps of the fixed / moving indices ((fixed_stride + moving_stride) * nw words), ~0: global
  uint32_t off_lvl;    // the candidates grouped by level, 4 bytes each (fixed | moving << 13 | lowe ok << 29) ...
  int lvl_cap;         // ... for cloud pairs with at most this many candidates (0: always global, 8 bytes each)
  uint32_t off_mx;     // scratch of the matrix-core dense phase (bf_matrix_phase1), dead behind it: the level lists lie over it
  // few cloud pairs: the dense phase is spread over `chunks` workgroups per pair (each takes a slice of the
  // moving cloud) that accumulate into global memory; one workgroup per pair then registers the candidates
  int chunks;
  uint32_t* g_acc;     // [batch][fixed_stride + moving_stride + 256 + 8]: candidates per fixed, per moving, per level, total
};

enum { kBfFused = 0, kBfDense = 1, kBfRegister = 2 };

// Lowe's ratio against the sorted distance list of one index (bruteforce_impl.cpp:157-199):
// a single-entry list passes (:181-184); otherwise the first strictly larger distance is the second
// best, none = equal = reject (:163-165), else best / second < maximum ratio in float.
template <bool LDS>
__device__ __forceinline__ bool lowe_ok(const uint32_t* bm, int nw, uint32_t count, int d, float max_ratio) {
  if (count == 1) {
    return true;
  }
  int second = -1;
  int w      = d >> 5;
  // the rows are updated with device-scope atomics: read them the same way (never from a stale L1 line)
  uint32_t word = (LDS ? bm[w] : __hip_atomic_load(&bm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~((2u << (d & 31)) - 1u);  // bits above d
  while (true) {
    if (word) {
      second = (w << 5) + __ffs(word) - 1;
      break;
    }
    if (++w >= nw) {
      break;
    }
    word = LDS ? bm[w] : __hip_atomic_load(&bm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (second < 0) {
    return false;
  }
  return (float) d / (float) second < max_ratio;
}

// dense phase of the fused shape on the matrix cores (defined with the matrix-core kernels below)
template <int THREADS, int CHUNK>
__device__ __forceinline__ void bf_matrix_phase1(const BfArgs& a, unsigned char* scratch, int nf, int nm, const uint32_t* gdf, const uint32_t* gdm, uint2* cand,
                                                 uint32_t* cnt_f, uint32_t* cnt_m, uint32_t* hist, uint32_t* counter, uint32_t* lbm_f, uint32_t* lbm_m);

// THREADS = 1024 (one workgroup per CU) everywhere but in the matrix-core fused shape for clouds that leave room for TWO 512-thread
// workgroups per CU (each at most 80 KiB of LDS): one scores while the other walks its latency-bound registration phases.
template <int KPT, int MODE, bool MX = false, int THREADS = kBfThreads>
__global__ __launch_bounds__(THREADS, 4) void bruteforce_kernel(const BfArgs a) {
  static_assert(!MX || (MODE == kBfFused && KPT == 1), "the matrix-core dense phase belongs to the fused shape");
  static_assert(THREADS == kBfThreads || MX, "the popcount dense phase owns one fixed row per thread of a 1024-thread workgroup");
  constexpr int kBfThreads = THREADS, kBfWaves = THREADS / 64;  // (shadow the file-scope constants inside the kernel)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* cnt_f  = reinterpret_cast<uint32_t*>(smem + a.off_cnt_f);   // candidates per fixed; later pool counts
  uint32_t* cnt_m  = reinterpret_cast<uint32_t*>(smem + a.off_cnt_m);
  uint8_t* reg_f   = smem + a.off_reg_f;                                 // registered_indices_fixed (:105)
  uint8_t* reg_m   = smem + a.off_reg_m;
  uint32_t* acc    = reinterpret_cast<uint32_t*>(smem + a.off_acc);     // per fixed: moving | level << 16, ~0 = none
  uint32_t* hist   = reinterpret_cast<uint32_t*>(smem + a.off_hist);    // [256] candidates per level
  uint32_t* lstart = hist + kBfLevels;                                   // [257]
  uint32_t* lfill  = lstart + kBfLevels + 1;                             // [256]
  uint32_t* hist2  = lfill + kBfLevels;                                  // [256] registrations per level
  uint32_t* misc   = hist2 + kBfLevels;                                  // [0] candidates, [1] registrations
  const bool bm_lds = MODE != kBfDense && a.off_bm != 0xffffffffu;       // (uniform)
  uint32_t* lbm_f  = reinterpret_cast<uint32_t*>(smem + (bm_lds ? a.off_bm : 0u));
  uint32_t* lbm_m  = lbm_f + (size_t) a.b.fixed_stride * a.nw;
  uint32_t* lvl    = reinterpret_cast<uint32_t*>(smem + a.off_lvl);
  const int tid    = threadIdx.x;
  const int lane   = tid & 63;
  const int wave   = tid >> 6;
  // scratch rows: per workgroup when fused (workgroups loop over pairs), per pair in the split shape
  const int scratch_row        = MODE == kBfDense ? (int) blockIdx.y : (int) blockIdx.x;
  uint2* __restrict__ cand     = a.cand + (size_t) scratch_row * a.cap;
  uint2* __restrict__ by_level = a.by_level + (size_t) scratch_row * a.cap;
  uint32_t* __restrict__ bm_f  = a.bitmaps + (size_t) scratch_row * (size_t) (a.b.fixed_stride + a.b.moving_stride) * a.nw;
  uint32_t* __restrict__ bm_m  = bm_f + (size_t) a.b.fixed_stride * a.nw;
  const size_t acc_row         = (size_t) (a.b.fixed_stride + a.b.moving_stride + kBfLevels + 8);
  uint32_t* __restrict__ g_cnt_f = MODE == kBfFused ? nullptr : a.g_acc + (size_t) scratch_row * acc_row;
  uint32_t* __restrict__ g_cnt_m = MODE == kBfFused ? nullptr : g_cnt_f + a.b.fixed_stride;
  uint32_t* __restrict__ g_hist  = MODE == kBfFused ? nullptr : g_cnt_m + a.b.moving_stride;
  uint32_t* __restrict__ g_total = MODE == kBfFused ? nullptr : g_hist + kBfLevels;

  const int frame_first = MODE == kBfDense ? (int) blockIdx.y : (int) blockIdx.x;
  const int frame_step  = MODE == kBfFused ? (int) gridDim.x : a.b.batch;  // split shapes: exactly one pair per workgroup
  for (int frame = frame_first; frame < a.b.batch; frame += frame_step) {
    int nf = a.b.n_fixed[frame];
    int nm = a.b.n_moving[frame];
    nf     = nf < 0 ? 0 : (nf > a.b.fixed_stride ? a.b.fixed_stride : nf);
    nm     = nm < 0 ? 0 : (nm > a.b.moving_stride ? a.b.moving_stride : nm);
    const int out_stride = a.b.fixed_stride < a.b.moving_stride ? a.b.fixed_stride : a.b.moving_stride;
    prs_corr* __restrict__ out = a.b.matches + (size_t) frame * out_stride;
    const uint32_t* __restrict__ gdf =
      reinterpret_cast<const uint32_t*>(a.b.fixed_desc + (size_t) frame * a.b.fixed_stride * PRS_DESC_BYTES);
    // the moving rows are read-only for the whole launch: constant address space + uniform index
    // = scalar loads (s_load_dwordx8), the row sits in SGPRs and feeds v_xor as the scalar operand
    typedef const uint32_t __attribute__((address_space(4))) const_u32;
    const_u32* gdm = (const_u32*) (uintptr_t) (a.b.moving_desc + (size_t) frame * a.b.moving_stride * PRS_DESC_BYTES);

    // ---- reset ----------------------------------------------------------------------------------
    if (MODE != kBfDense) {
      for (int i = tid; i < nf; i += kBfThreads) {
        cnt_f[i] = MODE == kBfRegister ? g_cnt_f[i] : 0u;
        reg_f[i] = 0;
        acc[i]   = 0xffffffffu;
      }
      for (int i = tid; i < nm; i += kBfThreads) {
        cnt_m[i] = MODE == kBfRegister ? g_cnt_m[i] : 0u;
        reg_m[i] = 0;
      }
      if (MODE == kBfFused || bm_lds) {
        for (int i = tid; i < (nf + nm) * a.nw; i += kBfThreads) {
          // fixed rows first, moving rows behind them (bm_m = bm_f + fixed_stride * nw)
          const int idx = i < nf * a.nw ? i : (a.b.fixed_stride * a.nw + (i - nf * a.nw));
          if (bm_lds) {
            lbm_f[idx] = MODE == kBfRegister ? bm_f[idx] : 0u;  // (the dense launch's rows, complete at the kernel boundary)
          } else {
            bm_f[idx] = 0;
          }
        }
      }
      for (int i = tid; i < 4 * kBfLevels + 8; i += kBfThreads) {
        hist[i] = (MODE == kBfRegister && i < kBfLevels) ? g_hist[i] : 0u;
      }
      __syncthreads();
      if (MODE == kBfRegister && tid == 0) {
        misc[0] = g_total[0];
      }
    }

    // ---- phase 1: all pairs (bruteforce_impl.cpp:32-79) ----------------------------------------
    if constexpr (MX) {
      bf_matrix_phase1<THREADS, THREADS == 1024 ? 64 : 32>(a, smem + a.off_mx, nf, nm, gdf,
                       reinterpret_cast<const uint32_t*>(a.b.moving_desc + (size_t) frame * a.b.moving_stride * PRS_DESC_BYTES), cand, cnt_f, cnt_m, hist, &misc[0],
                       lbm_f, lbm_m);
    } else {
    uint32_t fd[KPT][8];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int f  = k * kBfThreads + tid;
      const int fc = f < nf ? f : (nf > 0 ? nf - 1 : 0);
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        fd[k][w] = nf > 0 ? gdf[8 * fc + w] : 0u;
      }
    }
    // the slice of the moving cloud this workgroup scores
    int m_begin = 0, m_end = nm;
    if (MODE == kBfDense) {
      const int per = (nm + a.chunks - 1) / a.chunks;
      m_begin       = (int) blockIdx.x * per;
      m_end         = m_begin + per < nm ? m_begin + per : nm;
    }
    if (MODE == kBfRegister) {
      m_begin = m_end = 0;
    }
    // Candidates are NOT handled where they are found: a lane shifts the outcome of every comparison into a 32-bit record (one
    // v_cmp + one v_addc per pair, no branch), and after 32 moving rows the wave drains its records: every lane with a marked row
    // re-scores it (the row through a per-lane load: the block's rows are hot in the cache) and publishes it, one slot range per wave
    // and round from the candidate counter.  On real descriptors 1.6 % of the pairs are candidates (two thirds of a wave's rows
    // hold one): handling them inside the row loop ran the publish path -- a returning atomic on the pair's one counter, five more
    // atomics, a store -- twenty times per 32 rows with one or two lanes busy; the drain runs it three or four times with a third
    // of the lanes busy.  (Uniform random rows: one candidate per fixed point, nothing to gain or lose.)
    if (MODE == kBfDense) {
      for (int i = tid; i < kBfLevels; i += kBfThreads) {
        hist[i] = 0u;  // (this workgroup's candidates by level: added to the pair's histogram once, behind the loop)
      }
      __syncthreads();
    }
    uint32_t hit[KPT], cnt_mine[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      hit[k]      = 0u;
      cnt_mine[k] = 0u;  // candidates of this thread's fixed row(s): a register, stored / added once
    }
    const uint32_t* __restrict__ gdm_lane =
      reinterpret_cast<const uint32_t*>(a.b.moving_desc + (size_t) frame * a.b.moving_stride * PRS_DESC_BYTES);  // (per-lane rows of the drain)
    auto drain = [&](const int block_first, const int block_rows) {
#pragma unroll
      for (int k = 0; k < KPT; ++k) {
        const int f = k * kBfThreads + tid;
        uint32_t h  = f < nf ? hit[k] : 0u;
        hit[k]      = 0u;
        // one slot range per wave and block: the lanes' counts summed along the wave (six DPP adds), ONE returning add on the candidate
        // counter, then every lane walks its own marks with no further exchange (non-returning atomics and stores only)
        const uint32_t mine = (uint32_t) __popc(h);
        uint32_t incl       = mine;
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111, 0xf, 0xf, false);  // row_shr:1
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112, 0xf, 0xf, false);  // row_shr:2
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114, 0xf, 0xf, false);  // row_shr:4
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118, 0xf, 0xf, false);  // row_shr:8
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
        incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
        const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
        if (total == 0u) {
          continue;  // (wave-uniform)
        }
        uint32_t base = 0;
        if (lane == 0) {
          base = MODE == kBfDense ? atomicAdd(g_total, total) : atomicAdd(&misc[0], total);
        }
        uint32_t slot = (uint32_t) __builtin_amdgcn_readfirstlane((int) base) + incl - mine;
        while (h != 0u) {  // :52
          const int b = 31 - __clz((int) h);  // the record's bit j holds row block_first + block_rows - 1 - j
          h &= ~(1u << b);
          const int m     = block_first + block_rows - 1 - b;
          const uint4* pm = reinterpret_cast<const uint4*>(gdm_lane + 8 * m);
          const uint4 m0 = pm[0], m1 = pm[1];
          const int d = __popc(fd[k][0] ^ m0.x) + __popc(fd[k][1] ^ m0.y) + __popc(fd[k][2] ^ m0.z) + __popc(fd[k][3] ^ m0.w) + __popc(fd[k][4] ^ m1.x) +
                        __popc(fd[k][5] ^ m1.y) + __popc(fd[k][6] ^ m1.z) + __popc(fd[k][7] ^ m1.w);
          if (slot < (uint32_t) a.cap) {
            cand[slot] = make_uint2((uint32_t) f | ((uint32_t) m << 16), (uint32_t) d);
          }
          ++slot;
          const uint32_t bit = 1u << (d & 31);
          if (bm_lds) {
            atomicOr(&lbm_f[f * a.nw + (d >> 5)], bit);
            atomicOr(&lbm_m[m * a.nw + (d >> 5)], bit);
          } else {
            atomicOr(&bm_f[f * a.nw + (d >> 5)], bit);
            atomicOr(&bm_m[m * a.nw + (d >> 5)], bit);
          }
          if (MODE == kBfDense) {
            atomicAdd(&g_cnt_m[m], 1u);
          } else {
            atomicAdd(&cnt_m[m], 1u);
          }
          atomicAdd(&hist[d], 1u);
        }
        cnt_mine[k] += mine;
      }
    };
    uint32_t md_next[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      md_next[w] = m_begin < m_end ? gdm[8 * m_begin + w] : 0u;
    }
    for (int block_first = m_begin; block_first < m_end; block_first += 32) {
      const int block_rows = m_end - block_first < 32 ? m_end - block_first : 32;
      // the block's 32 rows (1 KB) touched once by the wave's 64 lanes: they reach the SCALAR cache through the row loop, the drain's
      // per-lane loads go through the vector L1 and would each wait for the L2 otherwise
      const int touch_row = block_first + (lane >> 1) < m_end ? block_first + (lane >> 1) : m_end - 1;
      const uint4 touched = *reinterpret_cast<const uint4*>(gdm_lane + 8 * touch_row + 4 * (lane & 1));
      for (int j = 0; j < block_rows; ++j) {
        const int m = block_first + j;
        // uniform address: the row travels through the scalar cache into SGPRs; the next row is
        // requested before this one is scored
        uint32_t md[8];
        const int mn = m + 1 < m_end ? m + 1 : m;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          md[w]      = md_next[w];
          md_next[w] = gdm[8 * mn + w];
        }
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
          int d = 0;
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            d += __popc(fd[k][w] ^ md[w]);
          }
          // record = 2 * record + (d < lim)
          asm("v_cmp_gt_i32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hit[k]) : "v"(d), "s"(a.lim) : "vcc");
        }
      }
      asm volatile("" ::"v"(touched.x), "v"(touched.y), "v"(touched.z), "v"(touched.w));  // (the touch has landed)
      drain(block_first, block_rows);
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int f = k * kBfThreads + tid;
      if (f < nf && MODE == kBfFused) {
        cnt_f[f] = cnt_mine[k];
      }
      if (f < nf && MODE == kBfDense && cnt_mine[k] != 0u) {
        atomicAdd(&g_cnt_f[f], cnt_mine[k]);
      }
    }
    if (MODE == kBfDense) {
      __syncthreads();
      for (int i = tid; i < kBfLevels; i += kBfThreads) {
        if (hist[i] != 0u) {
          atomicAdd(&g_hist[i], hist[i]);
        }
      }
      __syncthreads();
    }
    }  // (popcount dense phase)
    if (MODE == kBfDense) {
      continue;  // the registration launch takes over
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t n_cand = misc[0];
    int status            = (nf == 0 || nm == 0) ? PRS_WARN_EMPTY_INPUT : PRS_OK;  // bruteforce_impl.cpp:217-226
    if (n_cand > (uint32_t) a.cap) {
      if (tid == 0) {
        a.b.n_matches[frame] = 0;
        a.b.status[frame]    = PRS_ERR_CAPACITY;
      }
      __syncthreads();
      continue;
    }
    if (n_cand == 0) {  // :83-85
      if (tid == 0) {
        a.b.n_matches[frame] = 0;
        a.b.status[frame]    = status | PRS_WARN_NO_MATCHES;
      }
      __syncthreads();
      continue;
    }

    // ---- phase 2: level offsets, Lowe flags, candidates grouped by level -----------------------
    if (wave == 0) {
      uint32_t carry = 0;
      for (int base = 0; base < kBfLevels; base += 64) {
        const uint32_t h   = hist[base + lane];
        uint32_t incl      = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(incl, o, 64);
          if (lane >= o) {
            incl += v;
          }
        }
        lstart[base + lane] = carry + incl - h;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) {
        lstart[kBfLevels] = carry;
      }
    }
    __syncthreads();
    const bool lvl_lds = n_cand <= (uint32_t) a.lvl_cap;  // (uniform) this pair's level lists fit the LDS
    for (uint32_t i = tid; i < n_cand; i += kBfThreads) {
      const uint2 c  = cand[i];
      const int f    = (int) (c.x & 0xffffu);
      const int m    = (int) (c.x >> 16);
      const int d    = (int) c.y;
      const bool ok  = bm_lds ? lowe_ok<true>(lbm_f + (size_t) f * a.nw, a.nw, cnt_f[f], d, a.max_ratio) &&
                                 lowe_ok<true>(lbm_m + (size_t) m * a.nw, a.nw, cnt_m[m], d, a.max_ratio)
                              : lowe_ok<false>(bm_f + (size_t) f * a.nw, a.nw, cnt_f[f], d, a.max_ratio) &&
                                 lowe_ok<false>(bm_m + (size_t) m * a.nw, a.nw, cnt_m[m], d, a.max_ratio);  // :279-284
      const uint32_t pos = lstart[d] + atomicAdd(&lfill[d], 1u);
      if (lvl_lds) {
        lvl[pos] = (uint32_t) f | ((uint32_t) m << 13) | (ok ? 1u << 29 : 0u);
      } else {
        by_level[pos] = make_uint2(c.x, ok ? 1u : 0u);
      }
    }
    __syncthreads();
    // the candidate counts are dead: the arrays now count pool members per index
    for (int i = tid; i < nf; i += kBfThreads) {
      cnt_f[i] = 0;
    }
    for (int i = tid; i < nm; i += kBfThreads) {
      cnt_m[i] = 0;
    }
    __threadfence_block();
    __syncthreads();

    // ---- phase 3: one pool per distinct distance, ascending (:113-157, :247-293) ----------------
    auto level_entry = [&](const uint32_t i, int& f, int& m) -> bool {  // -> Lowe's ratio holds on both sides
      if (lvl_lds) {
        const uint32_t x = lvl[i];
        f = (int) (x & 0x1fffu);
        m = (int) ((x >> 13) & 0xffffu);
        return (x >> 29) != 0u;
      }
      const uint2 c = by_level[i];
      f = (int) (c.x & 0xffffu);
      m = (int) (c.x >> 16);
      return c.y != 0u;
    };
    // Two barriers per level.  The pool counts of consecutive non-empty levels live in the two 16-bit halves of a count word (a count
    // never exceeds the other cloud's size <= 65535): while level d is counted into its half, the entries of the previous non-empty
    // level clear theirs (they were last read before the barrier that ended that level), so no third pass and barrier is needed to
    // reset them; and a candidate that registers may raise its flags at once -- its pool counts are 1 / 1, no other member of the
    // pool carries either index, and later levels only start behind the barrier.
    int half = 0;
    uint32_t p0 = 0, p1 = 0;  // the previous non-empty level's entries
    for (int d = 0; d < a.lim && d < kBfLevels; ++d) {
      const uint32_t s0 = lstart[d], s1 = lstart[d + 1];
      if (s0 == s1) {
        continue;  // uniform
      }
      const uint32_t one = 1u << (16 * half), keep_mine = 0xffffu << (16 * half);
      for (uint32_t i = p0 + tid; i < p1; i += kBfThreads) {
        int f, m;
        (void) level_entry(i, f, m);
        atomicAnd(&cnt_f[f], keep_mine);  // (the other half: this level's counts may already be growing in the same word)
        atomicAnd(&cnt_m[m], keep_mine);
      }
      // pool = candidates of this distance whose indices are both unregistered (:117-118)
      for (uint32_t i = s0 + tid; i < s1; i += kBfThreads) {
        int f, m;
        (void) level_entry(i, f, m);
        if (!reg_f[f] && !reg_m[m]) {
          atomicAdd(&cnt_f[f], one);
          atomicAdd(&cnt_m[m], one);
        }
      }
      __syncthreads();
      // unique in the pool (:256-266) + Lowe on both sides -> registered (:285-289)
      for (uint32_t i = s0 + tid; i < s1; i += kBfThreads) {
        int f, m;
        const bool ok = level_entry(i, f, m);
        if (!reg_f[f] && !reg_m[m] && ((cnt_f[f] >> (16 * half)) & 0xffffu) == 1u && ((cnt_m[m] >> (16 * half)) & 0xffffu) == 1u && ok) {
          acc[f]   = (uint32_t) m | ((uint32_t) d << 16);
          reg_f[f] = 1;
          reg_m[m] = 1;
          atomicAdd(&hist2[d], 1u);
        }
      }
      __syncthreads();
      p0   = s0;
      p1   = s1;
      half ^= 1;
    }

    // ---- phase 4: emit ordered by (distance, fixed index) ---------------------------------------
    if (wave == 0) {
      uint32_t carry = 0;
      for (int base = 0; base < kBfLevels; base += 64) {
        const uint32_t h   = hist2[base + lane];
        uint32_t incl      = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(incl, o, 64);
          if (lane >= o) {
            incl += v;
          }
        }
        lfill[base + lane] = carry + incl - h;  // first output slot of the level
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) {
        misc[1] = carry;
      }
    }
    __syncthreads();
    for (int d = wave; d < a.lim && d < kBfLevels; d += kBfWaves) {
      if (hist2[d] == 0) {
        continue;
      }
      uint32_t running = lfill[d];
      for (int base = 0; base < nf; base += 64) {
        const int f      = base + lane;
        const uint32_t r = f < nf ? acc[f] : 0xffffffffu;
        const bool mine  = r != 0xffffffffu && (int) (r >> 16) == d;
        const uint64_t mask = __ballot(mine);
        if (mine) {
          const uint32_t slot = running + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
          prs_corr c;
          c.fixed_idx  = f;
          c.moving_idx = (int) (r & 0xffffu);
          c.response   = (float) d;
          out[slot]    = c;
        }
        running += (uint32_t) __popcll(mask);
      }
    }
    if (tid == 0) {
      const uint32_t n = misc[1];
      a.b.n_matches[frame] = (int) n;
      a.b.status[frame]    = status | (n == 0 ? PRS_WARN_NO_MATCHES : 0);  // :237-242
    }
    __syncthreads();
  }
}

// ---- round 6: the dense phase on the matrix cores, split shape (forced mode PRS_BF_DENSE_MATRIX below a full batch) ---------
// hamming(a, b) = pop(a) + pop(b) - 2 a.b.  With the fixed rows as 0 / 1 bytes and the moving rows as +1 / -1 bytes (b' = 1 - 2 b),
//     sum_k a_k b'_k = pop(a) - 2 a.b,     so     hamming(a, b) = (A B'^T)[a][b] + pop(b):
// a 16 x 16 tile of distances is four v_mfma_i32_16x16x64_i8 (K = 256 bits) and "candidate" (d < lim, bruteforce_impl.cpp:52) is
// acc < lim - pop(b), one compare per accumulator against a per-lane threshold (a lane's four accumulators share their column).
// Integer products and sums: exact, the candidate set is the one the popcount kernel finds.
// Shape: 512 threads = 8 waves per workgroup, a wave owns 64 fixed rows (4 A tiles, expanded once into 64 registers), the workgroup
// walks the moving cloud in chunks of 64 rows that all waves expand into LDS (double-buffered, one barrier per chunk: 64 MFMAs per
// wave).  The operand layout inside K is free as long as A and B agree (a dot product does not care about the order of its terms):
// lane (i = l & 15, g = l >> 4) holds, for K block kb, the 16 bits [64 kb + 16 g, +16) of row i as 16 bytes; the result layout is the
// one tools/probes/mfma_i8_probe.hip pins (register r of lane l = row 4 (l >> 4) + r, column l & 15).
// Candidates go through the same global counters / bitmaps / list as the split popcount shape (MODE kBfDense);
// bruteforce_kernel<.., kBfRegister> then registers them pair by pair.  Built for one candidate per fixed point (uniform random
// rows): the entries it parks are re-scored from memory at a flush behind barriers, which real descriptors make the bulk of its time.
typedef int bf_v4i __attribute__((ext_vector_type(4)));
constexpr int kBfmThreads  = 512;                                  // 8 waves; two workgroups per CU: one scores while the other flushes / waits at its barrier
constexpr int kBfmRowsWave = 64;                                   // fixed rows per wave (4 A tiles)
constexpr int kBfmRowsWg   = kBfmRowsWave * (kBfmThreads / 64);    // 1024 fixed rows per workgroup
constexpr int kBfmChunk    = 64;                                   // moving rows per LDS chunk (4 B tiles; 128 rows per chunk: two more registers, spills, slower)
// LDS image of a chunk: four planes (one per 16-bit slice g of a 64-bit K block), a row of a plane = its four K blocks (64 B) + 16 B of
// pad.  A ds_read_b128 is served in groups of 16 lanes that mix two values of g ({0-3, 12-15, 20-27}, ...): with the planes a multiple
// of 256 B apart the bank of a lane depends on its row alone, and 16 rows at an 80-byte stride cover the 64 banks exactly once
// (rows of 272 B with the slices side by side: a two-way conflict in every group, SQ_LDS_BANK_CONFLICT = a third of the LDS cycles)
constexpr int kBfmPlaneRow = 64 + 16;
constexpr int kBfmPlane    = kBfmChunk * kBfmPlaneRow;
constexpr int kBfmWaveList = 1024;                                 // (column, row group, tiles) entries a WAVE parks in its own LDS segment between flushes (8 x 4 KB)
constexpr int kBfmFlushAt  = 256;                                  // a flush is due when a segment holds this many at the end of a chunk

// where a candidate of one cloud pair goes: the pair's list, bitmaps and counters (what MODE kBfDense of the popcount kernel
// updates per candidate).  Passed BY VALUE to the out-of-line overflow path: a reference to the kernel's argument block would
// force a copy of it onto the stack, and every later read of it through scratch memory.
struct BfSink {
  uint2* cand;
  uint32_t *bm_f, *bm_m, *g_cnt_f, *g_cnt_m, *g_hist, *g_total;
  int nw, cap;
};
__device__ __forceinline__ void bf_publish(const BfSink k, const uint2 e, const uint32_t slot, const bool with_histogram = true) {
  const int f = (int) (e.x & 0xffffu), m = (int) (e.x >> 16), d = (int) e.y;
  if (slot < (uint32_t) k.cap) {
    k.cand[slot] = e;
  }
  const uint32_t bit = 1u << (d & 31);
  atomicOr(&k.bm_f[f * k.nw + (d >> 5)], bit);
  atomicOr(&k.bm_m[m * k.nw + (d >> 5)], bit);
  atomicAdd(&k.g_cnt_f[f], 1u);
  atomicAdd(&k.g_cnt_m[m], 1u);
  if (with_histogram) {
    atomicAdd(&k.g_hist[d], 1u);
  }
}
// The waves' parked entries -> the pair's candidate list (all threads, behind the barrier that published the waves' counts `wcnt`).
// Out of line: inlined, its registers pushed the scoring loop of the kernel into spills.  The eight segments are walked as one list
// (entry v of the concatenation -> segment by the running sums of the counts), ONE ENTRY PER THREAD: the 4 pairs (lane's 4 rows of a
// tile x the entry's column) of the entry's lowest marked tile are scored exactly from the packed rows -- the candidate test of the
// reference (bruteforce_impl.cpp:52); the matrix-core distances only SELECTED the entry -- then the next marked tile while any lane of
// the wave has one.  The scoring loop parks one entry per (lane, tile), so the first pass has every lane busy; sixteen threads per
// entry, one pair each behind the tile mask, ran the scoring code once per 4 entries with 3 lanes in 4 idle (a sixth of the kernel's
// vector instructions).  Candidates of a pass take ONE slot range per wave from the pair's global counter and one add per distance
// level and flush (a returning atomic per candidate on the one counter of a cloud pair serialises in the L2: 0.4 of 1.3 ms).
__device__ __attribute__((noinline)) void bf_flush(const BfSink sink, const uint32_t* gdf, const uint32_t* gdm, const int nf, const int lim, const uint32_t* clist,
                                                   uint32_t* lhist, const uint32_t* wcnt) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t end[kBfmThreads / 64];  // running sums of the waves' counts
  uint32_t n = 0;
#pragma unroll
  for (int w = 0; w < kBfmThreads / 64; ++w) {
    n += wcnt[w];
    end[w] = n;
  }
  if (n == 0) {
    return;  // (block-uniform)
  }
  static_assert(kBfLevels <= kBfmThreads, "one thread per distance level");
  if (tid < kBfLevels) {
    lhist[tid] = 0;
  }
  __syncthreads();
  for (uint32_t v0 = 0; v0 < n; v0 += kBfmThreads) {
    const uint32_t v = v0 + (uint32_t) tid;
    uint32_t entry   = 0;
    if (v < n) {
      uint32_t seg = 0, first = 0;
#pragma unroll
      for (int w = 0; w + 1 < kBfmThreads / 64; ++w) {
        if (v >= end[w]) {
          seg   = (uint32_t) w + 1u;
          first = end[w];
        }
      }
      entry = clist[seg * kBfmWaveList + (v - first)];
    }
    uint32_t tiles = entry >> 28;
    const int m    = (int) (entry & 0xffffu);
    const int f00  = (int) (((entry >> 16) & 0x7ffu) << 2);
    uint4 m0 = make_uint4(0u, 0u, 0u, 0u), m1 = m0;
    if (tiles != 0u) {
      const uint4* pm = reinterpret_cast<const uint4*>(gdm + 8 * m);
      m0 = pm[0];
      m1 = pm[1];
    }
    while (__any(tiles != 0u)) {  // (wave-uniform)
      int d[4]    = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
      int f_first = 0;
      if (tiles != 0u) {
        const int t = __ffs((int) tiles) - 1;
        tiles &= tiles - 1u;
        f_first = f00 + 16 * t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (f_first + r < nf) {
            const uint4* pf = reinterpret_cast<const uint4*>(gdf + 8 * (f_first + r));
            const uint4 a0 = pf[0], a1 = pf[1];
            d[r] = __popc(a0.x ^ m0.x) + __popc(a0.y ^ m0.y) + __popc(a0.z ^ m0.z) + __popc(a0.w ^ m0.w) + __popc(a1.x ^ m1.x) + __popc(a1.y ^ m1.y) +
                   __popc(a1.z ^ m1.z) + __popc(a1.w ^ m1.w);
          }
        }
      }
      const uint32_t mine = (d[0] < lim ? 1u : 0u) + (d[1] < lim ? 1u : 0u) + (d[2] < lim ? 1u : 0u) + (d[3] < lim ? 1u : 0u);
      uint32_t incl       = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) {
          incl += up;
        }
      }
      const uint32_t total = __shfl(incl, 63, 64);
      if (total != 0u) {  // (wave-uniform)
        uint32_t base = 0;
        if (lane == 0) {
          base = atomicAdd(sink.g_total, total);
        }
        uint32_t slot = (uint32_t) __shfl((int) base, 0, 64) + incl - mine;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (d[r] < lim) {
            bf_publish(sink, make_uint2((uint32_t) (f_first + r) | ((uint32_t) m << 16), (uint32_t) d[r]), slot++, false);
            atomicAdd(&lhist[d[r]], 1u);
          }
        }
      }
    }
  }
  __syncthreads();
  if (tid < kBfLevels && lhist[tid] != 0) {
    atomicAdd(&sink.g_hist[tid], lhist[tid]);
  }
  __syncthreads();  // (the histogram and the segments are free again)
}

__global__ __launch_bounds__(kBfmThreads, 4) void bruteforce_dense_mfma_kernel(const BfArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char bbuf[2][4 * kBfmPlane];
  static_assert(kBfmPlane % 256 == 0, "planes must not shift the banks");
  __shared__ int popm[2][kBfmChunk];
  __shared__ uint32_t lut_a[16], lut_b[16];  // 4 bits -> 4 bytes: 0 / 1 (fixed side), +1 / -1 (moving side)
  // candidates wait here for a bulk flush: half of a wave's tile rows meet one (a candidate per fixed row and cloud pair is one
  // per 1024 pairs), and a slot from the GLOBAL counter costs the wave a trip to memory the matrix pipe idles through.  Every wave
  // appends to its OWN segment and keeps its count in a scalar register: no atomic, no wait in the scoring loop (a shared list paid a
  // returning LDS atomic per tile row that met the threshold).  The counts meet in `wcnt` once per chunk, double-buffered by chunk
  // parity so that ONE barrier per chunk serves both the staged rows and the uniform "flush now" decision.
  __shared__ uint32_t clist[(kBfmThreads / 64) * kBfmWaveList];  // moving row | (first fixed row of the lane's 16) / 4 << 16 | tiles that met the threshold << 28
  __shared__ uint32_t wcnt[2][kBfmThreads / 64];
  __shared__ uint32_t lhist[kBfLevels];  // candidates of a flush by distance (one global add per level and flush)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int frame = (int) blockIdx.y;
  int nf = a.b.n_fixed[frame];
  int nm = a.b.n_moving[frame];
  nf     = nf < 0 ? 0 : (nf > a.b.fixed_stride ? a.b.fixed_stride : nf);
  nm     = nm < 0 ? 0 : (nm > a.b.moving_stride ? a.b.moving_stride : nm);
  const int row0_wg = (int) blockIdx.x * kBfmRowsWg;
  if (row0_wg >= nf || nm == 0) {
    return;  // (block-uniform)
  }
  const uint32_t* __restrict__ gdf = reinterpret_cast<const uint32_t*>(a.b.fixed_desc + (size_t) frame * a.b.fixed_stride * PRS_DESC_BYTES);
  const uint32_t* __restrict__ gdm = reinterpret_cast<const uint32_t*>(a.b.moving_desc + (size_t) frame * a.b.moving_stride * PRS_DESC_BYTES);
  BfSink sink;
  sink.cand    = a.cand + (size_t) frame * a.cap;
  sink.bm_f    = a.bitmaps + (size_t) frame * (size_t) (a.b.fixed_stride + a.b.moving_stride) * a.nw;
  sink.bm_m    = sink.bm_f + (size_t) a.b.fixed_stride * a.nw;
  sink.g_cnt_f = a.g_acc + (size_t) frame * (size_t) (a.b.fixed_stride + a.b.moving_stride + kBfLevels + 8);
  sink.g_cnt_m = sink.g_cnt_f + a.b.fixed_stride;
  sink.g_hist  = sink.g_cnt_m + a.b.moving_stride;
  sink.g_total = sink.g_hist + kBfLevels;
  sink.nw      = a.nw;
  sink.cap     = a.cap;

  if (tid < 16) {
    uint32_t v01 = 0, vpm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      v01 |= ((tid >> b) & 1 ? 0x01u : 0x00u) << (8 * b);
      vpm |= ((tid >> b) & 1 ? 0xffu : 0x01u) << (8 * b);
    }
    lut_a[tid] = v01;
    lut_b[tid] = vpm;
  }
  __syncthreads();
  auto expand16 = [](const uint32_t* lut, const uint32_t bits16) -> bf_v4i {
    bf_v4i v;
    v.x = (int) lut[bits16 & 15u];
    v.y = (int) lut[(bits16 >> 4) & 15u];
    v.z = (int) lut[(bits16 >> 8) & 15u];
    v.w = (int) lut[(bits16 >> 12) & 15u];
    return v;
  };
  // ---- this wave's fixed rows: A[t][kb] = bits [64 kb + 16 lg, +16) of row row0 + 16 t + li (rows past the end: zeros) ----
  // Thread-index values are RE-DERIVED where the chunk loop uses them (the lane from v_mbcnt behind a volatile asm the compiler cannot
  // hoist, the wave in a scalar register): kept live beside the 64 registers of A they were spilled, and a reload from scratch ahead of
  // the scoring loop waits for the vector-memory counter -- i.e. for the NEXT chunk's words, requested to be in flight across it.
  auto lane_now = []() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const int wave_s     = __builtin_amdgcn_readfirstlane(wave);
  const int row0       = row0_wg + wave_s * kBfmRowsWave;
  const bool wave_live = row0 < nf;
  bf_v4i A[4][4];
  auto build_a = [&]() {
    const int l = lane_now();
    uint32_t w[4][4];  // (all sixteen words requested before the first is expanded: rows past the end read row nf - 1 and are zeroed)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int f  = row0 + 16 * t + (l & 15);
      const int fr = f < nf ? f : nf - 1;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        w[t][kb] = gdf[8 * fr + 2 * kb + (l >> 5)];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool live = row0 + 16 * t + (l & 15) < nf;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        A[t][kb] = expand16(lut_a, live ? (w[t][kb] >> (16 * ((l >> 4) & 1))) & 0xffffu : 0u);
      }
    }
  };
  build_a();
  // ---- the moving cloud, chunk by chunk ----
  const int n_chunks = (nm + kBfmChunk - 1) / kBfmChunk;
  // A chunk = 64 rows x 8 words = one 32-bit word per thread: thread (row = tid >> 3, word j = tid & 7) owns bits [32 j, +32) of moving
  // row 64 c + row = the two 16-bit pieces (K block j >> 1, slices 2 (j & 1) and 2 (j & 1) + 1: the same spot of two neighbouring planes).
  // One coalesced load at (chunk base + 4 tid), one population count reduced over the 8 lanes of a row.  The word is REQUESTED before a
  // chunk is scored and EXPANDED behind it (fetch / stage): requested and consumed back to back, every wave of the workgroup sits out a
  // trip to memory per chunk with the matrix pipe idle.  (Two 16-bit pieces per thread from two rows, the first shape of this kernel,
  // spent ~110 of a wave's ~235 vector instructions per chunk on staging -- and the kernel is bound by vector issue, not by the matrix
  // pipe: tools/probes/mfma_valu_overlap_probe.hip.)
  static_assert(kBfmThreads == kBfmChunk * 8, "one word of the chunk per thread");
  const uint32_t last_word = 32u * (uint32_t) nm - 4u;  // (byte offset of the cloud's last word: rows past the end read it and are masked)
  auto fetch = [&](const int c) -> uint32_t {
    const uint32_t off = (uint32_t) c * (kBfmChunk * 32u) + 256u * (uint32_t) wave_s + 4u * (uint32_t) lane_now();
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(gdm) + (off < last_word ? off : last_word));
  };
  auto stage = [&](const int c, const int buf, const uint32_t w) {
    const int l = lane_now(), row = 8 * wave_s + (l >> 3), j = l & 7;
    unsigned char* dst = &bbuf[buf][(2 * (j & 1)) * kBfmPlane + __mul24(row, kBfmPlaneRow) + 16 * (j >> 1)];
    *reinterpret_cast<bf_v4i*>(dst)             = expand16(lut_b, w & 0xffffu);
    *reinterpret_cast<bf_v4i*>(dst + kBfmPlane) = expand16(lut_b, w >> 16);
    // pop(b) of the row: its 8 words sit on 8 neighbouring lanes
    int pop = __popc(w);
    pop += __builtin_amdgcn_update_dpp(0, pop, 0xb1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
    pop += __builtin_amdgcn_update_dpp(0, pop, 0x4e, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
    pop += __builtin_amdgcn_update_dpp(0, pop, 0x141, 0xf, 0xf, true);  // row_half_mirror
    if (j == 0) {
      popm[buf][row] = c * kBfmChunk + row < nm ? pop : (1 << 20);  // a row past the end never meets the threshold
    }
  };
  uint32_t my_count = 0;  // entries in this wave's segment (wave-uniform: a scalar register)
  auto flush = [&](const int parity) {
    bf_flush(sink, gdf, gdm, nf, a.lim, clist, lhist, wcnt[parity]);
    my_count = 0;
  };
  stage(0, 0, fetch(0));
  __syncthreads();
  const bf_v4i zero = {0, 0, 0, 0};
  for (int c = 0; c < n_chunks; ++c) {
    const int buf = c & 1;
    uint32_t w_next = 0u;
    if (c + 1 < n_chunks) {
      w_next = fetch(c + 1);  // in flight while this chunk is scored
    }
    if (wave_live) {
      const int l = lane_now(), li = l & 15, lg = l >> 4;
#pragma unroll 1
      for (int bt = 0; bt < kBfmChunk / 16; ++bt) {
        if (c * kBfmChunk + 16 * bt >= nm) {
          break;  // (uniform) tiles past the end of the moving cloud
        }
        const unsigned char* brow = &bbuf[buf][__mul24(lg, kBfmPlane) + __mul24(16 * bt + li, kBfmPlaneRow)];
        const int pop_b           = popm[buf][16 * bt + li];
        const int thr             = a.lim - pop_b;  // candidate  <=>  acc < thr
        bf_v4i B[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          B[kb] = *reinterpret_cast<const bf_v4i*>(brow + 16 * kb);
        }
        bf_v4i acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][0], B[0], zero, 0, 0, 0);
        }
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][kb], B[kb], acc[t], 0, 0, 0);
          }
        }
        bool any_t[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          any_t[t] = (acc[t].x < thr) | (acc[t].y < thr) | (acc[t].z < thr) | (acc[t].w < thr);
        }
        // A lane whose column met the threshold in one of its 4 rows of a tile parks (column, row group, tile): the 4 pairs are scored
        // again, exactly, from the packed rows when the list is flushed.  No per-pair branch here: half of a wave's tile rows hold a
        // candidate (one per fixed row and cloud pair = one per 1024 pairs), and telling the accumulators of a lane apart in this
        // loop cost the matrix pipe a third of its time.  One entry per (lane, tile) while the segment has room for a worst-case tile
        // row (4 x 64), else one per lane with all its tiles marked: a chunk then adds at most 256 + 256 + 64 + 64 to a count below
        // kBfmFlushAt, so a segment never overflows.
        const unsigned long long mask_t[4] = {__ballot(any_t[0]), __ballot(any_t[1]), __ballot(any_t[2]), __ballot(any_t[3])};
        const unsigned long long anymask   = mask_t[0] | mask_t[1] | mask_t[2] | mask_t[3];
        if (anymask != 0ull) {  // (wave-uniform)
          const int lane            = lane_now();
          const uint32_t base_entry = (uint32_t) (c * kBfmChunk + 16 * bt + (lane & 15)) | ((uint32_t) ((row0 + 4 * (lane >> 4)) >> 2) << 16);
          uint32_t* segment         = &clist[wave_s * kBfmWaveList];
          auto rank = [](const unsigned long long mk) -> uint32_t {  // set bits below this lane
            return __builtin_amdgcn_mbcnt_hi((uint32_t) (mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mk, 0u));
          };
          if (my_count <= (uint32_t) kBfmWaveList / 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (mask_t[t] != 0ull) {  // (wave-uniform)
                if (any_t[t]) {
                  segment[my_count + rank(mask_t[t])] = base_entry | (1u << (28 + t));
                }
                my_count += (uint32_t) __popcll(mask_t[t]);
              }
            }
          } else {
            if (any_t[0] | any_t[1] | any_t[2] | any_t[3]) {
              const uint32_t tmask = (any_t[0] ? 1u : 0u) | (any_t[1] ? 2u : 0u) | (any_t[2] ? 4u : 0u) | (any_t[3] ? 8u : 0u);
              segment[my_count + rank(anymask)] = base_entry | (tmask << 28);
            }
            my_count += (uint32_t) __popcll(anymask);
          }
        }
      }
    }
    if (c + 1 < n_chunks) {
      stage(c + 1, buf ^ 1, w_next);
    }
    if (lane_now() == 0) {
      wcnt[buf][wave_s] = my_count;
    }
    __syncthreads();  // chunk c + 1 is staged, chunk c is consumed, the counts of this chunk's parity are published
    // (the next write to wcnt[buf] is two chunks away, behind the next barrier: every wave reads the same eight counts)
    bool crowded = false;
#pragma unroll
    for (int w = 0; w < kBfmThreads / 64; ++w) {
      crowded |= wcnt[buf][w] >= (uint32_t) kBfmFlushAt;
    }
    if (crowded) {
      flush(buf);
      build_a();  // (A is not kept across the call: 64 registers the calling convention would spill and reload around it)
    }
  }
  // (after the loop the last barrier has published the final counts in wcnt[(n_chunks - 1) & 1], zeros after a flush)
  if (lane_now() == 0) {
    wcnt[n_chunks & 1][wave_s] = my_count;
  }
  __syncthreads();
  flush(n_chunks & 1);
}

// ---- the dense phase of the FUSED shape on the matrix cores ------------------------------------------------------------
// bruteforce_kernel<1, kBfFused, true, THREADS>: one workgroup per cloud pair -- 1024 threads on 64-row chunks, or two workgroups of
// 512 threads per CU on 32-row chunks where the clouds leave room for two in the LDS --, the registration state in LDS as in the
// popcount shape, the N_f x N_m distances from v_mfma_i32_16x16x64_i8 as in bruteforce_dense_mfma_kernel (a wave owns 64 fixed rows,
// the workgroup walks the moving cloud in chunks that its first CHUNK / 8 waves expand into LDS, one 32-bit word per thread), fixed
// rows beyond THREADS in further passes.
// What differs is what happens to a candidate.  Real descriptors put 1.6 % of the pairs below a threshold of 50 bits: every tile of
// every tile row holds one, so nothing here is rare.  A lane that met the threshold in one of its 4 rows of a tile parks ONE 16-byte
// entry in its wave's LDS segment: (column, row group, tile), pop(b) and the tile's four accumulators as 16-bit halves (two v_perm; the
// distances acc + pop(b) are EXACT) -- six vector instructions per visited tile, where packing the distances to bytes took twelve.  A
// wave drains its own segment when a worst-case pair of tiles (2 x 64 entries) might not fit: one slot range per 64 entries from the
// pair's LDS counter, the bitmaps / counts / histogram updated by LDS atomics.  No re-scoring from memory, no barrier, no global
// atomic: the split kernel's flush spent 1.3 of its 1.7 ms per 1024 real cloud pairs on those.
constexpr int kMxSeg        = 192;  // entries of a wave's segment (16 bytes each): drained when more than kMxSeg - 128 wait in it
constexpr uint32_t bf_mx_bytes(const int threads, const int chunk) {  // LDS scratch of bf_matrix_phase1<threads, chunk>
  return 2u * 4u * (uint32_t) (chunk * kBfmPlaneRow) + 2u * (uint32_t) chunk * 4u + 2u * 16u * 4u + (uint32_t) (threads / 64) * kMxSeg * 16u;
}
constexpr uint32_t kMxBytes = bf_mx_bytes(kBfThreads, 64), kMxBytesDual = bf_mx_bytes(512, 32);

template <int THREADS, int CHUNK>
__device__ __forceinline__ void bf_matrix_phase1(const BfArgs& a, unsigned char* scratch, const int nf, const int nm, const uint32_t* __restrict__ gdf,
                                                 const uint32_t* __restrict__ gdm, uint2* __restrict__ cand, uint32_t* cnt_f, uint32_t* cnt_m, uint32_t* hist,
                                                 uint32_t* counter, uint32_t* lbm_f, uint32_t* lbm_m) {
  constexpr int kPlane = CHUNK * kBfmPlaneRow;  // a plane of the chunk image (CHUNK rows at the 80-byte stride: a multiple of 256 B)
  static_assert(kPlane % 256 == 0 && CHUNK % 16 == 0 && (CHUNK * 8) % 64 == 0 && CHUNK * 8 <= THREADS, "chunk shape");
  unsigned char* bbuf = scratch;                                                      // [2][4 planes][CHUNK rows x 80 B]
  int* popm           = reinterpret_cast<int*>(scratch + 2 * 4 * kPlane);             // [2][CHUNK]
  uint32_t* lut_a     = reinterpret_cast<uint32_t*>(popm + 2 * CHUNK);                // [16]
  uint32_t* lut_b     = lut_a + 16;                                                   // [16]
  uint4* segments     = reinterpret_cast<uint4*>(lut_b + 16);                         // [waves][kMxSeg]
  auto lane_now = []() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const int tid    = threadIdx.x;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 16) {
    uint32_t v01 = 0, vpm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      v01 |= ((tid >> b) & 1 ? 0x01u : 0x00u) << (8 * b);
      vpm |= ((tid >> b) & 1 ? 0xffu : 0x01u) << (8 * b);
    }
    lut_a[tid] = v01;
    lut_b[tid] = vpm;
  }
  __syncthreads();
  if (nf == 0 || nm == 0) {
    return;  // (block-uniform)
  }
  auto expand16 = [](const uint32_t* lut, const uint32_t bits16) -> bf_v4i {
    bf_v4i v;
    v.x = (int) lut[bits16 & 15u];
    v.y = (int) lut[(bits16 >> 4) & 15u];
    v.z = (int) lut[(bits16 >> 8) & 15u];
    v.w = (int) lut[(bits16 >> 12) & 15u];
    return v;
  };
  uint4* segment    = segments + wave_s * kMxSeg;
  uint32_t my_count = 0;  // entries in this wave's segment (wave-uniform: a scalar register)
  // this wave's parked entries -> the pair's candidate list and registration state (:52-69)
  auto drain = [&]() {
    for (uint32_t i0 = 0; i0 < my_count; i0 += 64u) {
      const int lane   = lane_now();
      const uint32_t i = i0 + (uint32_t) lane;
      const uint4 e    = i < my_count ? segment[i] : make_uint4(0u, 1u << 20, 0u, 0u);  // (pop(b) of a row past the end: above every threshold)
      const int m      = (int) (e.x & 0xffffu);
      const int f0     = (int) (((e.x >> 16) & 0x7ffu) << 2) + 16 * (int) ((e.x >> 27) & 3u);
      int d[4];
      d[0] = (int) (short) (e.z & 0xffffu) + (int) e.y;
      d[1] = (int) (short) (e.z >> 16) + (int) e.y;
      d[2] = (int) (short) (e.w & 0xffffu) + (int) e.y;
      d[3] = (int) (short) (e.w >> 16) + (int) e.y;
      uint32_t mine = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (d[r] >= a.lim || f0 + r >= nf) {
          d[r] = -1;  // (a row past the end scores as an all-zero row)
        } else {
          ++mine;
        }
      }
      uint32_t incl = mine;
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x111, 0xf, 0xf, false);  // row_shr:1
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x112, 0xf, 0xf, false);  // row_shr:2
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x114, 0xf, 0xf, false);  // row_shr:4
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x118, 0xf, 0xf, false);  // row_shr:8
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
      incl += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
      const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
      if (total == 0u) {
        continue;  // (wave-uniform)
      }
      uint32_t base = 0;
      if (lane == 0) {
        base = atomicAdd(counter, total);
      }
      uint32_t slot = (uint32_t) __builtin_amdgcn_readfirstlane((int) base) + incl - mine;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (d[r] >= 0) {
          const int f = f0 + r;
          if (slot < (uint32_t) a.cap) {
            cand[slot] = make_uint2((uint32_t) f | ((uint32_t) m << 16), (uint32_t) d[r]);
          }
          ++slot;
          const uint32_t bit = 1u << (d[r] & 31);
          atomicOr(&lbm_f[f * a.nw + (d[r] >> 5)], bit);
          atomicOr(&lbm_m[m * a.nw + (d[r] >> 5)], bit);
          atomicAdd(&cnt_f[f], 1u);
          atomicAdd(&cnt_m[m], 1u);
          atomicAdd(&hist[d[r]], 1u);
        }
      }
    }
    my_count = 0;
  };
  const int n_chunks       = (nm + CHUNK - 1) / CHUNK;
  const uint32_t last_word = 32u * (uint32_t) nm - 4u;  // (byte offset of the cloud's last word: rows past the end read it and are masked)
  auto fetch = [&](const int c) -> uint32_t {           // waves 0-7: one 32-bit word of the chunk per thread
    const uint32_t off = (uint32_t) c * (CHUNK * 32u) + 256u * (uint32_t) wave_s + 4u * (uint32_t) lane_now();
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(gdm) + (off < last_word ? off : last_word));
  };
  auto stage = [&](const int c, const int buf, const uint32_t w) {
    const int l = lane_now(), row = 8 * wave_s + (l >> 3), j = l & 7;
    unsigned char* dst = bbuf + buf * (4 * kPlane) + (2 * (j & 1)) * kPlane + __mul24(row, kBfmPlaneRow) + 16 * (j >> 1);
    *reinterpret_cast<bf_v4i*>(dst)             = expand16(lut_b, w & 0xffffu);
    *reinterpret_cast<bf_v4i*>(dst + kPlane) = expand16(lut_b, w >> 16);
    int pop = __popc(w);
    pop += __builtin_amdgcn_update_dpp(0, pop, 0xb1, 0xf, 0xf, true);   // quad_perm [1, 0, 3, 2]
    pop += __builtin_amdgcn_update_dpp(0, pop, 0x4e, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
    pop += __builtin_amdgcn_update_dpp(0, pop, 0x141, 0xf, 0xf, true);  // row_half_mirror
    if (j == 0) {
      popm[buf * CHUNK + row] = c * CHUNK + row < nm ? pop : (1 << 20);  // a row past the end never meets the threshold
    }
  };
  const bool stager = wave_s < CHUNK * 8 / 64;  // (512 words per chunk)
  const bf_v4i zero = {0, 0, 0, 0};
  for (int pass_first = 0; pass_first < nf; pass_first += THREADS) {
    const int row0       = pass_first + wave_s * kBfmRowsWave;
    const bool wave_live = row0 < nf;
    bf_v4i A[4][4];
    {
      const int l = lane_now();
      uint32_t w[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f  = row0 + 16 * t + (l & 15);
        const int fr = f < nf ? f : nf - 1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          w[t][kb] = gdf[8 * fr + 2 * kb + (l >> 5)];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool live = row0 + 16 * t + (l & 15) < nf;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          A[t][kb] = expand16(lut_a, live ? (w[t][kb] >> (16 * ((l >> 4) & 1))) & 0xffffu : 0u);
        }
      }
    }
    __syncthreads();  // (the previous pass has left the chunk buffers)
    if (stager) {
      stage(0, 0, fetch(0));
    }
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
      const int buf   = c & 1;
      uint32_t w_next = 0u;
      if (stager && c + 1 < n_chunks) {
        w_next = fetch(c + 1);  // in flight while this chunk is scored
      }
      if (wave_live) {
        const int l = lane_now(), li = l & 15, lg = l >> 4;
#pragma unroll 1
        for (int bt = 0; bt < CHUNK / 16; ++bt) {
          if (c * CHUNK + 16 * bt >= nm) {
            break;  // (uniform) tiles past the end of the moving cloud
          }
          const unsigned char* brow = bbuf + buf * (4 * kPlane) + __mul24(lg, kPlane) + __mul24(16 * bt + li, kBfmPlaneRow);
          const int pop_b           = popm[buf * CHUNK + 16 * bt + li];
          const int thr             = a.lim - pop_b;  // candidate  <=>  acc < thr
          bf_v4i B[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) {
            B[kb] = *reinterpret_cast<const bf_v4i*>(brow + 16 * kb);
          }
          bf_v4i acc[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][0], B[0], zero, 0, 0, 0);
          }
#pragma unroll
          for (int kb = 1; kb < 4; ++kb) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][kb], B[kb], acc[t], 0, 0, 0);
            }
          }
          bool any_t[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            any_t[t] = (acc[t].x < thr) | (acc[t].y < thr) | (acc[t].z < thr) | (acc[t].w < thr);
          }
          const unsigned long long mask_t[4] = {__ballot(any_t[0]), __ballot(any_t[1]), __ballot(any_t[2]), __ballot(any_t[3])};
          if ((mask_t[0] | mask_t[1] | mask_t[2] | mask_t[3]) != 0ull) {  // (wave-uniform)
            const int lane            = lane_now();
            const uint32_t base_entry = (uint32_t) (c * CHUNK + 16 * bt + (lane & 15)) | ((uint32_t) ((row0 + 4 * (lane >> 4)) >> 2) << 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if ((t & 1) == 0 && (mask_t[t] | mask_t[t + 1]) != 0ull && my_count > (uint32_t) (kMxSeg - 128)) {
                drain();  // (wave-uniform: room for a worst-case pair of tiles)
              }
              if (mask_t[t] != 0ull) {  // (wave-uniform)
                if (any_t[t]) {
                  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (mask_t[t] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask_t[t], 0u));
                  segment[my_count + rank] = make_uint4(base_entry | ((uint32_t) t << 27), (uint32_t) pop_b,
                                                        __builtin_amdgcn_perm((uint32_t) acc[t].y, (uint32_t) acc[t].x, 0x05040100u),
                                                        __builtin_amdgcn_perm((uint32_t) acc[t].w, (uint32_t) acc[t].z, 0x05040100u));
                }
                my_count += (uint32_t) __popcll(mask_t[t]);
              }
            }
          }
        }
      }
      if (stager && c + 1 < n_chunks) {
        stage(c + 1, buf ^ 1, w_next);
      }
      __syncthreads();  // chunk c + 1 is staged, chunk c is consumed
    }
    drain();
  }
}

static inline uint32_t bf_align16(uint32_t v) {
  return (v + 15u) & ~15u;
}

int bruteforce_batch_launch(prs_context* ctx, const prs_bruteforce_params* params, const prs_bruteforce_batch* batch) {
  if (!params || !batch || !batch->fixed_desc || !batch->n_fixed) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: fixed not set");  // bruteforce_impl.cpp:204-206
  }
  if (!batch->moving_desc || !batch->n_moving) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: moving not set");  // :207-210
  }
  if (!batch->matches || !batch->n_matches || !batch->status) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_bruteforce_match: correspondences not set");  // :211-214
  }
  if (batch->batch <= 0) {
    return PRS_OK;
  }
  if (batch->fixed_stride <= 0 || batch->moving_stride <= 0 || batch->fixed_stride > 8192 || batch->moving_stride > 65535) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_bruteforce_match: fixed_stride must be in [1,8192], moving_stride in [1,65535]");
  }
  BfArgs a;
  a.b         = *batch;
  a.max_ratio = params->maximum_distance_ratio_to_second_best;
  // (float) d < maximum_descriptor_distance for integer d  <=>  d < lim
  int lim = 0;
  while (lim <= 257 && (float) lim < params->maximum_descriptor_distance) {
    ++lim;
  }
  if (lim > kBfLevels) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_bruteforce_match: maximum_descriptor_distance above 256 bits");
  }
  a.lim = lim;
  a.nw  = lim > 0 ? (lim + 31) / 32 : 1;
  const int big = batch->fixed_stride > batch->moving_stride ? batch->fixed_stride : batch->moving_stride;
  a.cap = batch->candidate_capacity > 0 ? batch->candidate_capacity : 16 * big;
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) {
      cus = prop.multiProcessorCount;
    }
  }
  // Which kernels score the pairs (prs_context_set_bruteforce_dense_phase).  PRS_BF_DENSE_MATRIX_WHEN_FULL, the default: a batch of
  // 32 or more cloud pairs takes the fused shape with its dense phase on the matrix cores
  // (bruteforce_kernel<1, kBfFused, true>: 1024 real cloud pairs 0.99 -> 0.62 ms, 1024 pairs of 2000 uniform random rows 2.75 -> 1.23 ms);
  // a handful of pairs keeps the popcount kernels, whose split shape spreads a pair over more workgroups (one real pair of 1350 points:
  // 0.11 ms against 0.23 ms).
  // LDS of the fused / registration workgroups: the registration state first
  uint32_t off = 0;
  a.off_cnt_f = off; off = bf_align16(off + (uint32_t) batch->fixed_stride * 4);
  a.off_cnt_m = off; off = bf_align16(off + (uint32_t) batch->moving_stride * 4);
  a.off_reg_f = off; off = bf_align16(off + (uint32_t) batch->fixed_stride);
  a.off_reg_m = off; off = bf_align16(off + (uint32_t) batch->moving_stride);
  a.off_acc   = off; off = bf_align16(off + (uint32_t) batch->fixed_stride * 4);
  a.off_hist  = off; off = bf_align16(off + (4 * kBfLevels + 8) * 4);
  if (off > 160u * 1024u) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_bruteforce_match: clouds do not fit the 160 KiB LDS");
  }
  const uint64_t bm_bytes  = (uint64_t) (batch->fixed_stride + batch->moving_stride) * (uint64_t) a.nw * 4u;
  const bool bm_fits       = !getenv("PRS_BF_GLOBAL_STATE") && off + bm_bytes + 4096u <= 160u * 1024u;
  const bool fused_regime  = !(batch->batch * 2 <= cus && batch->moving_stride >= 256);  // (else: few pairs, each spread over several workgroups)
  const bool matrix_forced = ctx->bf_mfma == PRS_BF_DENSE_MATRIX, matrix_when_full = ctx->bf_mfma == PRS_BF_DENSE_MATRIX_WHEN_FULL;
  // the fused shape with its dense phase on the matrix cores (bruteforce_kernel<1, kBfFused, true>): the registration state AND the
  // phase's scratch must fit the LDS
  // (by default from 32 cloud pairs on: one round of fused workgroups -- 0.14 ms for <= 256 real pairs of ~750 points -- beats the
  //  popcount kernels' split shape from ~27 pairs of that size on, earlier for larger clouds: 128 pairs 0.27 ms, 136 pairs 0.14 ms
  //  when the switch sat at half the CUs)
  const bool fused_matrix = ((matrix_forced && fused_regime) || (matrix_when_full && batch->batch >= 32 && batch->fixed_stride >= 256 && batch->moving_stride >= 64)) &&
                            bm_fits && ((off + bm_bytes + 255u) & ~(uint64_t) 255u) + kMxBytes <= 160u * 1024u;
  // (the split matrix-core kernel, bruteforce_dense_mfma_kernel + the registration launch: only when forced and the fused shape is
  //  not taken -- it is the fastest on uniform random rows and the slowest on real ones, see the header)
  const bool mfma = !fused_matrix && matrix_forced;
  // ... as TWO 512-thread workgroups per CU where the clouds leave room (each at most 80 KiB of LDS: strides up to ~1024 at a
  // threshold below 64 bits): a pair's registration phases are chains of dependent LDS trips with a sixth of the threads busy, and the
  // other workgroup's dense phase fills the CU meanwhile
  const uint32_t lds_limit_dual = 80u * 1024u - 512u;
  // (for more cloud pairs than CUs: 1024 real pairs 0.54 -> 0.47 ms, 384 pairs 0.27 -> 0.24; with one pair per CU the half-sized
  //  workgroup only takes longer, 128 real pairs 0.14 -> 0.20 ms.  PRS_BF_TWO_WORKGROUPS=1 / 0 forces it where it fits / never: tests, A-B)
  const char* two_wg = getenv("PRS_BF_TWO_WORKGROUPS");
  const bool dual = fused_matrix && (two_wg ? two_wg[0] == '1' : batch->batch > cus) &&
                    ((off + bm_bytes + 255u) & ~(uint64_t) 255u) + kMxBytesDual <= lds_limit_dual;
  const int grid = mfma ? batch->batch : (dual ? (batch->batch < 2 * cus ? batch->batch : 2 * cus) : (batch->batch < cus ? batch->batch : cus));
  // few pairs: spread the dense phase of each pair over several workgroups (slices of >= 32 moving rows: one drain block)
  a.chunks = 1;
  if (mfma) {
    a.chunks = 2;  // (any value > 1: per-pair scratch rows + global accumulators, as in the split popcount shape)
  } else if (!fused_matrix && batch->batch * 2 <= cus && batch->moving_stride >= 256) {
    int c = cus / batch->batch;
    const int most = batch->moving_stride / 32;
    a.chunks = c < most ? c : most;
  }
  const size_t b_cand = (size_t) grid * a.cap * sizeof(uint2);
  const size_t b_bm   = (size_t) grid * (size_t) (batch->fixed_stride + batch->moving_stride) * a.nw * sizeof(uint32_t);
  const size_t b_acc  = a.chunks > 1 ? (size_t) batch->batch * (size_t) (batch->fixed_stride + batch->moving_stride + kBfLevels + 8) * sizeof(uint32_t) : 0;
  a.cand     = static_cast<uint2*>(ctx_device_scratch_slot(ctx, 0, b_cand));
  a.by_level = static_cast<uint2*>(ctx_device_scratch_slot(ctx, 1, b_cand));
  a.bitmaps  = static_cast<uint32_t*>(ctx_device_scratch_slot(ctx, 2, b_bm + b_acc));
  a.g_acc    = a.bitmaps ? a.bitmaps + b_bm / sizeof(uint32_t) : nullptr;
  if (!a.cand || !a.by_level || !a.bitmaps) {
    return ctx_fail(ctx, PRS_ERR_HIP, "prs_bruteforce_match: scratch allocation failed");
  }
  // one workgroup per CU (the grid never exceeds the CUs): what the arrays above leave of the 160 KiB holds the distance bitmaps, then
  // the level lists of as many candidates as fit (pairs with more keep the 8-byte lists in global memory); the scratch of the
  // matrix-core dense phase is dead when the lists are written and lies under them
  a.off_bm  = 0xffffffffu;
  a.off_lvl = 0;
  a.lvl_cap = 0;
  a.off_mx  = 0;
  if (bm_fits) {
    a.off_bm  = off;
    off       = bf_align16(off + (uint32_t) bm_bytes);
    off       = fused_matrix ? (off + 255u) & ~255u : off;
    a.off_lvl = off;
    a.off_mx  = off;
    const uint32_t room = ((dual ? lds_limit_dual : 160u * 1024u) - off) / 4u;
    a.lvl_cap = (int) (room < (uint32_t) a.cap ? room : (uint32_t) a.cap);
    const uint32_t lists = 4u * (uint32_t) a.lvl_cap, mx = dual ? kMxBytesDual : kMxBytes;
    off += fused_matrix && mx > lists ? mx : lists;
  }
  const int kpt = (batch->fixed_stride + kBfThreads - 1) / kBfThreads;
  hipStream_t stream = ctx_stream(ctx);
  hipError_t e       = hipSuccess;
  auto launch = [&](auto kernel, dim3 g, const int threads = kBfThreads) {
    if (off > 64u * 1024u) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) off);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(kernel, g, dim3(threads), off, stream, a);
      e = hipGetLastError();
    }
  };
  auto launch_mode = [&](auto mode, dim3 g) {
    constexpr int M = decltype(mode)::value;
    if (kpt <= 1) {
      launch(bruteforce_kernel<1, M>, g);
    } else if (kpt <= 2) {
      launch(bruteforce_kernel<2, M>, g);
    } else if (kpt <= 4) {
      launch(bruteforce_kernel<4, M>, g);
    } else {
      launch(bruteforce_kernel<8, M>, g);
    }
  };
  if (a.chunks > 1) {
    const size_t acc_bytes = (size_t) batch->batch * (size_t) (batch->fixed_stride + batch->moving_stride + kBfLevels + 8) * sizeof(uint32_t);
    e = hipMemsetAsync(a.bitmaps, 0, b_bm + acc_bytes, stream);  // (the bitmaps and the accumulators behind them: one launch)
    if (e == hipSuccess && mfma) {
      hipLaunchKernelGGL(bruteforce_dense_mfma_kernel, dim3((batch->fixed_stride + kBfmRowsWg - 1) / kBfmRowsWg, batch->batch), dim3(kBfmThreads), 0, stream, a);
      e = hipGetLastError();
    } else if (e == hipSuccess) {
      launch_mode(std::integral_constant<int, kBfDense>{}, dim3(a.chunks, batch->batch));
    }
    if (e == hipSuccess) {
      launch_mode(std::integral_constant<int, kBfRegister>{}, dim3(batch->batch));
    }
  } else {
    if (dual) {
      launch(bruteforce_kernel<1, kBfFused, true, 512>, dim3(grid), 512);
    } else if (fused_matrix) {
      launch(bruteforce_kernel<1, kBfFused, true>, dim3(grid));
    } else {
      launch_mode(std::integral_constant<int, kBfFused>{}, dim3(grid));
    }
  }
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_bruteforce_match launch");
  }
  return PRS_OK;
}

}  // namespace prs
