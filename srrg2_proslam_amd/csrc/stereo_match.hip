// stereo_match.hip -- batched stereo epipolar descriptor matcher for gfx950 (MI355X).
//
// Replaces CorrespondenceFinderDescriptorBasedEpipolar<..>::compute
// (registration/correspondence_finders/correspondence_finder_descriptor_based_epipolar_impl.cpp:46-219)
// with optional fused stereo-adaptor assembly + rectified triangulation epilogue
// (sensor_processing/raw_data_preprocessor_stereo_projective.cpp:107-132,
//  mapping/triangulator_rigid_stereo.cpp:7-85).
//
// One 1024-thread workgroup owns one stereo pair; a launch covers `batch` independent frames.
//   1. keypoint coordinates are read coalesced (8 B/lane) and truncated to (row, col);
//      a per-row histogram (LDS atomics) + block scan + in-row rank gives the reference's
//      (row, col, index)-sorted feature vectors without a comparison sort;
//   2. the 256-bit descriptor rows are staged into LDS with 16 B/lane coalesced reads that stay
//      in flight while step 1 runs (each HBM byte is read exactly once);
//   3. epipolar rows are independent: one lane walks one row's short serial chain
//      (index_right = best + 1, epipolar_impl.cpp:181) scoring candidates with popcounts on LDS rows;
//   4. matches are compacted in sorted-left traversal order with a block scan and written as
//      prs_corr; the epilogue emits the (uL,vL,uR,vR) fixed cloud, its descriptors and the
//      triangulated points.
// The result is bit-identical to the sequential reference algorithm (see tests/test_stereo_match_gpu.py).
#include "prs_device.h"
#include "prs_host.h"

namespace prs {

struct StereoArgs {
  prs_stereo_params p;
  prs_stereo_batch b;
  prs_triangulator_params tri;
  int epilogue;
  int sort_cap;  // entries in each sorted/bucket array (>= stride, >= image_rows + 1)
  uint32_t off_desc_l, off_desc_r, off_sorted_l, off_sorted_r, off_bucket_l, off_bucket_r;
  uint32_t off_rowstart_l, off_rowstart_r, off_scratch;
  unsigned long long* stamps;  // diagnostic: [batch][16] shader-clock stamps of thread 0 (NULL = off)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kStereoThreads = 1024;
constexpr float kFloatMax    = 3.402823466e+38f;

#define PRS_STAMP(i)                                              \
  do {                                                            \
    if (a.stamps && tid == 0) {                                   \
      a.stamps[(size_t) frame * 16 + (i)] = (unsigned long long) clock64(); \
    }                                                             \
  } while (0)

template <int KPT, bool STAGE>
__global__ __launch_bounds__(kStereoThreads) void stereo_match_kernel(const StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid    = threadIdx.x;
  const int frame  = blockIdx.x;
  const int stride = a.b.stride;
  const int rows   = a.p.image_rows;
  int nL           = a.b.n_left[frame];
  int nR           = a.b.n_right[frame];
  nL               = nL < 0 ? 0 : (nL > stride ? stride : nL);
  nR               = nR < 0 ? 0 : (nR > stride ? stride : nR);
  const size_t base = (size_t) frame * (size_t) stride;
  const prs_kp2* __restrict__ kpL = a.b.left_kp + base;
  const prs_kp2* __restrict__ kpR = a.b.right_kp + base;
  const uint4* __restrict__ gdL   = reinterpret_cast<const uint4*>(a.b.left_desc + base * PRS_DESC_BYTES);
  const uint4* __restrict__ gdR   = reinterpret_cast<const uint4*>(a.b.right_desc + base * PRS_DESC_BYTES);

  uint4* ldL         = reinterpret_cast<uint4*>(smem + a.off_desc_l);
  uint4* ldR         = reinterpret_cast<uint4*>(smem + a.off_desc_r);
  uint32_t* sortedL  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_l);
  uint32_t* sortedR  = reinterpret_cast<uint32_t*>(smem + a.off_sorted_r);
  uint32_t* bucketL  = reinterpret_cast<uint32_t*>(smem + a.off_bucket_l);
  uint32_t* bucketR  = reinterpret_cast<uint32_t*>(smem + a.off_bucket_r);
  uint16_t* rsL      = reinterpret_cast<uint16_t*>(smem + a.off_rowstart_l);
  uint16_t* rsR      = reinterpret_cast<uint16_t*>(smem + a.off_rowstart_r);
  uint64_t* scratch  = reinterpret_cast<uint64_t*>(smem + a.off_scratch);
  int* misc          = reinterpret_cast<int*>(scratch + 20);
  uint32_t* histL    = sortedL;  // the histograms die before the sorted arrays are born
  uint32_t* histR    = sortedR;
  uint32_t* rec      = bucketL;  // match record per sorted-left position (after the sort)
  uint8_t* matchedL  = reinterpret_cast<uint8_t*>(bucketR);  // pass number + 1, 0 = unmatched
  uint8_t* matchedR  = matchedL + a.sort_cap;

  PRS_STAMP(0);
  // ---- issue every global read of this frame up front ----------------------------------------
  prs_kp2 cL[KPT], cR[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = k * kStereoThreads + tid;
    cL[k]       = i < nL ? kpL[i] : prs_kp2{0.f, 0.f};
    cR[k]       = i < nR ? kpR[i] : prs_kp2{0.f, 0.f};
  }
  u32x4 stgL[STAGE ? 2 * KPT : 1], stgR[STAGE ? 2 * KPT : 1];
  if (STAGE) {
    // unconditional loads (tail lanes re-read the last valid 16 B: same cache line, no extra HBM
    // traffic) keep the staging registers out of scratch memory
    const int lastL = 2 * nL > 0 ? 2 * nL - 1 : 0;
    const int lastR = 2 * nR > 0 ? 2 * nR - 1 : 0;
#pragma unroll
    for (int k = 0; k < 2 * KPT; ++k) {
      const int i = k * kStereoThreads + tid;
      stgL[k]     = reinterpret_cast<const u32x4*>(gdL)[i < lastL ? i : lastL];
      stgR[k]     = reinterpret_cast<const u32x4*>(gdR)[i < lastR ? i : lastR];
    }
  }

  // ---- a1: Feature{row,col,unsorted_index} + counting sort by row ----------------------------
  for (int i = tid; i <= rows; i += kStereoThreads) {
    histL[i] = 0;
    histR[i] = 0;
  }
  if (tid == 0) {
    misc[0] = 0;
  }
  __syncthreads();

  PRS_STAMP(1);
  int rowL[KPT], rowR[KPT];
  uint32_t keyL[KPT], keyR[KPT], slotL[KPT], slotR[KPT];
  bool bad = false;
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int i = k * kStereoThreads + tid;
    rowL[k]     = -1;
    rowR[k]     = -1;
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
    return;
  }

  PRS_STAMP(2);
  // exclusive scan of both histograms at once (left in the low, right in the high word)
  {
    const int ipt   = (rows + 1 + kStereoThreads - 1) / kStereoThreads;
    const int start = tid * ipt;
    uint64_t sum    = 0;
    for (int j = 0; j < ipt; ++j) {
      const int r = start + j;
      if (r <= rows) {
        sum += (uint64_t) histL[r] | ((uint64_t) histR[r] << 32);
      }
    }
    uint64_t total;
    uint64_t run = block_exclusive_scan_u64(sum, scratch, total);
    for (int j = 0; j < ipt; ++j) {
      const int r = start + j;
      if (r <= rows) {
        const uint64_t h = (uint64_t) histL[r] | ((uint64_t) histR[r] << 32);
        rsL[r]           = (uint16_t) (run & 0xffffffffu);
        rsR[r]           = (uint16_t) (run >> 32);
        run += h;
      }
    }
  }
  __syncthreads();

  PRS_STAMP(3);
  // scatter into row buckets (arbitrary order inside a row) ...
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    if (rowL[k] >= 0) {
      bucketL[rsL[rowL[k]] + slotL[k]] = keyL[k];
    }
    if (rowR[k] >= 0) {
      bucketR[rsR[rowR[k]] + slotR[k]] = keyR[k];
    }
  }
  __syncthreads();
  // ... then rank inside the row by (col, unsorted index): epipolar_impl.cpp:36-41 + canonical tie-break
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    if (rowL[k] >= 0) {
      const int s = rsL[rowL[k]], e = rsL[rowL[k] + 1];
      int rank = 0;
      for (int j = s; j < e; ++j) {
        rank += bucketL[j] < keyL[k] ? 1 : 0;
      }
      sortedL[s + rank] = keyL[k];
    }
    if (rowR[k] >= 0) {
      const int s = rsR[rowR[k]], e = rsR[rowR[k] + 1];
      int rank = 0;
      for (int j = s; j < e; ++j) {
        rank += bucketR[j] < keyR[k] ? 1 : 0;
      }
      sortedR[s + rank] = keyR[k];
    }
  }
  PRS_STAMP(4);
  // descriptor rows land in LDS (the loads were issued before the sort)
  if (STAGE) {
#pragma unroll
    for (int k = 0; k < 2 * KPT; ++k) {
      const int i = k * kStereoThreads + tid;
      if (i < 2 * nL) {
        reinterpret_cast<u32x4*>(ldL)[i] = stgL[k];
      }
      if (i < 2 * nR) {
        reinterpret_cast<u32x4*>(ldR)[i] = stgR[k];
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < nL; i += kStereoThreads) {
    matchedL[i] = 0;
  }
  for (int i = tid; i < nR; i += kStereoThreads) {
    matchedR[i] = 0;
  }
  __syncthreads();

  PRS_STAMP(5);
  // ---- a2: per-offset epipolar scan ----------------------------------------------------------
  const float max_dist  = a.p.maximum_descriptor_distance;
  const float max_ratio = a.p.maximum_distance_ratio_to_second_best;
  const int max_disp    = a.p.maximum_disparity_pixels;
  const int thickness   = a.p.epipolar_line_thickness_pixels > 0 ? a.p.epipolar_line_thickness_pixels : 0;
  const int n_offsets   = 1 + 2 * thickness;
  prs_corr* __restrict__ out = a.b.matches + base;
  int out_base               = 0;
  int fixed_base             = 0;

  for (int o = 0; o < n_offsets; ++o) {
    const int off = o == 0 ? 0 : ((o & 1) ? (o + 1) / 2 : -(o / 2));  // 0,+1,-1,+2,-2 (epipolar_impl.cpp:71-79)
    for (int r = tid; r < rows; r += kStereoThreads) {
      const int rr = r + off;
      if (rr < 0 || rr >= rows) {
        continue;
      }
      const int ls = rsL[r], le = rsL[r + 1];
      int c        = rsR[rr];
      const int re = rsR[rr + 1];
      int lo       = c;  // right features before lo are further than max_disp left of every remaining left feature
      for (int p = ls; p < le && c < re; ++p) {
        if (matchedL[p]) {
          continue;  // pruned by an earlier pass (epipolar_impl.cpp:188-196)
        }
        const uint32_t kl = sortedL[p];
        const int col_l   = (int) (kl >> 16);
        const int idx_l   = (int) (kl & 0xffffu);
        uint4 d0, d1;
        if (STAGE) {
          d0 = ldL[2 * idx_l];
          d1 = ldL[2 * idx_l + 1];
        } else {
          d0 = gdL[2 * idx_l];
          d1 = gdL[2 * idx_l + 1];
        }
        // columns are non-decreasing along the row, so candidates skipped for exceeding the
        // disparity range (epipolar_impl.cpp:146-149) stay skipped for all later left features:
        // remember where the in-range window starts instead of rescanning from the cursor
        while (lo < re && col_l - (int) (sortedR[lo] >> 16) > max_disp) {
          ++lo;
        }
        float best = kFloatMax, second = kFloatMax;
        int best_q = -1;
        for (int q = c > lo ? c : lo; q < re; ++q) {
          if (matchedR[q]) {
            continue;  // pruned (epipolar_impl.cpp:197-205)
          }
          const uint32_t kr = sortedR[q];
          const int disp    = col_l - (int) (kr >> 16);
          if (disp < 0) {
            break;  // epipolar_impl.cpp:141-143
          }
          if (disp > max_disp) {
            continue;  // epipolar_impl.cpp:146-149
          }
          const int idx_r = (int) (kr & 0xffffu);
          uint4 e0, e1;
          if (STAGE) {
            e0 = ldR[2 * idx_r];
            e1 = ldR[2 * idx_r + 1];
          } else {
            e0 = gdR[2 * idx_r];
            e1 = gdR[2 * idx_r + 1];
          }
          const float d = (float) hamming256(d0, d1, e0, e1);
          if (d < best) {  // epipolar_impl.cpp:158-164
            second = best;
            best   = d;
            best_q = q;
          } else if (d < second) {
            second = d;
          }
        }
        if (best < max_dist && best / second < max_ratio) {  // epipolar_impl.cpp:171-173
          rec[p]           = ((sortedR[best_q] & 0xffffu) << 16) | (uint32_t) (int) best;
          matchedL[p]      = (uint8_t) (o + 1);
          matchedR[best_q] = 1;
          c                = best_q + 1;  // epipolar_impl.cpp:181
        }
      }
    }
    __syncthreads();
    PRS_STAMP(6);

    // compaction in sorted-left traversal order (+ optional adaptor/triangulator epilogue)
    uint32_t m_rec[KPT];
    float4 m_uvuv[KPT];
    uint64_t cnt = 0;
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      const int p = tid * KPT + k;
      m_rec[k]    = 0xffffffffu;
      m_uvuv[k]   = make_float4(0.f, 0.f, 0.f, -1.f);
      if (p < nL && matchedL[p] == (uint8_t) (o + 1)) {
        m_rec[k] = rec[p];
        cnt += 1;
        if (a.epilogue) {
          const prs_kp2 l   = kpL[sortedL[p] & 0xffffu];
          const prs_kp2 rgt = kpR[m_rec[k] >> 16];
          // raw_data_preprocessor_stereo_projective.cpp:117-125
          const float hd = l.u - rgt.u, vd = l.v - rgt.v;
          if (!(hd < 0.0f || vd < 0.0f)) {
            m_uvuv[k] = make_float4(l.u, l.v, rgt.u, rgt.v);
            cnt += (uint64_t) 1 << 32;
          } else {
            m_uvuv[k].w = -2.0f;  // dropped
          }
        }
      }
    }
    uint64_t total;
    uint64_t pre = block_exclusive_scan_u64(cnt, scratch, total);
    int w_match  = out_base + (int) (pre & 0xffffffffu);
    int w_fixed  = fixed_base + (int) (pre >> 32);
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
      if (m_rec[k] != 0xffffffffu) {
        const int p     = tid * KPT + k;
        const int idx_l = (int) (sortedL[p] & 0xffffu);
        prs_corr cr;
        cr.fixed_idx  = idx_l;
        cr.moving_idx = (int) (m_rec[k] >> 16);
        cr.response   = (float) (m_rec[k] & 0xffffu);
        out[w_match++] = cr;
        if (a.epilogue && m_uvuv[k].w != -2.0f) {
          const size_t w = base + (size_t) w_fixed++;
          reinterpret_cast<float4*>(a.b.fixed_uvuv)[w] = m_uvuv[k];
          uint4* fd = reinterpret_cast<uint4*>(a.b.fixed_desc) + 2 * w;
          if (STAGE) {
            fd[0] = ldL[2 * idx_l];
            fd[1] = ldL[2 * idx_l + 1];
          } else {
            fd[0] = gdL[2 * idx_l];
            fd[1] = gdL[2 * idx_l + 1];
          }
          // triangulator_rigid_stereo.cpp:39-45,60-85 (operation order kept)
          const float x_L = m_uvuv[k].x, y_L = m_uvuv[k].y, x_R = m_uvuv[k].z, y_R = m_uvuv[k].w;
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
          reinterpret_cast<float4*>(a.b.fixed_xyz)[w] = pt;
        }
      }
    }
    out_base += (int) (total & 0xffffffffu);
    fixed_base += (int) (total >> 32);
  }

  PRS_STAMP(7);
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
  hipLaunchKernelGGL(kernel, dim3(a.b.batch), dim3(kStereoThreads), lds, stream, a);
  return hipGetLastError();
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
  StereoArgs a;
  a.p        = *params;
  a.b        = *batch;
  a.epilogue = (batch->fixed_uvuv && batch->fixed_desc && batch->n_fixed && batch->fixed_xyz && batch->triangulator) ? 1 : 0;
  if (a.epilogue) {
    a.tri = *batch->triangulator;
  } else {
    a.tri = prs_triangulator_params{1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  }
  a.stamps = ctx_stamps(ctx, (size_t) batch->batch * 16 * sizeof(unsigned long long));
  const int kpt        = stride <= 1024 ? 1 : (stride <= 2048 ? 2 : (stride <= 4096 ? 4 : 8));
  const uint32_t rows1 = (uint32_t) params->image_rows + 1;
  a.sort_cap           = (int) (((uint32_t) stride > rows1 ? (uint32_t) stride : rows1));
  // LDS carve; every offset is a multiple of 16 (cdna_hip_programming.md Guideline 17)
  auto carve = [&](bool stage, StereoArgs& s) -> size_t {
    uint32_t off    = 0;
    const uint32_t dbytes = stage ? (uint32_t) stride * PRS_DESC_BYTES : 0;
    s.off_desc_l    = off; off = align_up(off + dbytes, 16);
    s.off_desc_r    = off; off = align_up(off + dbytes, 16);
    s.off_sorted_l  = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_sorted_r  = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_bucket_l  = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_bucket_r  = off; off = align_up(off + (uint32_t) s.sort_cap * 4, 16);
    s.off_rowstart_l = off; off = align_up(off + (rows1 + 1) * 2, 16);
    s.off_rowstart_r = off; off = align_up(off + (rows1 + 1) * 2, 16);
    s.off_scratch   = off; off = align_up(off + 24 * 8, 16);
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
    ctx_report_stamps(ctx, batch->batch, 8, "stereo_match: issue-loads | zero+barrier | coords+hist | scan | scatter+rank | stage-write+init | chain | compaction");
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
