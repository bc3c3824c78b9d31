// scene_clip.hip -- SceneClipperProjective3D::compute on the device
// (mapping/scene_clipper_projective_3d.cpp:9-67): frustum clip of a local map + the
// local->global index map, i.e. an ordered stream compaction.  SURVEY.md section 8f #2.
//
// HBM-bound by construction: every scene point (16 B coordinates + 32 B descriptor row) is read
// once, every kept point (16 + 32 + 4 B) written once.  Survivors keep ascending source order,
// which is the order the reference's projector emits them in (one sequential loop).
//
// Two launch shapes:
//   * one 256-thread workgroup per scene walking its tiles with a running offset (batched path:
//     many independent sequences, no inter-workgroup dependency), or
//   * (few scenes, large maps) tile-parallel count + scatter: grid (tiles, scenes), the scatter
//     pass re-evaluates the 16-B coordinates (descriptor rows are still read once).
#include "prs_device.h"
#include "prs_host.h"
#include "prs_se3.h"

namespace prs {

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));

constexpr int kClipThreads = 256;
constexpr int kClipSub     = 4;                        // sub-tiles (coalesced 256-point slabs) per tile
constexpr int kClipTile    = kClipThreads * kClipSub;  // points per tile
constexpr int kClipWaves   = kClipThreads / 64;

struct ClipArgs {
  prs_projector proj;
  prs_clip_batch b;
  float S[16];      // sensor_in_robot
  int to_robot;     // sensor_in_robot != identity (scene_clipper_projective_3d.cpp:61)
  int tiles;        // tiles per scene (tile-parallel shape)
  int* counts;      // [batch][tiles] survivors per tile (tile-parallel shape)
  const float* info_lut;  // information scale by landmark age (4096 entries), used when scene_n_opt is set
};

enum { kClipWalk = 0, kClipCount = 1, kClipScatter = 2 };

struct ClipPose {
  float W[12];  // local map -> camera, rows 0..2
};

// one tile: keep test per point, ordered slots, optional writes.  Returns the tile's survivor count.
template <int MODE>
__device__ __forceinline__ int clip_tile(const ClipArgs& a,
                                         const ClipPose& pose,
                                         const float4* __restrict__ in_xyzw,
                                         const u32x4c* __restrict__ in_desc,
                                         int n,
                                         int tile_base,
                                         int out_base,
                                         float4* __restrict__ out_xyzw,
                                         u32x4c* __restrict__ out_desc,
                                         int32_t* __restrict__ out_index,
                                         const uint32_t* __restrict__ in_nopt,
                                         int* wave_counts /* LDS [kClipSub * kClipWaves] */) {
  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float cols = (float) a.proj.canvas_cols;
  const float rows = (float) a.proj.canvas_rows;
  float4 p[kClipSub];
  float cx[kClipSub], cy[kClipSub], cz[kClipSub];
  bool keep[kClipSub];
  uint64_t mask[kClipSub];
#pragma unroll
  for (int k = 0; k < kClipSub; ++k) {
    const int i = tile_base + k * kClipThreads + tid;
    p[k]        = in_xyzw[i < n ? i : (n - 1)];
  }
#pragma unroll
  for (int k = 0; k < kClipSub; ++k) {
    const int i = tile_base + k * kClipThreads + tid;
    // PointProjectorPinhole_::compute as restated in SURVEY.md appendix A (same as the finder's projection)
    const float x = ((pose.W[0] * p[k].x + pose.W[1] * p[k].y) + pose.W[2] * p[k].z) + pose.W[3];
    const float y = ((pose.W[4] * p[k].x + pose.W[5] * p[k].y) + pose.W[6] * p[k].z) + pose.W[7];
    const float z = ((pose.W[8] * p[k].x + pose.W[9] * p[k].y) + pose.W[10] * p[k].z) + pose.W[11];
    bool ok       = i < n;
    if (z < a.proj.range_min || z > a.proj.range_max) {
      ok = false;
    }
    const float hx = a.proj.fx * x + a.proj.cx * z;
    const float hy = a.proj.fy * y + a.proj.cy * z;
    const float u  = hx / z;
    const float v  = hy / z;
    if (u < 0.0f || u >= cols || v < 0.0f || v >= rows) {
      ok = false;
    }
    keep[k] = ok;
    cx[k]   = x;
    cy[k]   = y;
    cz[k]   = z;
    mask[k] = __ballot(ok);
    if (lane == 0) {
      wave_counts[k * kClipWaves + wave] = __popcll(mask[k]);
    }
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int j = 0; j < kClipSub * kClipWaves; ++j) {
    total += wave_counts[j];  // entries are in (sub-tile, wave) order = ascending source index
  }
  if (MODE != kClipCount) {
#pragma unroll
    for (int k = 0; k < kClipSub; ++k) {
      int base = out_base;
#pragma unroll
      for (int j = 0; j < kClipSub * kClipWaves; ++j) {
        base += (j < k * kClipWaves + wave) ? wave_counts[j] : 0;
      }
      if (keep[k]) {
        const int i    = tile_base + k * kClipThreads + tid;
        const int slot = base + __popcll(mask[k] & ((1ull << lane) - 1ull));
        float ox = cx[k], oy = cy[k], oz = cz[k];
        if (a.to_robot) {  // transformInPlace<Isometry>(sensor_in_robot), scene_clipper_projective_3d.cpp:61-63
          ox = ((a.S[0] * cx[k] + a.S[1] * cy[k]) + a.S[2] * cz[k]) + a.S[3];
          oy = ((a.S[4] * cx[k] + a.S[5] * cy[k]) + a.S[6] * cz[k]) + a.S[7];
          oz = ((a.S[8] * cx[k] + a.S[9] * cy[k]) + a.S[10] * cz[k]) + a.S[11];
        }
        float w = p[k].w;
        if (in_nopt) {  // aligner_slice_processor_projective.cpp:46-52
          const uint32_t n = in_nopt[i];
          w                = a.info_lut[n < 4095u ? n : 4095u];
        }
        out_xyzw[slot]  = make_float4(ox, oy, oz, w);
        out_index[slot] = i;
        if (in_desc && out_desc) {
          const u32x4c d0        = in_desc[2 * i];
          const u32x4c d1        = in_desc[2 * i + 1];
          out_desc[2 * slot]     = d0;
          out_desc[2 * slot + 1] = d1;
        }
      }
    }
  }
  __syncthreads();  // wave_counts is rewritten by the next tile
  return total;
}

template <int MODE>
__global__ __launch_bounds__(kClipThreads) void scene_clip_kernel(const ClipArgs a) {
  __shared__ int wave_counts[kClipSub * kClipWaves];
  __shared__ int red[kClipWaves];
  const int scene = MODE == kClipWalk ? blockIdx.x : blockIdx.y;
  const int tid   = threadIdx.x;
  int n           = a.b.n_scene[scene];
  n               = n < 0 ? 0 : (n > a.b.stride ? a.b.stride : n);
  if (n == 0) {
    // empty global scene: status Ready, nothing is cleared (scene_clipper_projective_3d.cpp:21-28)
    if (tid == 0 && (MODE == kClipWalk || (MODE == kClipScatter && blockIdx.x == 0))) {
      a.b.status[scene] = PRS_WARN_EMPTY_INPUT;
    }
    if (MODE == kClipCount && tid == 0) {
      a.counts[(size_t) scene * a.tiles + blockIdx.x] = 0;
    }
    return;
  }
  const size_t base = (size_t) scene * (size_t) a.b.stride;
  const float4* __restrict__ in_xyzw = reinterpret_cast<const float4*>(a.b.scene_xyzw) + base;
  const u32x4c* __restrict__ in_desc =
    a.b.scene_desc ? reinterpret_cast<const u32x4c*>(a.b.scene_desc + base * PRS_DESC_BYTES) : nullptr;
  float4* __restrict__ out_xyzw = reinterpret_cast<float4*>(a.b.clipped_xyzw) + base;
  u32x4c* __restrict__ out_desc =
    a.b.clipped_desc ? reinterpret_cast<u32x4c*>(a.b.clipped_desc + base * PRS_DESC_BYTES) : nullptr;
  int32_t* __restrict__ out_index = a.b.global_indices + base;
  const uint32_t* __restrict__ in_nopt = a.b.scene_n_opt ? a.b.scene_n_opt + base : nullptr;

  // projector->setCameraPose(_robot_in_local_map * _sensor_in_robot), :46; points go through its inverse
  ClipPose pose;
  {
    float R[16], C[16], W[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      R[i] = a.b.robot_in_local_map[(size_t) scene * 16 + i];
    }
    se3_mul(R, a.S, C);
    se3_inverse(C, W);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      pose.W[i] = W[i];
    }
  }

  if (MODE == kClipWalk) {
    int running = 0;
    for (int tile_base = 0; tile_base < n; tile_base += kClipTile) {
      running += clip_tile<kClipWalk>(a, pose, in_xyzw, in_desc, n, tile_base, running, out_xyzw, out_desc, out_index, in_nopt, wave_counts);
    }
    if (tid == 0) {
      a.b.n_clipped[scene] = running;
      a.b.status[scene]    = running == 0 ? PRS_WARN_NO_PROJECTION : PRS_OK;  // :55-58
    }
  } else if (MODE == kClipCount) {
    const int tile_base = blockIdx.x * kClipTile;
    int total           = 0;
    if (tile_base < n) {
      total = clip_tile<kClipCount>(a, pose, in_xyzw, in_desc, n, tile_base, 0, out_xyzw, out_desc, out_index, in_nopt, wave_counts);
    }
    if (tid == 0) {
      a.counts[(size_t) scene * a.tiles + blockIdx.x] = total;
    }
  } else {
    // survivors of the tiles before this one
    const int* __restrict__ cnt = a.counts + (size_t) scene * a.tiles;
    int part                    = 0;
    for (int j = tid; j < (int) blockIdx.x; j += kClipThreads) {
      part += cnt[j];
    }
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o, 64);
    }
    if ((tid & 63) == 0) {
      red[tid >> 6] = part;
    }
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < kClipWaves; ++w) {
      before += red[w];
    }
    const int tile_base = blockIdx.x * kClipTile;
    int total           = 0;
    if (tile_base < n) {
      total = clip_tile<kClipScatter>(a, pose, in_xyzw, in_desc, n, tile_base, before, out_xyzw, out_desc, out_index, in_nopt, wave_counts);
    }
    if (tid == 0 && (int) blockIdx.x == a.tiles - 1) {
      a.b.n_clipped[scene] = before + total;
      a.b.status[scene]    = (before + total) == 0 ? PRS_WARN_NO_PROJECTION : PRS_OK;
    }
  }
}

int scene_clip_launch(prs_context* ctx, const prs_projector* projector, const float* sensor_in_robot, const prs_clip_batch* batch) {
  if (!projector) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing projector");  // scene_clipper_projective_3d.cpp:12-14
  }
  if (!batch || !batch->clipped_xyzw || !batch->global_indices || !batch->n_clipped || !batch->status) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing clipped scene");  // :15-17
  }
  if (!batch->scene_xyzw || !batch->n_scene || !batch->robot_in_local_map || !sensor_in_robot) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: missing global scene");  // :18-20
  }
  if ((batch->scene_desc == nullptr) != (batch->clipped_desc == nullptr)) {
    return ctx_fail(ctx, PRS_ERR_NULL, "prs_scene_clip: descriptor rows need both the scene and the clipped buffer");
  }
  if (batch->batch <= 0) {
    return PRS_OK;
  }
  if (batch->stride <= 0) {
    return ctx_fail(ctx, PRS_ERR_UNSUPPORTED, "prs_scene_clip: stride must be positive");
  }
  ClipArgs a;
  a.proj = *projector;
  a.b    = *batch;
  bool differs = false;
  for (int i = 0; i < 16; ++i) {
    a.S[i]  = sensor_in_robot[i];
    differs = differs || (sensor_in_robot[i] != ((i % 5 == 0) ? 1.0f : 0.0f));
  }
  a.to_robot = differs ? 1 : 0;
  a.tiles    = (batch->stride + kClipTile - 1) / kClipTile;
  a.counts   = nullptr;
  a.info_lut = nullptr;
  if (batch->scene_n_opt) {
    a.info_lut = ctx_info_scale_table(ctx);
    if (!a.info_lut) {
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_scene_clip: information scale table allocation failed");
    }
  }
  hipStream_t stream = ctx_stream(ctx);
  // many scenes (or short ones): one workgroup walks a scene; few long scenes: tile-parallel
  const bool walk = batch->batch >= 128 || a.tiles <= 2;
  if (walk) {
    hipLaunchKernelGGL(scene_clip_kernel<kClipWalk>, dim3(batch->batch), dim3(kClipThreads), 0, stream, a);
  } else {
    a.counts = static_cast<int*>(ctx_device_scratch_slot(ctx, 3, (size_t) batch->batch * a.tiles * sizeof(int)));
    if (!a.counts) {
      return ctx_fail(ctx, PRS_ERR_HIP, "prs_scene_clip: scratch allocation failed");
    }
    hipLaunchKernelGGL(scene_clip_kernel<kClipCount>, dim3(a.tiles, batch->batch), dim3(kClipThreads), 0, stream, a);
    hipLaunchKernelGGL(scene_clip_kernel<kClipScatter>, dim3(a.tiles, batch->batch), dim3(kClipThreads), 0, stream, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    return ctx_fail_hip(ctx, e, "prs_scene_clip launch");
  }
  return PRS_OK;
}

}  // namespace prs
