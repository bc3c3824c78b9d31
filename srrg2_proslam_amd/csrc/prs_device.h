// prs_device.h -- device-side helpers shared by the gfx950 kernels (wave64, LDS, exact float ops).
// Everything that must agree bit-for-bit with a plain sequential float evaluation is written as
// explicit two-operand expressions and the library is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/proslam_hip.h"

#define PRS_WAVE 64

namespace prs {

// 256-bit Hamming distance of two 32-byte rows held as 2 x uint4 each
// (replaces srrg2_core PointDescriptorField::distance; call site CF/..epipolar_impl.cpp:157)
// 1.0f / x, correctly rounded (the value the IEEE division of the CPU side produces), in 3 + 2 instructions instead of the twelve of
// the compiler's expansion (v_div_scale x 2, v_rcp, 4 x v_fma, v_div_fmas, v_div_fixup).  tools/probes/rcp_exact_probe.hip runs ALL 2^32
// bit patterns on gfx950: v_rcp_f32 followed by ONE Newton step in fused multiply-adds equals the IEEE quotient for every x whose biased
// exponent is 1 .. 252 (2^-126 <= |x| < 2^126; both signs, every mantissa); zeros, denormals, the two largest binades (denormal
// quotients), infinities and NaNs differ.  A wave that holds such an operand takes the compiler's division instead: the depths and
// chi-squares of live correspondences never are, but the stand-in rows of a partially filled wave and of points behind the camera
// carry chi = 0, so those waves take the long form (same bits).  Feeding the unread lanes a benign operand was measured slower
// (round 5: 13.54 / 13.64 ms against 13.34 / 13.38: the select costs every wave what the long form costs a few).
__device__ __forceinline__ float recip_exact(const float x) {
  const float ax  = __builtin_fabsf(x);
  const bool plain = ax >= 0x1p-126f && ax < 0x1p126f;
  if (__builtin_expect(__ballot(!plain) != 0ull, 0)) {
    return 1.0f / x;
  }
  const float y = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
}

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
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

// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32 has the accumulate operand; left to itself the compiler
// reassociates eight of them into a tree of plain popcounts and three-operand adds: 19 instead of 16 instructions per
// 256-bit Hamming distance.  Used where instruction issue is the bound (the projective search); the matcher's scoring
// phase is latency-bound and 3 % faster with the tree)
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

__device__ __forceinline__ uint64_t wave_inclusive_scan_u64(uint64_t v) {
  const int lane = threadIdx.x & (PRS_WAVE - 1);
#pragma unroll
  for (int d = 1; d < PRS_WAVE; d <<= 1) {
    const uint64_t o = __shfl_up((unsigned long long) v, d, PRS_WAVE);
    if (lane >= d) {
      v += o;
    }
  }
  return v;
}

// exclusive prefix over all threads of the block (blockDim.x multiple of 64, <= 1024).
// scratch: >= 17 uint64_t in LDS.  Contains two __syncthreads().
__device__ __forceinline__ uint64_t block_exclusive_scan_u64(uint64_t v, uint64_t* scratch, uint64_t& total) {
  const int lane   = threadIdx.x & (PRS_WAVE - 1);
  const int wave   = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  const uint64_t incl = wave_inclusive_scan_u64(v);
  if (lane == PRS_WAVE - 1) {
    scratch[wave] = incl;
  }
  __syncthreads();
  uint64_t base = 0;
  uint64_t all  = 0;
  for (int w = 0; w < nwaves; ++w) {
    const uint64_t t = scratch[w];
    if (w < wave) {
      base += t;
    }
    all += t;
  }
  total = all;
  __syncthreads();
  return base + incl - v;
}

} // namespace prs
