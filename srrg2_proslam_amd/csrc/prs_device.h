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
