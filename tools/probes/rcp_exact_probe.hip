// rcp_exact_probe.hip -- over ALL 2^32 float bit patterns: where does v_rcp_f32 + one / two Newton steps (FMA) equal the correctly
// rounded 1.0f / x of the compiler's IEEE expansion (v_div_scale / v_rcp / 4 x v_fma / v_div_fmas / v_div_fixup, 12 instructions)?
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/bin/rcp_exact_probe tools/probes/rcp_exact_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__device__ __forceinline__ float rcp1(float x) {
  const float y0 = __builtin_amdgcn_rcpf(x);
  const float e  = __builtin_fmaf(-x, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rcp2(float x) {
  const float y1 = rcp1(x);
  const float e  = __builtin_fmaf(-x, y1, 1.0f);
  return __builtin_fmaf(e, y1, y1);
}

// per biased exponent of x (0 .. 255): mismatches of the one-step and of the two-step form; + an example of each
__global__ void probe(unsigned long long* bad1, unsigned long long* bad2, uint32_t* ex1, uint32_t* ex2) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const uint32_t bits = (uint32_t) i;
    const float x = __uint_as_float(bits);
    const float want = 1.0f / x;
    const uint32_t w = __float_as_uint(want);
    const bool nan_w = (w & 0x7fffffffu) > 0x7f800000u;
    const uint32_t a = __float_as_uint(rcp1(x)), b = __float_as_uint(rcp2(x));
    const bool ok1 = a == w || (nan_w && (a & 0x7fffffffu) > 0x7f800000u);
    const bool ok2 = b == w || (nan_w && (b & 0x7fffffffu) > 0x7f800000u);
    const int e = (bits >> 23) & 0xff;
    if (!ok1) {
      atomicAdd(&bad1[e], 1ull);
      ex1[e] = bits;
    }
    if (!ok2) {
      atomicAdd(&bad2[e], 1ull);
      ex2[e] = bits;
    }
  }
}

int main() {
  unsigned long long *b1, *b2, h1[256], h2[256];
  uint32_t *e1, *e2, x1[256], x2[256];
  hipMalloc(&b1, sizeof h1); hipMalloc(&b2, sizeof h2); hipMalloc(&e1, sizeof x1); hipMalloc(&e2, sizeof x2);
  hipMemset(b1, 0, sizeof h1); hipMemset(b2, 0, sizeof h2); hipMemset(e1, 0, sizeof x1); hipMemset(e2, 0, sizeof x2);
  hipLaunchKernelGGL(probe, dim3(256 * 32), dim3(256), 0, 0, b1, b2, e1, e2);
  hipMemcpy(h1, b1, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(h2, b2, sizeof h2, hipMemcpyDeviceToHost);
  hipMemcpy(x1, e1, sizeof x1, hipMemcpyDeviceToHost); hipMemcpy(x2, e2, sizeof x2, hipMemcpyDeviceToHost);
  unsigned long long t1 = 0, t2 = 0;
  for (int e = 0; e < 256; ++e) {
    t1 += h1[e]; t2 += h2[e];
    if (h1[e] || h2[e]) {
      float f1, f2; memcpy(&f1, &x1[e], 4); memcpy(&f2, &x2[e], 4);
      printf("biased exponent %3d: one step %10llu wrong (e.g. %08x = %g), two steps %10llu wrong (e.g. %08x = %g)\n", e, h1[e], x1[e], f1, h2[e], x2[e], f2);
    }
  }
  printf("total: one step %llu, two steps %llu of 2^32 (both signs)\n", t1, t2);
  return 0;
}
