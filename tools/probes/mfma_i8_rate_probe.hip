// mfma_i8_rate_probe.hip -- cycles per v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8 on gfx950 (what prices the brute-force matcher's
// dense phase): W waves per SIMD issue N MFMAs each on independent accumulators, every CU busy; wall clock and s_memtime.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_i8_rate_probe tools/probes/mfma_i8_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void loop(int* out, long long* cyc, int iters, v4i a, v4i b) {
  v4i acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
  v16i big[2];
  for (int i = 0; i < 16; ++i) { big[0][i] = i; big[1][i] = -i; }
  a.x += threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (SHAPE == 16) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) big[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, big[k], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
  for (int k = 0; k < 2; ++k) for (int i = 0; i < 16; ++i) s += big[k][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 256 * 8 * 1024 * sizeof(int)); hipMalloc(&cyc, 4096 * sizeof(long long));
  const v4i a = {0x01010101, 0x01000100, 0x00010001, 0x01010000}, b = {0x01ff01ff, -1, 0x01010101, 0x01ffff01};
  for (int shape : {16, 32}) {
    for (int waves : {1, 2, 4}) {  // waves per SIMD: blocks of 64 threads, 4 * waves per CU
      const int iters = 20000, blocks = 256 * 4 * waves;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto k = shape == 16 ? loop<16> : loop<32>;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, 100, a, b);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, a, b);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c0; hipMemcpy(&c0, cyc, sizeof c0, hipMemcpyDeviceToHost);
      const double mfmas = (double) blocks * iters * (shape == 16 ? 4 : 2);
      const double ops = mfmas * (shape == 16 ? 16 * 16 * 64 * 2.0 : 32 * 32 * 32 * 2.0);
      printf("v_mfma_i32_%s_i8, %d wave(s) per SIMD: %.3f ms, %.1f TOP/s, %.1f wall-ns per MFMA and SIMD, readcyclecounter ticks per MFMA of wave 0: %.1f\n",
             shape == 16 ? "16x16x64" : "32x32x32", waves, ms, ops / ms / 1e9, ms * 1e6 / (mfmas / 1024.0), (double) c0 / (iters * (shape == 16 ? 4 : 2)));
    }
  }
  return 0;
}
