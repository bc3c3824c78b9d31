// mfma_valu_overlap_probe.hip -- does the vector ALU issue while the matrix pipe of the same SIMD executes a v_mfma_i32_16x16x64_i8 on gfx950?
// (what decides whether the brute-force matcher's dense phase, ~3.2 vector instructions per MFMA, can approach the matrix rate.)
// A 512-thread workgroup per CU = two waves per SIMD (wave w on SIMD w % 4).  Modes:
//   M   waves 0-3 issue MFMAs (four independent accumulators), waves 4-7 exit
//   V   waves 4-7 issue VALU (v_add_u32 on twelve independent registers), waves 0-3 exit
//   MV  both at once (one MFMA wave + one VALU wave per SIMD)
//   I   waves 0-3 interleave: one MFMA, then R VALU, in one instruction stream; waves 4-7 exit
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_valu_overlap_probe tools/probes/mfma_valu_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(one))

template <int MODE, int R>
__global__ __launch_bounds__(512) void probe(int* out, int iters, v4i a, v4i b) {
  const int wave = threadIdx.x >> 6;
  v4i acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
  int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, r8 = 8, r9 = 9, r10 = 10, r11 = 11;
  const int one = 1 + (int) (threadIdx.x & 1);
  a.x += threadIdx.x;
  const bool mf = wave < 4;
  if (MODE == 0 && !mf) return;
  if (MODE == 1 && mf) return;
  if (MODE == 3 && !mf) return;
  if (MODE == 3) {
    for (int i = 0; i < iters; ++i) {
      MFMA(acc0); if (R >= 1) VADD(r0); if (R >= 2) VADD(r1); if (R >= 3) VADD(r2); if (R >= 4) VADD(r3); if (R >= 5) VADD(r4); if (R >= 6) VADD(r5);
      MFMA(acc1); if (R >= 1) VADD(r6); if (R >= 2) VADD(r7); if (R >= 3) VADD(r8); if (R >= 4) VADD(r9); if (R >= 5) VADD(r10); if (R >= 6) VADD(r11);
      MFMA(acc2); if (R >= 1) VADD(r0); if (R >= 2) VADD(r1); if (R >= 3) VADD(r2); if (R >= 4) VADD(r3); if (R >= 5) VADD(r4); if (R >= 6) VADD(r5);
      MFMA(acc3); if (R >= 1) VADD(r6); if (R >= 2) VADD(r7); if (R >= 3) VADD(r8); if (R >= 4) VADD(r9); if (R >= 5) VADD(r10); if (R >= 6) VADD(r11);
    }
  } else if (mf) {
    for (int i = 0; i < iters; ++i) {
      MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3);
    }
  } else {
    for (int i = 0; i < iters; ++i) {  // 4 * R VALU per iteration, matching mode I's count
#pragma unroll
      for (int k = 0; k < R; ++k) {
        VADD(r0); VADD(r1); VADD(r2); VADD(r3);
      }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc0.x + acc1.y + acc2.z + acc3.w + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11;
}

template <int MODE, int R>
static float run(int* out, int iters, v4i a, v4i b) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, R>), dim3(256), dim3(512), 0, 0, out, 100, a, b);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, R>), dim3(256), dim3(512), 0, 0, out, iters, a, b);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int R>
static void row(int* out, int iters, v4i a, v4i b) {
  const float m = run<0, R>(out, iters, a, b), v = run<1, R>(out, iters, a, b), mv = run<2, R>(out, iters, a, b), il = run<3, R>(out, iters, a, b);
  const double per = 1e6 / (4.0 * iters);  // ns per MFMA slot
  printf("R=%d VALU per MFMA: MFMA wave alone %.2f ns/MFMA, VALU wave alone %.2f ns per %d VALU, both waves on the SIMD %.2f, one interleaved stream %.2f   (sum %.2f, max %.2f)\n", R,
         m * per, v * per, R, mv * per, il * per, (m + v) * per, (m > v ? m : v) * per);
}

int main() {
  int* out;
  hipMalloc(&out, 256 * 512 * sizeof(int));
  const v4i a = {0x01010101, 0x01000100, 0x00010001, 0x01010000}, b = {0x01ff01ff, -1, 0x01010101, 0x01ffff01};
  const int iters = 200000;
  row<1>(out, iters, a, b);
  row<2>(out, iters, a, b);
  row<3>(out, iters, a, b);
  row<4>(out, iters, a, b);
  row<6>(out, iters, a, b);
  return 0;
}
