// mfma_fp4_probe.hip -- v_mfma_scale_f32_16x16x128_f8f6f4 with 4-bit (E2M1) operands on gfx950, as a binary dot-product engine:
// (1) do 0 / +1 / -1 decode as 0x0 / 0x2 / 0xA, do A and B pair position p of lane group g with each other, is the C/D layout the
// usual one (row 4 (l >> 4) + r, column l & 15), are the sums exact; (2) cycles per instruction, and how much vector issue it leaves.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/p tools/probes/mfma_fp4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x, i = l & 15;
  // A: row i has its first i positions (of the lane's 32) = 1.0, B: column j has its first j positions = -1.0, the rest +1.0
  unsigned long long a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
  for (int p = 0; p < 32; ++p) {
    const unsigned long long an = p < i ? 0x2ull : 0x0ull, bn = p < i ? 0xAull : 0x2ull;
    if (p < 16) { a_lo |= an << (4 * p); b_lo |= bn << (4 * p); } else { a_hi |= an << (4 * (p - 16)); b_hi |= bn << (4 * (p - 16)); }
  }
  v8i a = {(int) a_lo, (int) (a_lo >> 32), (int) a_hi, (int) (a_hi >> 32), 0, 0, 0, 0};
  v8i b = {(int) b_lo, (int) (b_lo >> 32), (int) b_hi, (int) (b_hi >> 32), 0, 0, 0, 0};
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) out[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

template <int R>
__global__ __launch_bounds__(256) void rate(int* out, int iters, v4i a, v4i b) {
  v4f acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
  int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5;
  const int one = 1 + (int) (threadIdx.x & 1), sc = 0x7f7f7f7f;
  a[0] += threadIdx.x & 1;
#define MF(acc) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(acc) : "v"(a), "v"(b), "v"(sc))
#define VA(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(one))
  for (int i = 0; i < iters; ++i) {
    MF(acc0); if (R >= 1) VA(r0); if (R >= 2) VA(r1); if (R >= 3) VA(r2); if (R >= 4) VA(r3); if (R >= 5) VA(r4); if (R >= 6) VA(r5);
    MF(acc1); if (R >= 1) VA(r0); if (R >= 2) VA(r1); if (R >= 3) VA(r2); if (R >= 4) VA(r3); if (R >= 5) VA(r4); if (R >= 6) VA(r5);
    MF(acc2); if (R >= 1) VA(r0); if (R >= 2) VA(r1); if (R >= 3) VA(r2); if (R >= 4) VA(r3); if (R >= 5) VA(r4); if (R >= 6) VA(r5);
    MF(acc3); if (R >= 1) VA(r0); if (R >= 2) VA(r1); if (R >= 3) VA(r2); if (R >= 4) VA(r3); if (R >= 5) VA(r4); if (R >= 6) VA(r5);
  }
  out[blockIdx.x * 256 + threadIdx.x] = (int) (acc0[0] + acc1[1] + acc2[2] + acc3[3]) + r0 + r1 + r2 + r3 + r4 + r5;
}

template <int R>
static void time_it(int* out, v4i a, v4i b) {
  const int iters = 100000;
  hipEvent_t e0, e1;
  (void) hipEventCreate(&e0);
  (void) hipEventCreate(&e1);
  hipLaunchKernelGGL((rate<R>), dim3(256), dim3(256), 0, 0, out, 100, a, b);  // one wave per SIMD
  (void) hipEventRecord(e0);
  hipLaunchKernelGGL((rate<R>), dim3(256), dim3(256), 0, 0, out, iters, a, b);
  (void) hipEventRecord(e1);
  (void) hipEventSynchronize(e1);
  float ms;
  (void) hipEventElapsedTime(&ms, e0, e1);
  printf("one stream per SIMD, %d VALU behind every fp4 16x16x128 MFMA: %.2f ns per MFMA slot\n", R, ms * 1e6 / (4.0 * iters));
}

int main() {
  float* out;
  (void) hipMalloc(&out, 256 * sizeof(float));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  float h[256];
  (void) hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      const float want = 4.0f * (float) (i - 2 * (i < j ? i : j));
      if (h[i * 16 + j] != want) { if (bad < 8) printf("  [%d][%d] = %g, expected %g\n", i, j, h[i * 16 + j], want); ++bad; }
    }
  printf("layout / exactness: %d of 256 outputs differ from 4 (i - 2 min(i, j))\n", bad);
  int* iout;
  (void) hipMalloc(&iout, 256 * 256 * sizeof(int));
  const v4i a = {0x22220202, 0x20202222, 0x02022020, 0x22222222}, b = {0x2a2a2a2a, (int) 0xa2a2a2a2, 0x22aa22aa, (int) 0xaaaa2222};
  time_it<0>(iout, a, b);
  time_it<1>(iout, a, b);
  time_it<2>(iout, a, b);
  time_it<3>(iout, a, b);
  time_it<4>(iout, a, b);
  time_it<6>(iout, a, b);
  return 0;
}
