// Issue-rate probe for gfx950: wave64 vector instructions per cycle ONE CU retires, per instruction kind, at 1, 2, 4 and 8 waves
// per SIMD (eight independent chains per wave, no memory).  The kernels of this repository are made of these instructions; the
// table (profiles/r04/valu_issue_rates.txt) says which of them are full rate (~1.7 per cycle and CU at 2-4 waves per SIMD, ~2.3 at 8)
// and which are half rate.
//   hipcc --offload-arch=gfx950 -O3 -w tools/valu_probe.hip -o tools/bin/valu_probe && tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHAIN8(OP, TAIL)                                                                                                        \
  asm volatile(OP " %0, %0, %8" TAIL "\n" OP " %1, %1, %8" TAIL "\n" OP " %2, %2, %8" TAIL "\n" OP " %3, %3, %8" TAIL "\n" OP \
                  " %4, %4, %8" TAIL "\n" OP " %5, %5, %8" TAIL "\n" OP " %6, %6, %8" TAIL "\n" OP " %7, %7, %8" TAIL "\n"   \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                                \
               : "v"(c))

enum { ADD, XOR, BCNT, ALIGNBIT, ALIGNBYTE, PERM, LSHL_ADD, ADD3, AND_OR, BFE, MAD24, MIN3, MAX3, PK_ADD16, PK_MIN16, PK_SUB16, DOT4, DOT2, SAD, MUL_LO,
       FMA, MULF, PK_FMA, PK_MULF, RCP, CNDMASK, MINU, FMAC, FMA3, CND_S, CMP_CND, AND, LSHL, MUL24, MAXF, CVT, SUBREV, N_KINDS };

template <int KIND>
__global__ __launch_bounds__(1024) void probe(uint32_t* out, long long* cycles, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  const uint32_t c = 0x3f800000u + (threadIdx.x & 1023u);  // (a normal float when read as one)
  const uint32_t c2 = 0x3f800100u + threadIdx.x;
  typedef float fpair __attribute__((ext_vector_type(2)));
  fpair p0 = {1.f, 2.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
  const fpair pc = {1.0000001f, 0.9999999f};
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == ADD) CHAIN8("v_add_u32", "");
      if (KIND == XOR) CHAIN8("v_xor_b32", "");
      if (KIND == BCNT) CHAIN8("v_bcnt_u32_b32", "");
      if (KIND == ALIGNBIT) CHAIN8("v_alignbit_b32", ", 31");
      if (KIND == ALIGNBYTE) CHAIN8("v_alignbyte_b32", ", 1");
      if (KIND == PERM) CHAIN8("v_perm_b32", ", %8");
      if (KIND == LSHL_ADD) CHAIN8("v_lshl_add_u32", ", 1");
      if (KIND == ADD3) CHAIN8("v_add3_u32", ", %8");
      if (KIND == AND_OR) CHAIN8("v_and_or_b32", ", %8");
      if (KIND == BFE) CHAIN8("v_bfe_u32", ", 5");
      if (KIND == MAD24) CHAIN8("v_mad_u32_u24", ", %8");
      if (KIND == MIN3) CHAIN8("v_min3_u32", ", %8");
      if (KIND == MAX3) CHAIN8("v_max3_u32", ", %8");
      if (KIND == PK_ADD16) CHAIN8("v_pk_add_u16", "");
      if (KIND == PK_MIN16) CHAIN8("v_pk_min_u16", "");
      if (KIND == PK_SUB16) CHAIN8("v_pk_sub_i16", "");
      if (KIND == DOT4) CHAIN8("v_dot4_u32_u8", ", %8");
      if (KIND == DOT2) CHAIN8("v_dot2_u32_u16", ", %8");
      if (KIND == SAD) CHAIN8("v_sad_u8", ", %8");
      if (KIND == MUL_LO) CHAIN8("v_mul_lo_u32", "");
      if (KIND == FMA) CHAIN8("v_fma_f32", ", %8");
      if (KIND == MULF) CHAIN8("v_mul_f32", "");
      if (KIND == RCP) {
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
      if (KIND == CNDMASK) CHAIN8("v_cndmask_b32", ", vcc");
      if (KIND == MINU) CHAIN8("v_min_u32", "");
      if (KIND == FMAC) CHAIN8("v_fmac_f32", "");  // VOP2: d += s0 * s1
      if (KIND == FMA3) {
        asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2));
      }
      if (KIND == CND_S) CHAIN8("v_cndmask_b32", ", s[20:21]");
      if (KIND == CMP_CND) {
        asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                     "v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");
      }
      if (KIND == AND) CHAIN8("v_and_b32", "");
      if (KIND == LSHL) CHAIN8("v_lshlrev_b32", "");
      if (KIND == MUL24) CHAIN8("v_mul_u32_u24", "");
      if (KIND == MAXF) CHAIN8("v_max_f32", "");
      if (KIND == SUBREV) CHAIN8("v_subrev_u32", "");
      if (KIND == CVT) {
        asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
      if (KIND == PK_FMA || KIND == PK_MULF) {
        if (KIND == PK_FMA) {
          asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                       "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        } else {
          asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                       "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        }
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t) (p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y);
  if ((threadIdx.x & 63) == 0) {
    atomicMax((unsigned long long*) &cycles[blockIdx.x], (unsigned long long) (t1 - t0));
  }
}

template <int KIND>
static void run(const char* name, uint32_t* out, long long* cyc, int cus) {
  const int iters = 2048;
  printf("%-18s", name);
  for (int waves_per_simd : {1, 2, 3, 4, 5, 6, 8}) {
    const int waves_per_cu  = 4 * waves_per_simd;
    const int blocks_per_cu = waves_per_cu > 16 ? 2 : 1;  // (two workgroups per CU beyond 1024 threads)
    const int threads       = 64 * waves_per_cu / blocks_per_cu;
    hipMemset(cyc, 0, sizeof(long long) * cus * 2);
    hipLaunchKernelGGL(probe<KIND>, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, cyc, sizeof(long long) * cus * blocks_per_cu, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < cus * blocks_per_cu; ++i) mean += (double) h[i];
    mean /= cus * blocks_per_cu;
    printf("  %5.2f", (double) waves_per_cu * iters * 64 / mean);
  }
  printf("\n");
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  uint32_t* out;
  long long* cyc;
  hipMalloc(&out, sizeof(uint32_t) * 1024 * cus * 2);
  hipMalloc(&cyc, sizeof(long long) * cus * 2);
  printf("gfx950, %d CUs: wave64 vector instructions per clock64 tick and CU (eight independent chains per wave, all CUs busy)\n", cus);
  printf("%-18s  1 w/SIMD  2      3      4      5      6      8\n", "instruction");
  run<ADD>("v_add_u32", out, cyc, cus);
  run<XOR>("v_xor_b32", out, cyc, cus);
  run<MINU>("v_min_u32", out, cyc, cus);
  run<CNDMASK>("v_cndmask_b32", out, cyc, cus);
  run<BCNT>("v_bcnt_u32_b32", out, cyc, cus);
  run<ALIGNBIT>("v_alignbit_b32", out, cyc, cus);
  run<ALIGNBYTE>("v_alignbyte_b32", out, cyc, cus);
  run<PERM>("v_perm_b32", out, cyc, cus);
  run<LSHL_ADD>("v_lshl_add_u32", out, cyc, cus);
  run<ADD3>("v_add3_u32", out, cyc, cus);
  run<AND_OR>("v_and_or_b32", out, cyc, cus);
  run<BFE>("v_bfe_u32", out, cyc, cus);
  run<MAD24>("v_mad_u32_u24", out, cyc, cus);
  run<MIN3>("v_min3_u32", out, cyc, cus);
  run<MAX3>("v_max3_u32", out, cyc, cus);
  run<PK_ADD16>("v_pk_add_u16", out, cyc, cus);
  run<PK_MIN16>("v_pk_min_u16", out, cyc, cus);
  run<PK_SUB16>("v_pk_sub_i16", out, cyc, cus);
  run<DOT4>("v_dot4_u32_u8", out, cyc, cus);
  run<DOT2>("v_dot2_u32_u16", out, cyc, cus);
  run<SAD>("v_sad_u8", out, cyc, cus);
  run<MUL_LO>("v_mul_lo_u32", out, cyc, cus);
  run<FMA>("v_fma_f32", out, cyc, cus);
  run<MULF>("v_mul_f32", out, cyc, cus);
  run<PK_FMA>("v_pk_fma_f32", out, cyc, cus);
  run<PK_MULF>("v_pk_mul_f32", out, cyc, cus);
  run<RCP>("v_rcp_f32", out, cyc, cus);
  run<FMAC>("v_fmac_f32 (VOP2)", out, cyc, cus);
  run<FMA3>("v_fma 3 regs", out, cyc, cus);
  run<CND_S>("v_cndmask sgpr", out, cyc, cus);
  run<CMP_CND>("cmp+cndmask x4", out, cyc, cus);
  run<AND>("v_and_b32", out, cyc, cus);
  run<LSHL>("v_lshlrev_b32", out, cyc, cus);
  run<MUL24>("v_mul_u32_u24", out, cyc, cus);
  run<MAXF>("v_max_f32", out, cyc, cus);
  run<SUBREV>("v_subrev_u32", out, cyc, cus);
  run<CVT>("v_cvt_f32_u32", out, cyc, cus);
  return 0;
}
