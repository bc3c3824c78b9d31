// mfma_i8_probe.hip -- operand / result lane maps of v_mfma_i32_16x16x32_i8 on gfx950, checked with exact integer data
// (the feature extractor's Gaussian runs on it: features.hip).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_i8_probe tools/probes/mfma_i8_probe.hip
// Assumed maps: lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15] in byte j of its 8-byte fragment;
// result register r of lane l is D[row 4 (l >> 4) + r][col l & 15].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const int8_t* A, const int8_t* B, const int* C, int* D) {  // A[16][32], B[32][16], C / D[16][16] row-major
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  long a = 0, b = 0;
  for (int j = 0; j < 8; ++j) {
    a |= (long) (uint8_t) A[i * 32 + 8 * g + j] << (8 * j);
    b |= (long) (uint8_t) B[(8 * g + j) * 16 + i] << (8 * j);
  }
  v4i c;
  for (int r = 0; r < 4; ++r) {
    c[r] = C[(4 * g + r) * 16 + i];
  }
  const v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    D[(4 * g + r) * 16 + i] = d[r];
  }
}

int main() {
  int8_t hA[16 * 32], hB[32 * 16];
  int hC[256], hD[256], ref[256];
  srand(7);
  for (int& c : hC) c = rand() % 100000 - 50000;
  for (int8_t& v : hA) v = (int8_t) (rand() % 256 - 128);
  for (int8_t& v : hB) v = (int8_t) (rand() % 256 - 128);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      int s = hC[i * 16 + j];
      for (int k = 0; k < 32; ++k) s += (int) hA[i * 32 + k] * (int) hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  int8_t *dA, *dB;
  int *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
  printf("v_mfma_i32_16x16x32_i8 with the assumed lane maps: %d of 256 results differ\n", bad);
  return bad != 0;
}
