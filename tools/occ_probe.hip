#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" __global__ __launch_bounds__(512) void k512(int* p) { extern __shared__ int s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) & 511]; }
extern "C" __global__ __launch_bounds__(256) void k256(int* p) { extern __shared__ int s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 1) & 255]; }
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n", pr.sharedMemPerBlock, pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock);
  for (size_t lds : {16384, 32768, 40000, 49152, 53000, 65008, 65536, 81920, 100000, 163840}) {
    hipFuncSetAttribute((const void*) k512, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipFuncSetAttribute((const void*) k256, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    int a = -1, b = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k512, 512, lds);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k256, 256, lds);
    printf("lds %zu: k512 %d blocks/CU, k256 %d blocks/CU\n", lds, a, b);
  }
  return 0;
}
