// What a burst of LDS stores costs a workgroup that is otherwise doing vector arithmetic (the "terms" phase of the latency pipeline,
// csrc/align.hip lat_gn_body): W waves each run F dependent-ish fused multiply-adds and then S stores of `width` bytes per lane,
// followed by a workgroup barrier; cycles per round as seen by wave 0.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_store_probe.hip -o /tmp/lds_store_probe && /tmp/lds_store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int WIDTH, int S>
__global__ __launch_bounds__(1024, 1) void probe(int waves, int flops, int rounds, float seed, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  float a0 = seed + tid, a1 = seed * 2.0f, a2 = seed * 3.0f, a3 = seed * 0.5f;
  unsigned long long t0 = 0;
  __syncthreads();
  if (tid == 0) {
    t0 = clock64();
  }
  for (int r = 0; r < rounds; ++r) {
    if (wave < waves) {
      for (int i = 0; i < flops / 4; ++i) {
        a0 = fmaf(a0, 1.0001f, a1);
        a1 = fmaf(a1, 0.9999f, a2);
        a2 = fmaf(a2, 1.0002f, a3);
        a3 = fmaf(a3, 0.9998f, a0);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (WIDTH == 16) {
          reinterpret_cast<float4*>(smem)[s * 1024 + tid] = make_float4(a0, a1, a2, a3 + s);
        } else if (WIDTH == 8) {
          reinterpret_cast<float2*>(smem)[s * 1024 + tid] = make_float2(a0, a1 + s);
        } else {
          reinterpret_cast<float*>(smem)[s * 1024 + tid] = a0 + s;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = clock64() - t0;
  }
  sink[tid] = a0 + a1 + a2 + a3 + reinterpret_cast<float*>(smem)[tid];
}

template <int WIDTH, int S>
void run(int waves, int flops, unsigned long long* d_out, float* d_sink) {
  const int rounds = 200;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<WIDTH, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<WIDTH, S>), dim3(1), dim3(1024), 128 * 1024, 0, waves, flops, rounds, 1.0f, d_out, d_sink);
  }
  unsigned long long h = 0;
  hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("waves %2d  fma %3d  stores %d x %2d B : %7.0f cycles per round\n", waves, flops, S, WIDTH, (double) h / rounds);
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 64);
  hipMalloc(&d_sink, 4096 * 4);
  for (int waves : {1, 4, 8, 16}) {
    run<16, 0>(waves, 200, d_out, d_sink);
    run<16, 7>(waves, 200, d_out, d_sink);
    run<16, 7>(waves, 0, d_out, d_sink);
    run<8, 14>(waves, 0, d_out, d_sink);
    run<4, 28>(waves, 0, d_out, d_sink);
    run<16, 3>(waves, 200, d_out, d_sink);
  }
  return 0;
}
