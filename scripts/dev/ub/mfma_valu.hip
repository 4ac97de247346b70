// Microbenchmark: does VALU work issued next to f32 MFMAs cost MFMA throughput on gfx950?
// Each iteration: 12 independent v_mfma_f32_16x16x4_f32 + NV dependent-free v_add_f32 / v_and_b32 (template), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int tid = threadIdx.x;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = tid * 1e-3f, b = 1.f + tid * 1e-4f;
  float v[8];
  int w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = tid + i; w[i] = tid * 7 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 12; ++j) {
        const int i = (m * (NV / 12) + j) & 7;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        else if (KIND == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[i]) : "v"(tid));
        else asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[i]) : "v"(tid));
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i] + (float)w[i];
  out[blockIdx.x * 512 + tid] = r;
}

template <int NV, int KIND>
double run(float* out, int iters, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return 12 * 2048.0 * iters * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads += 256) {
    printf("waves/SIMD=%d  v_add_f32 per 12 MFMA: 0 -> %.1f TF | 12 -> %.1f | 24 -> %.1f | 48 -> %.1f | 96 -> %.1f\n", threads / 256,
           run<0, 0>(out, iters, threads), run<12, 0>(out, iters, threads), run<24, 0>(out, iters, threads), run<48, 0>(out, iters, threads),
           run<96, 0>(out, iters, threads));
    printf("              v_and_b32 per 12 MFMA: 12 -> %.1f | 24 -> %.1f | 48 -> %.1f | 96 -> %.1f ;  v_mul_lo_u32: 12 -> %.1f | 24 -> %.1f\n",
           run<12, 1>(out, iters, threads), run<24, 1>(out, iters, threads), run<48, 1>(out, iters, threads), run<96, 1>(out, iters, threads),
           run<12, 2>(out, iters, threads), run<24, 2>(out, iters, threads));
  }
  return 0;
}
