// Microbenchmark (round 3): do ds_read_b128 / v_add_f32 fillers cost less MFMA time when the accumulators live in AGPRs
// (the LDS return data and the VALU results are written to VGPRs; the MFMA's C / D then use the other half of the file)?
// 12 x v_mfma_f32_16x16x4_f32 per iteration on 4 accumulator tiles, NV v_add_f32 / NL ds_read_b128 spread between them,
// two waves per SIMD (512 threads, one block per CU).  ACC = 0: "+v" accumulators, 1: "+a".
//   hipcc --offload-arch=gfx950 -O3 mfma_agpr.hip -o mfma_agpr && ./mfma_agpr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC, int NV, int NL>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f - 2.f;
  __syncthreads();
  float a = 0.37f + tid * 1e-3f, b = 1.f - tid * 1e-4f;
  float v[8];
  f32x4 ld[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = tid + i;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      else     asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV / 12; ++j) {
        const int i = (m * (NV / 12) + j) & 7;
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      }
#pragma unroll
      for (int j = 0; j < NL / 12; ++j) {
        const int i = (m * (NL / 12) + j) & 3;
        f32x4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((tid & 255) * 16)));
        ld[i] = t;
      }
    }
    if (NL) {
      asm volatile("s_waitcnt lgkmcnt(0)");
      a += ld[0].x * 1e-30f;
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + ld[i].y;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 512 + tid] = r;
}

template <int ACC, int NV, int NL>
double run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ACC, NV, NL>), dim3(256), dim3(512), 32768, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<ACC, NV, NL>), dim3(256), dim3(512), 32768, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return 12 * 2048.0 * iters * 8 * 256 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  printf("two waves per SIMD, TFLOP/s       bare   12 v_add  24 v_add  12 ds_read_b128  24 ds_read_b128  24 v_add + 12 ds_read\n");
  printf("accumulators in VGPRs          %7.1f  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f\n", run<0, 0, 0>(out, iters), run<0, 12, 0>(out, iters),
         run<0, 24, 0>(out, iters), run<0, 0, 12>(out, iters), run<0, 0, 24>(out, iters), run<0, 24, 12>(out, iters));
  printf("accumulators in AGPRs          %7.1f  %7.1f  %7.1f  %7.1f  %7.1f  %7.1f\n", run<1, 0, 0>(out, iters), run<1, 12, 0>(out, iters),
         run<1, 24, 0>(out, iters), run<1, 0, 12>(out, iters), run<1, 0, 24>(out, iters), run<1, 24, 12>(out, iters));
  return 0;
}
