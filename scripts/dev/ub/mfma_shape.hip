// Microbenchmark (round 3): is v_mfma_f32_32x32x2_f32 a better carrier than v_mfma_f32_16x16x4_f32 for loops that also
// issue VALU / LDS-read instructions?  Same FLOP per iteration in both shapes (12 x 16x16x4 = 6 x 32x32x2 = 24576 FLOP
// per wave), operands on random (non-trivial) data, NV v_add_f32 (or NL ds_read_b128) spread evenly between the MFMAs,
// one or two waves per SIMD.  Prints TFLOP/s chip-wide: the shape whose number drops less per filler hides them better.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NV, int NL>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f - 2.f;
  __syncthreads();
  const f32x4* s4 = reinterpret_cast<const f32x4*>(smem) + (tid & 255);
  float a = 0.37f + tid * 1e-3f, b = 1.f - tid * 1e-4f;
  float v[8];
  f32x4 ld[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = tid + i;
  constexpr int NM = SHAPE == 16 ? 12 : 6;
  f32x4 acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 acc16[2] = {{0}, {0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (SHAPE == 16) acc4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m & 3], 0, 0, 0);
      else acc16[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[m & 1], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / NM; ++j) {
        const int i = (m * (NV / NM) + j) & 7;
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      }
#pragma unroll
      for (int j = 0; j < NL / NM; ++j) {
        const int i = (m * (NL / NM) + j) & 3;
        // conflict-free ds_read_b128 (consecutive lanes, consecutive granules); the asm keeps one read per slot
        f32x4 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(((tid & 255) * 16))), "n"(0));
        ld[i] = t;
      }
    }
    if (NL) {   // consume the reads once per iteration (one wait per iteration, as a pipelined loop would)
      asm volatile("s_waitcnt lgkmcnt(0)");
      a += ld[0].x * 1e-30f;
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w + ld[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc16[0][i] + acc16[1][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 512 + tid] = r + s4[0].x;
}

template <int SHAPE, int NV, int NL>
double run(float* out, int iters, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(256), dim3(threads), 32768, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, NV, NL>), dim3(256), dim3(threads), 32768, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return 12 * 2048.0 * iters * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads += 256) {
    printf("waves/SIMD=%d  v_add_f32 per 24576 FLOP:   0      12      24      48      96\n", threads / 256);
    printf("   16x16x4 (12 MFMA)             %7.1f %7.1f %7.1f %7.1f %7.1f TF\n", run<16, 0, 0>(out, iters, threads),
           run<16, 12, 0>(out, iters, threads), run<16, 24, 0>(out, iters, threads), run<16, 48, 0>(out, iters, threads),
           run<16, 96, 0>(out, iters, threads));
    printf("   32x32x2 ( 6 MFMA)             %7.1f %7.1f %7.1f %7.1f %7.1f TF\n", run<32, 0, 0>(out, iters, threads),
           run<32, 12, 0>(out, iters, threads), run<32, 24, 0>(out, iters, threads), run<32, 48, 0>(out, iters, threads),
           run<32, 96, 0>(out, iters, threads));
    printf("               ds_read_b128 per 24576 FLOP:        6      12      24\n");
    printf("   16x16x4                                     - %7.1f %7.1f TF\n", run<16, 0, 12>(out, iters, threads), run<16, 0, 24>(out, iters, threads));
    printf("   32x32x2                               %7.1f %7.1f %7.1f TF\n", run<32, 0, 6>(out, iters, threads), run<32, 0, 12>(out, iters, threads),
           run<32, 0, 24>(out, iters, threads));
    printf("   both: 24 v_add_f32 + 12 ds_read_b128:  16x16x4 %7.1f   32x32x2 %7.1f TF\n", run<16, 24, 12>(out, iters, threads),
           run<32, 24, 12>(out, iters, threads));
  }
  return 0;
}
