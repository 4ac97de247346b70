// Microbenchmark: LDS-fed f32 MFMA loops (no global traffic) to find the practical ceiling of the conv inner loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// variant 0: per iteration 4 ds_read_b128 + 12 mfma16 (conv2 fwd pattern), software pipelined
// variant 1: per iteration 2 ds_read_b128 + 4 mfma 32x32x2 (same FLOPs as 8 mfma16)
// variant 2: mfma16 only (operands in registers)         variant 3: mfma32 only
template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  __syncthreads();
  const f32x4* s4 = reinterpret_cast<const f32x4*>(smem);
  const int base = lane + (tid >> 6) * 64;
  if (V == 0 || V == 2) {
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 a = s4[base], b0 = s4[base + 512], b1 = s4[base + 1024], b2 = s4[base + 1536];
    for (int it = 0; it < iters; ++it) {
      f32x4 an = a, b0n = b0, b1n = b1, b2n = b2;
      if (V == 0) {
        const int o = (it & 7) * 64;
        an = s4[(base + o) & 4095]; b0n = s4[(base + 512 + o) & 4095]; b1n = s4[(base + 1024 + o) & 4095]; b2n = s4[(base + 1536 + o) & 4095];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[s], a[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[s], a[s], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2[s], a[s], acc[2], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a = an; b0 = b0n; b1 = b1n; b2 = b2n;
    }
    f32x4 r = acc[0] + acc[1] + acc[2];
    out[blockIdx.x * 512 + tid] = r.x + r.y + r.z + r.w;
  } else {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 a = s4[base], b0 = s4[base + 512], b1 = s4[base + 1024];
    for (int it = 0; it < iters; ++it) {
      f32x4 an = a, b0n = b0, b1n = b1;
      if (V == 1) {
        const int o = (it & 7) * 64;
        an = s4[(base + o) & 4095]; b0n = s4[(base + 512 + o) & 4095]; b1n = s4[(base + 1024 + o) & 4095];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a[s], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a = an; b0 = b0n; b1 = b1n;
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    out[blockIdx.x * 512 + tid] = r;
  }
}

template <int V>
double run(float* out, int iters, int blocks_per_cu, double flop_per_iter_wave) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  return flop_per_iter_wave * iters * 8.0 * blocks / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 512 * 4);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 20000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    printf("blocks/CU=%d (waves/SIMD=%d): mfma16+lds %.1f TF | mfma32+lds %.1f TF | mfma16 only %.1f TF | mfma32 only %.1f TF\n", bpc, 2 * bpc,
           run<0>(out, iters, bpc, 12 * 2048.0), run<1>(out, iters, bpc, 8 * 4096.0), run<2>(out, iters, bpc, 12 * 2048.0),
           run<3>(out, iters, bpc, 8 * 4096.0));
  }
  return 0;
}
