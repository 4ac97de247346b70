// Microbenchmark: the K-loop STRUCTURE of the conv GEMM kernels with no global traffic.
// 256 threads = 4 waves (2 x 2), block tile 128 x 64, wave tile 64 x 32 (TJ = 4, TI = 2), f32 MFMA 16x16x4.
// Per K-step (BK = 16): 4 A + 2 B ds_read_b128 fragments, 32 MFMAs.   Variants:
//   0: barrier; reads; MFMAs                      (what conv_gemm / conv_dma do)
//   1: reads for step k+1 issued before the MFMAs of step k (register double buffer), barrier per step
//   2: as 0 without the barrier                   3: as 1 without the barrier
//   4: as 1 with BK = 32 (one barrier per 64 MFMAs)
//   5: MFMAs only (operands stay in registers)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 64, D = 4;
constexpr int STAGE_F4 = 4 * BM + 4 * BN;   // 768 float4 = 12 KiB

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, int lds_extra) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < D * STAGE_F4 * 4; i += 256) smem[i] = (float)(((i + 17) * 2654435761u) >> 20) * 1e-3f - 2.f;
  __syncthreads();
  const f32x4* s4 = reinterpret_cast<const f32x4*>(smem);
  const int r = lane & 15, q = lane >> 4;
  const int pixbase = (wid & 1) * 64, cobase = (wid >> 1) * 32;
  f32x4 acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  auto rd = [&](int buf, int kb, f32x4 (&xf)[4], f32x4 (&wf)[2]) {
    const f32x4* a = s4 + buf * STAGE_F4;
    const f32x4* b = a + 4 * BM;
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = a[((kb * 4 + q) & 3) * BM + pixbase + j * 16 + r];
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = b[((kb * 4 + q) & 3) * BN + cobase + i * 16 + r];
  };
  auto mm = [&](f32x4 (&xf)[4], f32x4 (&wf)[2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s], xf[j][s], acc[i][j], 0, 0, 0);
  };
  f32x4 xa[4], wa[2], xb[4], wb[2];
  int buf = 0;
  if (V == 0 || V == 2) {
    for (int it = 0; it < iters; ++it) {
      if (V == 0) asm volatile("s_barrier" ::: "memory");
      rd(buf, 0, xa, wa);
      mm(xa, wa);
      buf = (buf + 1) & 3;
    }
  } else if (V == 1 || V == 3) {
    rd(0, 0, xa, wa);
    for (int it = 0; it < iters; it += 2) {
      if (V == 1) asm volatile("s_barrier" ::: "memory");
      rd((buf + 1) & 3, 0, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      mm(xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      if (V == 1) asm volatile("s_barrier" ::: "memory");
      rd((buf + 2) & 3, 0, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      mm(xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      buf = (buf + 2) & 3;
    }
  } else if (V == 4) {
    rd(0, 0, xa, wa);
    for (int it = 0; it < iters; it += 2) {
      asm volatile("s_barrier" ::: "memory");
      rd((buf + 1) & 3, 0, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      mm(xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      rd((buf + 2) & 3, 0, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      mm(xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      buf = (buf + 2) & 3;
    }
  } else {
    rd(0, 0, xa, wa);
    for (int it = 0; it < iters; ++it) mm(xa, wa);
  }
  f32x4 t = {0, 0, 0, 0};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + tid] = t.x + t.y + t.z + t.w;
}

template <int V>
double run(float* out, int iters, int bpc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // LDS per block chosen so that exactly bpc blocks fit a CU (160 KiB)
  int lds = 160 * 1024 / bpc - 256;
  if (lds < D * STAGE_F4 * 16) lds = D * STAGE_F4 * 16;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int blocks = 256 * bpc;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, out, iters, 0);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, out, iters, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  return 32 * 2048.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000;
  for (int bpc = 1; bpc <= 4; ++bpc)
    printf("blocks/CU=%d (waves/SIMD=%d): v0 %.1f | v1(prefetch) %.1f | v2(no bar) %.1f | v3(prefetch,no bar) %.1f | v4(bk32) %.1f | v5(mfma only) %.1f TF\n",
           bpc, bpc, run<0>(out, iters, bpc), run<1>(out, iters, bpc), run<2>(out, iters, bpc), run<3>(out, iters, bpc),
           run<4>(out, iters, bpc), run<5>(out, iters, bpc));
  return 0;
}
