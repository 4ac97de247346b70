// Microbenchmark: how fast can one 512-thread block per CU pull 2-D halo tiles into LDS by LDS-DMA when it does nothing
// else?  Pattern of conv2's wgrad / forward: x [N][256][256][32] fp32 (805 MB at N = 96), tile = 9 rows x 33 pixels x
// 128 B (39 pieces of 1 KiB), one barrier per tile, DEPTH tiles in flight.  Compared with the ~5 B/clk/CU those kernels see.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int DEPTH, int NWAVE_DMA>
__global__ __launch_bounds__(512) void k(const float* x, int N, int tiles_per_block, float* sink) {
  constexpr int H = 256, W = 256, C = 32, PIECES = 39;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tiles_x = W / 32, tiles_y = H / 8, per_img = tiles_x * tiles_y;
  int tile = blockIdx.x * tiles_per_block;
  const int tend = tile + tiles_per_block;
  auto issue = [&](int t, int buf) {
    const int n = t / per_img, rem = t - n * per_img, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const float* base = x + (((long long)n * H + ty * 8) * W + tx * 32) * C;
    if (wid < NWAVE_DMA) {
      for (int k = wid; k < PIECES; k += NWAVE_DMA) {
        const int sl = k * 64 + lane;               // float4 slot: row = sl / 264, (pixel, quad) = sl % 264
        const int row = sl / 264, rem2 = sl - row * 264;
        int iy = ty * 8 + row, ix = tx * 32 + (rem2 >> 3);
        const float* src = base + ((long long)row * W) * C + rem2 * 4;
        if (iy >= H || ix >= W) src = x;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (buf * PIECES + k) * 256), 16, 0, 0);
      }
    }
  };
  for (int d = 0; d < DEPTH - 1 && tile + d < tend; ++d) issue(tile + d, d);
  int buf = 0;
  float acc = 0.f;
  for (; tile < tend; ++tile) {
    if (tile + DEPTH - 1 < tend) issue(tile + DEPTH - 1, (buf + DEPTH - 1) % DEPTH);
    // wait until at most (DEPTH - 1) tiles' pieces of this wave are outstanding: simple version waits for all when DEPTH == 2
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __syncthreads();
    acc += smem[buf * PIECES * 256 + tid];
    __syncthreads();
    buf = (buf + 1) % DEPTH;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 123.456f) sink[0] = acc;
}

template <int DEPTH, int NW>
double run(const float* x, int N, float* sink) {
  const int total = N * 8 * 32, tpb = total / 256;
  hipFuncSetAttribute((const void*)k<DEPTH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 39 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<DEPTH, NW>), dim3(256), dim3(512), DEPTH * 39 * 1024, 0, x, N, tpb, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<DEPTH, NW>), dim3(256), dim3(512), DEPTH * 39 * 1024, 0, x, N, tpb, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  return (double)tpb * 256 * 39 * 1024 / (ms * 1e-3) / 1e12;
}

int main() {
  const int N = 96;
  float *x, *sink;
  hipMalloc(&x, (size_t)N * 256 * 256 * 32 * 4 + (1 << 20));
  hipMalloc(&sink, 64);
  hipMemset(x, 0, (size_t)N * 256 * 256 * 32 * 4);
  printf("LDS-DMA of 9x33-pixel halo tiles (39 KiB each), 1 block of 512 threads per CU, TB/s of staged bytes:\n");
  printf("  8 waves issue, 2 tiles deep: %.2f   3 deep: %.2f\n", run<2, 8>(x, N, sink), run<3, 8>(x, N, sink));
  printf("  4 waves issue, 2 tiles deep: %.2f   3 deep: %.2f\n", run<2, 4>(x, N, sink), run<3, 4>(x, N, sink));
  printf("  1 wave issues, 2 tiles deep: %.2f   3 deep: %.2f\n", run<2, 1>(x, N, sink), run<3, 1>(x, N, sink));
  return 0;
}
