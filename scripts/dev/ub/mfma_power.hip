// Microbenchmark (round 5): what does the f32 MFMA pipe sustain when its OPERANDS CHANGE from instruction to instruction, as they do in a
// real convolution, instead of staying the same per lane (scripts/dev/ub/mfma_shape.hip)?  12 x v_mfma_f32_16x16x4_f32 per iteration, two
// waves per SIMD, one block per CU, nothing else in the loop.  Operand sets: all zeros; one value per lane, constant over time; 16 different
// random values per lane cycled through (every MFMA sees new A and B).  Prints TFLOP/s and the mean shader clock of the launch
// (s_memrealtime is a constant 100 MHz counter, s_memtime counts shader cycles): is 157.3 TFLOP/s (2.4 GHz) what the chip clocks at under
// this load, or less?   build: hipcc -O3 --offload-arch=gfx950 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 3: mode 2 + one ds_read_b128 per MFMA; MODE 4: mode 3 + one 16-byte global load per lane per 4 MFMAs, streaming through `big`
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, unsigned long long* clk, int iters,
                                         const f32x4* __restrict__ big = nullptr, long long big_n = 0) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];     // 32 KB, sized at the launch (an unreferenced static array would be dropped)
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 512) lds[i] = f32x4{src[i & 8191], src[(i + 1) & 8191], src[(i + 2) & 8191], src[(i + 3) & 8191]};
  __syncthreads();
  f32x4 sink = {0, 0, 0, 0}, pend = {0, 0, 0, 0};
  long long gpos = ((long long)blockIdx.x * 512 + tid);
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float va = src[(tid * 16 + i) & 8191], vb = src[(8192 + tid * 16 + i * 7) & 16383];
    a[i] = MODE == 0 ? 0.f : MODE == 1 ? src[tid & 8191] : va;
    b[i] = MODE == 0 ? 0.f : MODE == 1 ? src[8192 + (tid & 8191)] : vb;
  }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 t[4];
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 15], b[(m * 5 + (m >> 4)) & 15], acc[m & 3], 0, 0, 0);
      if (MODE >= 3) {
        // (every result register stays reserved until the wait: a read whose result is dead would still land, later, in a register
        // the compiler has meanwhile given to something else)
        asm volatile("ds_read_b128 %0, %1" : "=v"(t[m & 3]) : "v"((unsigned)(((tid + 64 * m) & 2047) * 16)));
        if ((m & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
        if (m == 47) sink += t[0] + t[1] + t[2] + t[3];
      }
      if (MODE == 4 && (m & 3) == 3) {
        sink += __builtin_nontemporal_load(big + gpos);
        gpos += 256 * 512;
        if (gpos >= big_n) gpos -= big_n;
      }
      // MODE 5 / 6 / 7: one 16-byte load per lane per 8 / 16 / 32 MFMAs, consumed one group later (nothing waits for a load just issued)
      if (MODE >= 5 && (m & ((8 << (MODE - 5)) - 1)) == (8 << (MODE - 5)) - 1) {
        sink += pend;
        pend = __builtin_nontemporal_load(big + gpos);
        gpos += 256 * 512;
        if (gpos >= big_n) gpos -= big_n;
      }
    }
    // keep the magnitudes bounded without touching the operand registers: scale the accumulators down now and then
    if ((it & 63) == 63) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] *= 1e-3f;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  sink += pend;
  out[blockIdx.x * 512 + tid] = r + sink.x + sink.y + sink.z + sink.w + lds[tid].x;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const float* src, float* out, unsigned long long* clk, int iters, const f32x4* big = nullptr, long long big_n = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 32768, 0, src, out, clk, iters, big, big_n);
  hipEventRecord(e0);
  for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 32768, 0, src, out, clk, iters, big, big_n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
  unsigned long long h[512]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  const double tf = 48 * 2048.0 * iters * 8 * 256 / (ms * 1e-3) / 1e12;
  const int per = MODE == 4 ? 4 : MODE >= 5 ? (8 << (MODE - 5)) : 0;
  const double tbs = per ? tf * 1e12 / 2048.0 / per * 16 * 64 / 1e12 : 0.0;      // MFMAs/s (per wave) / per x 1 KiB per wave-load
  printf("%-60s %7.1f TFLOP/s  %7.2f ms  clock %.3f GHz  loads %.2f TB/s\n", name, tf, ms, cyc / real * 0.1, tbs);
}

int main() {
  float* src; float* out; unsigned long long* clk;
  hipMalloc(&src, 16384 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
  float h[16384];
  srand(7);
  for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  f32x4* big; const long long big_n = (1ll << 30) / 16; hipMalloc(&big, 1ll << 30); hipMemset(big, 0, 1ll << 30);
  const int iters = 120000;      // ~20 ms per launch: long enough for the clock governor
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("operands all zero", src, out, clk, iters);
    run<1>("one value per lane, constant over time", src, out, clk, iters);
    run<2>("16 random values per lane, new A and B per MFMA", src, out, clk, iters);
    run<3>("... + one ds_read_b128 per MFMA", src, out, clk, iters / 4);
    run<4>("... + 16 B per lane from HBM per 4 MFMAs", src, out, clk, iters / 4, big, big_n);
    run<5>("... + 16 B per lane per 8 MFMAs, consumed a group later", src, out, clk, iters / 4, big, big_n);
    run<6>("... + 16 B per lane per 16 MFMAs, consumed a group later", src, out, clk, iters / 4, big, big_n);
    run<7>("... + 16 B per lane per 32 MFMAs, consumed a group later", src, out, clk, iters / 4, big, big_n);
  }
  return 0;
}
