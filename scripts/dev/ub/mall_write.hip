// Microbenchmark: does the 256 MB memory-side cache (MALL / Infinity Cache) absorb a write burst that fits it?
// conv1's forward writes 805 MB per step at HBM speed; cut into 4 chunks of 201 MB with conv2's (MFMA-bound) forward of
// the chunk in between, its stores could land in the cache and drain to HBM behind the next kernel - if the cache takes
// writes faster than HBM does.
//   (a) one streaming write of S MB, S = 50 .. 805, fresh region each time (rotating through 1.6 GB);
//   (b) the same with a ~100 us ALU-only kernel between the writes (time for the cache to drain).
//   hipcc --offload-arch=gfx950 -O3 mall_write.hip -o mall_write && ./mall_write
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(f32x4* __restrict__ out, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = v;
}
__global__ __launch_bounds__(256) void spin(float* out, int iters) {      // ALU only: ~ iters * 4 cycles per wave
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 123.456f) out[0] = a;
}

int main() {
  const long long total = 1610612736ll;     // 1.5 GiB pool
  char* pool;
  float* dummy;
  hipMalloc(&pool, total);
  hipMalloc(&dummy, 64);
  hipMemset(pool, 0, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int delay = 0; delay < 2; ++delay)
    for (long long mb : {50ll, 100ll, 201ll, 402ll, 805ll}) {
      const long long bytes = mb * 1000000ll / 4096 * 4096, n4 = bytes / 16;
      long long off = 0;
      double sum = 0;
      float best = 1e9f;
      const int reps = 12;
      for (int rep = 0; rep < reps; ++rep) {
        if (off + bytes > total) off = 0;
        if (delay) hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, dummy, 60000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(wr, dim3(8192), dim3(256), 0, 0, (f32x4*)(pool + off), n4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
        off += bytes;
      }
      printf("%s %4lld MB per write, fresh region: mean %.1f us = %.2f TB/s (best %.1f us = %.2f TB/s)\n",
             delay ? "after a ~100 us ALU kernel," : "back to back,              ", mb, sum / (reps - 2) * 1e3,
             bytes / (sum / (reps - 2) * 1e-3) / 1e12, best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
  // how long is the spin kernel?
  hipEventRecord(e0);
  hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, dummy, 60000);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("spin kernel: %.1f us\n", ms * 1e3);
  return 0;
}
