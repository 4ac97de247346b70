// Microbenchmark: what does a pure streaming WRITE of conv1's output size (805 MB) reach, with plain and with
// non-temporal 16-byte stores, alone and beside a 145 MB read stream (conv1's forward moves 830 MB out + 145 MB in
// in 165-190 us = 5.1-5.9 TB/s)?
//   hipcc --offload-arch=gfx950 -O3 hbm_write.hip -o hbm_write && ./hbm_write
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool READ>
__global__ __launch_bounds__(256) void k(f32x4* __restrict__ out, const f32x4* __restrict__ in, long long n4, long long nin4) {
  const long long stride = (long long)gridDim.x * 256;
  f32x4 acc = {1.f, 2.f, 3.f, 4.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    if (READ && (i & 7) == 0) acc += in[(i >> 3) % nin4];       // one 16-byte read per eight 16-byte writes (~ 145 : 830 would be 1 : 5.7)
    if (NT) __builtin_nontemporal_store(acc, out + i); else out[i] = acc;
  }
}

int main() {
  const long long n4 = 805306368ll / 16, nin4 = 150994944ll / 16;
  f32x4 *out, *in;
  hipMalloc(&out, n4 * 16);
  hipMalloc(&in, nin4 * 16);
  hipMemset(in, 0, nin4 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 4; ++variant)
    for (int blocks : {2048, 8192, 32768}) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        switch (variant) {
          case 0: hipLaunchKernelGGL((k<false, false>), dim3(blocks), dim3(256), 0, 0, out, in, n4, nin4); break;
          case 1: hipLaunchKernelGGL((k<true, false>), dim3(blocks), dim3(256), 0, 0, out, in, n4, nin4); break;
          case 2: hipLaunchKernelGGL((k<false, true>), dim3(blocks), dim3(256), 0, 0, out, in, n4, nin4); break;
          default: hipLaunchKernelGGL((k<true, true>), dim3(blocks), dim3(256), 0, 0, out, in, n4, nin4); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%s stores%s, %5d blocks: %.1f us = %.2f TB/s written\n", (variant & 1) ? "non-temporal" : "plain       ",
             variant >= 2 ? " + read stream" : "              ", blocks, best * 1e3, n4 * 16 / (best * 1e-3) / 1e12);
    }
  return 0;
}
