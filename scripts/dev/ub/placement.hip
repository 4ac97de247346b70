// Microbenchmark: where do the blocks of a SMALL grid land?  Each block notes its (XCC, SE, CU) and spins 30k shader
// clocks (~14 us: every block is still resident when the last one starts);
// the host counts blocks per CU.  Question behind it: a launch of 432 blocks that could all be co-resident (4-8 per
// CU by registers / LDS) - is it dealt one or two per CU over all 256 CUs, or packed onto a part of the chip?
//   hipcc --offload-arch=gfx950 -O2 placement.hip -o placement && ./placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>

__global__ __launch_bounds__(256) void k(unsigned* where, unsigned long long* when, long long spin) {
  extern __shared__ float smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID[3:0]
    where[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    when[blockIdx.x] = t0;
  }
  smem[threadIdx.x] = 1.f;
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 0) when[65536 + blockIdx.x] = __builtin_amdgcn_s_memtime();
}

int main() {
  unsigned* d_where;
  unsigned long long* d_when;
  hipMalloc(&d_where, 65536 * 4);
  hipMalloc(&d_when, 2 * 65536 * 8);
  static unsigned h[65536];
  static unsigned long long hw[2 * 65536];
  const int grids[] = {96, 288, 432, 768, 1056, 1152};
  const int ldss[] = {1024, 40 * 1024, 80 * 1024};
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int lds : ldss)
    for (int grid : grids) {
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_where, d_when, 30000ll);   // s_memtime counts shader clocks here
      hipDeviceSynchronize();
      hipMemcpy(h, d_where, grid * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hw, d_when, 2 * 65536 * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per_cu;
      std::map<unsigned, int> per_xcc;
      for (int b = 0; b < grid; ++b) {
        per_cu[h[b] & 0xfffff00u]++;       // xcc, se, sh, cu
        per_xcc[h[b] >> 16]++;
      }
      int hist[16] = {0};
      for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
      printf("lds %3d KiB grid %4d: %3zu CUs used, blocks per CU histogram:", lds / 1024, grid, per_cu.size());
      for (int i = 1; i < 16; ++i)
        if (hist[i]) printf(" %dx%d", hist[i], i);
      printf("; per XCC:");
      for (auto& kv : per_xcc) printf(" %d", kv.second);
      printf("\n");
    }
  return 0;
}
