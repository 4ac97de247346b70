// Probe (development only, not part of the product): Adam's arithmetic as a NARROW persistent grid -- `blocks` x 256 threads,
// grid-stride -- so that its waves can sit beside the two 209-VGPR waves per SIMD of the fused encoder-bottom backward and use the
// HBM bandwidth that MFMA-bound kernel leaves idle.  scripts/dev/adam_beside_bottom.py drives it.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o scripts/dev/ub/libnarrow_adam.so scripts/dev/ub/narrow_adam.hip
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void narrow_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long long n4, const float* __restrict__ scal,
                                                          float b1, float b2, float eps, float gscale) {
  const float lr_t = scal[0];
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    f32x4 pv[UNROLL], gv[UNROLL], mv[UNROLL], vv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long e = i + u * stride;
      pv[u] = reinterpret_cast<f32x4*>(p)[e];
      gv[u] = reinterpret_cast<const f32x4*>(g)[e];
      mv[u] = reinterpret_cast<f32x4*>(m)[e];
      vv[u] = reinterpret_cast<f32x4*>(v)[e];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long e = i + u * stride;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gg = gv[u][c] * gscale;
        mv[u][c] = b1 * mv[u][c] + (1.f - b1) * gg;
        vv[u][c] = b2 * vv[u][c] + (1.f - b2) * gg * gg;
        pv[u][c] = pv[u][c] - lr_t * mv[u][c] / (sqrtf(vv[u][c]) + eps);
      }
      reinterpret_cast<f32x4*>(p)[e] = pv[u];
      reinterpret_cast<f32x4*>(m)[e] = mv[u];
      reinterpret_cast<f32x4*>(v)[e] = vv[u];
    }
  }
  for (; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float gg = gv[c] * gscale;
      mv[c] = b1 * mv[c] + (1.f - b1) * gg;
      vv[c] = b2 * vv[c] + (1.f - b2) * gg * gg;
      pv[c] = pv[c] - lr_t * mv[c] / (sqrtf(vv[c]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}

extern "C" int narrow_adam(float* p, const float* g, float* m, float* v, long long n, const float* scal, int blocks, int unroll, void* stream) {
  const long long n4 = n / 4;
  if (unroll == 1) hipLaunchKernelGGL(narrow_adam_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, scal, 0.9f, 0.999f, 1e-8f, 1.f);
  else if (unroll == 2) hipLaunchKernelGGL(narrow_adam_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, scal, 0.9f, 0.999f, 1e-8f, 1.f);
  else hipLaunchKernelGGL(narrow_adam_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, scal, 0.9f, 0.999f, 1e-8f, 1.f);
  return (int)hipGetLastError();
}
