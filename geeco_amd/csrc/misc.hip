// Small HBM-bound helpers: kernel transposes/padding, fused TF-style Adam, reductions.
#include "geeco_common.h"

// ---- [G][9][A][B] -> [G][9][B][A] ----------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_taps_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                             long long gs_w, long long gs_wt, int A, int B) {
  __shared__ float tile[32][33];
  const int g = blockIdx.z / 9, tap = blockIdx.z % 9;
  const float* src = w + (long long)g * gs_w + (long long)tap * A * B;
  float* dst = wt + (long long)g * gs_wt + (long long)tap * A * B;
  const int tilesB = (B + 31) / 32;
  const int ta = blockIdx.x / tilesB, tb = blockIdx.x % tilesB;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int a = ta * 32 + ty + i, b = tb * 32 + tx;
    if (a < A && b < B) tile[ty + i][tx] = src[(long long)a * B + b];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int b = tb * 32 + ty + i, a = ta * 32 + tx;
    if (a < A && b < B) dst[(long long)b * A + a] = tile[tx][ty + i];
  }
}

extern "C" int geeco_transpose_hwio(const float* w, float* wt, int groups, int64_t gs_w, int64_t gs_wt, int Cin,
                                    int Cout, void* stream) {
  GEECO_CHECK_ARG(w && wt && groups >= 1 && Cin >= 1 && Cout >= 1, "transpose_hwio: bad arguments");
  dim3 grid((unsigned)(cdiv(Cin, 32) * cdiv(Cout, 32)), 1, (unsigned)(groups * 9));
  hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, (long long)gs_w,
                     (long long)gs_wt, Cin, Cout);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- [A][B][C] -> [A][Bd][C] -----------------------------------------------------------------------
__global__ void pad_mid_kernel(const float* __restrict__ src, float* __restrict__ dst, long long A, int B, int Bd,
                               int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = A * Bd * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long ab = i / C;
  const int b = (int)(ab % Bd);
  const long long a = ab / Bd;
  dst[i] = b < B ? src[(a * B + b) * C + c] : 0.f;
}

extern "C" int geeco_pad_mid(const float* src, float* dst, int64_t A, int B, int Bd, int C, void* stream) {
  GEECO_CHECK_ARG(src && dst && A >= 1 && B >= 1 && Bd >= 1 && C >= 1, "pad_mid: bad arguments");
  const long long total = (long long)A * Bd * C;
  hipLaunchKernelGGL(pad_mid_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (long long)A, B, Bd, C);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- every derived weight copy of an encoder stack in ONE launch -----------------------------------
// The per-step derive chain (conv1 kernel padded to 4 input channels + the per-tap transposes of
// conv2..8 for the dgrad GEMMs) is ~10 launches of a few microseconds each on the critical path of
// the Adam graph; here a block looks up its job in a small table passed by value.
#define GEECO_MAX_DERIVE 8
struct DeriveParams {
  const float* w[GEECO_MAX_DERIVE];
  float* wt[GEECO_MAX_DERIVE];
  long long gs_wt[GEECO_MAX_DERIVE];
  int A[GEECO_MAX_DERIVE], B[GEECO_MAX_DERIVE];      // [9][A][B] -> [9][B][A]
  int block0[GEECO_MAX_DERIVE + 2];                    // first block of job l; [n] = first pad block; [n+1] = grid size
  int n, groups;
  long long gs_w;
  const float* pad_src;      // [G][9][Bs][C] at stride gs_w -> pad_dst [G][9][Bd][C] at stride gs_pad
  float* pad_dst;
  long long gs_pad;
  int pad_Bs, pad_Bd, pad_C;
};

__global__ __launch_bounds__(256) void derive_weights_kernel(const DeriveParams p) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  if (b >= p.block0[p.n]) {      // pad job: one thread per destination element
    const long long per_g = 9ll * p.pad_Bd * p.pad_C;
    const long long i = (long long)(b - p.block0[p.n]) * 256 + threadIdx.x;
    if (i >= per_g * p.groups) return;
    const int g = (int)(i / per_g);
    const long long e = i - g * per_g;
    const int c = (int)(e % p.pad_C);
    const int bb = (int)((e / p.pad_C) % p.pad_Bd);
    const int a = (int)(e / ((long long)p.pad_C * p.pad_Bd));
    p.pad_dst[g * p.gs_pad + e] = bb < p.pad_Bs ? p.pad_src[g * p.gs_w + ((long long)a * p.pad_Bs + bb) * p.pad_C + c] : 0.f;
    return;
  }
  int l = 0;
#pragma unroll
  for (int k = 1; k < GEECO_MAX_DERIVE; ++k)
    if (k < p.n && b >= p.block0[k]) l = k;
  const int A = p.A[l], B = p.B[l];
  const int tilesB = (B + 31) / 32, tpt = ((A + 31) / 32) * tilesB;
  const int local = b - p.block0[l];
  const int gt = local / tpt, t = local - gt * tpt;       // gt = g * 9 + tap
  const int g = gt / 9, tap = gt - g * 9;
  const float* src = p.w[l] + (long long)g * p.gs_w + (long long)tap * A * B;
  float* dst = p.wt[l] + (long long)g * p.gs_wt[l] + (long long)tap * A * B;
  const int ta = t / tilesB, tb = t - ta * tilesB;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int a = ta * 32 + ty + i, bb = tb * 32 + tx;
    if (a < A && bb < B) tile[ty + i][tx] = src[(long long)a * B + bb];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    int bb = tb * 32 + ty + i, a = ta * 32 + tx;
    if (a < A && bb < B) dst[(long long)bb * A + a] = tile[tx][ty + i];
  }
}

extern "C" int geeco_derive_conv_weights(int nlayers, const float* const* w, float* const* wt, const int* Cin,
                                         const int* Cout, const int64_t* gs_wt, int groups, int64_t gs_w,
                                         const float* pad_src, float* pad_dst, int pad_Cin, int pad_Cin_padded,
                                         int pad_Cout, int64_t gs_pad, void* stream) {
  GEECO_CHECK_ARG(nlayers >= 0 && nlayers <= GEECO_MAX_DERIVE && groups >= 1, "derive_conv_weights: bad arguments");
  GEECO_CHECK_ARG(nlayers == 0 || (w && wt && Cin && Cout && gs_wt), "derive_conv_weights: null table");
  DeriveParams p = {};
  p.n = nlayers; p.groups = groups; p.gs_w = gs_w;
  int blocks = 0;
  for (int l = 0; l < nlayers; ++l) {
    GEECO_CHECK_ARG(w[l] && wt[l] && Cin[l] >= 1 && Cout[l] >= 1, "derive_conv_weights: layer %d", l);
    p.w[l] = w[l]; p.wt[l] = wt[l]; p.A[l] = Cin[l]; p.B[l] = Cout[l]; p.gs_wt[l] = gs_wt[l];
    p.block0[l] = blocks;
    blocks += groups * 9 * cdiv(Cin[l], 32) * cdiv(Cout[l], 32);
  }
  p.block0[nlayers] = blocks;
  if (pad_src && pad_dst) {
    GEECO_CHECK_ARG(pad_Cin >= 1 && pad_Cin_padded >= pad_Cin && pad_Cout >= 1, "derive_conv_weights: bad pad dims");
    p.pad_src = pad_src; p.pad_dst = pad_dst; p.gs_pad = gs_pad;
    p.pad_Bs = pad_Cin; p.pad_Bd = pad_Cin_padded; p.pad_C = pad_Cout;
    blocks += (int)cdiv64(9ll * pad_Cin_padded * pad_Cout * groups, 256);
  }
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(derive_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- Adam (tf.train.AdamOptimizer semantics) -------------------------------------------------------
// scal[0] = lr_t for this step; the step counter lives in device memory so that a captured
// hipGraph replays with the right bias correction.
__global__ void adam_prepare_kernel(long long* step, float lr, float b1, float b2, float* scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long t = *step + 1;
    *step = t;
    double b1t = pow((double)b1, (double)t), b2t = pow((double)b2, (double)t);
    scal[0] = (float)((double)lr * sqrt(1.0 - b2t) / (1.0 - b1t));
  }
}

extern "C" int geeco_adam_prepare(int64_t* global_step_dev, float lr, float beta1, float beta2, float* scal_dev,
                                  void* stream) {
  GEECO_CHECK_ARG(global_step_dev && scal_dev, "adam_prepare: null pointer");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)global_step_dev, lr,
                     beta1, beta2, scal_dev);
  GEECO_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   const float* __restrict__ scal, float b1, float b2, float eps,
                                                   float gscale, float l2) {
  const float lr_t = scal[0];
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
    float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = ge[k] * gscale + l2 * pe[k];
      me[k] = b1 * me[k] + (1.f - b1) * gg;
      ve[k] = b2 * ve[k] + (1.f - b2) * gg * gg;
      pe[k] = pe[k] - lr_t * me[k] / (sqrtf(ve[k]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = f32x4{pe[0], pe[1], pe[2], pe[3]};
    reinterpret_cast<f32x4*>(m)[i] = f32x4{me[0], me[1], me[2], me[3]};
    reinterpret_cast<f32x4*>(v)[i] = f32x4{ve[0], ve[1], ve[2], ve[3]};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float gg = g[i] * gscale + l2 * p[i];
    float mm = b1 * m[i] + (1.f - b1) * gg;
    float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - lr_t * mm / (sqrtf(vv) + eps);
  }
}

extern "C" int geeco_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_t_dev,
                             float beta1, float beta2, float eps, float grad_scale, float l2, void* stream) {
  GEECO_CHECK_ARG(p && g && m && v && lr_t_dev && n >= 1, "adam_tf: bad arguments");
  GEECO_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                  "adam_tf: arenas must be 16-byte aligned");
  long long blocks = cdiv64(n >> 2, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     lr_t_dev, beta1, beta2, eps, grad_scale, l2);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// The same update over up to GEECO_ADAM_SEGMENTS_MAX pieces of the arena, each with its own gradient source (data parallel: the
// 99 % of the variables whose gradients arrived with the early bucket are updated while the late bucket is still on the wire;
// conv1 / conv2 of the encoders follow, their gradients read straight from the late bucket's staging buffer).  Element by element the
// arithmetic of adam_kernel: any partition of the arena gives bitwise the one-pass result.
struct AdamSegs {
  const float* g[GEECO_ADAM_SEGMENTS_MAX];
  long long p4[GEECO_ADAM_SEGMENTS_MAX];        // first float4 of the piece in the arena
  long long n4[GEECO_ADAM_SEGMENTS_MAX];        // its float4 count
  int n;
};

__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ p, float* __restrict__ g_out, float* __restrict__ m,
                                                            float* __restrict__ v, const AdamSegs s, const float* __restrict__ scal,
                                                            float b1, float b2, float eps, float gscale, float l2) {
  const float lr_t = scal[0];
  // Every block walks piece after piece with the whole grid's stride, as adam_kernel walks the arena (pieces dealt to block groups side by
  // side measured the same: 36.4-36.8 us for the early pieces of the bench model against 33.8 for the undivided pass in the same
  // place of the step).  The piece loop is unrolled: static indices into the by-value argument (a dynamic index would copy it to scratch).
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < GEECO_ADAM_SEGMENTS_MAX; ++k) {
    if (k >= s.n) break;
    const long long e0 = s.p4[k], cnt = s.n4[k];
    const float* __restrict__ gs = s.g[k];
    for (long long i = i0; i < cnt; i += stride) {
      const long long e = e0 + i;
      f32x4 pv = reinterpret_cast<f32x4*>(p)[e];
      f32x4 gv = reinterpret_cast<const f32x4*>(gs)[i];
      f32x4 mv = reinterpret_cast<f32x4*>(m)[e];
      f32x4 vv = reinterpret_cast<f32x4*>(v)[e];
      if (g_out) reinterpret_cast<f32x4*>(g_out)[e] = gv;
      float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
      float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float gg = ge[c] * gscale + l2 * pe[c];
        me[c] = b1 * me[c] + (1.f - b1) * gg;
        ve[c] = b2 * ve[c] + (1.f - b2) * gg * gg;
        pe[c] = pe[c] - lr_t * me[c] / (sqrtf(ve[c]) + eps);
      }
      reinterpret_cast<f32x4*>(p)[e] = f32x4{pe[0], pe[1], pe[2], pe[3]};
      reinterpret_cast<f32x4*>(m)[e] = f32x4{me[0], me[1], me[2], me[3]};
      reinterpret_cast<f32x4*>(v)[e] = f32x4{ve[0], ve[1], ve[2], ve[3]};
    }
  }
}

extern "C" int geeco_adam_tf_segments(float* p, float* g_out, float* m, float* v, const geeco_adam_segment* segs, int nseg,
                                      const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, float l2,
                                      void* stream) {
  GEECO_CHECK_ARG(p && m && v && lr_t_dev && segs && nseg >= 1 && nseg <= GEECO_ADAM_SEGMENTS_MAX, "adam_tf_segments: bad arguments (1..%d pieces)",
                  GEECO_ADAM_SEGMENTS_MAX);
  GEECO_CHECK_ARG((((uintptr_t)p | (uintptr_t)g_out | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_tf_segments: arenas must be 16-byte aligned");
  AdamSegs s = {};
  s.n = nseg;
  long long total4 = 0;
  for (int k = 0; k < nseg; ++k) {
    GEECO_CHECK_ARG(segs[k].g && ((uintptr_t)segs[k].g & 15) == 0 && segs[k].p_off >= 0 && segs[k].p_off % 4 == 0 && segs[k].count >= 4 &&
                        segs[k].count % 4 == 0,
                    "adam_tf_segments: piece %d: offset / count must be multiples of 4 floats, the gradient source 16-byte aligned", k);
    s.g[k] = segs[k].g;
    s.p4[k] = segs[k].p_off / 4;
    s.n4[k] = segs[k].count / 4;
    total4 += s.n4[k];
  }
  long long grid = 1;
  for (int k = 0; k < nseg; ++k) grid = cdiv64(s.n4[k], 256) > grid ? cdiv64(s.n4[k], 256) : grid;
  if (grid > 2048) grid = 2048;       // as geeco_adam_tf
  const int b = (int)grid;
  hipLaunchKernelGGL(adam_segments_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, p, g_out, m, v, s, lr_t_dev, beta1, beta2,
                     eps, grad_scale, l2);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- sum of squares (L2 regularisation loss, graph.py:13-15) ---------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ p, long long n, float* out) {
  __shared__ float sw[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += p[i] * p[i];
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sw[0] + sw[1] + sw[2] + sw[3]);
}

extern "C" int geeco_sumsq(const float* p, int64_t n, float* out, void* stream) {
  GEECO_CHECK_ARG(p && out && n >= 1, "sumsq: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  long long blocks = cdiv64(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, (long long)n, out);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- column sums: out[j] = sum_i a[i][j] -----------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, long long lda, int M, int N,
                                                     float* __restrict__ out, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= M; i += 8) {          // 8 loads in flight; the sum keeps the row order
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = a[(long long)(i + u) * lda + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; i < M; ++i) s += a[(long long)i * lda + j];
  out[j] = accumulate ? out[j] + s : s;
}

extern "C" int geeco_colsum(const float* a, int64_t lda, int M, int N, float* out, int accumulate, void* stream) {
  GEECO_CHECK_ARG(a && out && M >= 1 && N >= 1, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     (long long)lda, M, N, out, accumulate);
  GEECO_LAUNCH_CHECK();
  return 0;
}
