// Dynamic image (temporal rank pooling) + per-sample min/max normalisation, HBM-bound streaming.
//
// Replaces reference src/models/e2evmc/graph.py:30-55 (dynimg) with coefficients from :17-28:
//   D[n] = sum_t alpha_t X[n][t];  out[n] = (D[n] - min D[n]) / (max D[n] - min D[n] + 1e-6)
// Pass 1 streams the K frames once (16 B per lane), writes D (channel-padded) and per-block
// min/max partials; pass 2 folds the partials (one wave reduce) and normalises in place.
// Algorithmic bytes: N*K*HW*C*4 read + N*HW*Cpad*4 written (+ one extra read/write of D).
#include "geeco_common.h"
#include <stdlib.h>

#define DYN_MAXK 64

struct DynParams {
  const float* frames;
  const float* frames2;
  const float* depth;      // optional 4th channel kept in its own tensor ([N][K][HW] / [N][HW]): RGB-D without packing
  const float* depth2;
  long long dsample_stride, dframe_stride;
  long long sample_stride, frame_stride;
  int N, K;
  long long HW;
  int C, Cpad;
  float* out;
  float* last;   // optional [N][HW][4]: the LAST frame of the stack channel-padded (conv1's input of the current frame)
  float* part;   // [N][nblk][2]
  int nblk;
  float alpha[DYN_MAXK];
};

__device__ __forceinline__ const float* dyn_frame_ptr(const DynParams& p, int n, int t) {
  if (p.frames2 && t == 1) return p.frames2 + (long long)n * p.HW * p.C;
  return p.frames + (long long)n * p.sample_stride + (long long)t * p.frame_stride;
}

__device__ __forceinline__ void block_minmax_store(float mn, float mx, float* dst) {
  __shared__ float smn[4], smx[4];
  mn = wave_reduce_min(mn);
  mx = wave_reduce_max(mx);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smn[wid] = mn;
    smx[wid] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dst[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    dst[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

// C == 3, Cpad == 4, HW % 4 == 0: one thread = 4 pixels = 3 float4 in, 4 float4 out.  DEPTH: a 4th channel comes from its
// own tensor (one more float4 = the depth of the 4 pixels per frame): rgb || depth (estimator.py:169,172) is formed in
// registers instead of packing all N * K frames to 4 channels first (1.07 GB read + 1.43 GB written per step at K = 32).
template <bool DEPTH>
__global__ __launch_bounds__(256) void dynimg_wsum3_kernel(const DynParams p) {
  const int n = blockIdx.y;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;   // 4-pixel unit
  const long long U = p.HW >> 2;
  float mn = INFINITY, mx = -INFINITY;
  if (u < U) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll 4
    for (int t = 0; t < p.K; ++t) {
      const f32x4* src = reinterpret_cast<const f32x4*>(dyn_frame_ptr(p, n, t)) + u * 3;
      const float w = p.alpha[t];
      f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
      a0 += w * v0;
      a1 += w * v1;
      a2 += w * v2;
      f32x4 v3 = {0.f, 0.f, 0.f, 0.f};
      if (DEPTH) {
        const float* dp = (p.depth2 && t == 1) ? p.depth2 + (long long)n * p.HW
                                               : p.depth + (long long)n * p.dsample_stride + (long long)t * p.dframe_stride;
        v3 = reinterpret_cast<const f32x4*>(dp)[u];
        a3 += w * v3;
      }
      if (p.last && t == p.K - 1) {     // the frame is in registers anyway: its channel-padded copy costs no extra read
        f32x4* lo = reinterpret_cast<f32x4*>(p.last + ((long long)n * p.HW + u * 4) * 4);
        lo[0] = f32x4{v0.x, v0.y, v0.z, v3.x};
        lo[1] = f32x4{v0.w, v1.x, v1.y, v3.y};
        lo[2] = f32x4{v1.z, v1.w, v2.x, v3.z};
        lo[3] = f32x4{v2.y, v2.z, v2.w, v3.w};
      }
    }
    float e[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    float d4[4] = {a3.x, a3.y, a3.z, a3.w};
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      mn = fminf(mn, e[i]);
      mx = fmaxf(mx, e[i]);
    }
    if (DEPTH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mn = fminf(mn, d4[i]);
        mx = fmaxf(mx, d4[i]);
      }
    }
    f32x4* dst = reinterpret_cast<f32x4*>(p.out + ((long long)n * p.HW + u * 4) * 4);
    dst[0] = f32x4{e[0], e[1], e[2], DEPTH ? d4[0] : 0.f};
    dst[1] = f32x4{e[3], e[4], e[5], DEPTH ? d4[1] : 0.f};
    dst[2] = f32x4{e[6], e[7], e[8], DEPTH ? d4[2] : 0.f};
    dst[3] = f32x4{e[9], e[10], e[11], DEPTH ? d4[3] : 0.f};
  }
  block_minmax_store(mn, mx, p.part + ((long long)n * p.nblk + blockIdx.x) * 2);
}

// Generic: one thread = one pixel, C <= Cpad <= 8 channels (C == 4: one float4 per frame).
__global__ __launch_bounds__(256) void dynimg_wsum_generic_kernel(const DynParams p) {
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  if (px < p.HW) {
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    if (p.C == 4) {
#pragma unroll 4
      for (int t = 0; t < p.K; ++t) {
        f32x4 v = reinterpret_cast<const f32x4*>(dyn_frame_ptr(p, n, t))[px];
        const float w = p.alpha[t];
        acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w;
      }
    } else {
      for (int t = 0; t < p.K; ++t) {
        const float* src = dyn_frame_ptr(p, n, t) + px * p.C;
        const float w = p.alpha[t];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < p.C) acc[c] += w * src[c];
      }
    }
    float* dst = p.out + ((long long)n * p.HW + px) * p.Cpad;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < p.C) {
        mn = fminf(mn, acc[c]);
        mx = fmaxf(mx, acc[c]);
      }
      if (c < p.Cpad) dst[c] = c < p.C ? acc[c] : 0.f;
    }
  }
  block_minmax_store(mn, mx, p.part + ((long long)n * p.nblk + blockIdx.x) * 2);
}

// Pass 2: fold partials, normalise in place.  One thread = 4 consecutive floats of out.
__global__ __launch_bounds__(256) void dynimg_norm_kernel(float* out, const float* part, int nblk, long long HW,
                                                          int C, int Cpad, float* out2, const float* part2) {
  const int n = blockIdx.y;
  if (blockIdx.z == 1) {      // second image of a pair (geeco_goal_dynimgs_fwd): same shape, its own partials
    out = out2;
    part = part2;
  }
  __shared__ float s_mn, s_rng;
  {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) {
      mn = fminf(mn, part[((long long)n * nblk + i) * 2]);
      mx = fmaxf(mx, part[((long long)n * nblk + i) * 2 + 1]);
    }
    __shared__ float smn[4], smx[4];
    mn = wave_reduce_min(mn);
    mx = wave_reduce_max(mx);
    if ((threadIdx.x & 63) == 0) {
      smn[threadIdx.x >> 6] = mn;
      smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
      float b = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
      s_mn = a;
      s_rng = b - a + 1e-6f;   // graph.py:49
    }
    __syncthreads();
  }
  const float mn = s_mn, rng = s_rng;
  const long long total = HW * Cpad;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  float* o = out + (long long)n * total + i4;
  if ((total & 3) == 0) {
    f32x4 v = *reinterpret_cast<f32x4*>(o);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = (int)((i4 + k) % Cpad);
      e[k] = c < C ? (e[k] - mn) / rng : 0.f;
    }
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int k = 0; k < 4 && i4 + k < total; ++k) {
      int c = (int)((i4 + k) % Cpad);
      o[k] = c < C ? (o[k] - mn) / rng : 0.f;
    }
  }
}

extern "C" void geeco_dynimg_alpha(int K, float* alpha) {
  // graph.py:17-28: float32 harmonic numbers summed in ascending order
  auto H = [](int t) {
    float h = 0.f;
    for (int i = 1; i <= t; ++i) h = h + 1.0f / (float)i;
    return h;
  };
  const float HT = H(K);
  for (int t = 1; t <= K; ++t) alpha[t - 1] = (float)(2 * (K - t + 1)) - (float)(K + 1) * (HT - H(t - 1));
}

static int dyn_nblk(long long HW, int C) {
  if (C == 3 && (HW & 3) == 0) return (int)cdiv64(HW >> 2, 256);
  return (int)cdiv64(HW, 256);
}

extern "C" int64_t geeco_dynimg_ws_bytes(int N, int64_t hwc) {
  // upper bound over both kernels: one partial pair per 256 pixels
  return (int64_t)N * (cdiv64(hwc, 256) + 1) * 2 * 4;
}

static int dynimg_fwd_impl(const float* frames, const float* frames2, int64_t sample_stride, int64_t frame_stride,
                           const float* alpha_host, int N, int K, int64_t HW, int C, int Cpad, float* out, float* last,
                           void* ws, void* stream);

extern "C" int geeco_dynimg_fwd(const float* frames, const float* frames2, int64_t sample_stride,
                                int64_t frame_stride, const float* alpha_host, int N, int K, int64_t HW, int C,
                                int Cpad, float* out, void* ws, void* stream) {
  return dynimg_fwd_impl(frames, frames2, sample_stride, frame_stride, alpha_host, N, K, HW, C, Cpad, out, nullptr, ws,
                         stream);
}

extern "C" int geeco_dynimg_fwd_last(const float* frames, int64_t sample_stride, int64_t frame_stride,
                                     const float* alpha_host, int N, int K, int64_t HW, float* out, float* last, void* ws,
                                     void* stream) {
  GEECO_CHECK_ARG(last, "dynimg_fwd_last: null pointer");
  GEECO_CHECK_ARG((HW & 3) == 0 && sample_stride % 4 == 0 && frame_stride % 4 == 0,
                  "dynimg_fwd_last: RGB frames, HW %% 4 == 0, 16-byte aligned strides");
  return dynimg_fwd_impl(frames, nullptr, sample_stride, frame_stride, alpha_host, N, K, HW, 3, 4, out, last, ws, stream);
}

static int dynimg_fwd_impl(const float* frames, const float* frames2, int64_t sample_stride, int64_t frame_stride,
                           const float* alpha_host, int N, int K, int64_t HW, int C, int Cpad, float* out, float* last,
                           void* ws, void* stream) {
  GEECO_CHECK_ARG(frames && alpha_host && out && ws, "dynimg_fwd: null pointer");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "dynimg_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 1 && C >= 1 && C <= Cpad && Cpad <= 8, "dynimg_fwd: bad dims");
  GEECO_CHECK_ARG(!frames2 || K == 2, "dynimg_fwd: frames2 only with K == 2");
  DynParams p = {};
  p.frames = frames; p.frames2 = frames2; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.N = N; p.K = K; p.HW = HW; p.C = C; p.Cpad = Cpad; p.out = out; p.last = last; p.part = (float*)ws;
  p.nblk = dyn_nblk(HW, C);
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)p.nblk, (unsigned)N);
  const bool aligned = (sample_stride % 4 == 0) && (frame_stride % 4 == 0);
  if (C == 3 && Cpad == 4 && (HW & 3) == 0 && aligned)
    hipLaunchKernelGGL(dynimg_wsum3_kernel<false>, grid, dim3(256), 0, s, p);
  else {
    GEECO_CHECK_ARG(C != 4 || aligned, "dynimg_fwd: C == 4 needs 16-byte aligned frames");
    p.nblk = (int)cdiv64(HW, 256);
    grid.x = p.nblk;
    hipLaunchKernelGGL(dynimg_wsum_generic_kernel, grid, dim3(256), 0, s, p);
  }
  GEECO_LAUNCH_CHECK();
  dim3 g2((unsigned)cdiv64(HW * Cpad, 1024), (unsigned)N);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, out, (const float*)ws, p.nblk, (long long)HW, C, Cpad, (float*)nullptr,
                     (const float*)nullptr);
  GEECO_LAUNCH_CHECK();
  return 0;
}

static int dynimg_rgbd_impl(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                            const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                            const float* alpha_host, int N, int K, int64_t HW, float* out, float* last, void* ws,
                            void* stream);

extern "C" int geeco_dynimg_rgbd_fwd(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                                     const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                                     const float* alpha_host, int N, int K, int64_t HW, float* out, void* ws, void* stream) {
  return dynimg_rgbd_impl(rgb, rgb2, sample_stride, frame_stride, depth, depth2, dsample_stride, dframe_stride, alpha_host, N,
                          K, HW, out, nullptr, ws, stream);
}

extern "C" int geeco_dynimg_rgbd_fwd_last(const float* rgb, int64_t sample_stride, int64_t frame_stride, const float* depth,
                                          int64_t dsample_stride, int64_t dframe_stride, const float* alpha_host, int N,
                                          int K, int64_t HW, float* out, float* last, void* ws, void* stream) {
  GEECO_CHECK_ARG(last, "dynimg_rgbd_fwd_last: null pointer");
  return dynimg_rgbd_impl(rgb, nullptr, sample_stride, frame_stride, depth, nullptr, dsample_stride, dframe_stride, alpha_host,
                          N, K, HW, out, last, ws, stream);
}

static int dynimg_rgbd_impl(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                            const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                            const float* alpha_host, int N, int K, int64_t HW, float* out, float* last, void* ws,
                            void* stream) {
  GEECO_CHECK_ARG(rgb && depth && alpha_host && out && ws, "dynimg_rgbd_fwd: null pointer");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "dynimg_rgbd_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "dynimg_rgbd_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG((!rgb2) == (!depth2) && (!rgb2 || K == 2), "dynimg_rgbd_fwd: rgb2 / depth2 come together, with K == 2");
  GEECO_CHECK_ARG(sample_stride % 4 == 0 && frame_stride % 4 == 0 && dsample_stride % 4 == 0 && dframe_stride % 4 == 0,
                  "dynimg_rgbd_fwd: 16-byte aligned frames");
  DynParams p = {};
  p.frames = rgb; p.frames2 = rgb2; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.depth = depth; p.depth2 = depth2; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW; p.C = 3; p.Cpad = 4; p.out = out; p.last = last; p.part = (float*)ws;
  p.nblk = dyn_nblk(HW, 3);
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dynimg_wsum3_kernel<true>, dim3((unsigned)p.nblk, (unsigned)N), dim3(256), 0, s, p);
  GEECO_LAUNCH_CHECK();
  dim3 g2((unsigned)cdiv64(HW * 4, 1024), (unsigned)N);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, out, (const float*)ws, p.nblk, (long long)HW, 4, 4, (float*)nullptr,
                     (const float*)nullptr);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// The goal model's three conv1 inputs (graph.py:386-401) in three launches instead of five: buffer image (+ the current
// frame's padded copy), diff image of (current frame, target), and ONE normalisation launch for both images.
extern "C" int geeco_goal_dynimgs_fwd(const float* rgb, int64_t sample_stride, int64_t frame_stride, const float* tgt_rgb,
                                      const float* depth, int64_t dsample_stride, int64_t dframe_stride,
                                      const float* tgt_depth, const float* alpha_host, const float* alpha2_host, int N, int K,
                                      int64_t HW, float* cur_out, float* buf_out, float* diff_out, void* ws, void* stream) {
  GEECO_CHECK_ARG(rgb && tgt_rgb && alpha_host && alpha2_host && cur_out && buf_out && diff_out && ws,
                  "goal_dynimgs_fwd: null pointer");
  GEECO_CHECK_ARG((!depth) == (!tgt_depth), "goal_dynimgs_fwd: depth and tgt_depth come together");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "goal_dynimgs_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "goal_dynimgs_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG(sample_stride % 4 == 0 && frame_stride % 4 == 0 && dsample_stride % 4 == 0 && dframe_stride % 4 == 0,
                  "goal_dynimgs_fwd: 16-byte aligned frames");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = dyn_nblk(HW, 3);
  float* part1 = (float*)ws;
  float* part2 = part1 + (long long)N * nblk * 2;
  DynParams p = {};
  p.frames = rgb; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.depth = depth; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW; p.C = 3; p.Cpad = 4; p.out = buf_out; p.last = cur_out; p.part = part1; p.nblk = nblk;
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  DynParams d = {};
  d.frames = rgb + (long long)(K - 1) * frame_stride; d.frames2 = tgt_rgb; d.sample_stride = sample_stride;
  if (depth) { d.depth = depth + (long long)(K - 1) * dframe_stride; d.depth2 = tgt_depth; d.dsample_stride = dsample_stride; }
  d.N = N; d.K = 2; d.HW = HW; d.C = 3; d.Cpad = 4; d.out = diff_out; d.part = part2; d.nblk = nblk;
  d.alpha[0] = alpha2_host[0]; d.alpha[1] = alpha2_host[1];
  const dim3 grid((unsigned)nblk, (unsigned)N);
  if (depth) {
    hipLaunchKernelGGL(dynimg_wsum3_kernel<true>, grid, dim3(256), 0, s, p);
    hipLaunchKernelGGL(dynimg_wsum3_kernel<true>, grid, dim3(256), 0, s, d);
  } else {
    hipLaunchKernelGGL(dynimg_wsum3_kernel<false>, grid, dim3(256), 0, s, p);
    hipLaunchKernelGGL(dynimg_wsum3_kernel<false>, grid, dim3(256), 0, s, d);
  }
  GEECO_LAUNCH_CHECK();
  const int C = depth ? 4 : 3;
  dim3 g2((unsigned)cdiv64(HW * 4, 1024), (unsigned)N, 2);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, buf_out, (const float*)part1, nblk, (long long)HW, C, 4, diff_out,
                     (const float*)part2);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- pixel packing: [n][HW][C1] (+ [n][HW][C2]) -> [n][HW][Cpad] ---------------------------------
__global__ __launch_bounds__(256) void pack_pixels_kernel(const float* src, long long s1, const float* src2,
                                                          long long s2, long long HW, int C1, int C2, int Cpad,
                                                          float* dst) {
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const float* a = src + (long long)n * s1 + px * C1;
  const float* b = src2 ? src2 + (long long)n * s2 + px * C2 : nullptr;
  float* o = dst + ((long long)n * HW + px) * Cpad;
  if (Cpad == 4) {
    float e[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) e[c] = c < C1 ? a[c] : (b && c - C1 < C2 ? b[c - C1] : 0.f);
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int c = 0; c < Cpad; ++c) o[c] = c < C1 ? a[c] : (b && c - C1 < C2 ? b[c - C1] : 0.f);
  }
}

extern "C" int geeco_pack_pixels(const float* src, int64_t src_sample_stride, const float* src2,
                                 int64_t src2_sample_stride, int N, int64_t HW, int C1, int C2, int Cpad,
                                 float* dst, void* stream) {
  GEECO_CHECK_ARG(src && dst, "pack_pixels: null pointer");
  GEECO_CHECK_ARG(N >= 1 && HW >= 1 && C1 >= 1 && C1 + (src2 ? C2 : 0) <= Cpad, "pack_pixels: bad dims");
  dim3 grid((unsigned)cdiv64(HW, 256), (unsigned)N);
  hipLaunchKernelGGL(pack_pixels_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (long long)src_sample_stride,
                     src2, (long long)src2_sample_stride, (long long)HW, C1, C2, Cpad, dst);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- on-device window builder ---------------------------------------------------------------------
// The reference materialises every K-frame window of an episode on the host (_window_v3,
// src/data/geeco_gym.py:615-631) and feeds 12.6 MB per sample over PCIe.  Here an episode's frames
// are uploaded ONCE (RGB as the uint8 values the recorder stored, data_recorder / tfrecord.py:73-74)
// and each batch's windows are gathered in HBM:  out[n][k][:] = conv(src[starts[n] + k][:]),
// conv(u8) = float(u8) / 255.0f  (the division of _parse_v4, geeco_gym.py:312, bit-exact).
template <typename T>
__global__ __launch_bounds__(256) void gather_windows_kernel(const T* __restrict__ src, const int* __restrict__ starts,
                                                             int K, long long frame_elems, float divisor,
                                                             float* __restrict__ out) {
  const int n = blockIdx.z, k = blockIdx.y;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= frame_elems) return;
  const T* s = src + (long long)(starts[n] + k) * frame_elems + i4;
  float* o = out + ((long long)n * K + k) * frame_elems + i4;
  if (i4 + 4 <= frame_elems) {
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = divisor != 1.f ? (float)s[j] / divisor : (float)s[j];
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int j = 0; i4 + j < frame_elems; ++j) o[j] = divisor != 1.f ? (float)s[j] / divisor : (float)s[j];
  }
}

extern "C" int geeco_gather_windows(const void* src, int src_is_u8, const int* starts_dev, int N, int K,
                                    int64_t frame_elems, float divisor, float* out, void* stream) {
  GEECO_CHECK_ARG(src && starts_dev && out, "gather_windows: null pointer");
  GEECO_CHECK_ARG(N >= 1 && K >= 1 && frame_elems >= 4 && frame_elems % 4 == 0, "gather_windows: bad dims");
  GEECO_CHECK_ARG(divisor != 0.f, "gather_windows: divisor == 0");
  dim3 grid((unsigned)cdiv64(frame_elems, 1024), (unsigned)K, (unsigned)N);
  if (src_is_u8)
    hipLaunchKernelGGL(gather_windows_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, starts_dev, K, (long long)frame_elems, divisor, out);
  else
    hipLaunchKernelGGL(gather_windows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src,
                       starts_dev, K, (long long)frame_elems, divisor, out);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- fused input stage of goal_e2evmc's dynimg branch (graph.py:386-402) -------------------------------------------
// One launch produces the three conv1 inputs of geeco-f from a batch of K-frame windows:
//   out_obs  = frames[:, K-1]                               (rgb_frame_list[-1], :387)  channel-padded to 4
//   out_dyn  = dynimg(frames)                                (:392)
//   out_diff = dynimg([frames[:, K-1], tgt])                 (:397-400)
// The separate kernels read the K frames once, D twice more (normalisation pass) and the current frame three times;
// here every input byte is read ONCE and every output byte written ONCE: a thread keeps its raw D values in registers
// across a per-sample rendezvous (the min / max of a sample spans all its blocks) and normalises them on the way out.
//   phase 1: stream the frames (16 B per lane), accumulate D_buf, keep cur, load tgt, D_diff = a0 cur + a1 tgt;
//            block min / max -> order-preserving unsigned encoding -> atomicMax on the sample's 4 sync words;
//   rendezvous: one lane per block releases + counts in, polls until all BPS blocks of ITS sample have arrived
//            (relaxed poll, one acquire), every block of a sample is resident by construction (grid <= 2 blocks per
//            CU, enforced on the host), the poll is bounded anyway;
//   phase 2: normalise from registers, write the three images; the last block to leave resets the sync words to 0,
//            so the workspace is zero again for the next launch / graph replay.
// The sync words must be ZERO before the first launch (ops.goal_inputs_ws zero-fills the workspace once).
struct GoalInParams {
  const float* frames;       // [N][K][HW][C] (strided)
  const float* tgt;          // [N][HW][C]
  long long sample_stride, frame_stride, tgt_stride;
  int N, K, BPS;
  long long items;           // per sample: 4-pixel units (C == 3) or pixels (C == 4)
  long long HW;
  float* out_obs;            // [N][HW][4]
  float* out_dyn;
  float* out_diff;
  unsigned* sync;            // [N][64] (one 256-byte line per sample): enc(max buf), enc(-min buf), enc(max diff), enc(-min diff), arrived, left, error
  float a2[2];
  float alpha[DYN_MAXK];
};

__device__ __forceinline__ unsigned f32_ordered(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);      // monotone; every float maps above 0
}
__device__ __forceinline__ float f32_unordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// NV = float4 per item and frame (3: four RGB pixels, 1: one RGBD pixel); IPT = items per thread; MINW = waves per
// SIMD the register budget must allow (4: up to 1024 resident blocks, 2: up to 512)
template <int NV, int IPT, int MINW>
__global__ __launch_bounds__(256, MINW) void goal_inputs_kernel(const GoalInParams p) {
  const int n = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const long long stride = (long long)p.BPS * 256;
  f32x4 acc[IPT][NV], dif[IPT][NV];
  float mnb = INFINITY, mxb = -INFINITY, mnd = INFINITY, mxd = -INFINITY;
  const float* fbase = p.frames + (long long)n * p.sample_stride;
  const float* tbase = p.tgt + (long long)n * p.tgt_stride;
  // items of this thread; the tail threads of a ragged sample re-read the last item (harmless) and skip min/max + stores
  long long it[IPT];
  bool ok[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const long long item = j * stride + (long long)b * 256 + tid;
    ok[j] = item < p.items;
    it[j] = ok[j] ? item : p.items - 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[j][v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // phase 1a: frames 0 .. K-2, all items of the thread per frame (IPT * NV independent 16-byte loads per frame)
#pragma unroll 2
  for (int t = 0; t < p.K - 1; ++t) {
    const f32x4* src = reinterpret_cast<const f32x4*>(fbase + (long long)t * p.frame_stride);
    const float w = p.alpha[t];
    f32x4 x[IPT][NV];
#pragma unroll
    for (int j = 0; j < IPT; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v) x[j][v] = src[it[j] * NV + v];
#pragma unroll
    for (int j = 0; j < IPT; ++j)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[j][v] += w * x[j][v];
  }
  // phase 1b: the current frame (copied out at once: it needs no normalisation) and the target frame
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(fbase + (long long)(p.K - 1) * p.frame_stride);
    const f32x4* tg = reinterpret_cast<const f32x4*>(tbase);
    const float w = p.alpha[p.K - 1];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      f32x4 cur[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        cur[v] = src[it[j] * NV + v];
        const f32x4 tv = tg[it[j] * NV + v];
        acc[j][v] += w * cur[v];
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        d += p.a2[0] * cur[v];
        d += p.a2[1] * tv;
        dif[j][v] = d;
        if (ok[j]) {
          mnb = fminf(mnb, fminf(fminf(acc[j][v].x, acc[j][v].y), fminf(acc[j][v].z, acc[j][v].w)));
          mxb = fmaxf(mxb, fmaxf(fmaxf(acc[j][v].x, acc[j][v].y), fmaxf(acc[j][v].z, acc[j][v].w)));
          mnd = fminf(mnd, fminf(fminf(d.x, d.y), fminf(d.z, d.w)));
          mxd = fmaxf(mxd, fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w)));
        }
      }
      if (ok[j]) {
        if (NV == 3) {
          const float* c = reinterpret_cast<const float*>(&cur[0]);
          float* o = p.out_obs + ((long long)n * p.HW + it[j] * 4) * 4;
#pragma unroll
          for (int px = 0; px < 4; ++px)
            *reinterpret_cast<f32x4*>(o + px * 4) = f32x4{c[px * 3], c[px * 3 + 1], c[px * 3 + 2], 0.f};
        } else {
          *reinterpret_cast<f32x4*>(p.out_obs + ((long long)n * p.HW + it[j]) * 4) = cur[0];
        }
      }
    }
  }
  // ---- block min / max -> the sample's sync words; rendezvous of the sample's BPS blocks -------------------------------
  __shared__ float red[4][4];
  __shared__ float res[4];
  mnb = wave_reduce_min(mnb); mxb = wave_reduce_max(mxb);
  mnd = wave_reduce_min(mnd); mxd = wave_reduce_max(mxd);
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = mxb; red[tid >> 6][1] = -mnb; red[tid >> 6][2] = mxd; red[tid >> 6][3] = -mnd;
  }
  __syncthreads();
  unsigned* sy = p.sync + (long long)n * 64;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m = fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]));
      if (m > -INFINITY) __hip_atomic_fetch_max(sy + k, f32_ordered(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_fetch_add(sy + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(sy + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.BPS) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > (1 << 22)) {          // never expected: every block of the grid is resident (host-side check)
        __hip_atomic_store(sy + 6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int k = 0; k < 4; ++k) res[k] = f32_unordered(__hip_atomic_load(sy + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  __syncthreads();
  const float bmn = -res[1], brng = res[0] - bmn + 1e-6f;     // graph.py:47-49
  const float dmn = -res[3], drng = res[2] - dmn + 1e-6f;
  // ---- phase 2: normalise from registers and write the two dynamic images ------------------------------------------------
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    if (!ok[j]) continue;
    if (NV == 3) {
      const float* a = reinterpret_cast<const float*>(&acc[j][0]);
      const float* d = reinterpret_cast<const float*>(&dif[j][0]);
      const long long o = ((long long)n * p.HW + it[j] * 4) * 4;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        *reinterpret_cast<f32x4*>(p.out_dyn + o + px * 4) =
            f32x4{(a[px * 3] - bmn) / brng, (a[px * 3 + 1] - bmn) / brng, (a[px * 3 + 2] - bmn) / brng, 0.f};
        *reinterpret_cast<f32x4*>(p.out_diff + o + px * 4) =
            f32x4{(d[px * 3] - dmn) / drng, (d[px * 3 + 1] - dmn) / drng, (d[px * 3 + 2] - dmn) / drng, 0.f};
      }
    } else {
      const long long o = ((long long)n * p.HW + it[j]) * 4;
      const f32x4 a = acc[j][0], d = dif[j][0];
      *reinterpret_cast<f32x4*>(p.out_dyn + o) = f32x4{(a.x - bmn) / brng, (a.y - bmn) / brng, (a.z - bmn) / brng, (a.w - bmn) / brng};
      *reinterpret_cast<f32x4*>(p.out_diff + o) = f32x4{(d.x - dmn) / drng, (d.y - dmn) / drng, (d.z - dmn) / drng, (d.w - dmn) / drng};
    }
  }
  // ---- leave: the last block of the sample zeroes its sync words (nobody reads them any more) --------------------------
  if (tid == 0) {
    const unsigned left = __hip_atomic_fetch_add(sy + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == (unsigned)p.BPS - 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) __hip_atomic_store(sy + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Every block of the grid must be resident at the rendezvous: launch bounds (256, 2) keep the kernel within 256
// registers, so 2 blocks of 256 threads fit every CU: at most 512 blocks.  (Measured at N=32, K=16, 256x256 RGB: 16 blocks
// per sample x 4 units per thread 161 us; 32 blocks x 2 units 203 us.)
struct GoalInPlan { int bps, ipt; };
static GoalInPlan goal_inputs_plan(int N, int64_t HW, int C) {
  GoalInPlan pl = {0, 0};
  if (N < 1 || N > 512 || !(C == 3 || C == 4) || (C == 3 && (HW & 3))) return pl;
  const long long items = C == 3 ? HW >> 2 : HW;
  int bps = 512 / N;
  if (bps > 16) bps = 16;
  if (bps >= 1 && cdiv64(items, (long long)bps * 256) <= (C == 3 ? 4 : 16)) {
    pl.bps = bps;
    pl.ipt = (int)cdiv64(items, (long long)bps * 256);
  }
  return pl;
}

extern "C" int64_t geeco_goal_inputs_ws_bytes(int N) { return (int64_t)N * 64 * 4; }

extern "C" int geeco_goal_inputs_supported(int N, int K, int64_t HW, int C) {
  if (K < 1 || K > DYN_MAXK) return 0;
  return goal_inputs_plan(N, HW, C).bps > 0;
}

template <int NV, int IPT, int MINW>
static void launch_goal_inputs(const GoalInParams& p, hipStream_t s) {
  geeco_note_kernel("goal_inputs_kernel<%d, %d, %d>", NV, IPT, MINW);
  hipLaunchKernelGGL((goal_inputs_kernel<NV, IPT, MINW>), dim3((unsigned)p.BPS, (unsigned)p.N), dim3(256), 0, s, p);
}

extern "C" int geeco_goal_inputs_fwd(const float* frames, int64_t sample_stride, int64_t frame_stride, const float* tgt,
                                     int64_t tgt_stride, const float* alpha_host, int N, int K, int64_t HW, int C,
                                     float* out_obs, float* out_dyn, float* out_diff, void* ws, void* stream) {
  GEECO_CHECK_ARG(frames && tgt && alpha_host && out_obs && out_dyn && out_diff && ws, "goal_inputs_fwd: null pointer");
  GEECO_CHECK_ARG(geeco_goal_inputs_supported(N, K, HW, C), "goal_inputs_fwd: unsupported shape N=%d K=%d HW=%lld C=%d", N, K,
                  (long long)HW, C);
  GEECO_CHECK_ARG(sample_stride % 4 == 0 && frame_stride % 4 == 0 && tgt_stride % 4 == 0, "goal_inputs_fwd: 16-byte aligned frames");
  GoalInParams p = {};
  p.frames = frames; p.tgt = tgt; p.sample_stride = sample_stride; p.frame_stride = frame_stride; p.tgt_stride = tgt_stride;
  const GoalInPlan pl = goal_inputs_plan(N, HW, C);
  p.N = N; p.K = K; p.HW = HW; p.BPS = pl.bps;
  p.items = C == 3 ? HW >> 2 : HW;
  p.out_obs = out_obs; p.out_dyn = out_dyn; p.out_diff = out_diff; p.sync = (unsigned*)ws;
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  geeco_dynimg_alpha(2, p.a2);
  hipStream_t s = (hipStream_t)stream;
  if (C == 3) {
    if (pl.ipt <= 1) launch_goal_inputs<3, 1, 2>(p, s);
    else if (pl.ipt <= 2) launch_goal_inputs<3, 2, 2>(p, s);
    else launch_goal_inputs<3, 4, 2>(p, s);
  } else {
    if (pl.ipt <= 4) launch_goal_inputs<1, 4, 2>(p, s);
    else if (pl.ipt <= 8) launch_goal_inputs<1, 8, 2>(p, s);
    else launch_goal_inputs<1, 16, 2>(p, s);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}
