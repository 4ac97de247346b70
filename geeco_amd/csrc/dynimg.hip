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
  // DIFF (geeco_goal_dynimgs_fwd): the pair image alpha2[0] * last frame + alpha2[1] * target (graph.py:397-400) from the same pass:
  // the last frame is in registers anyway, so the pair image costs one read of the target instead of a launch that reads both
  const float* tgt;       // [N][HW][3]
  const float* tgt_depth; // [N][HW] (DEPTH)
  float* diff_out;        // [N][HW][4]
  float* part2;           // [N][nblk][2]
  float alpha2[2];
  // U8 (geeco_goal_dynimgs_u8_fwd): the RGB frames are the recorder's uint8 values, still in the episode's resident frames:
  // win[n] = address of the first frame of window n ([K][HW][3] bytes, consecutive frames), tgt_u8[n] = its target frame
  const unsigned char* const* win;
  const unsigned char* const* tgt_u8;
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

__device__ __forceinline__ void block_minmax_store2(float mn, float mx, float* dst, float mn2, float mx2, float* dst2) {
  __shared__ float s4[4][4];
  mn = wave_reduce_min(mn);
  mx = wave_reduce_max(mx);
  mn2 = wave_reduce_min(mn2);
  mx2 = wave_reduce_max(mx2);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s4[wid][0] = mn; s4[wid][1] = mx; s4[wid][2] = mn2; s4[wid][3] = mx2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dst[0] = fminf(fminf(s4[0][0], s4[1][0]), fminf(s4[2][0], s4[3][0]));
    dst[1] = fmaxf(fmaxf(s4[0][1], s4[1][1]), fmaxf(s4[2][1], s4[3][1]));
    dst2[0] = fminf(fminf(s4[0][2], s4[1][2]), fminf(s4[2][2], s4[3][2]));
    dst2[1] = fmaxf(fmaxf(s4[0][3], s4[1][3]), fmaxf(s4[2][3], s4[3][3]));
  }
}

// float(u8) / 255.0f (_parse_v4, geeco_gym.py:312) without the division sequence: one Newton correction of a * (1/255) is the
// correctly rounded quotient for every a in 0..255 (tests/test_kernels_gpu.py::test_u8_unit_conversion_is_the_division checks all
// 256 values bitwise against the division of geeco_gather_windows).
__device__ __forceinline__ float u8_unit(float a) {
  const float r = 1.0f / 255.0f;
  const float q = a * r;
  const float e = __builtin_fmaf(-255.0f, q, a);
  return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ f32x4 u8x4_unit(unsigned int b) {
  return f32x4{u8_unit((float)(b & 255u)), u8_unit((float)((b >> 8) & 255u)), u8_unit((float)((b >> 16) & 255u)),
               u8_unit((float)(b >> 24))};
}

// 4 pixels = 12 bytes = three dwords of a uint8 RGB frame -> the three float4 the fp32 path loads
__device__ __forceinline__ void load_u8_unit(const unsigned char* frame, long long u, f32x4& v0, f32x4& v1, f32x4& v2) {
  const unsigned int* s = reinterpret_cast<const unsigned int*>(frame) + u * 3;
  const unsigned int b0 = s[0], b1 = s[1], b2 = s[2];
  v0 = u8x4_unit(b0);
  v1 = u8x4_unit(b1);
  v2 = u8x4_unit(b2);
}

// C == 3, Cpad == 4, HW % 4 == 0: one thread = 4 pixels = 3 float4 in, 4 float4 out.  DEPTH: a 4th channel comes from its
// own tensor (one more float4 = the depth of the 4 pixels per frame): rgb || depth (estimator.py:169,172) is formed in
// registers instead of packing all N * K frames to 4 channels first (1.07 GB read + 1.43 GB written per step at K = 32).
// U8: the RGB source is uint8 frames behind a per-sample address table (the window is never materialised as fp32: a quarter of
// the bytes, and no gather launch in front); everything downstream of the load is the fp32 path, so the images are bitwise equal.
template <bool DEPTH, bool DIFF = false, bool U8 = false>
__global__ __launch_bounds__(256) void dynimg_wsum3_kernel(const DynParams p) {
  const int n = blockIdx.y;
  [[maybe_unused]] const unsigned char* wbase = U8 ? p.win[n] : nullptr;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;   // 4-pixel unit
  const long long U = p.HW >> 2;
  float mn = INFINITY, mx = -INFINITY;
  [[maybe_unused]] float mn2 = INFINITY, mx2 = -INFINITY;
  if (u < U) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    [[maybe_unused]] f32x4 d0 = a0, d1 = a0, d2 = a0, d3 = a0;
#pragma unroll 4
    for (int t = 0; t < p.K; ++t) {
      const float w = p.alpha[t];
      f32x4 v0, v1, v2;
      if (U8) {
        load_u8_unit(wbase + (long long)t * p.HW * 3, u, v0, v1, v2);
      } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(dyn_frame_ptr(p, n, t)) + u * 3;
        v0 = src[0]; v1 = src[1]; v2 = src[2];
      }
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
      if (DIFF && t == p.K - 1) {       // the pair image, summed in the order of the two-frame pass: 0 + alpha2[0] * current, + alpha2[1] * target
        f32x4 t0, t1, t2;
        if (U8) {
          load_u8_unit(p.tgt_u8[n], u, t0, t1, t2);
        } else {
          const f32x4* ts = reinterpret_cast<const f32x4*>(p.tgt + (long long)n * p.HW * 3) + u * 3;
          t0 = ts[0]; t1 = ts[1]; t2 = ts[2];
        }
        const float w0 = p.alpha2[0], w1 = p.alpha2[1];
        d0 += w0 * v0; d1 += w0 * v1; d2 += w0 * v2;
        d0 += w1 * t0; d1 += w1 * t1; d2 += w1 * t2;
        if (DEPTH) {
          const f32x4 t3 = reinterpret_cast<const f32x4*>(p.tgt_depth + (long long)n * p.HW)[u];
          d3 += w0 * v3;
          d3 += w1 * t3;
        }
      }
    }
    if (DIFF) {
      float e2[12] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w};
      float q4[4] = {d3.x, d3.y, d3.z, d3.w};
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        mn2 = fminf(mn2, e2[i]);
        mx2 = fmaxf(mx2, e2[i]);
      }
      if (DEPTH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          mn2 = fminf(mn2, q4[i]);
          mx2 = fmaxf(mx2, q4[i]);
        }
      }
      f32x4* dd = reinterpret_cast<f32x4*>(p.diff_out + ((long long)n * p.HW + u * 4) * 4);
      dd[0] = f32x4{e2[0], e2[1], e2[2], DEPTH ? q4[0] : 0.f};
      dd[1] = f32x4{e2[3], e2[4], e2[5], DEPTH ? q4[1] : 0.f};
      dd[2] = f32x4{e2[6], e2[7], e2[8], DEPTH ? q4[2] : 0.f};
      dd[3] = f32x4{e2[9], e2[10], e2[11], DEPTH ? q4[3] : 0.f};
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
  if (DIFF)
    block_minmax_store2(mn, mx, p.part + ((long long)n * p.nblk + blockIdx.x) * 2, mn2, mx2,
                        p.part2 + ((long long)n * p.nblk + blockIdx.x) * 2);
  else
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

// The goal model's three conv1 inputs (graph.py:386-401) in TWO launches (round 3: three; before: five): one pass over the window
// writes the buffer image, the current frame's padded copy AND the pair image of (current frame, target); one normalisation
// launch serves both images.
static int goal_dynimgs_launch(DynParams& p, const float* alpha_host, const float* alpha2_host, bool u8, float* cur_out,
                               float* buf_out, float* diff_out, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nblk = dyn_nblk(p.HW, 3);
  float* part1 = (float*)ws;
  float* part2 = part1 + (long long)p.N * nblk * 2;
  p.C = 3; p.Cpad = 4; p.out = buf_out; p.last = cur_out; p.part = part1; p.nblk = nblk;
  for (int t = 0; t < p.K; ++t) p.alpha[t] = alpha_host[t];
  // ONE pass for both images (round 4): the pair image needs the current frame, which this pass holds in registers at t = K - 1
  p.diff_out = diff_out; p.part2 = part2;
  p.alpha2[0] = alpha2_host[0]; p.alpha2[1] = alpha2_host[1];
  const dim3 grid((unsigned)nblk, (unsigned)p.N);
  const bool depth = p.depth != nullptr;
  if (u8) {
    if (depth)
      hipLaunchKernelGGL((dynimg_wsum3_kernel<true, true, true>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((dynimg_wsum3_kernel<false, true, true>), grid, dim3(256), 0, s, p);
  } else {
    if (depth)
      hipLaunchKernelGGL((dynimg_wsum3_kernel<true, true>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((dynimg_wsum3_kernel<false, true>), grid, dim3(256), 0, s, p);
  }
  GEECO_LAUNCH_CHECK();
  const int C = depth ? 4 : 3;
  dim3 g2((unsigned)cdiv64(p.HW * 4, 1024), (unsigned)p.N, 2);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, buf_out, (const float*)part1, nblk, p.HW, C, 4, diff_out,
                     (const float*)part2);
  GEECO_LAUNCH_CHECK();
  return 0;
}

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
  DynParams p = {};
  p.frames = rgb; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.depth = depth; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW;
  p.tgt = tgt_rgb; p.tgt_depth = tgt_depth;
  return goal_dynimgs_launch(p, alpha_host, alpha2_host, false, cur_out, buf_out, diff_out, ws, stream);
}

// The same input stage fed from the episodes' resident uint8 frames (the data path's "next" row: the window of
// _window_v3, geeco_gym.py:615-631, and the / 255 of _parse_v4, :312, happen inside the load): win_ptrs_dev / tgt_ptrs_dev are
// DEVICE arrays of N addresses (window n = K consecutive [HW][3] uint8 frames starting at win_ptrs_dev[n]; its target frame
// at tgt_ptrs_dev[n]), so a captured graph keeps replaying while the host repoints the tables between steps.  Depth (float32)
// stays a dense [N][K][HW] / [N][HW] tensor.  Outputs are bitwise those of geeco_gather_windows + geeco_goal_dynimgs_fwd.
extern "C" int geeco_goal_dynimgs_u8_fwd(const void* const* win_ptrs_dev, const void* const* tgt_ptrs_dev, const float* depth,
                                         int64_t dsample_stride, int64_t dframe_stride, const float* tgt_depth,
                                         const float* alpha_host, const float* alpha2_host, int N, int K, int64_t HW,
                                         float* cur_out, float* buf_out, float* diff_out, void* ws, void* stream) {
  GEECO_CHECK_ARG(win_ptrs_dev && tgt_ptrs_dev && alpha_host && alpha2_host && cur_out && buf_out && diff_out && ws,
                  "goal_dynimgs_u8_fwd: null pointer");
  GEECO_CHECK_ARG((!depth) == (!tgt_depth), "goal_dynimgs_u8_fwd: depth and tgt_depth come together");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "goal_dynimgs_u8_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "goal_dynimgs_u8_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG(dsample_stride % 4 == 0 && dframe_stride % 4 == 0, "goal_dynimgs_u8_fwd: 16-byte aligned depth frames");
  DynParams p = {};
  p.win = reinterpret_cast<const unsigned char* const*>(win_ptrs_dev);
  p.tgt_u8 = reinterpret_cast<const unsigned char* const*>(tgt_ptrs_dev);
  p.depth = depth; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW;
  p.tgt_depth = tgt_depth;
  return goal_dynimgs_launch(p, alpha_host, alpha2_host, true, cur_out, buf_out, diff_out, ws, stream);
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
