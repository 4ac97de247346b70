// 3x3 convolution forward / input-gradient as an implicit GEMM on f32 MFMA (gfx950).
//
// Replaces tf.layers.conv2d(kernel_size=3, padding='SAME') + bias + ReLU
// (reference src/models/e2evmc/graph.py:76-115) and its Conv2DBackpropInput + ReluGrad.
//
// GEMM view:  out[m][n] = sum_k A[m][k] * B[k][n]
//   m = (image, Y', X') output position,  k = (tap, channel),  n = output channel.
//   A is never materialised: each K-step gathers, per row m, BK contiguous channels of the source
//   pixel addressed by the tap (zero outside the image => TF SAME padding).
//   B = kernel slabs [tap][C][Nout] row-major (HWIO for forward, HWOI for dgrad).
// Forward and dgrad differ only by the tap table and the source/destination strides (dgrad of a
// stride-s conv = s*s parity classes, each a dense gather-GEMM over its own subset of taps), so
// one kernel serves both.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32): roles are swapped so that D = W^T . A^T, i.e. the MFMA
// "row" index is the output channel and the "column" (lane & 15) is the pixel: each lane then owns
// 4 consecutive output channels of one pixel => one 16-byte NHWC store per tile.
//
// LDS images (double buffered):
//   sA[plane = k/4][row m][4 floats], rows XOR-swizzled by (plane & 3): ds_read_b128 of a 16-row
//       fragment hits 16 distinct 16-byte slots (conflict free), ds_write_b128 is 2-way.
//   sB[k][BN + 4]: ds_read_b32, row pitch = 4 (mod 8) dwords => the two k-rows of a 32-lane half
//       land in different bank halves (conflict free).
#include "geeco_common.h"

struct ConvGemmParams {
  const float* x;
  const float* w;
  const float* bias;
  const float* mask;
  float* out;
  long long gs_x, gs_w, gs_b, gs_out;
  int N, Hs, Ws, C;     // source tensor [N][Hs][Ws][C]
  int Hd, Wd, Nout;     // destination tensor [N][Hd][Wd][Nout]
  int Hc, Wc;           // iteration grid: rows enumerate (n, Y', X')
  int ss;               // source pixel = (Y'*ss + dy, X'*ss + dx)
  int ds, oy0, ox0;     // destination pixel = (Y'*ds + oy0, X'*ds + ox0)
  int ntaps;
  int relu;
  long long M;          // N*Hc*Wc
  int Ktot;             // ntaps*C
  int dy[9], dx[9], wslab[9];
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
  constexpr int SPR = BK / 4;     // float4 slots per row per K-step
  constexpr int RPP = 256 / SPR;  // rows staged per pass
  constexpr int PA = BM / RPP;    // A staging passes
  constexpr int LDB = BN + 4;
  constexpr int BN4 = BN / 4;
  constexpr int NB4 = BK * BN4;   // float4s in the B tile
  constexpr int PB = (NB4 + 255) / 256;
  constexpr int WPIX = BM / WM;   // pixels per wave
  constexpr int WCO = BN / WN;    // output channels per wave
  constexpr int TJ = WPIX / 16;
  constexpr int TI = WCO / 16;
  constexpr int KB = BK / 16;
  static_assert(PA >= 1 && TJ >= 1 && TI >= 1 && WM * WN == 4, "bad tile");

  __shared__ __attribute__((aligned(16))) float smem[2 * (BM * BK + BK * LDB) + 48];
  float* sA = smem;
  float* sB = smem + 2 * BM * BK;
  int* sTap = reinterpret_cast<int*>(smem + 2 * (BM * BK + BK * LDB));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int g = blockIdx.z;
  const float* __restrict__ xg = p.x + (long long)g * p.gs_x;
  const float* __restrict__ wg = p.w + (long long)g * p.gs_w;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int C = p.C, C4 = p.C >> 2;

  if (tid < 9) {
    int t = tid;
    bool ok = t < p.ntaps;
    sTap[t] = ok ? p.dy[t] : 0;
    sTap[12 + t] = ok ? p.dx[t] : 0;
    sTap[24 + t] = ok ? (p.dy[t] * p.Ws + p.dx[t]) * C : 0;
    sTap[36 + t] = ok ? p.wslab[t] * C : 0;
  }
  __syncthreads();

  // ---- per-thread staging state -----------------------------------------------------------
  const int kq = tid % SPR;
  long long rb[PA];
  int iy0[PA], ix0[PA];
  {
    const long long HcWc = (long long)p.Hc * p.Wc;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      long long m = m0 + tid / SPR + j * RPP;
      if (m < p.M) {
        long long n = m / HcWc;
        int rem = (int)(m - n * HcWc);
        int yp = rem / p.Wc;
        int xp = rem - yp * p.Wc;
        iy0[j] = yp * p.ss;
        ix0[j] = xp * p.ss;
        rb[j] = ((n * p.Hs + iy0[j]) * p.Ws + ix0[j]) * (long long)C;
      } else {
        iy0[j] = -(1 << 28);
        ix0[j] = -(1 << 28);
        rb[j] = 0;
      }
    }
  }
  int a_tap = kq / C4;
  int a_cq = kq - a_tap * C4;

  int b_row[PB], b_c4[PB], b_tap[PB], b_c[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int idx = tid + i * 256;
    b_row[i] = idx / BN4;
    b_c4[i] = idx - b_row[i] * BN4;
    b_tap[i] = b_row[i] / C;
    b_c[i] = b_row[i] - b_tap[i] * C;
  }

  f32x4 ra[PA], rbv[PB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_tiles = [&]() {
    {
      const int t = a_tap < 9 ? a_tap : 8;
      const bool tv = a_tap < p.ntaps;
      const int dy = sTap[t], dx = sTap[12 + t], toff = sTap[24 + t] + a_cq * 4;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        int iy = iy0[j] + dy, ix = ix0[j] + dx;
        bool v = tv && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
        ra[j] = v ? *reinterpret_cast<const f32x4*>(xg + rb[j] + toff) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      bool v = (tid + i * 256 < NB4) && (b_tap[i] < p.ntaps) && (n0 + b_c4[i] * 4 < p.Nout);
      const int t = b_tap[i] < 9 ? b_tap[i] : 8;
      rbv[i] = v ? *reinterpret_cast<const f32x4*>(wg + (long long)(sTap[36 + t] + b_c[i]) * p.Nout + n0 +
                                                   b_c4[i] * 4)
                 : zero4;
    }
  };
  auto advance = [&]() {
    a_cq += SPR;
    while (a_cq >= C4) {
      a_cq -= C4;
      ++a_tap;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      b_c[i] += BK;
      while (b_c[i] >= C) {
        b_c[i] -= C;
        ++b_tap[i];
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* a = sA + buf * (BM * BK);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      int row = tid / SPR + j * RPP;
      *reinterpret_cast<f32x4*>(a + (kq * BM + (row ^ (kq & 3))) * 4) = ra[j];
    }
    float* b = sB + buf * (BK * LDB);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (tid + i * 256 < NB4) *reinterpret_cast<f32x4*>(b + b_row[i] * LDB + b_c4[i] * 4) = rbv[i];
    }
  };

  // ---- main loop ---------------------------------------------------------------------------
  const int r = lane & 15, q = lane >> 4;
  const int wm = wid % WM, wn = wid / WM;
  const int pixbase = wm * WPIX;
  const int cobase = wn * WCO;

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = zero4;

  const int nk = (p.Ktot + BK - 1) / BK;
  if (nk > 0) {
    load_tiles();
    advance();
    store_tiles(0);
  }
  __syncthreads();

  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) {
      load_tiles();
      advance();
    }
    const float* a = sA + buf * (BM * BK);
    const float* b = sB + buf * (BK * LDB);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      f32x4 xf[TJ];
      float wf[TI][4];
#pragma unroll
      for (int j = 0; j < TJ; ++j)
        xf[j] = *reinterpret_cast<const f32x4*>(a + ((kb * 4 + q) * BM + ((pixbase + j * 16 + r) ^ q)) * 4);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) wf[i][s] = b[(kb * 16 + 4 * q + s) * LDB + cobase + i * 16 + r];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s], xf[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns pixel (lane & 15) of tile j, channels 4*(lane>>4)..+3 of tile i ----
  float* __restrict__ og = p.out + (long long)g * p.gs_out;
  const float* __restrict__ mg = p.mask ? p.mask + (long long)g * p.gs_out : nullptr;
  const float* __restrict__ bg = p.bias ? p.bias + (long long)g * p.gs_b : nullptr;
  const long long HcWc = (long long)p.Hc * p.Wc;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    long long m = m0 + pixbase + j * 16 + r;
    if (m >= p.M) continue;
    long long n = m / HcWc;
    int rem = (int)(m - n * HcWc);
    int yp = rem / p.Wc;
    int xp = rem - yp * p.Wc;
    long long opix = (n * p.Hd + (yp * p.ds + p.oy0)) * p.Wd + (xp * p.ds + p.ox0);
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      int co = n0 + cobase + i * 16 + 4 * q;
      if (co >= p.Nout) continue;
      f32x4 v = acc[i][j];
      if (bg) v += *reinterpret_cast<const f32x4*>(bg + co);
      if (p.relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      }
      if (mg) {
        f32x4 mk = *reinterpret_cast<const f32x4*>(mg + opix * p.Nout + co);
        v.x = mk.x > 0.f ? v.x : 0.f;
        v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f;
        v.w = mk.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(og + opix * p.Nout + co) = v;
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN>
static void launch_cfg(const ConvGemmParams& p, int groups, hipStream_t s) {
  dim3 grid((unsigned)cdiv64(p.M, BM), (unsigned)cdiv(p.Nout, BN), (unsigned)groups);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN>), grid, dim3(256), 0, s, p);
}

static int launch_conv_gemm(const ConvGemmParams& p, int groups, hipStream_t s) {
  if (p.M <= 0) return 0;
  const bool small = p.M * groups < 128 * 256;  // not enough 128-row tiles to fill the chip
  if (p.Nout % 64 == 0) {
    if (small)
      launch_cfg<64, 64, 16, 2, 2>(p, groups, s);
    else
      launch_cfg<128, 64, 16, 2, 2>(p, groups, s);
  } else if (p.Nout % 48 == 0) {
    launch_cfg<128, 48, 16, 4, 1>(p, groups, s);
  } else if (p.Nout % 32 == 0) {
    launch_cfg<128, 32, 16, 4, 1>(p, groups, s);
  } else {
    launch_cfg<128, 16, 16, 4, 1>(p, groups, s);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_conv3x3_fwd(const float* x, const float* w, const float* b, float* y, int groups,
                                 int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int N, int H,
                                 int W, int Cin, int Cout, int stride, int relu, void* stream) {
  GEECO_CHECK_ARG(x && w && y, "conv3x3_fwd: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 1 && W >= 1, "conv3x3_fwd: bad dims");
  GEECO_CHECK_ARG(Cin % 4 == 0 && Cin >= 4, "conv3x3_fwd: Cin=%d must be a multiple of 4", Cin);
  GEECO_CHECK_ARG(Cout % 16 == 0, "conv3x3_fwd: Cout=%d must be a multiple of 16", Cout);
  GEECO_CHECK_ARG(stride >= 1 && stride <= 4, "conv3x3_fwd: stride=%d", stride);
  ConvGemmParams p = {};
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, &Ho, &pt);
  same_pad(W, 3, stride, &Wo, &pl);
  p.x = x; p.w = w; p.bias = b; p.mask = nullptr; p.out = y;
  p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_out = gs_y;
  p.N = N; p.Hs = H; p.Ws = W; p.C = Cin;
  p.Hd = Ho; p.Wd = Wo; p.Nout = Cout;
  p.Hc = Ho; p.Wc = Wo; p.ss = stride; p.ds = 1; p.oy0 = 0; p.ox0 = 0;
  p.ntaps = 9; p.relu = relu;
  p.M = (long long)N * Ho * Wo;
  p.Ktot = 9 * Cin;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      p.dy[ky * 3 + kx] = ky - pt;
      p.dx[ky * 3 + kx] = kx - pl;
      p.wslab[ky * 3 + kx] = ky * 3 + kx;
    }
  return launch_conv_gemm(p, groups, (hipStream_t)stream);
}

extern "C" int geeco_conv3x3_dgrad(const float* dz, const float* wt, const float* ymask, float* dx,
                                   int groups, int64_t gs_dz, int64_t gs_wt, int64_t gs_dx, int N, int H,
                                   int W, int Cin, int Cout, int stride, void* stream) {
  GEECO_CHECK_ARG(dz && wt && dx, "conv3x3_dgrad: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 1 && W >= 1, "conv3x3_dgrad: bad dims");
  GEECO_CHECK_ARG(Cout % 4 == 0, "conv3x3_dgrad: Cout=%d must be a multiple of 4", Cout);
  GEECO_CHECK_ARG(Cin % 16 == 0, "conv3x3_dgrad: Cin=%d must be a multiple of 16", Cin);
  GEECO_CHECK_ARG(stride >= 1 && stride <= 4, "conv3x3_dgrad: stride=%d", stride);
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, &Ho, &pt);
  same_pad(W, 3, stride, &Wo, &pl);
  const int s = stride;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      ConvGemmParams p = {};
      p.x = dz; p.w = wt; p.bias = nullptr; p.mask = ymask; p.out = dx;
      p.gs_x = gs_dz; p.gs_w = gs_wt; p.gs_b = 0; p.gs_out = gs_dx;
      p.N = N; p.Hs = Ho; p.Ws = Wo; p.C = Cout;
      p.Hd = H; p.Wd = W; p.Nout = Cin;
      p.Hc = (H - py + s - 1) / s;
      p.Wc = (W - px + s - 1) / s;
      p.ss = 1; p.ds = s; p.oy0 = py; p.ox0 = px; p.relu = 0;
      int nt = 0;
      for (int ky = 0; ky < 3; ++ky) {
        int vy = py + pt - ky;
        if (((vy % s) + s) % s != 0) continue;
        for (int kx = 0; kx < 3; ++kx) {
          int vx = px + pl - kx;
          if (((vx % s) + s) % s != 0) continue;
          p.dy[nt] = (vy >= 0 ? vy : vy - (s - 1)) / s;   // exact (vy % s == 0)
          p.dx[nt] = (vx >= 0 ? vx : vx - (s - 1)) / s;
          p.wslab[nt] = ky * 3 + kx;
          ++nt;
        }
      }
      p.ntaps = nt;
      p.M = (long long)N * p.Hc * p.Wc;
      p.Ktot = nt * Cout;
      if (p.M <= 0) continue;
      int rc = launch_conv_gemm(p, groups, (hipStream_t)stream);
      if (rc) return rc;
    }
  return 0;
}
