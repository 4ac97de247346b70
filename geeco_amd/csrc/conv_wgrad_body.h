// The generic filter-gradient tile body (see conv_wgrad.hip for the algorithm) as a __device__ function, so that it can serve
// three kernels: the plain launch, the paired conv7 + conv8 launch (conv_wgrad.hip) and the heterogeneous top-of-the-backward
// launch beside conv7's input gradient (conv_gemm.hip).  The caller provides the LDS image (conv_wgrad_smem_floats<..>() floats).
#pragma once
#include "geeco_common.h"

struct WgradParams {
  const float* x;
  const float* dz;
  float* part;   // [G][S][Krows*Cout + Cout]
  float* dw;     // S == 1: the block's tile is final and goes straight to dw / db (no slab, no reduce launch)
  float* db;
  long long gs_x, gs_dz, gs_dw, gs_db;
  int N, H, W, Cin, Ho, Wo, Cout, stride, pt, pl;
  long long M;          // N*Ho*Wo
  long long m_per_split;
  int S;
  int Krows;            // 9*Cin
  int row_tiles, col_tiles;
};

template <int BR, int BC, int MK>
constexpr int conv_wgrad_smem_floats() {
  constexpr int LDA = (BR % 32 == 16) ? BR : BR + 16;
  constexpr int LDBZ = (BC % 32 == 16) ? BC : BC + 16;
  return 2 * MK * (LDA + LDBZ);
}

template <int BR, int BC, int MK>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const int g, const int by, const int split, float* smem) {
  constexpr int LDA = (BR % 32 == 16) ? BR : BR + 16;
  constexpr int LDBZ = (BC % 32 == 16) ? BC : BC + 16;
  constexpr int BR4 = BR / 4, BC4 = BC / 4;
  constexpr int NA4 = MK * BR4, NB4 = MK * BC4;
  constexpr int PAS = (NA4 + 255) / 256, PBS = (NB4 + 255) / 256;
  constexpr int TI = BC / 16;   // co tiles per wave
  constexpr int TJW = BR / 64;  // 16-row k-strips per wave (wave w owns strips w*TJW .. +TJW-1)
  static_assert(BR % 64 == 0, "whole strips per wave");

  float* sA = smem;
  float* sB = smem + 2 * MK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int rt = by / p.col_tiles;
  const int ct = by - rt * p.col_tiles;
  const float* __restrict__ xg = p.x + (long long)g * p.gs_x;
  const float* __restrict__ zg = p.dz + (long long)g * p.gs_dz;
  const int Cin = p.Cin, C4 = Cin >> 2, Cout = p.Cout;
  const long long mbeg = (long long)split * p.m_per_split;
  long long mend = mbeg + p.m_per_split;
  if (mend > p.M) mend = p.M;

  // ---- staging state: A (x gather) -------------------------------------------------------------
  int a_ky[PAS], a_kx[PAS], a_coff[PAS], a_mrow[PAS], a_slot[PAS];
  bool a_ok[PAS];
  int a_n[PAS], a_oy[PAS], a_ox[PAS];
  const long long HoWo = (long long)p.Ho * p.Wo;
  const int adv_x = MK % p.Wo, adv_y = (MK / p.Wo) % p.Ho, adv_n = MK / (p.Wo * p.Ho);   // MK pixels as (columns, rows, frames)
#pragma unroll
  for (int i = 0; i < PAS; ++i) {
    int idx = tid + i * 256;
    a_mrow[i] = idx / BR4;
    a_slot[i] = idx - a_mrow[i] * BR4;
    int sg = rt * BR4 + a_slot[i];
    int tap = sg / C4;
    a_ok[i] = (idx < NA4) && (tap < 9);
    if (tap > 8) tap = 8;
    a_ky[i] = tap / 3 - p.pt;
    a_kx[i] = tap % 3 - p.pl;
    a_coff[i] = (sg - (sg / C4) * C4) * 4;
    long long m = mbeg + a_mrow[i];
    long long n = m / HoWo;
    int rem = (int)(m - n * HoWo);
    a_n[i] = (int)n;
    a_oy[i] = rem / p.Wo;
    a_ox[i] = rem - a_oy[i] * p.Wo;
  }
  // ---- staging state: B (dz rows) --------------------------------------------------------------
  int b_mrow[PBS], b_c4[PBS];
#pragma unroll
  for (int i = 0; i < PBS; ++i) {
    int idx = tid + i * 256;
    b_mrow[i] = idx / BC4;
    b_c4[i] = idx - b_mrow[i] * BC4;
  }
  const int co0 = ct * BC;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[PAS], rz[PBS];
  f32x4 dbsum[PBS];
#pragma unroll
  for (int i = 0; i < PBS; ++i) dbsum[i] = zero4;

  auto load_tiles = [&](long long mb) {
#pragma unroll
    for (int i = 0; i < PAS; ++i) {
      int iy = a_oy[i] * p.stride + a_ky[i];
      int ix = a_ox[i] * p.stride + a_kx[i];
      bool v = a_ok[i] && (mb + a_mrow[i] < mend) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      ra[i] = v ? *reinterpret_cast<const f32x4*>(xg + (((long long)a_n[i] * p.H + iy) * p.W + ix) * Cin + a_coff[i])
                : zero4;
      // advance this thread's pixel by MK for the next stage: carries instead of a wrap loop (conv8: Wo = 2, i.e. 16
      // divergent loop trips per entry and stage - VALU work that the f32 MFMAs pay for)
      a_ox[i] += adv_x;
      const int cx = a_ox[i] >= p.Wo ? 1 : 0;
      a_ox[i] -= cx * p.Wo;
      a_oy[i] += adv_y + cx;
      const int cy = a_oy[i] >= p.Ho ? 1 : 0;
      a_oy[i] -= cy * p.Ho;
      a_n[i] += adv_n + cy;
    }
#pragma unroll
    for (int i = 0; i < PBS; ++i) {
      long long m = mb + b_mrow[i];
      bool v = (tid + i * 256 < NB4) && (m < mend) && (co0 + b_c4[i] * 4 < Cout);
      rz[i] = v ? *reinterpret_cast<const f32x4*>(zg + m * Cout + co0 + b_c4[i] * 4) : zero4;
    }
  };
  auto store_tiles = [&](int buf) {
    float* a = sA + buf * MK * LDA;
    float* b = sB + buf * MK * LDBZ;
#pragma unroll
    for (int i = 0; i < PAS; ++i)
      if (tid + i * 256 < NA4) *reinterpret_cast<f32x4*>(a + a_mrow[i] * LDA + a_slot[i] * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < PBS; ++i)
      if (tid + i * 256 < NB4) {
        *reinterpret_cast<f32x4*>(b + b_mrow[i] * LDBZ + b_c4[i] * 4) = rz[i];
        dbsum[i] += rz[i];
      }
  };

  f32x4 acc[TJW][TI];
#pragma unroll
  for (int j = 0; j < TJW; ++j)
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[j][i] = zero4;
  const int r = lane & 15, q = lane >> 4;

  const long long nchunk = (mend > mbeg) ? (mend - mbeg + MK - 1) / MK : 0;
  if (nchunk > 0) {
    load_tiles(mbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (long long c = 0; c < nchunk; ++c) {
    const int buf = (int)(c & 1);
    const bool more = c + 1 < nchunk;
    if (more) load_tiles(mbeg + (c + 1) * MK);
    const float* a = sA + buf * MK * LDA;
    const float* b = sB + buf * MK * LDBZ;
#pragma unroll
    for (int blk = 0; blk < MK / 4; ++blk) {
      float xv[TJW], zv[TI];
#pragma unroll
      for (int j = 0; j < TJW; ++j) xv[j] = a[(blk * 4 + q) * LDA + (wid * TJW + j) * 16 + r];
#pragma unroll
      for (int i = 0; i < TI; ++i) zv[i] = b[(blk * 4 + q) * LDBZ + i * 16 + r];
#pragma unroll
      for (int j = 0; j < TJW; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[i], xv[j], acc[j][i], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns k-row (lane & 15) of its wave's strip, co = 16 i + 4 q .. +3 ---------
  const long long slab = (long long)p.Krows * Cout + Cout;
  const bool direct = p.S == 1;
  float* __restrict__ part = direct ? p.dw + (long long)g * p.gs_dw : p.part + ((long long)g * p.S + split) * slab;
  float* __restrict__ bpart = direct ? (p.db ? p.db + (long long)g * p.gs_db : nullptr) : part + (long long)p.Krows * Cout;
#pragma unroll
  for (int j = 0; j < TJW; ++j) {
    const int krow = rt * BR + (wid * TJW + j) * 16 + r;
    if (krow < p.Krows) {
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        int co = co0 + i * 16 + 4 * q;
        if (co < Cout) *reinterpret_cast<f32x4*>(part + (long long)krow * Cout + co) = acc[j][i];
      }
    }
  }
  if (rt == 0 && bpart) {
    // bias gradient: per-thread dz sums -> LDS [pixel row][BC] -> fixed-order column sums
    // (the main loop ended with a barrier, so sB is free to reuse)
    float* sT = sB;
#pragma unroll
    for (int i = 0; i < PBS; ++i)
      if (tid + i * 256 < NB4) *reinterpret_cast<f32x4*>(sT + b_mrow[i] * LDBZ + b_c4[i] * 4) = dbsum[i];
    __syncthreads();
    if (tid < BC && co0 + tid < Cout) {
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MK; ++m) s += sT[m * LDBZ + tid];
      bpart[co0 + tid] = s;
    }
  }
}

// two problems of one tile shape as one grid: blocks [0, blocks0) serve q0 (the longer one), the rest q1; inside a problem the
// linear index decodes as (split, tile, group) like the 3-D grid of the single launch
struct WgradPairParams {
  WgradParams q0, q1;
  int blocks0;
};
template <int BR, int BC, int MK>
__device__ __forceinline__ void conv_wgrad_pair_body(const WgradPairParams& pp, const int b, float* smem) {
  if (b < pp.blocks0) {
    const int tiles = pp.q0.row_tiles * pp.q0.col_tiles;
    const int split = b % pp.q0.S, t = b / pp.q0.S;
    conv_wgrad_body<BR, BC, MK>(pp.q0, t / tiles, t % tiles, split, smem);
  } else {
    const int l = b - pp.blocks0;
    const int tiles = pp.q1.row_tiles * pp.q1.col_tiles;
    const int split = l % pp.q1.S, t = l / pp.q1.S;
    conv_wgrad_body<BR, BC, MK>(pp.q1, t / tiles, t % tiles, split, smem);
  }
}

// plan of the generic kernel for one problem (conv_wgrad.hip): fills the geometry fields of *p; *bc = BC + 1000 * BR
void geeco_wgrad_plan(int groups, int N, int H, int W, int Cin, int Cout, int stride, WgradParams* p, int* bc);
int geeco_wgrad_pair_fill(WgradPairParams* pp, const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0,
                          int64_t gs_dz0, int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                          const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                          int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1, int groups,
                          int stride, long long* blocks);
int geeco_wgrad_pair_finish(const WgradPairParams& pp, int groups, hipStream_t s, geeco_slab_reduce* pending2);
