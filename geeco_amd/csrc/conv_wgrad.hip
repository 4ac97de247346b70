// 3x3 convolution filter/bias gradient (Conv2DBackpropFilter + BiasAddGrad) on f32 MFMA.
//
// Autodiff of reference src/models/e2evmc/graph.py:76-115 taken by
// tf.train.AdamOptimizer.minimize (src/models/e2evmc/estimator.py:243-244).
//
//   dw[(tap, ci)][co] = sum_m x[pix(m, tap)][ci] * dz[m][co]        db[co] = sum_m dz[m][co]
//
// GEMM view: rows = (tap, ci) ("k-rows", 9*Cin of them), cols = co, reduction over output
// pixels m.  A block owns a BR x BC tile of dw for one slice of m (split-K); it streams MK = 16
// pixels per stage through LDS (x rows gathered per tap, dz rows dense), and finally writes its
// partial tile to a slab; wgrad_reduce sums the slabs in a fixed order (bitwise reproducible).
//
// MFMA roles (16x16x4): "row" i = co (operand from sB), "col" j = k-row (operand from sA), the
// 4-deep k of one MFMA = 4 consecutive pixels.  Each lane then owns 4 consecutive co of one
// k-row => one 16-byte store.  Both LDS images are [pixel][channels + pad] read with
// ds_read_b32; the row pitch is 16 (mod 32) dwords so the two pixels of a 32-lane half use
// different bank halves.
#include "geeco_common.h"
#include <stdlib.h>

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
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const int g, const int by, const int split) {
  constexpr int LDA = (BR % 32 == 16) ? BR : BR + 16;
  constexpr int LDBZ = (BC % 32 == 16) ? BC : BC + 16;
  constexpr int BR4 = BR / 4, BC4 = BC / 4;
  constexpr int NA4 = MK * BR4, NB4 = MK * BC4;
  constexpr int PAS = (NA4 + 255) / 256, PBS = (NB4 + 255) / 256;
  constexpr int TI = BC / 16;   // co tiles per wave
  constexpr int TJW = BR / 64;  // 16-row k-strips per wave (wave w owns strips w*TJW .. +TJW-1)
  static_assert(BR % 64 == 0, "whole strips per wave");

  __shared__ __attribute__((aligned(16))) float smem[2 * MK * (LDA + LDBZ)];
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

template <int BR, int BC, int MK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  conv_wgrad_body<BR, BC, MK>(p, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x);
}

// TWO independent filter-gradient problems of the same tile shape as ONE grid (round 4: conv7's and conv8's, both ready once
// conv8's input gradient exists; each alone is 432 blocks on 256 CUs for 30 / 12 us: together the short blocks fill the long
// ones' tail and one launch boundary goes).  Blocks [0, blocks0) serve problem 0 (put the longer one first), the rest problem 1;
// inside a problem the linear index decodes as (split, tile, group) like the 3-D grid of the single launch: bitwise the same.
struct WgradPairParams {
  WgradParams q0, q1;
  int blocks0;
};
template <int BR, int BC, int MK>
__global__ __launch_bounds__(256) void conv_wgrad_pair_kernel(const WgradPairParams pp) {
  const int b = (int)blockIdx.x;
  if (b < pp.blocks0) {
    const int tiles = pp.q0.row_tiles * pp.q0.col_tiles;
    const int split = b % pp.q0.S, t = b / pp.q0.S;
    conv_wgrad_body<BR, BC, MK>(pp.q0, t / tiles, t % tiles, split);
  } else {
    const int l = b - pp.blocks0;
    const int tiles = pp.q1.row_tiles * pp.q1.col_tiles;
    const int split = l % pp.q1.S, t = l / pp.q1.S;
    conv_wgrad_body<BR, BC, MK>(pp.q1, t / tiles, t % tiles, split);
  }
}

// Sums the S partial slabs in a fixed order (bitwise reproducible).  Streaming float4 kernel: a wave
// covers 256 consecutive elements; SPLIT (many slabs of a small tensor: conv1 / conv2) deals the slabs
// to the 4 waves of a block and combines them through LDS as ((w0 + w1) + (w2 + w3)), otherwise every
// wave sums all S slabs of its own 256 elements.  KC and the slab pitch are multiples of 4.
template <bool SPLIT>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw,
                                                  float* __restrict__ db, long long gs_dw, long long gs_db, int S,
                                                  long long KC, int Cout, int g, int bx, f32x4 (*sred)[64]) {
  const long long slab = KC + Cout;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long i = SPLIT ? ((long long)bx * 64 + lane) * 4 : ((long long)bx * 256 + threadIdx.x) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < slab) {
    const float* src = part + (long long)g * S * slab + i;
    int k0 = 0, k1 = S;
    if (SPLIT) {
      const int per = (S + 3) / 4;
      k0 = wid * per;
      k1 = (k0 + per < S) ? k0 + per : S;
    }
    int k = k0;
    for (; k + 4 <= k1; k += 4) {       // 4 independent loads in flight; the sum keeps the slab order
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (long long)(k + u) * slab);
#pragma unroll
      for (int u = 0; u < 4; ++u) s += t[u];
    }
    for (; k < k1; ++k) s += *reinterpret_cast<const f32x4*>(src + (long long)k * slab);
  }
  if (SPLIT) {
    sred[wid][lane] = s;
    __syncthreads();
    if (wid != 0) return;
    s = (sred[0][lane] + sred[1][lane]) + (sred[2][lane] + sred[3][lane]);
  }
  if (i < KC) {
    *reinterpret_cast<f32x4*>(dw + (long long)g * gs_dw + i) = s;
  } else if (i < slab && db) {
    *reinterpret_cast<f32x4*>(db + (long long)g * gs_db + (i - KC)) = s;
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           float* __restrict__ db, long long gs_dw, long long gs_db,
                                                           int S, long long KC, int Cout) {
  __shared__ f32x4 sred[SPLIT ? 4 : 1][64];
  wgrad_reduce_body<SPLIT>(part, dw, db, gs_dw, gs_db, S, KC, Cout, blockIdx.y, blockIdx.x, sred);
}

// few float4 columns and many slabs: split the slabs over the waves to get enough parallelism
static bool reduce_splits_slabs(int S, long long KC, int Cout, int groups) {
  return S >= 16 && (KC + Cout) / 4 * groups < 64 * 1024;
}

// Several pending slab sums in one launch (geeco_slab_reduce_batch): block -> (item, group, column block) through the
// prefix table; every item is summed exactly as its own wgrad_reduce_kernel launch would.
struct SlabReduceBatch {
  geeco_slab_reduce it[GEECO_SLAB_REDUCE_MAX];
  int first[GEECO_SLAB_REDUCE_MAX];   // first block of item i
  int bpg[GEECO_SLAB_REDUCE_MAX];     // blocks per group
  int split[GEECO_SLAB_REDUCE_MAX];
  int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const SlabReduceBatch b) {
  __shared__ f32x4 sred[4][64];
  const int bid = blockIdx.x;
  // static indices only (a dynamically indexed by-value argument would be copied to scratch)
  geeco_slab_reduce it = b.it[0];
  int first = 0, bpg = b.bpg[0], split = b.split[0];
#pragma unroll
  for (int i = 1; i < GEECO_SLAB_REDUCE_MAX; ++i) {
    if (i < b.n && bid >= b.first[i]) {
      it = b.it[i]; first = b.first[i]; bpg = b.bpg[i]; split = b.split[i];
    }
  }
  const int lb = bid - first;
  const int g = lb / bpg, bx = lb - g * bpg;
  if (split)
    wgrad_reduce_body<true>(it.part, it.dw, it.db, it.gs_dw, it.gs_db, it.S, it.KC, it.Cout, g, bx, sred);
  else
    wgrad_reduce_body<false>(it.part, it.dw, it.db, it.gs_dw, it.gs_db, it.S, it.KC, it.Cout, g, bx, sred);
}

// while a *_partial entry point runs on this thread, the slab sum is recorded here instead of launched
static thread_local geeco_slab_reduce* tl_pending = nullptr;
void geeco_set_pending_reduce(geeco_slab_reduce* p) { tl_pending = p; }

void geeco_launch_wgrad_reduce(const float* part, float* dw, float* db, long long gs_dw, long long gs_db, int S,
                               long long KC, int Cout, int groups, hipStream_t s) {
  if (tl_pending) {
    geeco_slab_reduce r = {part, dw, db, gs_dw, gs_db, KC, S, Cout, groups, 0};
    *tl_pending = r;
    return;
  }
  const long long n4 = (KC + Cout) / 4;
  if (reduce_splits_slabs(S, KC, Cout, groups)) {
    dim3 rgrid((unsigned)cdiv64(n4, 64), (unsigned)groups);
    geeco_note_kernel("wgrad_reduce_kernel<true>");
    hipLaunchKernelGGL(wgrad_reduce_kernel<true>, rgrid, dim3(256), 0, s, part, dw, db, gs_dw, gs_db, S, KC, Cout);
  } else {
    dim3 rgrid((unsigned)cdiv64(n4, 256), (unsigned)groups);
    geeco_note_kernel("wgrad_reduce_kernel<false>");
    hipLaunchKernelGGL(wgrad_reduce_kernel<false>, rgrid, dim3(256), 0, s, part, dw, db, gs_dw, gs_db, S, KC, Cout);
  }
}

extern "C" int geeco_slab_reduce_batch(const geeco_slab_reduce* items, int n, void* stream) {
  GEECO_CHECK_ARG(n >= 0 && n <= GEECO_SLAB_REDUCE_MAX && (items || n == 0), "slab_reduce_batch: n = %d (0..%d)", n,
                  GEECO_SLAB_REDUCE_MAX);
  SlabReduceBatch b = {};
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    const geeco_slab_reduce& r = items[i];
    if (r.S == 0) continue;             // nothing pending for this one
    GEECO_CHECK_ARG(r.part && r.dw && r.S >= 1 && r.groups >= 1 && r.KC >= 4 && r.KC % 4 == 0 && r.Cout % 4 == 0,
                    "slab_reduce_batch: item %d is malformed", i);
    const int k = b.n++;
    b.it[k] = r;
    b.split[k] = reduce_splits_slabs(r.S, r.KC, r.Cout, r.groups) ? 1 : 0;
    b.bpg[k] = (int)cdiv64((r.KC + r.Cout) / 4, b.split[k] ? 64 : 256);
    b.first[k] = (int)blocks;
    blocks += (long long)b.bpg[k] * r.groups;
  }
  if (b.n == 0) return 0;
  geeco_note_kernel("wgrad_reduce_batch_kernel");
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b);
  GEECO_LAUNCH_CHECK();
  return 0;
}

int64_t geeco_halo_wgrad_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride);
int geeco_try_halo_wgrad(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x,
                         int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout,
                         int stride, void* ws, hipStream_t stream, int* handled);

int64_t geeco_conv1_wgrad_ws_bytes(int groups, int Cin, int Cout, int stride);
int geeco_try_conv1_wgrad(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x,
                          int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout,
                          int stride, void* ws, hipStream_t stream, int* handled);

int64_t geeco_wgrad_lds_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride);
int geeco_try_wgrad_lds(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x, int64_t gs_dz,
                        int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout, int stride, void* ws,
                        hipStream_t stream, int* handled);

static void wgrad_plan(int groups, int N, int H, int W, int Cin, int Cout, int stride, WgradParams* p, int* bc) {
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, &Ho, &pt);
  same_pad(W, 3, stride, &Wo, &pl);
  p->N = N; p->H = H; p->W = W; p->Cin = Cin; p->Ho = Ho; p->Wo = Wo; p->Cout = Cout;
  p->stride = stride; p->pt = pt; p->pl = pl;
  p->M = (long long)N * Ho * Wo;
  p->Krows = 9 * Cin;
  int BC = (Cout % 128 == 0) ? 128 : (Cout % 64 == 0) ? 64 : (Cout % 48 == 0) ? 48 : (Cout % 32 == 0) ? 32 : 16;
  // 128-row tiles halve the dz traffic per MFMA; use them unless padding 9*Cin up to 128 wastes > 12 %
  int BR = (BC >= 64 && (long long)cdiv(p->Krows, 128) * 128 * 100 <= (long long)p->Krows * 112) ? 128 : 64;
  // measured on MI355X: the 128-row / 128-col tiles lose ~1.5 % of the step (occupancy beats traffic here)
  static const int big_tiles = geeco_dev_getenv("GEECO_WGRAD_BIG") ? 1 : 0;
  if (!big_tiles) { BR = 64; if (BC == 128) BC = 64; }
  *bc = BC + 1000 * BR;
  p->row_tiles = cdiv(p->Krows, BR);
  p->col_tiles = cdiv(Cout, BC);
  long long tiles = (long long)groups * p->row_tiles * p->col_tiles;
  long long S = 1024 / tiles;
  if (S < 1) S = 1;
  long long maxS = p->M / 512;          // at least 512 pixels per slice
  if (maxS < 1) maxS = 1;
  if (S > maxS) S = maxS;
  long long mps = cdiv64(p->M, S);
  mps = cdiv64(mps, 64) * 64;   // multiple of every MK
  S = cdiv64(p->M, mps);
  p->S = (int)S;
  p->m_per_split = mps;
}

extern "C" int64_t geeco_conv3x3_wgrad_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride) {
  WgradParams p = {};
  int bc;
  wgrad_plan(groups, N, H, W, Cin, Cout, stride, &p, &bc);
  int64_t a = (int64_t)groups * p.S * ((int64_t)p.Krows * Cout + Cout) * 4;
  int64_t b = geeco_halo_wgrad_ws_bytes(groups, N, H, W, Cin, Cout, stride);
  int64_t c = geeco_conv1_wgrad_ws_bytes(groups, Cin, Cout, stride);
  int64_t d = geeco_wgrad_lds_ws_bytes(groups, N, H, W, Cin, Cout, stride);
  a = a > b ? a : b;
  a = a > d ? a : d;
  return a > c ? a : c;
}

extern "C" int geeco_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, int groups,
                                   int64_t gs_x, int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H,
                                   int W, int Cin, int Cout, int stride, void* ws, void* stream) {
  GEECO_CHECK_ARG(x && dz && dw && ws, "conv3x3_wgrad: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 1 && W >= 1, "conv3x3_wgrad: bad dims");
  GEECO_CHECK_ARG(Cin % 4 == 0 && Cin >= 4, "conv3x3_wgrad: Cin=%d must be a multiple of 4", Cin);
  GEECO_CHECK_ARG(Cout % 16 == 0, "conv3x3_wgrad: Cout=%d must be a multiple of 16", Cout);
  {
    int handled = 0;
    int rc = geeco_try_halo_wgrad(x, dz, dw, db, groups, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, stride, ws,
                                  (hipStream_t)stream, &handled);
    if (rc || handled) return rc;
    rc = geeco_try_conv1_wgrad(x, dz, dw, db, groups, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, stride, ws,
                               (hipStream_t)stream, &handled);
    if (rc || handled) return rc;
    rc = geeco_try_wgrad_lds(x, dz, dw, db, groups, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, stride, ws,
                             (hipStream_t)stream, &handled);
    if (rc || handled) return rc;
  }
  WgradParams p = {};
  int BC;
  wgrad_plan(groups, N, H, W, Cin, Cout, stride, &p, &BC);
  p.x = x; p.dz = dz; p.part = (float*)ws; p.gs_x = gs_x; p.gs_dz = gs_dz;
  p.dw = dw; p.db = db; p.gs_dw = gs_dw; p.gs_db = gs_db;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)p.S, (unsigned)(p.row_tiles * p.col_tiles), (unsigned)groups);
  switch (BC) {
    case 128128: geeco_note_kernel("conv_wgrad_kernel<128, 128, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 16>), grid, dim3(256), 0, s, p); break;
    case 128064: geeco_note_kernel("conv_wgrad_kernel<128, 64, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 16>), grid, dim3(256), 0, s, p); break;
    case 64128: geeco_note_kernel("conv_wgrad_kernel<64, 128, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 16>), grid, dim3(256), 0, s, p); break;
    case 64064: geeco_note_kernel("conv_wgrad_kernel<64, 64, 32>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 32>), grid, dim3(256), 0, s, p); break;
    case 64048: geeco_note_kernel("conv_wgrad_kernel<64, 48, 32>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 48, 32>), grid, dim3(256), 0, s, p); break;
    case 64032: geeco_note_kernel("conv_wgrad_kernel<64, 32, 64>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 32, 64>), grid, dim3(256), 0, s, p); break;
    default: geeco_note_kernel("conv_wgrad_kernel<64, 16, 64>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 16, 64>), grid, dim3(256), 0, s, p); break;
  }
  GEECO_LAUNCH_CHECK();
  if (p.S > 1) {
    geeco_launch_wgrad_reduce((const float*)ws, dw, db, gs_dw, gs_db, p.S, (long long)p.Krows * Cout, Cout, groups, s);
    GEECO_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int geeco_conv3x3_wgrad_partial(const float* x, const float* dz, float* dw, float* db, int groups,
                                           int64_t gs_x, int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H,
                                           int W, int Cin, int Cout, int stride, void* ws, void* stream,
                                           geeco_slab_reduce* pending, int reserved_cus) {
  GEECO_CHECK_ARG(pending, "conv3x3_wgrad_partial: null pending");
  if (int e = geeco_enter_reserved_cus(reserved_cus)) return e;
  geeco_slab_reduce none = {};
  *pending = none;
  geeco_set_pending_reduce(pending);
  const int rc = geeco_conv3x3_wgrad(x, dz, dw, db, groups, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, stride, ws,
                                     stream);
  geeco_set_pending_reduce(nullptr);
  geeco_leave_reserved_cus();
  return rc;
}


// conv7's + conv8's filter gradients in one launch (see conv_wgrad_pair_kernel).  Only the shapes the generic kernel serves with
// 64 x 64 tiles (Cin = 256: the LDS-staged kernels take the layers below); anything else: GEECO_ENOSUP, the caller launches twice.
extern "C" int geeco_conv3x3_wgrad_pair(const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0, int64_t gs_dz0,
                                        int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                                        const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                                        int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1,
                                        int groups, int stride, void* stream, geeco_slab_reduce* pending2) {
  GEECO_CHECK_ARG(x0 && dz0 && dw0 && ws0 && x1 && dz1 && dw1 && ws1, "conv3x3_wgrad_pair: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N0 >= 1 && N1 >= 1 && H0 >= 1 && W0 >= 1 && H1 >= 1 && W1 >= 1, "conv3x3_wgrad_pair: bad dims");
  if (!(stride == 2 && Cin0 == 256 && Cin1 == 256 && Cout0 % 64 == 0 && Cout1 % 64 == 0 && Cout0 >= 64 && Cout1 >= 64)) {
    geeco_set_error("conv3x3_wgrad_pair: shapes outside the paired kernel (Cin = 256, Cout %% 64 == 0, stride 2)");
    return GEECO_ENOSUP;
  }
  WgradPairParams pp = {};
  int bc0, bc1;
  wgrad_plan(groups, N0, H0, W0, Cin0, Cout0, stride, &pp.q0, &bc0);
  wgrad_plan(groups, N1, H1, W1, Cin1, Cout1, stride, &pp.q1, &bc1);
  if (bc0 != 64064 || bc1 != 64064) {
    geeco_set_error("conv3x3_wgrad_pair: the two problems do not share the 64 x 64 tile kernel");
    return GEECO_ENOSUP;
  }
  pp.q0.x = x0; pp.q0.dz = dz0; pp.q0.part = (float*)ws0; pp.q0.gs_x = gs_x0; pp.q0.gs_dz = gs_dz0;
  pp.q0.dw = dw0; pp.q0.db = db0; pp.q0.gs_dw = gs_dw0; pp.q0.gs_db = gs_db0;
  pp.q1.x = x1; pp.q1.dz = dz1; pp.q1.part = (float*)ws1; pp.q1.gs_x = gs_x1; pp.q1.gs_dz = gs_dz1;
  pp.q1.dw = dw1; pp.q1.db = db1; pp.q1.gs_dw = gs_dw1; pp.q1.gs_db = gs_db1;
  const long long b0 = (long long)pp.q0.S * pp.q0.row_tiles * pp.q0.col_tiles * groups;
  const long long b1 = (long long)pp.q1.S * pp.q1.row_tiles * pp.q1.col_tiles * groups;
  pp.blocks0 = (int)b0;
  hipStream_t s = (hipStream_t)stream;
  geeco_note_kernel("conv_wgrad_pair_kernel<64, 64, 32>");
  hipLaunchKernelGGL((conv_wgrad_pair_kernel<64, 64, 32>), dim3((unsigned)(b0 + b1)), dim3(256), 0, s, pp);
  GEECO_LAUNCH_CHECK();
  const WgradParams* q[2] = {&pp.q0, &pp.q1};
  float* dws[2] = {dw0, dw1};
  float* dbs[2] = {db0, db1};
  const int64_t gsw[2] = {gs_dw0, gs_dw1}, gsb[2] = {gs_db0, gs_db1};
  void* wss[2] = {ws0, ws1};
  const int couts[2] = {Cout0, Cout1};
  for (int i = 0; i < 2; ++i) {
    if (pending2) {
      geeco_slab_reduce none = {};
      pending2[i] = none;
    }
    if (q[i]->S > 1) {
      if (pending2) geeco_set_pending_reduce(&pending2[i]);
      geeco_launch_wgrad_reduce((const float*)wss[i], dws[i], dbs[i], gsw[i], gsb[i], q[i]->S, (long long)q[i]->Krows * couts[i],
                                couts[i], groups, s);
      if (pending2) geeco_set_pending_reduce(nullptr);
      GEECO_LAUNCH_CHECK();
    }
  }
  return 0;
}
