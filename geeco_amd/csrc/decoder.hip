// Policy head: state concat, LSTM cell (MFMA gate GEMM + fused gate math), fc1 + heads + losses.
//
// Replaces reference src/models/e2evmc/graph.py:123-192 (concats), :198-260 (lstm_decoder),
// :452-500 (losses) and the loss composition of src/models/e2evmc/estimator.py:206-239.
#include "geeco_common.h"
#include <atomic>
#include <stdlib.h>

// =====================================================================================================
// state concat (graph.py:138-141, 162-165, 187-190)
// =====================================================================================================
struct ConcatParams {
  const float* feats[3];
  float* dfeats[3];
  int ch[3];
  int off[3];       // channel offset of feature i inside a cell
  int nfeat, jnt_off, J, Ctot;
  const float* jnt;
  long long jnt_stride;
  const float* sub_from;
  int N, cells;
  float* state;
  long long state_stride;
  int accumulate;
  float scale;
};

__global__ __launch_bounds__(256) void concat_fwd_kernel(const ConcatParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)p.cells * p.Ctot;
  if (i >= per * p.N) return;
  const int n = (int)(i / per);
  const int rem = (int)(i - (long long)n * per);
  const int cell = rem / p.Ctot, c = rem - cell * p.Ctot;
  float v;
  if (c >= p.jnt_off && c < p.jnt_off + p.J) {
    v = p.jnt[(long long)n * p.jnt_stride + (c - p.jnt_off)];
  } else {
    int f = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k)
      if (k < p.nfeat && c >= p.off[k]) f = k;
    const int cc = c - p.off[f];
    const long long idx = ((long long)n * p.cells + cell) * p.ch[f] + cc;
    v = p.feats[f][idx];
    if (f == 0 && p.sub_from) v = p.sub_from[idx] - v;
  }
  p.state[(long long)n * p.state_stride + rem] = v;
}

// blockIdx.y = feature map (all of them in one launch; maps without a gradient buffer are skipped)
__global__ __launch_bounds__(256) void concat_bwd_kernel(const ConcatParams p) {
  const int f = blockIdx.y;
  if (!p.dfeats[f]) return;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)p.cells * p.ch[f];
  if (i >= per * p.N) return;
  const int n = (int)(i / per);
  const int rem = (int)(i - (long long)n * per);
  const int cell = rem / p.ch[f], c = rem - cell * p.ch[f];
  float d = p.state[(long long)n * p.state_stride + cell * p.Ctot + p.off[f] + c];
  d = p.feats[f][i] > 0.f ? d * p.scale : 0.f;      // ReluGrad of the encoder's last layer
  p.dfeats[f][i] = p.accumulate ? p.dfeats[f][i] + d : d;
}

static int fill_concat(ConcatParams* p, const int* feat_ch, int nfeat, int jnt_pos, int J) {
  int off = 0;
  for (int i = 0; i < nfeat; ++i) {
    if (i == jnt_pos) {
      p->jnt_off = off;
      off += J;
    }
    p->ch[i] = feat_ch[i];
    p->off[i] = off;
    off += feat_ch[i];
  }
  if (jnt_pos >= nfeat) {
    p->jnt_off = off;
    off += J;
  }
  p->Ctot = off;
  p->nfeat = nfeat;
  p->J = J;
  return off;
}

extern "C" int geeco_state_concat_fwd(const float* const* feats, const int* feat_ch, int nfeat, int jnt_pos,
                                      const float* jnt, int64_t jnt_stride, int J, const float* sub_from, int N,
                                      int cells, float* state, int64_t state_stride, void* stream) {
  GEECO_CHECK_ARG(feats && feat_ch && jnt && state, "state_concat_fwd: null pointer");
  GEECO_CHECK_ARG(nfeat >= 1 && nfeat <= 3 && jnt_pos >= 0 && jnt_pos <= nfeat, "state_concat_fwd: nfeat/jnt_pos");
  ConcatParams p = {};
  fill_concat(&p, feat_ch, nfeat, jnt_pos, J);
  for (int i = 0; i < nfeat; ++i) p.feats[i] = feats[i];
  p.jnt = jnt; p.jnt_stride = jnt_stride; p.sub_from = sub_from; p.N = N; p.cells = cells;
  p.state = state; p.state_stride = state_stride;
  GEECO_CHECK_ARG(state_stride >= (int64_t)cells * p.Ctot, "state_concat_fwd: state_stride too small");
  const long long total = (long long)N * cells * p.Ctot;
  hipLaunchKernelGGL(concat_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_state_concat_bwd(const float* dstate, int64_t dstate_stride, const float* const* feats_fwd,
                                      float* const* dfeats, const int* feat_ch, int nfeat, int jnt_pos, int J, int N,
                                      int cells, int accumulate, float scale, void* stream) {
  GEECO_CHECK_ARG(dstate && feats_fwd && dfeats && feat_ch, "state_concat_bwd: null pointer");
  GEECO_CHECK_ARG(nfeat >= 1 && nfeat <= 3 && jnt_pos >= 0 && jnt_pos <= nfeat, "state_concat_bwd: nfeat/jnt_pos");
  ConcatParams p = {};
  fill_concat(&p, feat_ch, nfeat, jnt_pos, J);
  p.state = const_cast<float*>(dstate); p.state_stride = dstate_stride; p.N = N; p.cells = cells;
  p.accumulate = accumulate;
  p.scale = scale;
  for (int i = 0; i < nfeat; ++i) {
    p.feats[i] = feats_fwd[i];
    p.dfeats[i] = dfeats[i];
  }
  long long most = 0;
  for (int f = 0; f < nfeat; ++f) {
    if (!dfeats[f]) continue;
    GEECO_CHECK_ARG(feats_fwd[f], "state_concat_bwd: feats_fwd[%d] is null", f);
    const long long total = (long long)N * cells * p.ch[f];
    if (total > most) most = total;
  }
  if (most == 0) return 0;
  hipLaunchKernelGGL(concat_bwd_kernel, dim3((unsigned)cdiv64(most, 256), (unsigned)nfeat), dim3(256), 0,
                     (hipStream_t)stream, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================
// dense f32 GEMM on MFMA 16x16x4 with split-K slabs (LSTM gate matmul and its backward)
// =====================================================================================================
struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  float* part;
  long long lda, ldb, ldc;
  int M, N, K, ta, tb, accumulate, S, k_per_split;
};

constexpr int GEMM_LD = 80, GEMM_BK = 16;   // LD = 16 (mod 32): conflict-free ds_read_b32

// One 64 x 64 tile (bx = N tile, by = M tile, bz = K split) of C = op(A) op(B).  `store(gm, gn, value)` is called for
// every element of an unsplit product (S == 1) after C has been written: the one-launch LSTM backward hangs the
// state-concat scatter on it.
template <class Store>
__device__ __forceinline__ void gemm_block(const GemmParams& p, int bx, int by, int bz, float (*sA)[GEMM_BK * GEMM_LD],
                                           float (*sB)[GEMM_BK * GEMM_LD], Store store) {
  constexpr int BMg = 64, BNg = 64, BKg = GEMM_BK, LD = GEMM_LD;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = by * BMg, n0 = bx * BNg;
  const int kbeg = bz * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;
  const int r = lane & 15, q = lane >> 4;
  const int wm = (wid & 1) * 32, wn = (wid >> 1) * 32;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ra[4], rb[4];
  // element e = tid + 256*i of a 16 x 64 tile.  For row-major-in-k sources (A not transposed, B
  // transposed) consecutive threads walk k; otherwise they walk m/n, which keeps loads coalesced.
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 256 * i;
      int kk, mm;
      if (!p.ta) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
      int gk = k0 + kk, gm = m0 + mm;
      bool v = gk < kend && gm < p.M;
      ra[i] = v ? (p.ta ? p.A[(long long)gk * p.lda + gm] : p.A[(long long)gm * p.lda + gk]) : 0.f;
      int kb, nn;
      if (p.tb) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
      int gkb = k0 + kb, gn = n0 + nn;
      bool vb = gkb < kend && gn < p.N;
      rb[i] = vb ? (p.tb ? p.B[(long long)gn * p.ldb + gkb] : p.B[(long long)gkb * p.ldb + gn]) : 0.f;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 256 * i;
      int kk, mm;
      if (!p.ta) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
      sA[buf][kk * LD + mm] = ra[i];
      int kb, nn;
      if (p.tb) { kb = e & 15; nn = e >> 4; } else { nn = e & 63; kb = e >> 6; }
      sB[buf][kb * LD + nn] = rb[i];
    }
  };

  const int nk = kend > kbeg ? (kend - kbeg + BKg - 1) / BKg : 0;
  if (nk > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    const bool more = ks + 1 < nk;
    if (more) load_tiles(kbeg + (ks + 1) * BKg);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = sA[buf][(blk * 4 + q) * LD + wm + i * 16 + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = sB[buf][(blk * 4 + q) * LD + wn + j * 16 + r];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
  }
  // D[i = m][j = n]: lane holds n = lane & 15, m = 4 (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn + j * 16 + r;
      if (gn >= p.N) continue;
      const float e[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int gm = m0 + wm + i * 16 + 4 * q + k;
        if (gm >= p.M) continue;
        if (p.S == 1) {
          float* c = p.C + (long long)gm * p.ldc + gn;
          const float v = p.accumulate ? *c + e[k] : e[k];
          *c = v;
          store(gm, gn, v);
        } else {
          p.part[((long long)bz * p.M + gm) * p.N + gn] = e[k];
        }
      }
    }
}

struct NoStore {
  __device__ __forceinline__ void operator()(int, int, float) const {}
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
  __shared__ float sA[2][GEMM_BK * GEMM_LD];
  __shared__ float sB[2][GEMM_BK * GEMM_LD];
  gemm_block(p, blockIdx.x, blockIdx.y, blockIdx.z, sA, sB, NoStore());
}

__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long MN = (long long)p.M * p.N;
  if (i >= MN) return;
  float s = 0.f;
  const float* src = p.part + i;
  int k = 0;
  for (; k + 8 <= p.S; k += 8) {      // 8 independent loads in flight; the sum keeps the slab order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * MN];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < p.S; ++k) s += src[(long long)k * MN];
  const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
  float* c = p.C + (long long)m * p.ldc + n;
  *c = p.accumulate ? *c + s : s;
}

static void gemm_plan(int M, int N, int K, int* S, int* kps) {
  long long tiles = (long long)cdiv(M, 64) * cdiv(N, 64);
  long long s = 512 / tiles;
  if (s < 1) s = 1;
  long long maxs = K / 64;
  if (maxs < 1) maxs = 1;
  if (s > maxs) s = maxs;
  int k = cdiv(cdiv(K, (int)s), 16) * 16;
  *kps = k;
  *S = cdiv(K, k);
}

extern "C" int64_t geeco_gemm_ws_bytes(int M, int N, int K) {
  int S, kps;
  gemm_plan(M, N, K, &S, &kps);
  return S > 1 ? (int64_t)S * M * N * 4 : 16;
}

extern "C" int geeco_gemm_f32(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb, float* C,
                              int64_t ldc, int M, int N, int K, int accumulate, void* ws, void* stream) {
  GEECO_CHECK_ARG(A && B && C, "gemm_f32: null pointer");
  GEECO_CHECK_ARG(M >= 1 && N >= 1 && K >= 1, "gemm_f32: bad dims");
  GemmParams p = {};
  p.A = A; p.B = B; p.C = C; p.part = (float*)ws; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.ta = ta; p.tb = tb; p.accumulate = accumulate;
  gemm_plan(M, N, K, &p.S, &p.k_per_split);
  GEECO_CHECK_ARG(p.S == 1 || ws, "gemm_f32: workspace required for split-K");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64), (unsigned)p.S);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, p);
  GEECO_LAUNCH_CHECK();
  if (p.S > 1) {
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)cdiv64((long long)M * N, 256)), dim3(256), 0, s, p);
    GEECO_LAUNCH_CHECK();
  }
  return 0;
}

// ---- one LSTM step's weight / input gradients in ONE launch --------------------------------------------------------
// With one LSTM step (the goal model's dynimg branch: graph.py:405-407) everything after the gate gradients dz depends
// on dz alone: dWx = X^T dz, db = column sums of dz, dX = dz Wx^T, and the scatter of dX into the encoders' feature
// gradients (state concat backward + ReluGrad of conv8).  As separate launches that is gemm, colsum, gemm + split-K
// reduce, concat_bwd = five dependent kernel boundaries of ~4.5 us each for ~0.2 GFLOP; here they are the blocks of one
// two grids: (1) [0, nA) tiles of dWx, [nA, nA + nB) split-K tiles of dX (slabs; a K loop of this latency-bound GEMM costs
// ~0.7 us per 16-deep step, so the 512-deep product keeps its 8-way split: unsplit it measured +22 us on the step), the
// rest the bias column sums; (2) the slab sum of dX with the feature-gradient scatter in its epilogue.
struct LstmBwdBatch {
  GemmParams a, b;          // a: dWx (S == 1), b: dX (split-K into b.part)
  int nA, nB, ax, bx, by;   // block counts and tile counts of the two products
  const float* dz; long long ldz; int Mz, Nz; float* db;       // column sums
  ConcatParams cc;          // scatter of dX (cc.dfeats[i] may be null); cc.nfeat == 0: no scatter
};

struct ConcatScatter {
  const ConcatParams& c;
  __device__ __forceinline__ void operator()(int n, int d, float v) const {
    const int cell = d / c.Ctot, ch = d - cell * c.Ctot;
    if (cell >= c.cells || (ch >= c.jnt_off && ch < c.jnt_off + c.J)) return;
    int f = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k)
      if (k < c.nfeat && ch >= c.off[k]) f = k;
    if (!c.dfeats[f]) return;
    const long long i = ((long long)n * c.cells + cell) * c.ch[f] + (ch - c.off[f]);
    c.dfeats[f][i] = c.feats[f][i] > 0.f ? v * c.scale : 0.f;       // ReluGrad of the encoder's last layer
  }
};

__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const LstmBwdBatch q) {
  __shared__ float sA[2][GEMM_BK * GEMM_LD];
  __shared__ float sB[2][GEMM_BK * GEMM_LD];
  const int blk = blockIdx.x;
  if (blk < q.nA) {
    gemm_block(q.a, blk % q.ax, blk / q.ax, 0, sA, sB, NoStore());
  } else if (blk < q.nA + q.nB) {
    const int l = blk - q.nA, t = l % (q.bx * q.by);
    if (q.b.S == 1 && q.cc.nfeat > 0)
      gemm_block(q.b, t % q.bx, t / q.bx, l / (q.bx * q.by), sA, sB, ConcatScatter{q.cc});
    else
      gemm_block(q.b, t % q.bx, t / q.bx, l / (q.bx * q.by), sA, sB, NoStore());
  } else {
    const int j = (blk - q.nA - q.nB) * 256 + threadIdx.x;
    if (j >= q.Nz) return;
    float s = 0.f;
    for (int i = 0; i < q.Mz; ++i) s += q.dz[(long long)i * q.ldz + j];     // row order, as geeco_colsum
    q.db[j] = s;
  }
}

// slab sum of dX (fixed slab order, as gemm_reduce_kernel) + the state-concat scatter of the sums
__global__ __launch_bounds__(256) void lstm_step_bwd_finish_kernel(const LstmBwdBatch q) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long MN = (long long)q.b.M * q.b.N;
  if (i >= MN) return;
  float s = 0.f;
  const float* src = q.b.part + i;
  int k = 0;
  for (; k + 8 <= q.b.S; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(k + u) * MN];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < q.b.S; ++k) s += src[(long long)k * MN];
  const int m = (int)(i / q.b.N), n = (int)(i - (long long)m * q.b.N);
  q.b.C[(long long)m * q.b.ldc + n] = s;
  if (q.cc.nfeat > 0) ConcatScatter{q.cc}(m, n, s);
}

extern "C" int64_t geeco_lstm_step_bwd_ws_bytes(int N, int D, int H4) { return geeco_gemm_ws_bytes(N, D, H4); }

extern "C" int geeco_lstm_step_bwd(const float* x, int64_t ldx, const float* dz, int64_t ldz, const float* wx, int64_t ldw,
                                   float* dwx, int64_t lddw, float* db, float* dx, int64_t lddx, int N, int D, int H4,
                                   const float* const* feats_fwd, float* const* dfeats, const int* feat_ch, int nfeat,
                                   int jnt_pos, int J, int cells, void* ws, void* stream) {
  GEECO_CHECK_ARG(x && dz && wx && dwx && db && dx, "lstm_step_bwd: null pointer");
  GEECO_CHECK_ARG(N >= 1 && D >= 1 && H4 >= 1, "lstm_step_bwd: bad dims");
  GEECO_CHECK_ARG(nfeat >= 0 && nfeat <= 3 && (nfeat == 0 || (feats_fwd && dfeats && feat_ch && jnt_pos >= 0 && jnt_pos <= nfeat)),
                  "lstm_step_bwd: concat description");
  LstmBwdBatch q = {};
  // dWx [D][4H] = X^T dz: A = X [N][D] transposed, B = dz [N][4H]
  q.a.A = x; q.a.lda = ldx; q.a.ta = 1; q.a.B = dz; q.a.ldb = ldz; q.a.tb = 0; q.a.C = dwx; q.a.ldc = lddw;
  q.a.M = D; q.a.N = H4; q.a.K = N; q.a.S = 1; q.a.k_per_split = cdiv(N, 16) * 16;
  // dX [N][D] = dz Wx^T: A = dz [N][4H], B = Wx [D][4H] transposed
  q.b.A = dz; q.b.lda = ldz; q.b.ta = 0; q.b.B = wx; q.b.ldb = ldw; q.b.tb = 1; q.b.C = dx; q.b.ldc = lddx;
  q.b.M = N; q.b.N = D; q.b.K = H4; q.b.part = (float*)ws;
  gemm_plan(N, D, H4, &q.b.S, &q.b.k_per_split);
  GEECO_CHECK_ARG(q.b.S == 1 || ws, "lstm_step_bwd: workspace required (geeco_lstm_step_bwd_ws_bytes)");
  q.ax = cdiv(H4, 64); q.nA = q.ax * cdiv(D, 64);
  q.bx = cdiv(D, 64); q.by = cdiv(N, 64); q.nB = q.bx * q.by * q.b.S;
  q.dz = dz; q.ldz = ldz; q.Mz = N; q.Nz = H4; q.db = db;
  if (nfeat > 0) {
    const int ctot = fill_concat(&q.cc, feat_ch, nfeat, jnt_pos, J);
    GEECO_CHECK_ARG((int64_t)cells * ctot <= D, "lstm_step_bwd: %d cells x %d channels exceed the state width %d", cells, ctot, D);
    q.cc.N = N; q.cc.cells = cells; q.cc.scale = 1.f;
    for (int i = 0; i < nfeat; ++i) {
      GEECO_CHECK_ARG(!dfeats[i] || feats_fwd[i], "lstm_step_bwd: feats_fwd[%d] is null", i);
      q.cc.feats[i] = feats_fwd[i];
      q.cc.dfeats[i] = dfeats[i];
    }
  }
  const int blocks = q.nA + q.nB + cdiv(H4, 256);
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, q);
  GEECO_LAUNCH_CHECK();
  if (q.b.S > 1) {
    hipLaunchKernelGGL(lstm_step_bwd_finish_kernel, dim3((unsigned)cdiv64((long long)N * D, 256)), dim3(256), 0,
                       (hipStream_t)stream, q);
    GEECO_LAUNCH_CHECK();
  }
  return 0;
}

// =====================================================================================================
// LSTM gate math (tf.nn.rnn_cell.LSTMCell, gate order i, j, f, o; forget_bias = 1)
// =====================================================================================================
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                             const float* __restrict__ c_prev, float* __restrict__ c,
                                                             float* __restrict__ h, float* __restrict__ gates, int N,
                                                             int H) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int n = i / H, u = i - n * H;
  const float* zr = z + (long long)n * 4 * H;
  const float zi = zr[u] + bias[u], zj = zr[H + u] + bias[H + u];
  const float zf = zr[2 * H + u] + bias[2 * H + u], zo = zr[3 * H + u] + bias[3 * H + u];
  const float si = sigmoidf_(zi), tj = tanhf(zj), sf = sigmoidf_(zf + 1.0f), so = sigmoidf_(zo);
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float cn = sf * cp + si * tj;
  c[i] = cn;
  h[i] = so * tanhf(cn);
  float* gr = gates + (long long)n * 4 * H;
  gr[u] = si; gr[H + u] = tj; gr[2 * H + u] = sf; gr[3 * H + u] = so;
}

__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ gates,
                                                             const float* __restrict__ c_prev, const float* __restrict__ c,
                                                             const float* __restrict__ dh, const float* __restrict__ dc,
                                                             float* __restrict__ dz, float* __restrict__ dc_prev, int N,
                                                             int H) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * H) return;
  const int n = i / H, u = i - n * H;
  const float* gr = gates + (long long)n * 4 * H;
  const float si = gr[u], tj = gr[H + u], sf = gr[2 * H + u], so = gr[3 * H + u];
  const float tc = tanhf(c[i]);
  const float dhv = dh ? dh[i] : 0.f;
  const float dct = (dc ? dc[i] : 0.f) + dhv * so * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[i] : 0.f;
  float* dr = dz + (long long)n * 4 * H;
  dr[u] = dct * tj * si * (1.f - si);
  dr[H + u] = dct * si * (1.f - tj * tj);
  dr[2 * H + u] = dct * cp * sf * (1.f - sf);
  dr[3 * H + u] = dhv * tc * so * (1.f - so);
  if (dc_prev) dc_prev[i] = dct * sf;
}

// The first LSTM step (zero state: z = X Wx alone) with the split-K slab sum of the input projection INSIDE the gate kernel: one
// dependent launch fewer (~5 us of a step whose decoder is pure launch latency).  Block = 64 (sample, unit) pairs x 4 gates:
// wave g sums gate g's slabs in the slab order of gemm_reduce_kernel (bitwise the same z), LDS hands the four sums to wave 0.
__global__ __launch_bounds__(256) void lstm_gates_fwd_slabs_kernel(const float* __restrict__ part, int S,
                                                                   const float* __restrict__ bias, float* __restrict__ z,
                                                                   float* __restrict__ c, float* __restrict__ h,
                                                                   float* __restrict__ gates, int N, int H) {
  __shared__ float sz[4][64];
  const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < N * H;
  const int n = live ? i / H : 0, u = live ? i - n * H : 0;
  const long long MN = (long long)N * 4 * H;
  const long long col = (long long)n * 4 * H + g * H + u;
  float s = 0.f;
  if (live) {
    const float* src = part + col;
    int k = 0;
    for (; k + 8 <= S; k += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[(long long)(k + q) * MN];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; k < S; ++k) s += src[(long long)k * MN];
    z[col] = s;
  }
  sz[g][lane] = s + (live ? bias[g * H + u] : 0.f);
  __syncthreads();
  if (g != 0 || !live) return;
  const float zi = sz[0][lane], zj = sz[1][lane], zf = sz[2][lane], zo = sz[3][lane];
  const float si = sigmoidf_(zi), tj = tanhf(zj), sf = sigmoidf_(zf + 1.0f), so = sigmoidf_(zo);
  const float cn = sf * 0.f + si * tj;
  c[i] = cn;
  h[i] = so * tanhf(cn);
  float* gr = gates + (long long)n * 4 * H;
  gr[u] = si; gr[H + u] = tj; gr[2 * H + u] = sf; gr[3 * H + u] = so;
}

extern "C" int geeco_lstm_input_step_fwd(const float* x, int64_t ldx, const float* wx, int64_t ldw, const float* bias, float* z,
                                         float* c, float* h, float* gates, int N, int H, int D, void* ws, void* stream) {
  GEECO_CHECK_ARG(x && wx && bias && z && c && h && gates, "lstm_input_step_fwd: null pointer");
  GEECO_CHECK_ARG(N >= 1 && H >= 1 && D >= 1 && ldx >= D && ldw >= 4 * (int64_t)H, "lstm_input_step_fwd: bad dims");
  GemmParams p = {};
  p.A = x; p.B = wx; p.C = z; p.part = (float*)ws; p.lda = ldx; p.ldb = ldw; p.ldc = 4 * H;
  p.M = N; p.N = 4 * H; p.K = D;
  gemm_plan(p.M, p.N, p.K, &p.S, &p.k_per_split);
  GEECO_CHECK_ARG(p.S == 1 || ws, "lstm_input_step_fwd: workspace required for split-K (geeco_gemm_ws_bytes(N, 4H, D))");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.M, 64), (unsigned)p.S), dim3(256), 0, s, p);
  GEECO_LAUNCH_CHECK();
  if (p.S > 1)
    hipLaunchKernelGGL(lstm_gates_fwd_slabs_kernel, dim3((unsigned)cdiv(N * H, 64)), dim3(256), 0, s, (const float*)p.part, p.S,
                       bias, z, c, h, gates, N, H);
  else
    hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3((unsigned)cdiv(N * H, 256)), dim3(256), 0, s, (const float*)z, bias,
                       (const float*)nullptr, c, h, gates, N, H);
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_lstm_gates_fwd(const float* z, const float* bias, const float* c_prev, float* c, float* h,
                                    float* gates, int N, int H, void* stream) {
  GEECO_CHECK_ARG(z && bias && c && h && gates && N >= 1 && H >= 1, "lstm_gates_fwd: bad arguments");
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3((unsigned)cdiv(N * H, 256)), dim3(256), 0, (hipStream_t)stream, z,
                     bias, c_prev, c, h, gates, N, H);
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c, const float* dh,
                                    const float* dc, float* dz, float* dc_prev, int N, int H, void* stream) {
  GEECO_CHECK_ARG(gates && c && dz && N >= 1 && H >= 1, "lstm_gates_bwd: bad arguments");
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3((unsigned)cdiv(N * H, 256)), dim3(256), 0, (hipStream_t)stream,
                     gates, c_prev, c, dh, dc, dz, dc_prev, N, H);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================
// fc1 + heads + losses, forward and backward, one workgroup (everything is tiny: N x 128)
// =====================================================================================================
#define GEECO_MAX_HEADS 5
struct HeadsParams {
  const float* h;
  const float* fc1_w;
  const float* fc1_b;
  const float* hw[GEECO_MAX_HEADS];
  const float* hb[GEECO_MAX_HEADS];
  const float* tgt[GEECO_MAX_HEADS];
  long long tstride[GEECO_MAX_HEADS];
  int size[GEECO_MAX_HEADS], off[GEECO_MAX_HEADS], kind[GEECO_MAX_HEADS];
  float weight[GEECO_MAX_HEADS];
  int nheads, OT;
  float loss_scale;
  int N, H, Hfc, backward;
  float* preds;
  float* losses;
  float* dh;
  float* d_fc1_w;
  float* d_fc1_b;
  float* dhw[GEECO_MAX_HEADS];
  float* dhb[GEECO_MAX_HEADS];
  float* a1;    // ws: [N][Hfc]
  float* da1;   // ws: [N][Hfc]
  float* dpred; // ws: [N][OT]
};

template <class T>
__device__ __forceinline__ T sel5(T const (&a)[GEECO_MAX_HEADS], int i) {
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : (i == 3 ? a[3] : a[4])));
}

// C[M][N] = A[M][K] B[K][N] inside ONE workgroup on MFMA 16x16x4: waves take 16x16 output tiles
// round-robin; operands are fetched straight from global memory (everything is L2-resident and
// tiny), 2 loads per MFMA per lane.  A(i, k) = a[i * a_rs + k * a_ks], B(k, j) = b[k * b_ks + j * b_cs]
// (pointer + strides, so the K loop is pure pointer bumps); st(i, j, v) stores.
template <class FS>
__device__ __forceinline__ void block_mfma_gemm(int M, int N, int K, const float* a, int a_rs, int a_ks,
                                                const float* b, int b_ks, int b_cs, FS st) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int tn = (N + 15) >> 4, nt = ((M + 15) >> 4) * tn;
  for (int t = wave; t < nt; t += nw) {
    const int ti = t / tn, tj = t - ti * tn;
    const int i = ti * 16 + r, j = tj * 16 + r;
    const bool iv = i < M, jv = j < N;
    const float* ap = a + (long long)(iv ? i : 0) * a_rs + q * a_ks;
    const float* bp = b + (long long)(jv ? j : 0) * b_cs + q * b_ks;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += 4) {
      const bool kv = k0 + q < K;
      const float av = (iv && kv) ? ap[(long long)k0 * a_ks] : 0.f;
      const float bv = (jv && kv) ? bp[(long long)k0 * b_ks] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int io = ti * 16 + 4 * q + k;
      if (io < M && jv) st(io, j, e[k]);
    }
  }
}

__global__ __launch_bounds__(1024) void heads_loss_kernel(const HeadsParams p) {
  const int tid = threadIdx.x, NT = 1024;
  const int N = p.N, H = p.H, F = p.Hfc, OT = p.OT;
  __shared__ float s_red[16][GEECO_MAX_HEADS];
  __shared__ float s_hb[32];
  __shared__ int s_hd[32], s_hc[32];
  float* whm = p.dpred + (long long)N * OT;     // ws: [OT][F] = the head kernels side by side, transposed
  // P0: per-output-column tables and the packed [OT][F] head matrix
  if (tid < OT) {
    int hd = 0;
#pragma unroll
    for (int k = 1; k < GEECO_MAX_HEADS; ++k)
      if (k < p.nheads && tid >= p.off[k]) hd = k;
    s_hd[tid] = hd;
    s_hc[tid] = tid - sel5(p.off, hd);
    s_hb[tid] = sel5(p.hb, hd)[tid - sel5(p.off, hd)];
  }
  __syncthreads();
  for (int e = tid; e < OT * F; e += NT) {
    const int o = e / F, f = e - o * F;
    const int hd = s_hd[o];
    whm[e] = sel5(p.hw, hd)[f * sel5(p.size, hd) + s_hc[o]];
  }
  // P1: a1 = relu(h W1 + b1)                                   graph.py:229-230
  block_mfma_gemm(N, F, H, p.h, H, 1, p.fc1_w, F, 1,
                  [&](int n, int j, float v) { p.a1[n * F + j] = fmaxf(v + p.fc1_b[j], 0.f); });
  __syncthreads();
  // P2: preds[n][sum of head sizes]                             graph.py:233-259
  block_mfma_gemm(N, OT, F, p.a1, F, 1, whm, 1, F, [&](int n, int o, float v) { p.preds[n * OT + o] = v + s_hb[o]; });
  __syncthreads();
  // P3: losses and d(loss)/d(pred)           graph.py:430-500, estimator.py:206-239
  //   kind 0: tf.losses.mean_squared_error (mean over N*size); kind 1: softmax cross-entropy against
  //   one_hot(rint(target) + 1) (mean over N)
  float lsum[GEECO_MAX_HEADS];
#pragma unroll
  for (int k = 0; k < GEECO_MAX_HEADS; ++k) lsum[k] = 0.f;
  const float invn = 1.f / N;
  for (int n = tid; n < N; n += NT) {
    const float* pr = p.preds + n * OT;
    float* dp = p.dpred + n * OT;
#pragma unroll
    for (int hd = 0; hd < GEECO_MAX_HEADS; ++hd) {
      if (hd >= p.nheads) break;
      const int sz = p.size[hd], of = p.off[hd];
      const float* tg = p.tgt[hd] + (long long)n * p.tstride[hd];
      const float wsc = p.weight[hd] * p.loss_scale;
      if (p.kind[hd] == 0) {
        const float c2 = 2.f / (float)(N * sz) * wsc;
        for (int c = 0; c < sz; ++c) {
          const float d = pr[of + c] - tg[c];
          lsum[hd] += d * d;
          dp[of + c] = d * c2;
        }
      } else {
        const int label = (int)rintf(tg[0]) + 1;             // estimator.py:213-215
        float mx = pr[of];
        for (int c = 1; c < sz; ++c) mx = fmaxf(mx, pr[of + c]);
        float se = 0.f;
        for (int c = 0; c < sz; ++c) se += expf(pr[of + c] - mx);
        const bool lv = label >= 0 && label < sz;             // one_hot of an out-of-range label is all-zero
        if (lv) lsum[hd] += mx + logf(se) - pr[of + label];
        for (int c = 0; c < sz; ++c)
          dp[of + c] = lv ? (expf(pr[of + c] - mx) / se - (c == label ? 1.f : 0.f)) * invn * wsc : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < GEECO_MAX_HEADS; ++k) {
    lsum[k] = wave_reduce_sum(lsum[k]);
    if ((tid & 63) == 0) s_red[tid >> 6][k] = lsum[k];
  }
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    for (int hd = 0; hd < p.nheads; ++hd) {
      float a = 0.f;
      for (int w = 0; w < 16; ++w) a += s_red[w][hd];
      a *= p.kind[hd] == 0 ? 1.f / (float)(N * p.size[hd]) : invn;
      p.losses[1 + hd] = a;
      total += p.weight[hd] * a;
    }
    p.losses[0] = total;
  }
  if (!p.backward) return;
  __syncthreads();
  // P4: head gradients  d_hw[f][o] = sum_n a1[n][f] dpred[n][o];  da1 = (dpred Wh^T) * relu'
  block_mfma_gemm(F, OT, N, p.a1, 1, F, p.dpred, OT, 1, [&](int f, int o, float v) {
    const int hd = s_hd[o];
    sel5(p.dhw, hd)[f * sel5(p.size, hd) + s_hc[o]] = v;
  });
  for (int o = tid; o < OT; o += NT) {
    float sum = 0.f;
    for (int n = 0; n < N; ++n) sum += p.dpred[n * OT + o];
    sel5(p.dhb, s_hd[o])[s_hc[o]] = sum;
  }
  block_mfma_gemm(N, F, OT, p.dpred, OT, 1, whm, F, 1,
                  [&](int n, int f, float v) { p.da1[n * F + f] = p.a1[n * F + f] > 0.f ? v : 0.f; });
  __syncthreads();
  // P5: fc1 gradients and d(h)
  block_mfma_gemm(H, F, N, p.h, 1, H, p.da1, F, 1, [&](int k, int j, float v) { p.d_fc1_w[k * F + j] = v; });
  for (int j = tid; j < F; j += NT) {
    float sum = 0.f;
    for (int n = 0; n < N; ++n) sum += p.da1[n * F + j];
    p.d_fc1_b[j] = sum;
  }
  block_mfma_gemm(N, H, F, p.da1, F, 1, p.fc1_w, 1, F, [&](int n, int k, float v) { p.dh[n * H + k] = v; });
}

// Same computation with every operand resident in LDS (N <= 32, H == Hfc == 128: the shapes of all
// BASELINE.json configs with batch <= 32; other sizes take heads_loss_kernel).  The phases of heads_loss_kernel hand their results
// over through global memory (~1-2 us of store->load latency per phase, and an L2 round trip per
// MFMA operand); here h, fc1/kernel, the packed head matrix and every intermediate live in LDS, so a
// phase costs LDS latency only.  Gradients and predictions are written straight to their outputs.
// Row pitches of the LDS arrays.  Every array is an MFMA operand in two roles: "row-strided" (lane (r, q) reads element
// [r][k0 + q]: bank r * P + q) and "k-strided" ([k0 + q][r]: bank q * P + r); ds_read_b32 serves 32 lanes (r = 0..15, q = 0..1)
// per cycle over 32 banks.  P = 146 = 18 (mod 32): r * 18 takes 16 distinct even banks (row-strided: conflict free) and
// q * 18 + r overlaps in 2 of 32 lanes only (k-strided: 1.06 cycles instead of 1).  Round 2's pitches (129 for the
// activations: 2-way in both roles; 144 for fc1/kernel: conflict free k-strided but 8-way row-strided in the d(h) product)
// made this one-workgroup kernel LDS-bandwidth bound: PMC 53 % conflict cycles, 2 b32 reads per MFMA.
constexpr int HL_NMAX = 32, HL_DMAX = 128, HL_OMAX = 24;
constexpr int HL_WP = 146;              // fc1/kernel [h][f]
constexpr int HL_RP = 146;              // [n][..] activations and the packed head matrix [o][f]
constexpr int HL_OP = 33;               // predictions / targets [n][o] (element-wise use only)
constexpr int HL_OPD = 50;              // d(loss)/d(pred) [n][o]: MFMA operand in both roles (18 mod 32)
constexpr int HL_LDS_FLOATS = HL_DMAX * HL_WP + 3 * HL_NMAX * HL_RP + HL_OMAX * HL_RP + 2 * HL_NMAX * HL_OP + HL_NMAX * HL_OPD;

__global__ __launch_bounds__(1024) void heads_loss_lds_kernel(const HeadsParams p) {
  const int tid = threadIdx.x, NT = 1024;
  constexpr int H = HL_DMAX, F = HL_DMAX;      // the launcher takes this path only for H == Hfc == 128 (the defaults)
  const int N = p.N, OT = p.OT;
  extern __shared__ __attribute__((aligned(16))) float hl_smem[];
  float* sW1 = hl_smem;                          // [H][HL_WP]   fc1/kernel [h][f]
  float* sH = sW1 + HL_DMAX * HL_WP;             // [N][HL_RP]   LSTM output
  float* sA1 = sH + HL_NMAX * HL_RP;             // [N][HL_RP]   relu(fc1)
  float* sDA = sA1 + HL_NMAX * HL_RP;            // [N][HL_RP]   d(loss)/d(fc1 pre-activation)
  float* sWh = sDA + HL_NMAX * HL_RP;            // [OT][HL_RP]  head kernels side by side, transposed
  float* sPr = sWh + HL_OMAX * HL_RP;            // [N][OP] predictions
  float* sDp = sPr + HL_NMAX * HL_OP;            // [N][OPD] d(loss)/d(pred)
  constexpr int OP = HL_OP, OPD = HL_OPD;
  __shared__ float s_red[16][GEECO_MAX_HEADS];
  __shared__ float s_hb[32], s_b1[HL_DMAX];
  __shared__ int s_hd[32], s_hc[32];
#ifdef GEECO_STAMPS
#define HSTAMP(i) do { if (tid == 0) reinterpret_cast<unsigned long long*>(p.dpred)[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP(i)
#endif
  float* sTg = sDp + HL_NMAX * HL_OPD;           // [N][OP] targets (kind 1: the label in the head's first column)
  HSTAMP(0);
  // P0: every global input is fetched up front (independent loads, issued together), then one barrier
  auto head_of = [&](int o) {
    int hd = 0;
#pragma unroll
    for (int k = 1; k < GEECO_MAX_HEADS; ++k)
      if (k < p.nheads && o >= p.off[k]) hd = k;
    return hd;
  };
  if (tid < OT) {
    const int hd = head_of(tid);
    s_hd[tid] = hd;
    s_hc[tid] = tid - sel5(p.off, hd);
    s_hb[tid] = sel5(p.hb, hd)[tid - sel5(p.off, hd)];
  }
  for (int e = tid; e < F; e += NT) s_b1[e] = p.fc1_b[e];
  if ((F & 3) == 0 && (H & 3) == 0) {
    constexpr int WV = HL_DMAX * HL_DMAX / 4 / 1024;      // float4 of fc1/kernel per thread (4)
    f32x4 wv[WV], hv;
    const int F4 = F >> 2;
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int e4 = tid + NT * i;
      wv[i] = e4 < H * F4 ? reinterpret_cast<const f32x4*>(p.fc1_w)[e4] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    hv = tid < (N * H >> 2) ? reinterpret_cast<const f32x4*>(p.h)[tid] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int e4 = tid + NT * i;
      if (e4 < H * F4) {      // the row pitch is even, not a multiple of 4 floats: two 8-byte stores
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2* d = reinterpret_cast<f32x2*>(sW1 + (e4 / F4) * HL_WP + (e4 % F4) * 4);
        d[0] = f32x2{wv[i].x, wv[i].y};
        d[1] = f32x2{wv[i].z, wv[i].w};
      }
    }
    if (tid < (N * H >> 2)) {
      const int e = tid * 4;
      float* d = sH + (e / H) * HL_RP + (e % H);
      d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
    }
  } else {
    for (int e = tid; e < H * F; e += NT) sW1[(e / F) * HL_WP + (e % F)] = p.fc1_w[e];
    for (int e = tid; e < N * H; e += NT) sH[(e / H) * HL_RP + (e % H)] = p.h[e];
  }
  for (int e = tid; e < OT * F; e += NT) {
    const int o = e / F, f = e - o * F;
    const int hd = head_of(o);
    sWh[o * HL_RP + f] = sel5(p.hw, hd)[f * sel5(p.size, hd) + (o - sel5(p.off, hd))];
  }
  for (int e = tid; e < N * OT; e += NT) {
    const int n = e / OT, o = e - n * OT;
    const int hd = head_of(o), c = o - sel5(p.off, hd);
    if (sel5(p.kind, hd) == 0 || c == 0) sTg[n * OP + o] = sel5(p.tgt, hd)[(long long)n * sel5(p.tstride, hd) + c];
  }
  __syncthreads();
  HSTAMP(1);
  HSTAMP(2);
  // The six small GEMMs run through ONE copy of the MFMA tile loop (operands and results in LDS, or a
  // global result): the kernel executes once per step from a cold instruction cache, so its run time
  // follows its code size - six inlined, unrolled GEMMs (47 KB of code) took 3x longer than this loop.
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nph = p.backward ? 6 : 2;
#pragma unroll 1
  for (int ph = 0; ph < nph; ++ph) {
    // C(i, j) = sum_k A(i, k) B(k, j);  A(i, k) = smem[a + i a_rs + k a_ks], B(k, j) = smem[b + k b_ks + j b_cs],
    // C -> smem[c + i c_rs + j c_cs] or, if gc, gc[i c_rs + j c_cs]
    int M, Nn, K, a, a_rs, a_ks, b, b_ks, b_cs, c = 0, c_rs, c_cs;
    float* gc = nullptr;
    const int oH = (int)(sH - hl_smem), oW1 = 0, oA1 = (int)(sA1 - hl_smem), oDA = (int)(sDA - hl_smem);
    const int oWh = (int)(sWh - hl_smem), oPr = (int)(sPr - hl_smem), oDp = (int)(sDp - hl_smem);
    switch (ph) {
      case 0:   // P1: fc1 pre-activation = h W1                                  graph.py:229-230
        M = N; Nn = F; K = H; a = oH; a_rs = HL_RP; a_ks = 1; b = oW1; b_ks = HL_WP; b_cs = 1; c = oA1; c_rs = HL_RP; c_cs = 1;
        break;
      case 1:   // P2: preds = a1 Wh                                              graph.py:233-259
        M = N; Nn = OT; K = F; a = oA1; a_rs = HL_RP; a_ks = 1; b = oWh; b_ks = 1; b_cs = HL_RP; c = oPr; c_rs = OP; c_cs = 1;
        break;
      case 2:   // d(a1) = dpred Wh^T (ReluGrad applied after the loop body)
        M = N; Nn = F; K = OT; a = oDp; a_rs = OPD; a_ks = 1; b = oWh; b_ks = HL_RP; b_cs = 1; c = oDA; c_rs = HL_RP; c_cs = 1;
        break;
      case 3:   // head kernel gradients [f][o] = a1^T dpred, into the (now free) packed head matrix as [o][f]
        M = F; Nn = OT; K = N; a = oA1; a_rs = 1; a_ks = HL_RP; b = oDp; b_ks = OPD; b_cs = 1; c = oWh; c_rs = 1; c_cs = HL_RP;
        break;
      case 4:   // d(h) = d(a1) W1^T
        M = N; Nn = H; K = F; a = oDA; a_rs = HL_RP; a_ks = 1; b = oW1; b_ks = 1; b_cs = HL_WP; gc = p.dh; c_rs = H; c_cs = 1;
        break;
      default:  // d(fc1/kernel) = h^T d(a1)
        M = H; Nn = F; K = N; a = oH; a_rs = 1; a_ks = HL_RP; b = oDA; b_ks = HL_RP; b_cs = 1; gc = p.d_fc1_w; c_rs = F; c_cs = 1;
        break;
    }
    const int tn = (Nn + 15) >> 4, nt = ((M + 15) >> 4) * tn;
#pragma unroll 1
    for (int t = wave; t < nt; t += 16) {
      const int ti = t / tn, tj = t - ti * tn;
      const int i = ti * 16 + r, j = tj * 16 + r;
      const bool iv = i < M, jv = j < Nn;
      // rows / columns beyond the matrix read row / column 0 (valid data); their results are never stored,
      // so only the K tail needs masking.  The K loop is pointer bumps: integer multiplies for the operand
      // addresses (quarter rate) made the VALU, not the MFMA pipe, the limit of these tiny GEMMs.
      const float* ap = hl_smem + a + (iv ? i : 0) * a_rs + q * a_ks;
      const float* bp = hl_smem + b + (jv ? j : 0) * b_cs + q * b_ks;
      const int a4 = 4 * a_ks, b4 = 4 * b_ks;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      int k0 = 0;
#pragma unroll 1
      for (; k0 + 16 <= K; k0 += 16) {      // 4 k-steps per pass, two accumulator chains
        const float a0 = ap[0], a1 = ap[a4], a2 = ap[2 * a4], a3 = ap[3 * a4];
        const float b0 = bp[0], b1 = bp[b4], b2 = bp[2 * b4], b3 = bp[3 * b4];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
        ap += 4 * a4;
        bp += 4 * b4;
      }
      for (; k0 < K; k0 += 4) {             // tail: clamped address, value zeroed by a 0/1 factor
        const int k = k0 + q;
        const float mk = k < K ? 1.f : 0.f;
        const int back = k < K ? 0 : k - (K - 1);     // steps past the last valid k
        const float av = ap[-back * a_ks] * mk, bv = bp[-back * b_ks] * mk;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
        ap += a4;
        bp += b4;
      }
      const f32x4 acc = acc0 + acc1;
      const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int io = ti * 16 + 4 * q + k;
        if (io < M && jv) {
          if (gc)
            gc[io * c_rs + j * c_cs] = e[k];
          else
            hl_smem[c + io * c_rs + j * c_cs] = e[k];
        }
      }
    }
    __syncthreads();
    HSTAMP(3 + ph);
    if (ph == 0) {            // a1 = relu(. + b1)
      for (int e = tid; e < N * F; e += NT) {
        const int n = e / F, f = e - n * F;
        sA1[n * HL_RP + f] = fmaxf(sA1[n * HL_RP + f] + s_b1[f], 0.f);
      }
      __syncthreads();
    } else if (ph == 1) {     // + head biases; predictions out; losses and d(loss)/d(pred)
      for (int e = tid; e < N * OT; e += NT) {
        const int n = e / OT, o = e - n * OT;
        const float v = sPr[n * OP + o] + s_hb[o];
        sPr[n * OP + o] = v;
        p.preds[e] = v;
      }
      __syncthreads();
      // P3           graph.py:430-500, estimator.py:206-239 (as heads_loss_kernel)
      float lsum[GEECO_MAX_HEADS];
#pragma unroll
      for (int k = 0; k < GEECO_MAX_HEADS; ++k) lsum[k] = 0.f;
      const float invn = 1.f / N;
      for (int n = tid; n < N; n += NT) {
        const float* pr = sPr + n * OP;
        float* dp = sDp + n * OPD;
#pragma unroll 1
        for (int hd = 0; hd < p.nheads; ++hd) {
          const int sz = sel5(p.size, hd), of = sel5(p.off, hd);
          const float* tg = sTg + n * OP + of;
          const float wsc = sel5(p.weight, hd) * p.loss_scale;
          float l = 0.f;
          if (sel5(p.kind, hd) == 0) {
            const float c2 = 2.f / (float)(N * sz) * wsc;
            for (int cc = 0; cc < sz; ++cc) {
              const float d = pr[of + cc] - tg[cc];
              l += d * d;
              dp[of + cc] = d * c2;
            }
          } else {
            const int label = (int)rintf(tg[0]) + 1;             // estimator.py:213-215
            float mx = pr[of];
            for (int cc = 1; cc < sz; ++cc) mx = fmaxf(mx, pr[of + cc]);
            float se = 0.f;
            for (int cc = 0; cc < sz; ++cc) se += expf(pr[of + cc] - mx);
            const bool lv = label >= 0 && label < sz;             // one_hot of an out-of-range label is all-zero
            if (lv) l = mx + logf(se) - pr[of + label];
            for (int cc = 0; cc < sz; ++cc)
              dp[of + cc] = lv ? (expf(pr[of + cc] - mx) / se - (cc == label ? 1.f : 0.f)) * invn * wsc : 0.f;
          }
#pragma unroll
          for (int k = 0; k < GEECO_MAX_HEADS; ++k)
            if (k == hd) lsum[k] += l;
        }
      }
#pragma unroll
      for (int k = 0; k < GEECO_MAX_HEADS; ++k) {
        lsum[k] = wave_reduce_sum(lsum[k]);
        if ((tid & 63) == 0) s_red[tid >> 6][k] = lsum[k];
      }
      __syncthreads();
      if (tid == 0) {
        float total = 0.f;
        for (int hd = 0; hd < p.nheads; ++hd) {
          float acc_l = 0.f;
          for (int w = 0; w < 16; ++w) acc_l += s_red[w][hd];
          acc_l *= sel5(p.kind, hd) == 0 ? 1.f / (float)(N * sel5(p.size, hd)) : invn;
          p.losses[1 + hd] = acc_l;
          total += sel5(p.weight, hd) * acc_l;
        }
        p.losses[0] = total;
      }
    } else if (ph == 2) {     // ReluGrad of fc1; head bias gradients
      for (int e = tid; e < N * F; e += NT) {
        const int n = e / F, f = e - n * F;
        if (!(sA1[n * HL_RP + f] > 0.f)) sDA[n * HL_RP + f] = 0.f;
      }
      for (int o = tid; o < OT; o += NT) {
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += sDp[n * OPD + o];
        sel5(p.dhb, s_hd[o])[s_hc[o]] = sum;
      }
      __syncthreads();
    } else if (ph == 3) {     // scatter the head kernel gradients to their variables
      for (int e = tid; e < OT * F; e += NT) {
        const int o = e / F, f = e - o * F;
        const int hd = s_hd[o];
        sel5(p.dhw, hd)[f * sel5(p.size, hd) + s_hc[o]] = sWh[o * HL_RP + f];
      }
    } else if (ph == 4) {     // d(fc1/bias)
      for (int jj = tid; jj < F; jj += NT) {
        float sum = 0.f;
        for (int n = 0; n < N; ++n) sum += sDA[n * HL_RP + jj];
        p.d_fc1_b[jj] = sum;
      }
    }
  }
  HSTAMP(10);
}

extern "C" int64_t geeco_heads_ws_bytes(int N, int H, int Hfc) {
  (void)H;
  return ((int64_t)2 * N * Hfc + (int64_t)N * 32 + (int64_t)32 * Hfc) * 4;
}

extern "C" int geeco_heads_loss_fwd_bwd(const float* h, const float* fc1_w, const float* fc1_b, int nheads,
                                        const float* const* heads_w, const float* const* heads_b,
                                        const int* head_size, const int* head_kind, const float* head_weight,
                                        const float* const* targets, const int64_t* target_stride, float loss_scale,
                                        int N, int H, int Hfc, float* preds, float* losses, int backward, float* dh,
                                        float* d_fc1_w, float* d_fc1_b, float* const* d_heads_w,
                                        float* const* d_heads_b, float* ws, void* stream) {
  GEECO_CHECK_ARG(h && fc1_w && fc1_b && heads_w && heads_b && head_size && head_kind && head_weight && targets &&
                      target_stride && preds && losses && ws, "heads_loss: null pointer");
  GEECO_CHECK_ARG(nheads >= 1 && nheads <= GEECO_MAX_HEADS, "heads_loss: nheads=%d outside 1..%d", nheads, GEECO_MAX_HEADS);
  GEECO_CHECK_ARG(N >= 1 && N <= 4096 && H >= 1 && Hfc >= 1, "heads_loss: bad dims");
  GEECO_CHECK_ARG(!backward || (dh && d_fc1_w && d_fc1_b && d_heads_w && d_heads_b), "heads_loss: null gradient pointer");
  HeadsParams p = {};
  p.h = h; p.fc1_w = fc1_w; p.fc1_b = fc1_b; p.nheads = nheads; p.loss_scale = loss_scale;
  p.N = N; p.H = H; p.Hfc = Hfc; p.backward = backward; p.preds = preds; p.losses = losses;
  p.dh = dh; p.d_fc1_w = d_fc1_w; p.d_fc1_b = d_fc1_b;
  int off = 0;
  for (int i = 0; i < nheads; ++i) {
    GEECO_CHECK_ARG(head_size[i] >= 1 && head_size[i] <= 16, "heads_loss: head %d size %d", i, head_size[i]);
    GEECO_CHECK_ARG(head_kind[i] == 0 || head_kind[i] == 1, "heads_loss: head %d kind %d", i, head_kind[i]);
    GEECO_CHECK_ARG(heads_w[i] && heads_b[i] && targets[i], "heads_loss: head %d null pointer", i);
    p.hw[i] = heads_w[i]; p.hb[i] = heads_b[i]; p.tgt[i] = targets[i]; p.tstride[i] = target_stride[i];
    p.size[i] = head_size[i]; p.off[i] = off; p.kind[i] = head_kind[i]; p.weight[i] = head_weight[i];
    off += head_size[i];
    if (backward) {
      GEECO_CHECK_ARG(d_heads_w[i] && d_heads_b[i], "heads_loss: head %d null gradient pointer", i);
      p.dhw[i] = d_heads_w[i]; p.dhb[i] = d_heads_b[i];
    }
  }
  GEECO_CHECK_ARG(off <= 32, "heads_loss: %d outputs > 32", off);
  p.OT = off;
  p.a1 = ws; p.da1 = ws + (long long)N * Hfc; p.dpred = ws + 2ll * N * Hfc;
  static const int no_lds = geeco_dev_getenv("GEECO_HEADS_NO_LDS") ? 1 : 0;
  if (!no_lds && N <= HL_NMAX && H == HL_DMAX && Hfc == HL_DMAX && off <= HL_OMAX) {
    const size_t lds = (size_t)HL_LDS_FLOATS * 4;
    static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&heads_loss_lds_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(heads_loss_lds_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(heads_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}
