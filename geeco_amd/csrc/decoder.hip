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

__device__ __forceinline__ void lstm_step_bwd_body(const LstmBwdBatch& q, int blk, float (*sA)[GEMM_BK * GEMM_LD],
                                                   float (*sB)[GEMM_BK * GEMM_LD]) {
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

__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const LstmBwdBatch q) {
  __shared__ float sA[2][GEMM_BK * GEMM_LD];
  __shared__ float sB[2][GEMM_BK * GEMM_LD];
  lstm_step_bwd_body(q, (int)blockIdx.x, sA, sB);
}

// slab sum of dX (fixed slab order, as gemm_reduce_kernel) + the state-concat scatter of the sums
__device__ __forceinline__ void lstm_step_bwd_finish_body(const LstmBwdBatch& q, int block) {
  const long long i = (long long)block * 256 + threadIdx.x;
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

__global__ __launch_bounds__(256) void lstm_step_bwd_finish_kernel(const LstmBwdBatch q) {
  lstm_step_bwd_finish_body(q, (int)blockIdx.x);
}

extern "C" int64_t geeco_lstm_step_bwd_ws_bytes(int N, int D, int H4) { return geeco_gemm_ws_bytes(N, D, H4); }

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
  float* dpred; // ws: [N][OT]  (the single-workgroup kernel keeps its packed head matrix behind it: [N * OT ...)
  float* lterm; // ws: [N][8] per-sample loss terms of every head (per-sample kernel -> finish role)
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

// =====================================================================================================
// fc1 + heads + losses per SAMPLE (round 5).  Everything from the LSTM output to d(loss)/d(h) is independent per sample
// (graph.py:229-259, 430-500): only the loss means and the weight / bias gradients sum over the batch.  Round 2-4 ran the
// whole tail in ONE workgroup (six dependent MFMA tile loops with a barrier between: 34 us of pure latency at N = 32, three
// MFLOP of work).  Here:
//   * heads_sample_kernel: one 1024-thread workgroup per sample, N workgroups side by side: fc1/kernel staged in LDS once
//     (read twice: forward and d(h)), every product a few hundred FMAs per thread on the vector ALU (a 1 x 128 row times a
//     128 x 128 matrix is no MFMA shape), partial sums folded through LDS / half-wave shuffles in a fixed order.  Writes the
//     predictions, the per-sample loss terms, a1, d(a1), d(pred) and d(h).
//     FUSE (one-step decoders, zero initial state: the goal model's dynimg branch): the same workgroup first sums the split-K
//     slabs of its sample's gate pre-activations and runs the gate math (lstm_gates_fwd_slabs_kernel's work, same slab
//     order), and at the end turns d(h) into the gate gradients dz (lstm_gates_bwd_kernel's work): two dependent launches
//     less around the heads.
//   * heads_finish_role: what sums over the batch -- d(fc1/kernel) = h^T d(a1) (row tiles), the head kernels' gradients,
//     the bias gradients and the loss means, each a sum over n in ascending order.  A handful of independent blocks that ride
//     at the end of another launch's grid (lstm_step_bwd_finish_kernel) or run as heads_finish_kernel.
// Shapes: H <= 128, Hfc in {64, 128} (the reference's defaults are 128 / 128, params.py:21-22); anything else takes the
// single-workgroup heads_loss_kernel above.
// =====================================================================================================
constexpr int HS_THREADS = 1024, HS_HMAX = 128, HS_FMAX = 128;
constexpr int HS_WP = HS_FMAX + 4;      // LDS row pitch of fc1/kernel: rows of two half-waves fall on different bank groups
constexpr size_t HS_LDS_BYTES = (size_t)(HS_HMAX * HS_WP + HS_THREADS * 4) * 4;

struct StepFuse {
  const float* part;        // split-K slabs of z = x Wx: [S][N][4H]
  int S;
  const float* bias;        // lstm_cell/bias [4H]
  float* z;                 // [N][4H] slab sums (kept: geeco_lstm_input_step_fwd writes them too)
  float* c; float* hout;    // [N][H]
  float* gates;             // [N][4H] activated gates i, j, f, o
  float* dz;                // backward: gate gradients [N][4H]
};

__device__ __forceinline__ int heads_head_of(const HeadsParams& p, int o) {
  int hd = 0;
#pragma unroll
  for (int k = 1; k < GEECO_MAX_HEADS; ++k)
    if (k < p.nheads && o >= p.off[k]) hd = k;
  return hd;
}

template <bool FUSE, int G>              // G = Hfc / 4: float4 column groups of fc1/kernel, 16 or 32
__global__ __launch_bounds__(HS_THREADS) void heads_sample_kernel(const HeadsParams p, const StepFuse sf) {
  const int tid = threadIdx.x, n = blockIdx.x;
  const int H = p.H, OT = p.OT, N = p.N;
  constexpr int F = 4 * G;
  constexpr int P = HS_THREADS / G;      // row parts (fc1 forward) / rows per pass (d(h))
  extern __shared__ __attribute__((aligned(16))) float hs_dyn[];           // HS_LDS_BYTES (more than the 64 KiB a static array may take)
  float* sW1 = hs_dyn;                                                        // [H][HS_WP] fc1/kernel
  float* sPart = hs_dyn + HS_HMAX * HS_WP;                                    // [P][F] partial sums of the fc1 forward
  __shared__ __attribute__((aligned(16))) float sH[HS_HMAX], sA1[HS_FMAX], sDA[HS_FMAX], sDH[HS_HMAX], sB1[HS_FMAX];
  __shared__ float sZ[FUSE ? 4 * HS_HMAX : 1];
  __shared__ float sPr[32], sDp[32], sHb[32], sTg[32];
  __shared__ int sHd[32], sHc[32];
  __shared__ float sWh[32 * HS_FMAX];     // the head kernels side by side, transposed: [o][f] (both uses walk f across lanes)
  // ---- P0: fc1/kernel -> LDS (issued first: independent of everything), small tables, the sample's LSTM output -----------
  f32x4 wv[4];
  const int W4 = H * G;                  // float4 of fc1/kernel (<= 4096 = 4 per thread)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e4 = tid + HS_THREADS * i;
    wv[i] = e4 < W4 ? reinterpret_cast<const f32x4*>(p.fc1_w)[e4] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // every global input of the sample is fetched here, up front: after this phase only LDS is read
  if (tid < OT) {
    const int hd = heads_head_of(p, tid), cc = tid - sel5(p.off, hd);
    sHd[tid] = hd;
    sHc[tid] = cc;
    sHb[tid] = sel5(p.hb, hd)[cc];
    if (sel5(p.kind, hd) == 0 || cc == 0) sTg[tid] = sel5(p.tgt, hd)[(long long)n * sel5(p.tstride, hd) + cc];
  }
  if (tid < F) sB1[tid] = p.fc1_b[tid];
  for (int e = tid; e < OT * F; e += HS_THREADS) {      // consecutive threads: consecutive o of one f (the variables are [f][size])
    const int f = e / OT, o = e - f * OT;
    const int hd = heads_head_of(p, o);
    sWh[o * HS_FMAX + f] = sel5(p.hw, hd)[(long long)f * sel5(p.size, hd) + (o - sel5(p.off, hd))];
  }
  [[maybe_unused]] float g_si = 0.f, g_tj = 0.f, g_sf = 0.f, g_so = 0.f, g_tc = 0.f;
  if (FUSE) {
    // gate pre-activations of this sample: column tid of [4H], slabs summed in slab order (as gemm_reduce_kernel)
    if (tid < 4 * H) {
      const long long MN = (long long)N * 4 * H;
      const float* src = sf.part + (long long)n * 4 * H + tid;
      float s = 0.f;
      int k = 0;
      for (; k + 16 <= sf.S; k += 16) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = src[(long long)(k + q) * MN];
#pragma unroll
        for (int q = 0; q < 16; ++q) s += v[q];
      }
      if (k < sf.S) {      // the remaining slabs (< 16) in ONE round of predicated loads; a skipped slab adds nothing
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = k + q < sf.S ? src[(long long)(k + q) * MN] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (k + q < sf.S) s += v[q];
      }
      sf.z[(long long)n * 4 * H + tid] = s;
      sZ[tid] = s + sf.bias[tid];
    }
  } else {
    if (tid < H) sH[tid] = p.h[(long long)n * H + tid];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e4 = tid + HS_THREADS * i;
    if (e4 < W4) *reinterpret_cast<f32x4*>(sW1 + (e4 / G) * HS_WP + (e4 % G) * 4) = wv[i];
  }
  __syncthreads();
  if (FUSE) {
    if (tid < H) {      // tf.nn.rnn_cell.LSTMCell from a zero state (graph.py:217-225): gate order i, j, f, o; forget_bias 1
      const int u = tid;
      g_si = sigmoidf_(sZ[u]); g_tj = tanhf(sZ[H + u]); g_sf = sigmoidf_(sZ[2 * H + u] + 1.0f); g_so = sigmoidf_(sZ[3 * H + u]);
      const float cn = g_sf * 0.f + g_si * g_tj;
      g_tc = tanhf(cn);
      const float hv = g_so * g_tc;
      const long long i = (long long)n * H + u;
      sf.c[i] = cn;
      sf.hout[i] = hv;
      float* gr = sf.gates + (long long)n * 4 * H;
      gr[u] = g_si; gr[H + u] = g_tj; gr[2 * H + u] = g_sf; gr[3 * H + u] = g_so;
      sH[u] = hv;
    }
    __syncthreads();
  }
  // ---- P1: a1 = relu(h W1 + b1)                                                              graph.py:229-230 ----------
  {
    const int pp = tid / G, g = tid - pp * G;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int hh = pp; hh < H; hh += P) acc += sH[hh] * *reinterpret_cast<const f32x4*>(sW1 + hh * HS_WP + 4 * g);
    *reinterpret_cast<f32x4*>(sPart + pp * F + 4 * g) = acc;
  }
  __syncthreads();
  if (tid < F) {
    float v = sB1[tid];
#pragma unroll 16
    for (int pp = 0; pp < P; ++pp) v += sPart[pp * F + tid];
    v = fmaxf(v, 0.f);
    sA1[tid] = v;
    if (p.backward) p.a1[(long long)n * F + tid] = v;
  }
  __syncthreads();
  // ---- P2: predictions = a1 Wh + bh (all heads side by side)                                   graph.py:233-259 ----------
  if (tid < OT * 32) {
    const int o = tid >> 5, l = tid & 31;
    float s = 0.f;
#pragma unroll
    for (int f = l; f < F; f += 32) s += sA1[f] * sWh[o * HS_FMAX + f];
#pragma unroll
    for (int m = 16; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (l == 0) {
      const float v = s + sHb[o];
      sPr[o] = v;
      p.preds[(long long)n * OT + o] = v;
    }
  }
  __syncthreads();
  // ---- P3: this sample's loss terms and d(loss)/d(pred)                 graph.py:430-500, estimator.py:206-239 ----------
  if (tid < p.nheads) {
    const int hd = tid;
    const int sz = sel5(p.size, hd), of = sel5(p.off, hd);
    const float* tg = sTg + of;
    const float wsc = sel5(p.weight, hd) * p.loss_scale;
    const float invn = 1.f / N;
    float l = 0.f;
    if (sel5(p.kind, hd) == 0) {
      const float c2 = 2.f / (float)(N * sz) * wsc;
      for (int cc = 0; cc < sz; ++cc) {
        const float d = sPr[of + cc] - tg[cc];
        l += d * d;
        sDp[of + cc] = d * c2;
      }
    } else {
      const int label = (int)rintf(tg[0]) + 1;             // estimator.py:213-215
      float mx = sPr[of];
      for (int cc = 1; cc < sz; ++cc) mx = fmaxf(mx, sPr[of + cc]);
      float se = 0.f;
      for (int cc = 0; cc < sz; ++cc) se += expf(sPr[of + cc] - mx);
      const bool lv = label >= 0 && label < sz;             // one_hot of an out-of-range label is all-zero
      if (lv) l = mx + logf(se) - sPr[of + label];
      for (int cc = 0; cc < sz; ++cc)
        sDp[of + cc] = lv ? (expf(sPr[of + cc] - mx) / se - (cc == label ? 1.f : 0.f)) * invn * wsc : 0.f;
    }
    p.lterm[(long long)n * 8 + hd] = l;
  }
  if (!p.backward) return;
  __syncthreads();
  // ---- P4: d(a1) = ReluGrad(dpred Wh^T) ----------------------------------------------------------------------------------
  if (tid < OT) p.dpred[(long long)n * OT + tid] = sDp[tid];
  if (tid < F) {
    float s = 0.f;
#pragma unroll 4
    for (int o = 0; o < OT; ++o) s += sDp[o] * sWh[o * HS_FMAX + tid];
    s = sA1[tid] > 0.f ? s : 0.f;
    sDA[tid] = s;
    p.da1[(long long)n * F + tid] = s;
  }
  __syncthreads();
  // ---- P5: d(h) = d(a1) W1^T: G lanes share a row of fc1/kernel, half-wave shuffle sum ----------------------------------------
  {
    const int r = tid / G, g = tid - r * G;
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDA + 4 * g);
    for (int hh = r; hh < H; hh += P) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(sW1 + hh * HS_WP + 4 * g);
      float s = d4.x * w4.x + d4.y * w4.y + d4.z * w4.z + d4.w * w4.w;
#pragma unroll
      for (int m = G >> 1; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
      if (g == 0) sDH[hh] = s;
    }
  }
  __syncthreads();
  if (tid < H) {
    const float dhv = sDH[tid];
    if (p.dh) p.dh[(long long)n * H + tid] = dhv;
    if (FUSE) {      // lstm_gates_bwd_kernel for the zero-state single step: dc = 0, c_prev = 0
      const int u = tid;
      const float dct = 0.f + dhv * g_so * (1.f - g_tc * g_tc);
      float* dr = sf.dz + (long long)n * 4 * H;
      dr[u] = dct * g_tj * g_si * (1.f - g_si);
      dr[H + u] = dct * g_si * (1.f - g_tj * g_tj);
      dr[2 * H + u] = dct * 0.f * g_sf * (1.f - g_sf);
      dr[3 * H + u] = dhv * g_tc * g_so * (1.f - g_so);
    }
  }
}

// ---- what sums over the batch: role blocks of 256 threads ---------------------------------------------------------------
// blocks [0, nW): rows of d(fc1/kernel) = h^T d(a1), 256 / G rows each; blocks [nW, nW + nWh): the head kernels' gradients, 256
// elements each; the last block: the bias gradients and the loss means.  Forward only: just the loss means (one block).
__host__ __device__ __forceinline__ int heads_finish_blocks(int H, int F, int OT, int backward) {
  return backward ? (H + 256 / (F >> 2) - 1) / (256 / (F >> 2)) + (F * OT + 255) / 256 + 1 : 1;
}

constexpr int HF_NC = 32;      // samples staged per chunk

// `lds`: HF_LDS_FLOATS floats of shared memory lent by the calling kernel (the GEMM tile buffers of lstm_step_bwd_heads_kernel: a
// block of that grid must not need more LDS than a tile block, or fewer of them fit on a CU)
constexpr int HF_LDS_FLOATS = HF_NC * HS_FMAX + HF_NC * 32;

__device__ __forceinline__ void heads_finish_role(const HeadsParams& p, const float* h, int role, float* lds) {
  const int tid = threadIdx.x;
  const int N = p.N, H = p.H, F = p.Hfc, OT = p.OT;
  const int G = F >> 2, R = 256 / G;
  const int nW = (H + R - 1) / R;
  // Operands come through LDS in chunks of HF_NC samples, fetched with independent coalesced loads (all in flight at once): a
  // thread that walks n with dependent global loads pays a memory latency per sample (the first form of these roles: 36 us).
  float* sA = lds;                             // a chunk of d(a1) or a1: [n][F]
  float* sB = lds + HF_NC * HS_FMAX;           // a chunk of h rows [n][R] or of dpred [n][OT]
  // (rows past the chunk's end are zero-filled and every loop below runs all HF_NC rows: constant trip counts, so the LDS reads of
  // consecutive samples are issued together instead of one latency per sample)
  auto stage = [&](const float* src, int n0, int nc) {                    // [nc][F] floats (F % 4 == 0, rows contiguous)
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (long long)n0 * F);
    for (int e = tid; e < HF_NC * G; e += 256) reinterpret_cast<f32x4*>(sA)[e] = e < nc * G ? s4[e] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if (p.backward && role < nW) {            // d(fc1/kernel)[hh][4g..] = sum_n h[n][hh] d(a1)[n][4g..]
    const int r = tid / G, g = tid - r * G, hh = role * R + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += HF_NC) {
      const int nc = N - n0 < HF_NC ? N - n0 : HF_NC;
      if (n0) __syncthreads();
      stage(p.da1, n0, nc);
      for (int e = tid; e < HF_NC * R; e += 256) {
        const int nn = e / R, rr = e - nn * R;
        sB[e] = (nn < nc && role * R + rr < H) ? h[(long long)(n0 + nn) * H + role * R + rr] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int nn = 0; nn < HF_NC; ++nn) acc += sB[nn * R + r] * reinterpret_cast<const f32x4*>(sA + nn * F)[g];
    }
    if (hh < H) reinterpret_cast<f32x4*>(p.d_fc1_w + (long long)hh * F)[g] = acc;
    return;
  }
  const int nWh = (F * OT + 255) / 256;
  if (p.backward && role < nW + nWh) {      // head kernel gradients [f][c] = sum_n a1[n][f] dpred[n][o], one element per thread
    const int e = (role - nW) * 256 + tid;
    const bool live = e < F * OT;
    const int f = live ? e / OT : 0, o = live ? e - f * OT : 0;
    float acc = 0.f;
    for (int n0 = 0; n0 < N; n0 += HF_NC) {
      const int nc = N - n0 < HF_NC ? N - n0 : HF_NC;
      if (n0) __syncthreads();
      stage(p.a1, n0, nc);
      for (int i = tid; i < HF_NC * OT; i += 256) sB[i] = i < nc * OT ? p.dpred[(long long)n0 * OT + i] : 0.f;
      __syncthreads();
#pragma unroll 8
      for (int nn = 0; nn < HF_NC; ++nn) acc += sA[nn * F + f] * sB[nn * OT + o];
    }
    if (live) {
      const int hd = heads_head_of(p, o);
      sel5(p.dhw, hd)[(long long)f * sel5(p.size, hd) + (o - sel5(p.off, hd))] = acc;
    }
    return;
  }
  // bias gradients and loss means
  float accb = 0.f, acco = 0.f, accl = 0.f;      // thread f < F: d(fc1/bias)[f]; thread o < OT: d(head bias)[o]; thread hd: loss sum
  for (int n0 = 0; n0 < N; n0 += HF_NC) {
    const int nc = N - n0 < HF_NC ? N - n0 : HF_NC;
    if (n0) __syncthreads();
    if (p.backward) {
      stage(p.da1, n0, nc);
      for (int e = tid; e < HF_NC * OT; e += 256) sB[e] = e < nc * OT ? p.dpred[(long long)n0 * OT + e] : 0.f;
    }
    __shared__ float sL[HF_NC * 8];
    for (int e = tid; e < HF_NC * 8; e += 256) sL[e] = (e < nc * 8 && (e & 7) < p.nheads) ? p.lterm[(long long)n0 * 8 + e] : 0.f;
    __syncthreads();
    if (p.backward) {
      if (tid < F) {
#pragma unroll 8
        for (int nn = 0; nn < HF_NC; ++nn) accb += sA[nn * F + tid];
      }
      if (tid < OT) {
#pragma unroll 8
        for (int nn = 0; nn < HF_NC; ++nn) acco += sB[nn * OT + tid];
      }
    }
    if (tid < p.nheads) {
#pragma unroll 8
      for (int nn = 0; nn < HF_NC; ++nn) accl += sL[nn * 8 + tid];
    }
  }
  if (p.backward) {
    if (tid < F) p.d_fc1_b[tid] = accb;
    if (tid < OT) {
      const int hd = heads_head_of(p, tid);
      sel5(p.dhb, hd)[tid - sel5(p.off, hd)] = acco;
    }
  }
  __shared__ float s_l[GEECO_MAX_HEADS];
  if (tid < p.nheads) {
    const float sum = accl * (sel5(p.kind, tid) == 0 ? 1.f / (float)(N * sel5(p.size, tid)) : 1.f / N);
    p.losses[1 + tid] = sum;
    s_l[tid] = sum;
  }
  __syncthreads();
  if (tid == 0) {
    float total = 0.f;
    for (int hd = 0; hd < p.nheads; ++hd) total += sel5(p.weight, hd) * s_l[hd];
    p.losses[0] = total;
  }
}

__global__ __launch_bounds__(256) void heads_finish_kernel(const HeadsParams p, const float* h) {
  __shared__ __attribute__((aligned(16))) float lds[HF_LDS_FLOATS];
  heads_finish_role(p, h, (int)blockIdx.x, lds);      // (forward only: one block, which falls through to the loss means)
}

extern "C" int64_t geeco_heads_ws_bytes(int N, int H, int Hfc) {
  (void)H;
  return ((int64_t)2 * N * Hfc + (int64_t)N * 32 + (int64_t)32 * Hfc + (int64_t)N * 8) * 4;
}

struct HeadsPending {       // what geeco_heads_finish (include/geeco_hip.h) holds
  HeadsParams p;
  const float* h;
  int valid;
};
static_assert(sizeof(HeadsPending) <= sizeof(geeco_heads_finish), "geeco_heads_finish is too small for HeadsPending");

// the per-sample kernel serves these shapes; the rest takes the single-workgroup kernel
static bool heads_sample_shapes(int H, int Hfc, int OT) {
  return H >= 1 && H <= HS_HMAX && (Hfc == 64 || Hfc == 128) && OT <= 32;
}

static int heads_fill(HeadsParams* pp, const float* h, const float* fc1_w, const float* fc1_b, int nheads,
                      const float* const* heads_w, const float* const* heads_b, const int* head_size, const int* head_kind,
                      const float* head_weight, const float* const* targets, const int64_t* target_stride, float loss_scale,
                      int N, int H, int Hfc, float* preds, float* losses, int backward, float* dh, float* d_fc1_w,
                      float* d_fc1_b, float* const* d_heads_w, float* const* d_heads_b, float* ws) {
  GEECO_CHECK_ARG(fc1_w && fc1_b && heads_w && heads_b && head_size && head_kind && head_weight && targets &&
                      target_stride && preds && losses && ws, "heads_loss: null pointer");
  GEECO_CHECK_ARG(nheads >= 1 && nheads <= GEECO_MAX_HEADS, "heads_loss: nheads=%d outside 1..%d", nheads, GEECO_MAX_HEADS);
  GEECO_CHECK_ARG(N >= 1 && N <= 4096 && H >= 1 && Hfc >= 1, "heads_loss: bad dims");
  GEECO_CHECK_ARG(!backward || (d_fc1_w && d_fc1_b && d_heads_w && d_heads_b), "heads_loss: null gradient pointer");
  HeadsParams& p = *pp;
  p = HeadsParams{};
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
  p.lterm = ws + 2ll * N * Hfc + (long long)N * 32 + 32ll * Hfc;
  return 0;
}

template <bool FUSE, int G>
static int launch_heads_sample_g(const HeadsParams& p, const StepFuse& sf, hipStream_t stream) {
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&heads_sample_kernel<FUSE, G>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)HS_LDS_BYTES);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", HS_LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  geeco_note_kernel("heads_sample_kernel<%s>", FUSE ? "true" : "false");
  hipLaunchKernelGGL((heads_sample_kernel<FUSE, G>), dim3((unsigned)p.N), dim3(HS_THREADS), HS_LDS_BYTES, stream, p, sf);
  GEECO_LAUNCH_CHECK();
  return 0;
}

template <bool FUSE>
static int launch_heads_sample(const HeadsParams& p, const StepFuse& sf, hipStream_t stream) {
  return p.Hfc == 128 ? launch_heads_sample_g<FUSE, 32>(p, sf, stream) : launch_heads_sample_g<FUSE, 16>(p, sf, stream);
}

extern "C" int geeco_heads_loss_fwd_bwd(const float* h, const float* fc1_w, const float* fc1_b, int nheads,
                                        const float* const* heads_w, const float* const* heads_b,
                                        const int* head_size, const int* head_kind, const float* head_weight,
                                        const float* const* targets, const int64_t* target_stride, float loss_scale,
                                        int N, int H, int Hfc, float* preds, float* losses, int backward, float* dh,
                                        float* d_fc1_w, float* d_fc1_b, float* const* d_heads_w,
                                        float* const* d_heads_b, float* ws, void* stream) {
  GEECO_CHECK_ARG(h && (!backward || dh), "heads_loss: null pointer");
  HeadsParams p;
  if (int rc = heads_fill(&p, h, fc1_w, fc1_b, nheads, heads_w, heads_b, head_size, head_kind, head_weight, targets, target_stride,
                          loss_scale, N, H, Hfc, preds, losses, backward, dh, d_fc1_w, d_fc1_b, d_heads_w, d_heads_b, ws))
    return rc;
  hipStream_t s = (hipStream_t)stream;
  if (heads_sample_shapes(H, Hfc, p.OT)) {
    if (int rc = launch_heads_sample<false>(p, StepFuse{}, s)) return rc;
    geeco_note_kernel("heads_finish_kernel");
    hipLaunchKernelGGL(heads_finish_kernel, dim3((unsigned)heads_finish_blocks(H, Hfc, p.OT, backward)), dim3(256), 0, s, p, h);
  } else {
    geeco_note_kernel("heads_loss_kernel");
    hipLaunchKernelGGL(heads_loss_kernel, dim3(1), dim3(1024), 0, s, p);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}

// One-step decoder (zero initial state): gate GEMM, then ONE per-sample launch for the gate math, fc1, the heads, the losses and --
// with `backward` -- everything back to the gate gradients dz.  `pending` non-null: the batch sums of the heads' backward (and
// the loss means) are left for geeco_lstm_step_bwd to run at the end of its second grid; null: heads_finish_kernel runs here.
extern "C" int geeco_lstm_step_heads_fwd_bwd(const float* x, int64_t ldx, const float* wx, int64_t ldw, const float* bias,
                                             float* z, float* c, float* h, float* gates, int N, int H, int D, void* gemm_ws,
                                             const float* fc1_w, const float* fc1_b, int nheads, const float* const* heads_w,
                                             const float* const* heads_b, const int* head_size, const int* head_kind,
                                             const float* head_weight, const float* const* targets,
                                             const int64_t* target_stride, float loss_scale, int Hfc, float* preds,
                                             float* losses, int backward, float* dz, float* d_fc1_w, float* d_fc1_b,
                                             float* const* d_heads_w, float* const* d_heads_b, float* heads_ws,
                                             geeco_heads_finish* pending, void* stream) {
  GEECO_CHECK_ARG(x && wx && bias && z && c && h && gates && (!backward || dz), "lstm_step_heads: null pointer");
  GEECO_CHECK_ARG(N >= 1 && H >= 1 && D >= 1 && ldx >= D && ldw >= 4 * (int64_t)H, "lstm_step_heads: bad dims");
  if (pending) reinterpret_cast<HeadsPending*>(pending)->valid = 0;
  HeadsPending hp;
  if (int rc = heads_fill(&hp.p, h, fc1_w, fc1_b, nheads, heads_w, heads_b, head_size, head_kind, head_weight, targets,
                          target_stride, loss_scale, N, H, Hfc, preds, losses, backward, nullptr, d_fc1_w, d_fc1_b, d_heads_w,
                          d_heads_b, heads_ws))
    return rc;
  if (!heads_sample_shapes(H, Hfc, hp.p.OT)) return GEECO_ENOSUP;      // nothing launched: the caller runs the separate entry points
  GemmParams g = {};
  g.A = x; g.B = wx; g.C = z; g.part = (float*)gemm_ws; g.lda = ldx; g.ldb = ldw; g.ldc = 4 * H;
  g.M = N; g.N = 4 * H; g.K = D;
  gemm_plan(g.M, g.N, g.K, &g.S, &g.k_per_split);
  GEECO_CHECK_ARG(g.S == 1 || gemm_ws, "lstm_step_heads: workspace required for split-K (geeco_gemm_ws_bytes(N, 4H, D))");
  hipStream_t s = (hipStream_t)stream;
  geeco_note_kernel("gemm_f32_kernel");
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)cdiv(g.N, 64), (unsigned)cdiv(g.M, 64), (unsigned)g.S), dim3(256), 0, s, g);
  GEECO_LAUNCH_CHECK();
  StepFuse sf = {};
  sf.part = g.S > 1 ? (const float*)g.part : (const float*)z;      // unsplit: the GEMM wrote z itself (one "slab")
  sf.S = g.S; sf.bias = bias; sf.z = z; sf.c = c; sf.hout = h; sf.gates = gates; sf.dz = dz;
  if (int rc = launch_heads_sample<true>(hp.p, sf, s)) return rc;
  if (pending) {
    hp.h = h;
    hp.valid = 1;
    *reinterpret_cast<HeadsPending*>(pending) = hp;
  } else {
    geeco_note_kernel("heads_finish_kernel");
    hipLaunchKernelGGL(heads_finish_kernel, dim3((unsigned)heads_finish_blocks(H, Hfc, hp.p.OT, backward)), dim3(256), 0, s, hp.p, (const float*)h);
    GEECO_LAUNCH_CHECK();
  }
  return 0;
}

// Grid 1 of geeco_lstm_step_bwd with the heads' pending batch sums as its FIRST blocks: a handful of independent ~5 us blocks
// beside the tiles of dWx / dX (13 us), in a launch that exists anyway (at the end of the SECOND grid, whose blocks take ~1 us,
// they set that launch's length: measured 15.7 instead of 5.0 us).
__global__ __launch_bounds__(256) void lstm_step_bwd_heads_kernel(const LstmBwdBatch q, const HeadsParams hp, const float* h, int hb) {
  static_assert(4 * GEMM_BK * GEMM_LD >= HF_LDS_FLOATS, "the finish roles borrow the tile buffers");
  __shared__ __attribute__((aligned(16))) float tiles[4][GEMM_BK * GEMM_LD];      // sA[2], sB[2]
  if ((int)blockIdx.x < hb)
    heads_finish_role(hp, h, (int)blockIdx.x, &tiles[0][0]);
  else
    lstm_step_bwd_body(q, (int)blockIdx.x - hb, &tiles[0], &tiles[2]);
}

extern "C" int geeco_lstm_step_bwd(const float* x, int64_t ldx, const float* dz, int64_t ldz, const float* wx, int64_t ldw,
                                   float* dwx, int64_t lddw, float* db, float* dx, int64_t lddx, int N, int D, int H4,
                                   const float* const* feats_fwd, float* const* dfeats, const int* feat_ch, int nfeat,
                                   int jnt_pos, int J, int cells, void* ws, const geeco_heads_finish* pending, void* stream) {
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
  const HeadsPending* hp = reinterpret_cast<const HeadsPending*>(pending);
  if (hp && hp->valid) {
    const int hb = heads_finish_blocks(hp->p.H, hp->p.Hfc, hp->p.OT, hp->p.backward);
    geeco_note_kernel("lstm_step_bwd_heads_kernel");
    hipLaunchKernelGGL(lstm_step_bwd_heads_kernel, dim3((unsigned)(hb + blocks)), dim3(256), 0, (hipStream_t)stream, q, hp->p, hp->h, hb);
  } else {
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, q);
  }
  GEECO_LAUNCH_CHECK();
  if (q.b.S > 1) {
    hipLaunchKernelGGL(lstm_step_bwd_finish_kernel, dim3((unsigned)cdiv64((long long)N * D, 256)), dim3(256), 0,
                       (hipStream_t)stream, q);
    GEECO_LAUNCH_CHECK();
  }
  return 0;
}

