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
#include "conv_wgrad_body.h"

template <int BR, int BC, int MK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  __shared__ __attribute__((aligned(16))) float smem[conv_wgrad_smem_floats<BR, BC, MK>()];
  conv_wgrad_body<BR, BC, MK>(p, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x, smem);
}

// TWO independent filter-gradient problems of the same tile shape as ONE grid (round 4: conv7's and conv8's, both ready once
// conv8's input gradient exists; each alone is 432 blocks on 256 CUs for 30 / 12 us: together the short blocks fill the long
// ones' tail and one launch boundary goes).  Bitwise the single launches.
template <int BR, int BC, int MK>
__global__ __launch_bounds__(256) void conv_wgrad_pair_kernel(const WgradPairParams pp) {
  __shared__ __attribute__((aligned(16))) float smem[conv_wgrad_smem_floats<BR, BC, MK>()];
  conv_wgrad_pair_body<BR, BC, MK>(pp, (int)blockIdx.x, smem);
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
  // optional rider (geeco_slab_reduce_batch_prepare): the optimiser's per-step scalars, geeco_adam_prepare's work, as block
  // `prep_block` of this grid -- it depends on nothing here and only has to precede the Adam launch: one dependent launch less
  int prep_block;                     // -1: none
  long long* step;
  float* scal;
  float lr, b1, b2;
};

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const SlabReduceBatch b) {
  __shared__ f32x4 sred[4][64];
  const int bid = blockIdx.x;
  if (bid == b.prep_block) {          // tf.train.AdamOptimizer's lr_t (misc.hip: adam_prepare_kernel)
    if (threadIdx.x == 0) {
      const long long t = *b.step + 1;
      *b.step = t;
      const double b1t = pow((double)b.b1, (double)t), b2t = pow((double)b.b2, (double)t);
      b.scal[0] = (float)((double)b.lr * sqrt(1.0 - b2t) / (1.0 - b1t));
    }
    return;
  }
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

static int slab_reduce_batch_impl(const geeco_slab_reduce* items, int n, int64_t* step, float lr, float b1, float b2, float* scal,
                                  void* stream);

extern "C" int geeco_slab_reduce_batch(const geeco_slab_reduce* items, int n, void* stream) {
  return slab_reduce_batch_impl(items, n, nullptr, 0.f, 0.f, 0.f, nullptr, stream);
}

extern "C" int geeco_slab_reduce_batch_prepare(const geeco_slab_reduce* items, int n, int64_t* global_step_dev, float lr, float beta1,
                                               float beta2, float* scal_dev, void* stream) {
  GEECO_CHECK_ARG(global_step_dev && scal_dev, "slab_reduce_batch_prepare: null pointer");
  return slab_reduce_batch_impl(items, n, global_step_dev, lr, beta1, beta2, scal_dev, stream);
}

static int slab_reduce_batch_impl(const geeco_slab_reduce* items, int n, int64_t* step, float lr, float b1, float b2, float* scal,
                                  void* stream) {
  GEECO_CHECK_ARG(n >= 0 && n <= GEECO_SLAB_REDUCE_MAX && (items || n == 0), "slab_reduce_batch: n = %d (0..%d)", n,
                  GEECO_SLAB_REDUCE_MAX);
  SlabReduceBatch b = {};
  b.prep_block = -1;
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
  if (step) {                           // the rider takes the block behind the last reduce block (a launch of its own if nothing is pending)
    b.prep_block = (int)blocks++;
    b.step = (long long*)step; b.scal = scal; b.lr = lr; b.b1 = b1; b.b2 = b2;
  }
  if (blocks == 0) return 0;
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

void geeco_wgrad_plan(int groups, int N, int H, int W, int Cin, int Cout, int stride, WgradParams* p, int* bc) {
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
  geeco_wgrad_plan(groups, N, H, W, Cin, Cout, stride, &p, &bc);
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
  geeco_wgrad_plan(groups, N, H, W, Cin, Cout, stride, &p, &BC);
  p.x = x; p.dz = dz; p.part = (float*)ws; p.gs_x = gs_x; p.gs_dz = gs_dz;
  p.dw = dw; p.db = db; p.gs_dw = gs_dw; p.gs_db = gs_db;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)p.S, (unsigned)(p.row_tiles * p.col_tiles), (unsigned)groups);
  switch (BC) {
#ifdef GEECO_DEV_KERNELS      // GEECO_WGRAD_BIG: 128-row / 128-column tiles (measured ~1.5 % of the step slower)
    case 128128: geeco_note_kernel("conv_wgrad_kernel<128, 128, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 16>), grid, dim3(256), 0, s, p); break;
    case 128064: geeco_note_kernel("conv_wgrad_kernel<128, 64, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 16>), grid, dim3(256), 0, s, p); break;
    case 64128: geeco_note_kernel("conv_wgrad_kernel<64, 128, 16>"); hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 16>), grid, dim3(256), 0, s, p); break;
#endif
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
int geeco_wgrad_pair_fill(WgradPairParams* pp, const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0,
                          int64_t gs_dz0, int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                          const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                          int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1, int groups,
                          int stride, long long* blocks) {
  const bool two = x1 != nullptr;        // x1 == NULL: ONE problem (the heterogeneous launch with a single filter gradient)
  GEECO_CHECK_ARG(x0 && dz0 && dw0 && ws0 && (!two || (dz1 && dw1 && ws1)), "conv3x3_wgrad_pair: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N0 >= 1 && H0 >= 1 && W0 >= 1 && (!two || (N1 >= 1 && H1 >= 1 && W1 >= 1)),
                  "conv3x3_wgrad_pair: bad dims");
  if (!(stride == 2 && Cin0 == 256 && Cout0 % 64 == 0 && Cout0 >= 64 && (!two || (Cin1 == 256 && Cout1 % 64 == 0 && Cout1 >= 64)))) {
    geeco_set_error("conv3x3_wgrad_pair: shapes outside the paired kernel (Cin = 256, Cout %% 64 == 0, stride 2)");
    return GEECO_ENOSUP;
  }
  int bc0, bc1 = 64064;
  geeco_wgrad_plan(groups, N0, H0, W0, Cin0, Cout0, stride, &pp->q0, &bc0);
  if (two) geeco_wgrad_plan(groups, N1, H1, W1, Cin1, Cout1, stride, &pp->q1, &bc1);
  if (bc0 != 64064 || bc1 != 64064) {
    geeco_set_error("conv3x3_wgrad_pair: the two problems do not share the 64 x 64 tile kernel");
    return GEECO_ENOSUP;
  }
  pp->q0.x = x0; pp->q0.dz = dz0; pp->q0.part = (float*)ws0; pp->q0.gs_x = gs_x0; pp->q0.gs_dz = gs_dz0;
  pp->q0.dw = dw0; pp->q0.db = db0; pp->q0.gs_dw = gs_dw0; pp->q0.gs_db = gs_db0;
  pp->q1.x = x1; pp->q1.dz = dz1; pp->q1.part = (float*)ws1; pp->q1.gs_x = gs_x1; pp->q1.gs_dz = gs_dz1;
  pp->q1.dw = dw1; pp->q1.db = db1; pp->q1.gs_dw = gs_dw1; pp->q1.gs_db = gs_db1;
  const long long b0 = (long long)pp->q0.S * pp->q0.row_tiles * pp->q0.col_tiles * groups;
  const long long b1 = two ? (long long)pp->q1.S * pp->q1.row_tiles * pp->q1.col_tiles * groups : 0;
  pp->blocks0 = (int)b0;
  *blocks = b0 + b1;
  return 0;
}

// the two problems' slab sums (launched, or recorded into pending2[0..1]; S == 1: the kernel wrote the gradient itself)
int geeco_wgrad_pair_finish(const WgradPairParams& pp, int groups, hipStream_t s, geeco_slab_reduce* pending2) {
  const WgradParams* q[2] = {&pp.q0, &pp.q1};
  for (int i = 0; i < 2; ++i) {
    if (pending2) {
      geeco_slab_reduce none = {};
      pending2[i] = none;
    }
    if (q[i]->x && q[i]->S > 1) {
      if (pending2) geeco_set_pending_reduce(&pending2[i]);
      geeco_launch_wgrad_reduce(q[i]->part, q[i]->dw, q[i]->db, q[i]->gs_dw, q[i]->gs_db, q[i]->S,
                                (long long)q[i]->Krows * q[i]->Cout, q[i]->Cout, groups, s);
      if (pending2) geeco_set_pending_reduce(nullptr);
      GEECO_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int geeco_conv3x3_wgrad_pair(const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0, int64_t gs_dz0,
                                        int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                                        const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                                        int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1,
                                        int groups, int stride, void* stream, geeco_slab_reduce* pending2) {
  GEECO_CHECK_ARG(x1, "conv3x3_wgrad_pair: two problems");
  WgradPairParams pp = {};
  long long blocks = 0;
  if (int rc = geeco_wgrad_pair_fill(&pp, x0, dz0, dw0, db0, gs_x0, gs_dz0, gs_dw0, gs_db0, N0, H0, W0, Cin0, Cout0, ws0, x1, dz1,
                                     dw1, db1, gs_x1, gs_dz1, gs_dw1, gs_db1, N1, H1, W1, Cin1, Cout1, ws1, groups, stride, &blocks))
    return rc;
  hipStream_t s = (hipStream_t)stream;
  geeco_note_kernel("conv_wgrad_pair_kernel<64, 64, 32>");
  hipLaunchKernelGGL((conv_wgrad_pair_kernel<64, 64, 32>), dim3((unsigned)blocks), dim3(256), 0, s, pp);
  GEECO_LAUNCH_CHECK();
  return geeco_wgrad_pair_finish(pp, groups, s, pending2);
}
