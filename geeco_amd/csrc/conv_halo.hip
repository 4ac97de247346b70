// LDS-halo kernels for the two big-spatial / small-channel layers of the encoder:
//   conv1: 3x3, stride 1,  4 -> 32 channels (RGB padded to 4)     graph.py:76-80
//   conv2: 3x3, stride 2, 32 -> 48 channels                        graph.py:81-85
// (reference src/models/e2evmc/graph.py; backward = autodiff via estimator.py:243-244).
//
// Why a second kernel family: at Cout = 32/48 the gather-GEMM of conv_gemm.hip moves
// 9*Cin*4 bytes of gathered input per output pixel for 2*9*Cin*Cout FLOP = Cout/2 FLOP per byte
// (24 FLOP/B for conv2): the L1/TA load path, not the MFMA pipe, sets its speed (PMC: 55 % MFMA
// busy, 1.6-3x HBM over-fetch).  Here a block stages the input HALO of its output tile in LDS once
// and every tap reads its fragments from there (2.25x fewer bytes for stride 2, 9x for stride 1),
// the kernel weights stay resident in LDS for the block's lifetime (persistent blocks walk the
// tiles), and the next tile's halo is prefetched into registers behind the current tile's MFMAs.
//
// MFMA: v_mfma_f32_16x16x4_f32, roles as in conv_gemm.hip (row i = output channel, column j =
// pixel) so every lane owns 4 consecutive NHWC channels of one pixel.
#include "geeco_common.h"
#include <stdlib.h>

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the
// vector-memory counter (vmcnt(0)), which would expose the latency of the epilogue's global stores
// and of the next tile's prefetch loads once per tile (cdna_hip_programming.md, "Pipelining across
// barriers").  The "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// conv2-type forward: stride 2, CIN % 16 == 0, COUT % 16 == 0, tile = 4 x 16 output pixels.
// LDS: W as [tap][cq][co][4] (b128 B-fragments), halo as [cq][hy][parity][hx/2] float4 planes
// (b128 A-fragments: consecutive output columns are consecutive 16-byte slots).
// ------------------------------------------------------------------------------------------------
struct HaloFwdParams {
  const float* x;
  const float* w;      // HWIO [G][9][CIN][COUT]
  const float* bias;
  float* y;
  long long gs_x, gs_w, gs_b, gs_y;
  int N, H, W, Ho, Wo;
  int tiles_x, tiles_y;      // tiles per image
  long long ntiles;          // G*N*tiles_y*tiles_x
  int tiles_per_group;       // N*tiles_y*tiles_x
  int relu;
  int debug;   // ablation switches (GEECO_HALO_DEBUG): 1 = skip MFMAs, 2 = skip halo loads, 4 = skip output stores
};

template <int CIN, int COUT>
__global__ __launch_bounds__(512, 2) void conv_s2_halo_fwd_kernel(const HaloFwdParams p) {
  constexpr int NT = 512;                             // 8 waves = 4 output rows x 2 halves of the channel (K) range
  constexpr int TH = 4, TW = 16;
  constexpr int CQ = CIN / 4;
  constexpr int HY = 2 * TH + 1, HX2 = TW + 1;        // halo rows; columns per parity plane
  constexpr int PLANE = HY * 2 * HX2;                 // float4 per channel-quad plane (306)
  constexpr int HALO_F4 = CQ * PLANE;
  constexpr int NPIX = HY * (2 * TW + 1);             // 9 * 33 halo pixels
  constexpr int NLOAD = (NPIX * CQ + NT - 1) / NT;    // float4 loads per thread per tile
  constexpr int W_F4 = 9 * CQ * COUT;
  constexpr int TI = COUT / 16;
  constexpr int KB = CIN / 16;
  constexpr int KBW = KB / 2;                         // 16-channel blocks per wave
  constexpr int NIT = 9 * KBW;
  constexpr int RED_F4 = 4 * TI * 64;                 // partial accumulators of waves 4..7
  static_assert(KB % 2 == 0 && NIT >= NLOAD + 2, "K split / write interleave");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                              // 2 halo buffers
  f32x4* sR = sH + 2 * HALO_F4;                       // 2 reduction buffers

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int strip = wid & 3, khalf = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // contiguous tile range per block (neighbouring tiles share halo rows in L2; no divisions in the loop)
  const long long per = (p.ntiles + gridDim.x - 1) / gridDim.x;
  long long tile = (long long)blockIdx.x * per;
  long long tend = tile + per < p.ntiles ? tile + per : p.ntiles;
  if (tile >= tend) return;
  int g, n, ty, tx;
  {
    g = (int)(tile / p.tiles_per_group);
    int rem = (int)(tile - (long long)g * p.tiles_per_group);
    int per_img = p.tiles_x * p.tiles_y;
    n = rem / per_img;
    rem -= n * per_img;
    ty = rem / p.tiles_x;
    tx = rem - ty * p.tiles_x;
  }
  auto advance = [&](int& g_, int& n_, int& ty_, int& tx_) {
    if (++tx_ == p.tiles_x) {
      tx_ = 0;
      if (++ty_ == p.tiles_y) {
        ty_ = 0;
        if (++n_ == p.N) {
          n_ = 0;
          ++g_;
        }
      }
    }
  };

  // per-thread halo slots: idx -> (pixel, cq); pixel -> (hy, hx)
  int l_off[NLOAD], l_src[NLOAD];
  short l_hy[NLOAD], l_hx[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    int idx = tid + NT * i;
    int pix = idx / CQ, cq = idx - pix * CQ;
    int hy = pix / (2 * TW + 1), hx = pix - hy * (2 * TW + 1);
    l_hy[i] = (short)hy; l_hx[i] = (short)hx;
    l_off[i] = (pix < NPIX) ? cq * PLANE + (hy * 2 + (hx & 1)) * HX2 + (hx >> 1) : -1;
    l_src[i] = (hy * p.W + hx) * CIN + cq * 4;
  }
  f32x4 stage[NLOAD];

  auto load_halo = [&](int g_, int n_, int ty_, int tx_) {
    const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;      // TF SAME, stride 2, even input: pad_before = 0
    const float* xg = p.x + (long long)g_ * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * CIN;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      bool v = l_off[i] >= 0 && iy0 + l_hy[i] < p.H && ix0 + l_hx[i] < p.W;
      stage[i] = v ? *reinterpret_cast<const f32x4*>(xg + l_src[i]) : zero4;
    }
  };
  auto load_weights = [&](int g_) {
    const float* wg = p.w + (long long)g_ * p.gs_w;
    // HWIO [tap][c][co] -> LDS [tap][c/4][co][c%4]
    for (int e = tid; e < 9 * CIN * COUT; e += NT) {
      int co = e % COUT;
      int tc = e / COUT;               // tap*CIN + c
      int c = tc % CIN, tap = tc / CIN;
      smem[((tap * CQ + (c >> 2)) * COUT + co) * 4 + (c & 3)] = wg[e];
    }
  };

  load_halo(g, n, ty, tx);
  load_weights(g);
  int g_w = g;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i)
    if (l_off[i] >= 0) sH[l_off[i]] = stage[i];
  __syncthreads();
  f32x4 bias_r[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g * p.gs_b + i * 16 + 4 * q);

  // lane r = output column of row `strip`; this wave sums channels [khalf*CIN/2, (khalf+1)*CIN/2)
  const int a_lane = (khalf * KBW * 4 + q) * PLANE + (4 * strip) * HX2 + r;
  const f32x4* hB = sW + (khalf * KBW * 4 + q) * COUT + r;
  int buf = 0;
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) {
      advance(g2, n2, ty2, tx2);
      if (!(p.debug & 2)) load_halo(g2, n2, ty2, tx2);
    }
    const bool reload_w = more && g2 != g_w;
    f32x4 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[i] = zero4;
    const f32x4* hA = sH + buf * HALO_F4 + a_lane;
    f32x4* hN = sH + (buf ^ 1) * HALO_F4;
    // software-pipelined fragment reads; the next tile's halo is written to the other LDS buffer one
    // 16-byte store per MFMA group in the second half of the loop (its global loads were issued above)
    f32x4 a_cur, b_cur[TI], a_nxt, b_nxt[TI];
    auto frag = [&](int it, f32x4& a, f32x4 (&b)[TI]) {
      const int tap = it / KBW, kb = it - tap * KBW;
      const int ky = tap / 3, kx = tap - ky * 3;
      a = hA[kb * 4 * PLANE + (ky * 2 + (kx & 1)) * HX2 + (kx >> 1)];
#pragma unroll
      for (int i = 0; i < TI; ++i) b[i] = hB[(tap * CQ + kb * 4) * COUT + i * 16];
    };
    frag(0, a_cur, b_cur);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + 1 < NIT) frag(it + 1, a_nxt, b_nxt);
      if (more && it >= NIT - NLOAD) {
        const int j = it - (NIT - NLOAD);
        if (l_off[j] >= 0) hN[l_off[j]] = stage[j];
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads / staging store ABOVE this group's MFMAs
      if (!(p.debug & 1)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TI; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_cur[i][s], a_cur[s], acc[i], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[i] += b_cur[i] * a_cur.x;
      }
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
#pragma unroll
      for (int i = 0; i < TI; ++i) b_cur[i] = b_nxt[i];
    }
    f32x4* red = sR + (int)(tile & 1) * RED_F4;
    if (khalf == 1) {
#pragma unroll
      for (int i = 0; i < TI; ++i) red[(strip * TI + i) * 64 + lane] = acc[i];
    }
    lds_barrier();   // partial sums visible; next halo complete; everyone is done with buf and sW
    if (khalf == 0) {
      // epilogue: pixel (oy, ox) = (ty*4 + strip, tx*16 + r); channels 16 i + 4 q .. +3
      const int oy = ty * TH + strip, ox = tx * TW + r;
      const bool ok = oy < p.Ho && ox < p.Wo;
      float* yo = p.y + (long long)g * p.gs_y + (((long long)n * p.Ho + oy) * p.Wo + ox) * COUT;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        f32x4 v = acc[i] + red[(strip * TI + i) * 64 + lane] + bias_r[i];
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (ok && !(p.debug & 4)) *reinterpret_cast<f32x4*>(yo + i * 16 + 4 * q) = v;
      }
    }
    if (!more) break;
    if (reload_w) {             // the range crosses into the next encoder: refresh the resident weights
      load_weights(g2);
      g_w = g2;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g2 * p.gs_b + i * 16 + 4 * q);
      __syncthreads();
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
    ++tile;
  }
}

template <int CIN, int COUT>
static int launch_s2_halo_fwd(HaloFwdParams& p, hipStream_t s) {
  constexpr int CQ = CIN / 4;
  constexpr int PLANE = 9 * 2 * 17;
  const size_t lds = (size_t)(9 * CQ * COUT + 2 * CQ * PLANE + 2 * 4 * (COUT / 16) * 64) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_fwd_kernel<CIN, COUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  long long blocks = p.ntiles < 256 ? p.ntiles : 256;
  hipLaunchKernelGGL((conv_s2_halo_fwd_kernel<CIN, COUT>), dim3((unsigned)blocks), dim3(512), lds, s, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// Returns 1 if handled, 0 if the shape is not covered (caller falls back to the gather-GEMM),
// or an error code < 0 / hipError.
int geeco_try_halo_fwd(const float* x, const float* w, const float* b, float* y, int groups, int64_t gs_x,
                       int64_t gs_w, int64_t gs_b, int64_t gs_y, int N, int H, int W, int Cin, int Cout, int stride,
                       int relu, hipStream_t stream, int* handled) {
  *handled = 0;
  static const int disabled = getenv("GEECO_NO_HALO") ? 1 : 0;
  if (disabled || !b) return 0;
  if (stride == 2 && Cin == 32 && Cout == 48 && (H % 2 == 0) && (W % 2 == 0)) {
    HaloFwdParams p = {};
    p.x = x; p.w = w; p.bias = b; p.y = y;
    p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_y = gs_y;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
    p.tiles_per_group = N * p.tiles_x * p.tiles_y;
    p.ntiles = (long long)groups * p.tiles_per_group;
    p.relu = relu;
    static const int dbg = getenv("GEECO_HALO_DEBUG") ? atoi(getenv("GEECO_HALO_DEBUG")) : 0;
    p.debug = dbg;
    int rc = launch_s2_halo_fwd<32, 48>(p, stream);
    if (rc) return rc;
    *handled = 1;
  }
  return 0;
}
