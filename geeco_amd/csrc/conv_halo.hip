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
// tiles), and the next tile's halo is fetched behind the current tile's MFMAs (conv2 forward:
// straight into LDS by LDS-DMA; the gradient kernels: through registers).
//
// MFMA: v_mfma_f32_16x16x4_f32, roles as in conv_gemm.hip (row i = output channel, column j =
// pixel) so every lane owns 4 consecutive NHWC channels of one pixel.
#include "geeco_common.h"
#include <type_traits>
#include <atomic>
#include <stdlib.h>
#include <stdio.h>

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the
// vector-memory counter (vmcnt(0)), which would expose the latency of the epilogue's global stores
// and of the next tile's prefetch loads once per tile (cdna_hip_programming.md, "Pipelining across
// barriers").  The "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// conv2-type forward: stride 2, CIN == 32, COUT % 16 == 0, tile = 4 x 16 output pixels.
// LDS: W as [tap][cq][co][4] (b128 B-fragments); halo (2 buffers, filled by LDS-DMA) as
// [row][pixel pair][16 float4 = 2 pixels x 8 channel quads, XOR-swizzled by the pair index]:
// a pixel's 128 bytes are fetched by 8 consecutive lanes and the b128 A-fragments of 16 consecutive
// output columns (input pixels 2 r + kx) fall on distinct 16-byte slots.
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
  unsigned long long* stamps;   // -DGEECO_STAMPS builds only: [block][2 waves][64] s_memtime timeline
  // optional ReLU sign fields of y (geeco_conv2_fwd_relu_fields): [G][N][fHp][fWp][4] uint16, field q bit 4 i + j <-> channel 16 i + 4 q + j
  unsigned short* fields;
  long long gs_fields;
  int fHp, fWp;
  // chunked forward (conv3): byte fields [G][N][Ho][Wo][COUT / 8]: byte (T >> 1) * 4 + q, bit 4 (T & 1) + j <-> channel 16 T + 4 q + j
  unsigned char* fields8;
  long long gs_fields8;
};

#ifdef GEECO_STAMPS
// Dev instrumentation (scripts/dev/halo_stamps.py): per-tile timeline of wave 0 (K half 0) and wave 4 (K half 1).
static unsigned long long* g_hstamps = nullptr;
extern "C" int geeco_debug_dump_halo_stamps(const char* path) {
  if (!g_hstamps) return 1;
  (void)hipDeviceSynchronize();
  const size_t n = 256 * 2 * 64;
  unsigned long long* h = (unsigned long long*)malloc(n * 8);
  (void)hipMemcpy(h, g_hstamps, n * 8, hipMemcpyDeviceToHost);
  FILE* f = fopen(path, "wb");
  if (!f) return 2;
  fwrite(h, 8, n, f);
  fclose(f);
  free(h);
  return 0;
}
#define HSTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (lane == 0 && wid < 8 && (wid & 3) == 0 && p.stamps && (i) < 64)                                  \
      p.stamps[((long long)blockIdx.x * 2 + (wid >> 2)) * 64 + (i)] = __builtin_amdgcn_s_memtime();      \
  } while (0)
#else
#define HSTAMP(i)
#endif

__device__ float g_zero_page[64];   // source of LDS-DMA lanes that fall outside the image (TF SAME zero padding)

// Output stores of the big layers.  Bit SITE of GEECO_NT selects a non-temporal store (the tensor is far larger than the
// caches and streams to HBM: measured on conv1's 805 MB output, 207 -> 190 us); sites: 0 conv1 fwd, 1 conv2 fwd,
// 2 conv3 fwd, 3 conv3 dgrad.
#ifndef GEECO_NT
#define GEECO_NT 1
#endif
template <int SITE>
__device__ __forceinline__ void stream_store(float* dst, const f32x4& v) {
  if constexpr ((GEECO_NT >> SITE) & 1)
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
  else
    *reinterpret_cast<f32x4*>(dst) = v;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Barrier that also retires this wave's LDS-DMA (global_load_lds) writes before anyone reads them.
__device__ __forceinline__ void dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// RW: the wave's kernel fragments (9 taps x COUT / 16 float4 = 108 registers) stay in VGPRs for the block's
// lifetime, so the K loop reads ONE ds_read_b128 per 12 MFMAs instead of four; costs the second block per CU.
template <int CIN, int COUT, bool RW>
__global__ __launch_bounds__(512, RW ? 1 : 2) void conv_s2_halo_fwd_kernel(const HaloFwdParams p) {
  constexpr int NT = 512;                             // 8 waves = 4 output rows x 2 halves of the channel (K) range
  constexpr int TH = 4, TW = 16;
  constexpr int CQ = CIN / 4;
  static_assert(CQ == 8, "pair-swizzled halo image is laid out for 8 channel quads");
  // Halo image (filled by LDS-DMA, no VGPR staging): row hy = 17 pixel PAIRS x 16 float4; the 16 quads of
  // a pair (2 pixels x 8 quads) are XOR-swizzled by (pair & 15) so that the b128 fragment reads of 16
  // consecutive output columns (input pixels 2 r + kx) hit distinct 16-byte slots.
  constexpr int HY = 2 * TH + 1;
  constexpr int ROW = 17 * 16;                        // float4 per halo row
  constexpr int HALO_USED = HY * ROW;                 // 2448
  constexpr int NDMA = (HALO_USED + 63) / 64;         // 1 KiB LDS-DMA pieces per tile (39)
  constexpr int HALO_F4 = NDMA * 64;                  // 2496 (the last piece spills into padding)
  constexpr int NSLOT = (NDMA + 7) / 8;               // pieces per wave
  constexpr int W_F4 = 9 * CQ * COUT;
  constexpr int TI = COUT / 16;
  constexpr int KB = CIN / 16;
  constexpr int KBW = KB / 2;                         // 16-channel blocks per wave
  constexpr int NIT = 9 * KBW;
  constexpr int RED_F4 = 4 * TI * 64;                 // partial accumulators of waves 4..7
  static_assert(KB % 2 == 0, "K split");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                              // 2 halo buffers
  f32x4* sR = sH + 2 * HALO_F4;                       // 2 reduction buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // this wave's LDS-DMA pieces: piece k = wid + 8 i covers halo slots [64 k, 64 k + 64); lane -> (hy, hx, cq)
  __builtin_assume(wid >= 0 && wid < 8);
  int d_src[NSLOT];
  short d_hy[NSLOT], d_hx[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int sl = (wid + 8 * i) * 64 + lane;
    const int row = sl / ROW, rem = sl - row * ROW;
    const int pair = rem >> 4, u = (rem & 15) ^ (pair & 15);
    const int hx = 2 * pair + (u >> 3), cq = u & 7;
    const bool ok = sl < HALO_USED && hx <= 2 * TW;
    d_hy[i] = (short)(ok ? row : 30000);              // out-of-range marker fails the per-tile bounds test
    d_hx[i] = (short)hx;
    d_src[i] = (row * p.W + hx) * CIN + cq * 4;
  }
  auto dma_halo = [&](int buf, int g_, int n_, int ty_, int tx_) {
    const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;      // TF SAME, stride 2, even input: pad_before = 0
    const float* xg = p.x + (long long)g_ * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * CIN;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (wid + 8 * i < NDMA) {                         // wave-uniform
        const bool v = iy0 + d_hy[i] < p.H && ix0 + d_hx[i] < p.W;
        const float* src = v ? xg + d_src[i] : g_zero_page;
        f32x4* dst = sH + buf * HALO_F4 + (wid + 8 * i) * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      }
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

  dma_halo(0, g, n, ty, tx);
  load_weights(g);
  int g_w = g;
  f32x4 bias_r[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g * p.gs_b + i * 16 + 4 * q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // lane r = output column of row `strip`; this wave sums channels [khalf*CIN/2, (khalf+1)*CIN/2)
  const int cq_lane = khalf * KBW * 4 + q;
  const f32x4* hB = sW + cq_lane * COUT + r;
  f32x4 wreg[RW ? NIT : 1][TI];
  auto load_wreg = [&]() {
    if constexpr (RW) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int tap = it / KBW, kb = it - tap * KBW;
#pragma unroll
        for (int i = 0; i < TI; ++i) wreg[it][i] = hB[(tap * CQ + kb * 4) * COUT + i * 16];
      }
    }
  };
  load_wreg();
  int buf = 0;
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) {
      advance(g2, n2, ty2, tx2);
      dma_halo(buf ^ 1, g2, n2, ty2, tx2);     // lands in the other buffer while this tile computes
    }
    const bool reload_w = more && g2 != g_w;
    f32x4 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[i] = zero4;
    const f32x4* hA = sH + buf * HALO_F4 + (2 * strip) * ROW;
    f32x4 a_cur, b_cur[TI], a_nxt, b_nxt[TI];
    auto frag = [&](int it, f32x4& a, f32x4 (&b)[TI]) {
      const int tap = it / KBW, kb = it - tap * KBW;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int pair = r + (kx >> 1);
      a = hA[ky * ROW + pair * 16 + (((((kx & 1) << 3) | (cq_lane + 4 * kb))) ^ (pair & 15))];
      if constexpr (!RW) {
#pragma unroll
        for (int i = 0; i < TI; ++i) b[i] = hB[(tap * CQ + kb * 4) * COUT + i * 16];
      }
    };
    frag(0, a_cur, b_cur);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + 1 < NIT) frag(it + 1, a_nxt, b_nxt);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ABOVE this group's MFMAs
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(RW ? wreg[RW ? it : 0][i][s] : b_cur[i][s], a_cur[s], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
      if constexpr (!RW) {
#pragma unroll
        for (int i = 0; i < TI; ++i) b_cur[i] = b_nxt[i];
      }
    }
    f32x4* red = sR + (int)(tile & 1) * RED_F4;
    if (khalf == 1) {
#pragma unroll
      for (int i = 0; i < TI; ++i) red[(strip * TI + i) * 64 + lane] = acc[i];
    }
    dma_barrier();   // partial sums visible; next halo landed; everyone is done with buf and sW
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
        if (ok) *reinterpret_cast<f32x4*>(yo + i * 16 + 4 * q) = v;
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
      load_wreg();
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
    ++tile;
  }
}

// ------------------------------------------------------------------------------------------------
// Warp-specialised conv2 forward (the default): 8 compute waves (4 output rows x 2 K halves, kernel fragments
// in registers, loaded straight from the HWIO kernel) + LW loader waves that do nothing but issue the LDS-DMA of
// the halos TWO tiles ahead into a ring of three buffers; the output strip is transposed through LDS so that
// every store instruction writes 1 KiB of consecutive bytes.
// Why (in-kernel timelines, scripts/dev/halo_stamps.py): in the variant above an LDS-DMA instruction holds the
// issuing wave for ~300-600 cycles, ~3k cycles per tile on the compute waves that issue the 39 pieces; they reach
// the tile barrier late and their partners idle (tile period 9.6k cycles for 6.9k cycles of MFMA work per SIMD).
// With loaders the compute waves' MFMA phase is 3.7k cycles (3.5k ideal); what remains is the CU's vector
// memory pipe: 39 KB in + 12 KB out per tile pass through it at ~5.5 B/clk whoever issues them (the epilogue's
// three store instructions wait ~3-4k cycles behind the loaders' pieces).  Measured: +3.5 % on the launch.
// All waves meet at ONE barrier per tile: loaders arrive once the halo of the NEXT tile has landed (`vmcnt`
// leaves the tile after it in flight), compute waves after their last fragment read of the current one.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm_imm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CIN, int COUT, int LW>
__global__ __launch_bounds__(512 + 64 * LW) void conv_s2_halo_fwd_ws_kernel(const HaloFwdParams p) {
  constexpr int TH = 4, TW = 16;
  constexpr int CQ = CIN / 4;
  static_assert(CQ == 8 && CIN == 32, "pair-swizzled halo image is laid out for 8 channel quads; 2 K halves of 16");
  constexpr int HY = 2 * TH + 1;
  constexpr int ROW = 17 * 16;
  constexpr int HALO_USED = HY * ROW;
  constexpr int NDMA = (HALO_USED + 63) / 64;         // 39 pieces of 1 KiB per tile
  constexpr int HALO_F4 = NDMA * 64;
  constexpr int NSLOT = (NDMA + LW - 1) / LW;         // pieces per loader wave
  constexpr int TI = COUT / 16;
  constexpr int RED_F4 = 4 * TI * 64;
  constexpr int NBUF = 3;                             // halo ring: the DMA runs two tiles ahead of the MFMAs
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sH = reinterpret_cast<f32x4*>(smem);         // NBUF halo buffers
  f32x4* sR = sH + NBUF * HALO_F4;                    // 2 reduction buffers
  constexpr int OP = COUT / 4 + 1;                    // float4 pitch of an output pixel in the store staging (odd)
  f32x4* sO = sR + 2 * RED_F4;                        // 4 strips x [16 pixels][OP]: output transposed for full-line stores

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wid >= 8;
  const int r = lane & 15, q = lane >> 4;
  const int strip = wid & 3, khalf = (wid >> 2) & 1;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

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
  if (loader) {
    // ===== loader waves ===================================================================================
    const int lw = wid - 8;
    int d_src[NSLOT];
    short d_hy[NSLOT], d_hx[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int sl = (lw + LW * i) * 64 + lane;
      const int row = sl / ROW, rem = sl - row * ROW;
      const int pair = rem >> 4, u = (rem & 15) ^ (pair & 15);
      const int hx = 2 * pair + (u >> 3), cq = u & 7;
      const bool ok = sl < HALO_USED && hx <= 2 * TW;
      d_hy[i] = (short)(ok ? row : 30000);              // out-of-range marker fails the per-tile bounds test
      d_hx[i] = (short)hx;
      d_src[i] = (row * p.W + hx) * CIN + cq * 4;
    }
    auto dma_halo = [&](int buf, int g_, int n_, int ty_, int tx_) {
      const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;      // TF SAME, stride 2, even input: pad_before = 0
      const float* xg = p.x + (long long)g_ * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * CIN;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        if (lw + LW * i < NDMA) {                         // wave-uniform
          const bool v = iy0 + d_hy[i] < p.H && ix0 + d_hx[i] < p.W;
          const float* src = v ? xg + d_src[i] : g_zero_page;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sH + (lw + LW * i) * 64 + buf * HALO_F4), 16, 0, 0);
        }
      }
    };
    {
      // LDS-DMA ring: tiles t+1, t+2 are in flight ahead of the compute waves
      const int npieces = (NDMA - lw + LW - 1) / LW;       // 10 or 9 (wave-uniform)
      int g1 = g, n1 = n, ty1 = ty, tx1 = tx;              // tile + 1
      dma_halo(0, g, n, ty, tx);
      const bool has1 = tile + 1 < tend;
      if (has1) {
        advance(g1, n1, ty1, tx1);
        dma_halo(1, g1, n1, ty1, tx1);
      }
      if (has1) {
        if (npieces == NSLOT) wait_vm_imm<NSLOT>(); else wait_vm_imm<NSLOT - 1>();
      } else {
        wait_vm_imm<0>();
      }
      asm volatile("s_barrier" ::: "memory");             // (A) halo 0 landed
      int g2 = g1, n2 = n1, ty2 = ty1, tx2 = tx1;          // tile + 2
      int slot = 2;                                        // ring slot of tile + 2
      for (;;) {
        const bool more1 = tile + 1 < tend, more2 = tile + 2 < tend;
        if (more2) {
          advance(g2, n2, ty2, tx2);
          dma_halo(slot, g2, n2, ty2, tx2);
          slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
        // tile barrier: the halo of tile + 1 must have landed (tile + 2 may stay in flight)
        if (more2) {
          if (npieces == NSLOT) wait_vm_imm<NSLOT>(); else wait_vm_imm<NSLOT - 1>();
        } else {
          wait_vm_imm<0>();
        }
        asm volatile("s_barrier" ::: "memory");
        if (!more1) break;
        ++tile;
      }
    }
    return;
  }

  // ===== compute waves ======================================================================================
  int g_w = g;
  f32x4 bias_r[TI];
  const int cq_lane = khalf * 4 + q;       // this wave sums channels [16 khalf, 16 khalf + 16)
  f32x4 wreg[9][TI];
  // kernel fragments straight from the HWIO kernel (no LDS staging): lane (r, q) of co tile i holds
  // w[tap][4 cq_lane + s][16 i + r], s = 0..3; 16 lanes read 64 consecutive bytes
  auto load_wreg = [&](int g_) {
    const float* wg = p.w + (long long)g_ * p.gs_w + (4 * cq_lane) * COUT + r;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float* w0 = wg + tap * CIN * COUT + 16 * i;
        wreg[tap][i] = f32x4{w0[0], w0[COUT], w0[2 * COUT], w0[3 * COUT]};
      }
#pragma unroll
    for (int i = 0; i < TI; ++i) bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g_ * p.gs_b + i * 16 + 4 * q);
  };
  load_wreg(g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");               // (A)
  int buf = 0;
  [[maybe_unused]] int tcount = 0;        // tile ordinal, for the dev stamps only
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    HSTAMP(tcount < 10 ? 6 * tcount + 0 : 64);
    if (more) advance(g2, n2, ty2, tx2);
    HSTAMP(tcount < 10 ? 6 * tcount + 1 : 64);
    f32x4 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[i] = zero4;
    const f32x4* hA = sH + buf * HALO_F4 + (2 * strip) * ROW;
    f32x4 a_cur, a_nxt;
    auto frag = [&](int tap, f32x4& a) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int pair = r + (kx >> 1);
      a = hA[ky * ROW + pair * 16 + ((((kx & 1) << 3) | cq_lane) ^ (pair & 15))];
    };
    frag(0, a_cur);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) frag(tap + 1, a_nxt);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch read ABOVE this group's MFMAs
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][i][s], a_cur[s], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
    }
    HSTAMP(tcount < 10 ? 6 * tcount + 2 : 64);
    f32x4* red = sR + (int)(tile & 1) * RED_F4;
    if (khalf == 1) {
#pragma unroll
      for (int i = 0; i < TI; ++i) red[(strip * TI + i) * 64 + lane] = acc[i];
    }
    HSTAMP(tcount < 10 ? 6 * tcount + 3 : 64);
    lds_barrier();
    HSTAMP(tcount < 10 ? 6 * tcount + 4 : 64);   // tile barrier: partial sums visible; everyone is done with buf; the loaders' next halo landed
    if (khalf == 0) {
      // epilogue: lane owns pixel (ty*4 + strip, tx*16 + r), channels 16 i + 4 q .. +3.  The strip's 16 x COUT
      // outputs are 3 KB of consecutive NHWC bytes: they are transposed through LDS so that every store instruction
      // writes 1 KiB of consecutive bytes instead of 16 separate 64-byte pieces (in-kernel timeline: the three
      // piecewise stores held the wave ~3.6k cycles per tile - the kernel's critical path).
      f32x4* so = sO + strip * 16 * OP;
      unsigned field = 0;        // sign bits of this lane's 4 TI outputs (after the ReLU: > 0 <=> non-zero bits)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        f32x4 v = acc[i] + red[(strip * TI + i) * 64 + lane] + bias_r[i];
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        so[r * OP + 4 * i + q] = v;
        if (p.fields) {
#pragma unroll
          for (int j = 0; j < 4; ++j) field |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
        }
      }
      if (p.fields) {
        // the consumer (conv3's input-gradient kernel) holds the same (pixel r, quad q) layout in its accumulators, so
        // every lane stores its own 16-bit field: no cross-lane assembly; 16 pixels x 4 fields = 128 consecutive bytes
        const int fy = ty * TH + strip, fx = tx * TW + r;
        if (fy < p.Ho && fx < p.Wo)
          p.fields[(long long)g * p.gs_fields + (((long long)n * p.fHp + fy) * p.fWp + fx) * 4 + q] = (unsigned short)field;
      }
      // same-wave LDS round trip: the compiler's lgkmcnt wait orders the reads behind the writes
      const int oy = ty * TH + strip;
      float* yo = p.y + (long long)g * p.gs_y + (((long long)n * p.Ho + oy) * p.Wo + tx * TW) * COUT;
      constexpr int C4 = COUT / 4;
#pragma unroll
      for (int jj = 0; jj < TI; ++jj) {          // 16 * C4 float4 = TI x 64 lanes
        const int m = lane + 64 * jj;
        const int px = m / C4, c4 = m - px * C4;
        const f32x4 v = so[px * OP + c4];
        if (oy < p.Ho && tx * TW + px < p.Wo) stream_store<1>(yo + m * 4, v);
      }
    }
    HSTAMP(tcount < 10 ? 6 * tcount + 5 : 64);
    ++tcount;
    if (!more) break;
    if (g2 != g_w) {             // the range crosses into the next encoder: new kernel fragments
      load_wreg(g2);
      g_w = g2;
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    ++tile;
  }
}

template <int CIN, int COUT, int LW>
static int launch_s2_halo_fwd_ws(HaloFwdParams& p, hipStream_t s) {
  constexpr int HALO_F4 = ((9 * 17 * 16 + 63) / 64) * 64;
  const size_t lds = (size_t)(3 * HALO_F4 + 2 * 4 * (COUT / 16) * 64 + 4 * 16 * (COUT / 4 + 1)) * 16;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_fwd_ws_kernel<CIN, COUT, LW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
#ifdef GEECO_STAMPS
  if (!g_hstamps) (void)hipMalloc(&g_hstamps, 256 * 2 * 64 * 8);
  (void)hipMemset(g_hstamps, 0, 256 * 2 * 64 * 8);
  p.stamps = g_hstamps;
#endif
  long long blocks = p.ntiles < 256 ? p.ntiles : 256;
  geeco_note_kernel("conv_s2_halo_fwd_ws_kernel<%d, %d, %d>", CIN, COUT, LW);
  hipLaunchKernelGGL((conv_s2_halo_fwd_ws_kernel<CIN, COUT, LW>), dim3((unsigned)blocks), dim3(512 + 64 * LW), lds, s, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

#ifdef GEECO_DEV_KERNELS      // the pre-warp-specialised conv2 forward (GEECO_HALO_WS=0, GEECO_HALO_RW): development build only
template <int CIN, int COUT, bool RW>
static int launch_s2_halo_fwd_v(HaloFwdParams& p, hipStream_t s) {
  constexpr int CQ = CIN / 4;
  constexpr int HALO_F4 = ((9 * 17 * 16 + 63) / 64) * 64;
  const size_t lds = (size_t)(9 * CQ * COUT + 2 * HALO_F4 + 2 * 4 * (COUT / 16) * 64) * 16;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_fwd_kernel<CIN, COUT, RW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  long long blocks = p.ntiles < 256 ? p.ntiles : 256;
  geeco_note_kernel("conv_s2_halo_fwd_kernel<%d, %d, %s>", CIN, COUT, RW ? "true" : "false");
  hipLaunchKernelGGL((conv_s2_halo_fwd_kernel<CIN, COUT, RW>), dim3((unsigned)blocks), dim3(512), lds, s, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}
#endif

template <int CIN, int COUT>
static int launch_s2_halo_fwd(HaloFwdParams& p, hipStream_t s) {
#ifdef GEECO_DEV_KERNELS
  static const int rw = geeco_dev_getenv("GEECO_HALO_RW") ? atoi(geeco_dev_getenv("GEECO_HALO_RW")) : 1;   // measured +2.4..3.6 % on the launch
  // loader waves (0 = the kernels above): 4 measured +3.5 % on the launch, 2 are too few (-5 %)
  static const int ws = geeco_dev_getenv("GEECO_HALO_WS") ? atoi(geeco_dev_getenv("GEECO_HALO_WS")) : 4;
  if (ws == 2) return launch_s2_halo_fwd_ws<CIN, COUT, 2>(p, s);
  if (ws != 4) return rw ? launch_s2_halo_fwd_v<CIN, COUT, true>(p, s) : launch_s2_halo_fwd_v<CIN, COUT, false>(p, s);
#endif
  return launch_s2_halo_fwd_ws<CIN, COUT, 4>(p, s);      // 8 compute waves + 4 loader waves
}

// ------------------------------------------------------------------------------------------------
// conv3-type forward (stride 2, CIN % 16 == 0, COUT % 32 == 0, kernel resident in LDS): the input halo of
// a 4 x 16 output tile is staged in 16-channel chunks - with 48 input channels the whole halo (twice) does
// not fit beside the 110 KB kernel.  Step (tile, chunk): 9 taps read their A fragments from the chunk image
// [row 9][pixel pair 17][8 float4 = 2 pixels x 4 quads, XOR-swizzled by pair & 7] while the next step's
// image lands in the other buffer by LDS-DMA (4 lanes fetch a pixel's 64 contiguous bytes).  Wave = (output
// row of the tile, half of the output channels); accumulators live across the chunks of a tile.
// Why: the gather GEMM fetches every input pixel 2.25 times from beyond L2 (PMC 744 MB for a 302 MB input)
// and the kernel slab once per block (340 MB), at the per-CU miss rate of the vector memory path.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv_s2_halo_fwd_chunked_kernel(const HaloFwdParams p) {
  constexpr int NT = 512;
  constexpr int TH = 4, TW = 16;
  constexpr int CQ = CIN / 4;
  constexpr int NCH = CIN / 16;                       // chunks (steps) per tile
  constexpr int HY = 2 * TH + 1;
  constexpr int ROW = 17 * 8;                         // float4 per image row
  constexpr int IMG_F4 = HY * ROW;                    // 1224
  constexpr int NPIECE = (IMG_F4 + 63) / 64;          // 20
  constexpr int BUF_F4 = NPIECE * 64;
  constexpr int NSLOT = (NPIECE + 7) / 8;
  constexpr int W_F4 = 9 * CQ * COUT;
  constexpr int TI = COUT / 32;                       // co tiles per wave (each wave: half of the channels)
  static_assert(CIN % 16 == 0 && COUT % 32 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                              // 2 chunk images

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int strip = wid & 3, cohalf = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

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

  __builtin_assume(wid >= 0 && wid < 8);
  int d_src[NSLOT];
  short d_hy[NSLOT], d_hx[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int sl = (wid + 8 * i) * 64 + lane;
    const int rw = sl / ROW, rem = sl - rw * ROW;
    const int pair = rem >> 3, u = (rem & 7) ^ (pair & 7);
    const int hx = 2 * pair + (u >> 2), cq4 = u & 3;
    const bool ok = sl < IMG_F4 && hx <= 2 * TW;
    d_hy[i] = (short)(ok ? rw : 30000);               // out-of-range marker fails the per-tile bounds test
    d_hx[i] = (short)hx;
    d_src[i] = (rw * p.W + hx) * CIN + cq4 * 4;
  }
  const float* zero_page = g_zero_page;             // its address ONCE, in scalar registers: referenced inside the tile loop the
  asm volatile("" : "+s"(zero_page));              // compiler re-fetches it through the GOT (s_getpc + s_load + s_waitcnt lgkmcnt(0)) per DMA piece
  auto dma_chunk = [&](int buf, int g_, int n_, int ty_, int tx_, int chunk) {
    const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;      // TF SAME, stride 2, even input: pad_before = 0
    const float* xg = p.x + (long long)g_ * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * CIN + chunk * 16;
    const long long zero_x = zero_page - xg;             // tile-only values in scalar registers (as conv_wgrad_halo.hip)
    const int hy = p.H - iy0, hx = p.W - ix0;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (8 * (i + 1) <= NPIECE || wid + 8 * i < NPIECE) {      // compile-time true except in the last slot (wid < 8)
        const bool v = d_hy[i] < hy && d_hx[i] < hx;
        const long long off = v ? (long long)d_src[i] : zero_x;
        __builtin_amdgcn_global_load_lds((gptr_t)(xg + off), (lptr_t)(sH + buf * BUF_F4 + (wid + 8 * i) * 64), 16, 0, 0);
      }
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

  dma_chunk(0, g, n, ty, tx, 0);
  load_weights(g);
  int g_w = g;
  f32x4 bias_r[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
    bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g * p.gs_b + cohalf * (COUT / 2) + i * 16 + 4 * q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const f32x4* hB = sW + q * COUT + cohalf * (COUT / 2) + r;     // + ((tap*CQ + 4 chunk) * COUT + 16 i)
  // One tile; b0 = LDS buffer of its first chunk.  The chunk loop is fully unrolled and the tile loop below alternates
  // b0 (NCH is odd for conv3: the parity flips per tile), so the buffer index is a compile-time constant everywhere:
  // the fragment addresses are loop-invariant registers + immediates instead of VALU adds per chunk.
  auto tile_body = [&](auto b0c) -> bool {
    constexpr int b0 = decltype(b0c)::value;
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) advance(g2, n2, ty2, tx2);
    f32x4 acc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[i] = zero4;
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const int buf = (b0 + chunk) & 1;
      if (chunk + 1 < NCH)
        dma_chunk(buf ^ 1, g, n, ty, tx, chunk + 1);
      else if (more)
        dma_chunk(buf ^ 1, g2, n2, ty2, tx2, 0);
      const f32x4* hA = sH + buf * BUF_F4 + (2 * strip) * ROW;
      const f32x4* hBc = hB + 4 * chunk * COUT;
      f32x4 a_cur, b_cur[TI], a_nxt, b_nxt[TI];
      auto frag = [&](int tap, f32x4& a, f32x4 (&b)[TI]) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pair = r + (kx >> 1);
        a = hA[ky * ROW + pair * 8 + ((((kx & 1) << 2) | q) ^ (pair & 7))];
#pragma unroll
        for (int i = 0; i < TI; ++i) b[i] = hBc[tap * CQ * COUT + i * 16];
      };
      frag(0, a_cur, b_cur);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) frag(tap + 1, a_nxt, b_nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TI; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_cur[i][s], a_cur[s], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
#pragma unroll
        for (int i = 0; i < TI; ++i) b_cur[i] = b_nxt[i];
      }
      if (chunk + 1 < NCH || more) dma_barrier();   // next image landed; everyone is done with this one
    }
    {
      // epilogue: pixel (oy, ox) = (ty*4 + strip, tx*16 + r); channels cohalf*COUT/2 + 16 i + 4 q .. +3
      const int oy = ty * TH + strip, ox = tx * TW + r;
      const bool ok = oy < p.Ho && ox < p.Wo;
      float* yo = p.y + (long long)g * p.gs_y + (((long long)n * p.Ho + oy) * p.Wo + ox) * COUT + cohalf * (COUT / 2);
      unsigned sign = 0;     // sign bits of this lane's 4 TI outputs (after the ReLU: > 0 <=> non-zero bits): bit 4 i + j
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        f32x4 v = acc[i] + bias_r[i];
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (ok) stream_store<2>(yo + i * 16 + 4 * q, v);
        if (p.fields8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) sign |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
        }
      }
      // sign fields for the next layer's input-gradient kernel (same (pixel r, quad q) accumulator layout: the lane owns
      // whole bytes: its TI = 2 channel tiles are tile pair `cohalf`)
      if (p.fields8 && ok) {
        static_assert(TI == 2, "one byte per lane = one pair of 16-channel tiles");
        p.fields8[(long long)g * p.gs_fields8 + (((long long)n * p.Ho + oy) * p.Wo + ox) * (COUT / 8) + cohalf * 4 + q] = (unsigned char)sign;
      }
    }
    if (!more) return false;
    if (g2 != g_w) {             // the range crosses into the next encoder: refresh the resident kernel
      load_weights(g2);
      g_w = g2;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g2 * p.gs_b + cohalf * (COUT / 2) + i * 16 + 4 * q);
      __syncthreads();
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    ++tile;
    return true;
  };
  for (;;) {
    if (!tile_body(std::integral_constant<int, 0>{})) break;
    if (!tile_body(std::integral_constant<int, NCH & 1>{})) break;
  }
}

template <int CIN, int COUT>
static int launch_s2_halo_fwd_chunked(HaloFwdParams& p, hipStream_t s) {
  constexpr int BUF_F4 = ((9 * 17 * 8 + 63) / 64) * 64;
  const size_t lds = (size_t)(9 * (CIN / 4) * COUT + 2 * BUF_F4) * 16;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_fwd_chunked_kernel<CIN, COUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  long long blocks = p.ntiles < 256 ? p.ntiles : 256;
  geeco_note_kernel("conv_s2_halo_fwd_chunked_kernel<%d, %d>", CIN, COUT);
  hipLaunchKernelGGL((conv_s2_halo_fwd_chunked_kernel<CIN, COUT>), dim3((unsigned)blocks), dim3(512), lds, s, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// does the dispatcher below take this shape?  (geeco_conv3x3_fwd_state asks: a layer these kernels serve must not go through
// the gather GEMM there while every other path runs it through them)
int geeco_halo_fwd_handles(int H, int W, int Cin, int Cout, int stride) {
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  static const int no_chunked = geeco_dev_getenv("GEECO_NO_HALO3") ? 1 : 0;
  if (disabled || stride != 2 || (H % 2) || (W % 2)) return 0;
  return (Cin == 32 && Cout == 48) || (Cin == 48 && Cout == 64 && !no_chunked);
}

// Returns 1 if handled, 0 if the shape is not covered (caller falls back to the gather-GEMM),
// or an error code < 0 / hipError.
int geeco_try_halo_fwd(const float* x, const float* w, const float* b, float* y, int groups, int64_t gs_x,
                       int64_t gs_w, int64_t gs_b, int64_t gs_y, int N, int H, int W, int Cin, int Cout, int stride,
                       int relu, hipStream_t stream, int* handled) {
  *handled = 0;
  if (!b) return 0;
  const bool conv2 = Cin == 32 && Cout == 48;
  if (geeco_halo_fwd_handles(H, W, Cin, Cout, stride)) {
    HaloFwdParams p = {};
    p.x = x; p.w = w; p.bias = b; p.y = y;
    p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_y = gs_y;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
    p.tiles_per_group = N * p.tiles_x * p.tiles_y;
    p.ntiles = (long long)groups * p.tiles_per_group;
    p.relu = relu;
    int rc = conv2 ? launch_s2_halo_fwd<32, 48>(p, stream) : launch_s2_halo_fwd_chunked<48, 64>(p, stream);
    if (rc) return rc;
    *handled = 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv2-type filter/bias gradient with an LDS halo (stride 2, CIN == 32, COUT % 16 == 0).
//   dw[tap][ci][co] = sum_pixels x[halo(pixel, tap)][ci] * dz[pixel][co]       db[co] = sum dz
// A block walks a contiguous range of 4x16-pixel tiles of ONE encoder and keeps its partial dw in
// registers (wave = (output row of the tile, 16-channel half of ci): 9 taps x COUT/16 MFMA tiles =
// 108 accumulator registers); per tile it stages the x halo (same LDS image as the forward kernel,
// read here with ds_read_b32: bank = 8 (cq % 4) + 4 (pixel % 8) + (ci % 4), conflict free) and the
// dz tile [64 pixels][COUT] (row pitch COUT = 16 mod 32).  MFMA k = 4 consecutive pixels.
// At the end the four row-waves of each ci half are summed through LDS and the block writes one
// slab; wgrad_reduce_kernel (conv_wgrad.hip) sums the slabs in a fixed order.
// ------------------------------------------------------------------------------------------------
struct HaloWgradParams {
  unsigned long long* stamps;   // -DGEECO_STAMPS builds only (scripts/dev/wgrad2_stamps.py)
  const float* x;
  const float* dz;
  float* part;               // [G][S][9*CIN*COUT + COUT]
  long long gs_x, gs_dz;
  int N, H, W, Ho, Wo;
  int tiles_x, tiles_y;
  int tiles_per_group;
  int S;                     // slabs per group (S0, + 1 with the remainder block)
  int S0, per, groups;       // block b < S0 * groups: group b / S0, tiles [per * (b % S0), + per); block S0 * groups (if
                             // launched): the remainder [S0 * per, tiles_per_group) of EVERY group, one after the other
};

#ifdef GEECO_STAMPS
#define WSTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (lane == 0 && (wid & 3) == 0 && wid < 8 && g == 0 && p.stamps && (i) < 64)                        \
      p.stamps[((long long)split * 2 + (wid >> 2)) * 64 + (i)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define WSTAMP(i)
#endif

// Measured on this kernel (in-kernel timeline, scripts/dev/wgrad2_stamps.py; tile = 9.8 k cycles, its 216 MFMAs per SIMD
// = 6.9 k): the seven DMA pieces per wave hold both waves of a SIMD in the vector-memory queue for 1.3 - 2 k cycles per
// tile; issued from inside the MFMA loop they lengthen the loop by the same amount; without any DMA the tile takes
// 8.1 k; with four extra loader waves doing all DMA the MFMA waves finish after 7.3 k and then wait at the tile barrier
// until 10.8 k for the 51 KB to land.  The CU ingests ~5 B/clk here (3.1 TB/s chip-wide for 1.2 GB, all of it
// compulsory): the kernel is bound by that, not by where the loads sit.
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv_s2_halo_wgrad_kernel(const HaloWgradParams p) {
  constexpr int NT = 512;
  constexpr int TH = 4, TW = 16;
  constexpr int HY = 2 * TH + 1;
  // LDS images, both filled by LDS-DMA (no VGPR staging, no ds_write):
  //   x halo  [row 9][pixel pair 17][16 float4 = 2 pixels x 8 channel quads]; the quad slot is XOR-swizzled by
  //           (pair & 3) << 2 so that the ds_read_b32 of a k-group (4 consecutive pixel pairs x 16 channels of one
  //           16-channel half) covers all 64 banks: bank = 4 ((half << 3 | cq) ^ swz) + (c % 4);
  //   dz tile [64 pixels][COUT] row-major (pitch COUT = 16 mod 32: the 4 pixels of a k-group use different banks).
  constexpr int ROW = 17 * 16;
  constexpr int HALO_USED = HY * ROW;                   // 2448 float4
  constexpr int NHP = (HALO_USED + 63) / 64;            // 39 pieces of 1 KiB
  constexpr int HALO_F4 = NHP * 64;
  constexpr int C4 = COUT / 4;
  constexpr int DZ_F4 = TH * TW * C4;                   // 768 float4 = 12 pieces
  constexpr int NZP = DZ_F4 / 64;
  constexpr int NDZ = (DZ_F4 + NT - 1) / NT;
  constexpr int NSLOT = (NHP + NZP + 7) / 8;            // DMA pieces per wave (halo pieces first, then dz)
  constexpr int TI = COUT / 16;
  static_assert(CIN == 32, "two ci halves <-> two wave groups; 8 quads per pixel");
  static_assert(COUT % 32 == 16 && DZ_F4 % 64 == 0, "dz row pitch must be 16 (mod 32) floats");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sH = reinterpret_cast<f32x4*>(smem);            // 2 halo buffers
  f32x4* sZ = sH + 2 * HALO_F4;                           // 2 dz tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int strip = wid & 3, cit = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // Regular blocks own `per` tiles of one encoder; the remainder block (at most one per launch) walks the tiles the
  // regular blocks of every encoder leave over - one segment, one slab per encoder (bottom_slices() below).
  // (The segment body stays inline in this loop, at the kernel's indentation: as a __forceinline__ function called from
  // the loop it measured +1 % here and in the fused bottom, called through a lambda or from two sites 1.5 - 2 x slower.)
  const int nreg = p.S0 * p.groups;
  const bool regular = (int)blockIdx.x < nreg;
  const int nseg = regular ? 1 : p.groups;
#pragma unroll 1
  for (int seg = 0; seg < nseg; ++seg) {
  const int g = regular ? (int)blockIdx.x / p.S0 : seg;
  const int split = regular ? (int)blockIdx.x - g * p.S0 : p.S0;
  int tile = regular ? split * p.per : p.S0 * p.per;
  const int tend = regular && tile + p.per < p.tiles_per_group ? tile + p.per : p.tiles_per_group;
  const long long slab = 9ll * CIN * COUT + COUT;
  float* part = p.part + ((long long)g * p.S + split) * slab;

  int n, ty, tx;
  {
    int per_img = p.tiles_x * p.tiles_y;
    n = tile / per_img;
    int rem = tile - n * per_img;
    ty = rem / p.tiles_x;
    tx = rem - ty * p.tiles_x;
  }
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    if (++tx_ == p.tiles_x) {
      tx_ = 0;
      if (++ty_ == p.tiles_y) {
        ty_ = 0;
        ++n_;
      }
    }
  };

  // this wave's DMA pieces: halo pieces kh = wid + 8 i (slots [64 kh, +64) of the halo image) and dz pieces kz = wid + 8 j
  // (float4 [64 kz, +64) of the dz tile), numbered separately: one role per slot for every wave, a (wave-uniform) range test only
  // in the last slot of each kind (as conv_wgrad_halo.hip / conv_dgrad_lds.hip)
  constexpr int NSH = (NHP + 7) / 8, NSZ = (NZP + 7) / 8;
  __builtin_assume(wid >= 0 && wid < 8);
  int h_src[NSH], z_src[NSZ];
  short h_a[NSH], h_b[NSH], z_a[NSZ], z_b[NSZ];       // halo: (row, hx); dz: (tile row, tile column)
#pragma unroll
  for (int i = 0; i < NSH; ++i) {
    const int sl = (wid + 8 * i) * 64 + lane;
    const int rw = sl / ROW, rem = sl - rw * ROW;
    const int pair = rem >> 4, u = (rem & 15) ^ ((pair & 3) << 2);
    const int hx = 2 * pair + (u >> 3), cq = u & 7;
    const bool ok = sl < HALO_USED && hx <= 2 * TW;
    h_a[i] = (short)(ok ? rw : 30000);                 // out-of-range marker fails the per-tile bounds test
    h_b[i] = (short)hx;
    h_src[i] = (rw * p.W + hx) * CIN + cq * 4;
  }
#pragma unroll
  for (int j = 0; j < NSZ; ++j) {
    const int f = (wid + 8 * j) * 64 + lane;
    const int px = f / C4, c4 = f - px * C4;
    z_a[j] = (short)(px >> 4);
    z_b[j] = (short)(px & 15);
    z_src[j] = ((px >> 4) * p.Wo + (px & 15)) * COUT + c4 * 4;
  }
  const float* zero_page = g_zero_page;             // its address ONCE, in scalar registers: referenced inside the tile loop the
  asm volatile("" : "+s"(zero_page));              // compiler re-fetches it through the GOT (s_getpc + s_load + s_waitcnt lgkmcnt(0)) per DMA piece
  auto dma_tile = [&](int buf, int n_, int ty_, int tx_) {
    const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;
    const float* xg = p.x + (long long)g * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * CIN;
    const float* zg = p.dz + (long long)g * p.gs_dz + (((long long)n_ * p.Ho + ty_ * TH) * p.Wo + tx_ * TW) * COUT;
    // everything that depends on the tile only in scalar registers, once per tile; a piece is then two compares, a select
    // between its own offset and the zero page's, and one 64-bit add (as conv_wgrad_halo.hip)
    const long long zero_x = zero_page - xg, zero_z = zero_page - zg;
    const int hy = p.H - iy0, hx = p.W - ix0, zy = p.Ho - ty_ * TH, zx = p.Wo - tx_ * TW;
#pragma unroll
    for (int i = 0; i < NSH; ++i)
      if (8 * (i + 1) <= NHP || wid + 8 * i < NHP) {       // compile-time true except in the last slot
        const bool v = h_a[i] < hy && h_b[i] < hx;
        const long long off = v ? (long long)h_src[i] : zero_x;
        __builtin_amdgcn_global_load_lds((gptr_t)(xg + off), (lptr_t)(sH + buf * HALO_F4 + (wid + 8 * i) * 64), 16, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < NSZ; ++j)
      if (8 * (j + 1) <= NZP || wid + 8 * j < NZP) {
        const bool v = z_a[j] < zy && z_b[j] < zx;
        const long long off = v ? (long long)z_src[j] : zero_z;
        __builtin_amdgcn_global_load_lds((gptr_t)(zg + off), (lptr_t)(sZ + buf * DZ_F4 + (wid + 8 * j) * 64), 16, 0, 0);
      }
  };

  f32x4 dbsum[NDZ];
#pragma unroll
  for (int i = 0; i < NDZ; ++i) dbsum[i] = zero4;
  f32x4 acc[9][TI];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < TI; ++i) acc[t][i] = zero4;

  if (tile < tend) dma_tile(0, n, ty, tx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // B-operand (x) of lane: channel 16 cit + r, pixel pair 4 s + q (+1 for kx = 2); A-operand (dz): co = r
  const int cq_lane = cit * 4 + (r >> 2);
  int xe[3];                                              // float offset of (pair q + (kx >> 1), half kx & 1, cq, c % 4)
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int pr = q + (kx >> 1);
    xe[kx] = ((pr * 16 + ((((kx & 1) << 3) | cq_lane) ^ ((pr & 3) << 2))) << 2) + (r & 3);
  }
  const int za_lane = (16 * strip + q) * COUT + r;
  int buf = 0;
  [[maybe_unused]] int tcount = 0;
  for (; tile < tend; ++tile, ++tcount) {
    const bool more = tile + 1 < tend;
    int n2 = n, ty2 = ty, tx2 = tx;
    WSTAMP(tcount < 10 ? 6 * tcount + 0 : 64);
    if (more) {
      advance(n2, ty2, tx2);
      dma_tile(buf ^ 1, n2, ty2, tx2);                    // lands behind this tile's MFMAs
    }
    WSTAMP(tcount < 10 ? 6 * tcount + 1 : 64);
    // bias gradient: every thread adds its share of the dz tile (NDZ float4 reads per tile)
#pragma unroll
    for (int i = 0; i < NDZ; ++i)
      if (tid + NT * i < DZ_F4) dbsum[i] += sZ[buf * DZ_F4 + tid + NT * i];
    const float* hx = reinterpret_cast<const float*>(sH + buf * HALO_F4) + (2 * strip) * ROW * 4;
    const float* hz = reinterpret_cast<const float*>(sZ + buf * DZ_F4) + za_lane;
    // operands of k-group s + 1 are read while the 27 MFMAs of k-group s run (the reads of a group issued right in
    // front of its MFMAs left ~150 cycles of LDS latency exposed four times per tile)
    float a_cur[TI], b_cur[9], a_nxt[TI], b_nxt[9];
    auto frag = [&](int s, float (&a)[TI], float (&b)[9]) {
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = hz[(4 * s) * COUT + 16 * i];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        b[t] = hx[(ky * ROW + 4 * s * 16) * 4 + xe[kx]];
      }
    };
    WSTAMP(tcount < 10 ? 6 * tcount + 2 : 64);
    frag(0, a_cur, b_cur);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) frag(s + 1, a_nxt, b_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[i], b_cur[t], acc[t][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TI; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int t = 0; t < 9; ++t) b_cur[t] = b_nxt[t];
    }
    WSTAMP(tcount < 10 ? 6 * tcount + 3 : 64);
    dma_barrier();
    WSTAMP(tcount < 10 ? 6 * tcount + 4 : 64);
    n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
  }

  // ---- block reduction: sum the 4 row-waves of each ci half, 3 taps per round, through LDS -------
  f32x4* sR = sH;     // [wave 8][k 3*TI][lane 64]
  constexpr int RK = 3 * TI;
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < TI; ++i) sR[(wid * RK + t * TI + i) * 64 + lane] = acc[rd * 3 + t][i];
    __syncthreads();
    for (int e = tid; e < 2 * RK * 64; e += NT) {
      const int ln = e & 63, k = (e >> 6) % RK, c = e / (64 * RK);
      f32x4 s4 = sR[((c * 4 + 0) * RK + k) * 64 + ln];
      s4 += sR[((c * 4 + 1) * RK + k) * 64 + ln];
      s4 += sR[((c * 4 + 2) * RK + k) * 64 + ln];
      s4 += sR[((c * 4 + 3) * RK + k) * 64 + ln];
      const int tap = rd * 3 + k / TI, ti = k % TI;
      const int ci = 16 * c + (ln & 15), co = 16 * ti + 4 * (ln >> 4);
      *reinterpret_cast<f32x4*>(part + ((long long)tap * CIN + ci) * COUT + co) = s4;
    }
  }
  // ---- bias gradient: per-thread dz sums -> LDS [pixel slot][COUT] -> fixed-order column sums ------
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NDZ; ++i)
    if (tid + NT * i < DZ_F4) sZ[tid + NT * i] = dbsum[i];
  __syncthreads();
  if (tid < COUT) {
    const float* zf = reinterpret_cast<const float*>(sZ);
    float s1 = 0.f;
    for (int px = 0; px < TH * TW; ++px) s1 += zf[px * COUT + tid];
    part[9ll * CIN * COUT + tid] = s1;
  }
  __syncthreads();     // the next segment stages into the images this one's sums were just read from
  }
}

void geeco_launch_wgrad_reduce(const float* part, float* dw, float* db, long long gs_dw, long long gs_db, int S,
                               long long KC, int Cout, int groups, hipStream_t s);

// Persistent one-block-per-CU kernels of the encoder bottom (conv2's filter gradient, the fused conv2-dgrad + conv1-wgrad):
// slices per encoder.  The entry points' `reserved_cus` argument k leaves k CUs free for a collective that runs beside them (data parallel:
// the early gradient bucket is reduced while these two kernels run; a grid that occupies every CU would make the
// collective's workgroups wait for - or delay - the persistent blocks).  The workspace is sized for k = 0.
struct BottomSlices {
  int S0, per, S, blocks;
};
// S0 = CUs / groups regular blocks per encoder.  When that leaves CUs over (three encoders on 256 CUs: one) and the tiles
// the regular blocks leave over (T mod S0 per encoder) fit ONE more block of the same length, that block takes them:
// bench shape, fused bottom: 85 x 49 tiles with the last two blocks short or empty and the 256th CU idle becomes
// 85 x 48 + 1 x (3 x 16); conv2's filter gradient 97 -> 96 tiles per block.  Otherwise ceil(T / S0) tiles per block.
static BottomSlices bottom_slices(int groups, long long T, bool for_ws = false) {
  static const int no_rem = geeco_dev_getenv("GEECO_NO_REMAINDER_BLOCK") ? 1 : 0;
  const int cus = 256 - (for_ws ? 0 : geeco_call_reserved_cus());
  BottomSlices b;
  b.S0 = cus / groups < 1 ? 1 : cus / groups;
  const long long fl = T / b.S0, rem = T - fl * b.S0;
  if (for_ws) {                       // upper bound over all T
    b.per = 0; b.S = b.S0 + 1; b.blocks = b.S0 * groups + 1;
    return b;
  }
  if (!no_rem && rem > 0 && fl >= 1 && cus - b.S0 * groups >= 1 && rem * groups <= fl) {
    b.per = (int)fl; b.S = b.S0 + 1; b.blocks = b.S0 * groups + 1;
  } else {
    b.per = (int)((T + b.S0 - 1) / b.S0); b.S = b.S0; b.blocks = b.S0 * groups;
  }
  return b;
}

int64_t geeco_halo_wgrad_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride) {
  if (stride == 2 && Cin == 32 && Cout == 48 && H % 2 == 0 && W % 2 == 0)
    return (int64_t)groups * bottom_slices(groups, 0, true).S * (9ll * Cin * Cout + Cout) * 4;
  return 0;
}

int geeco_try_halo_wgrad(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x,
                         int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout,
                         int stride, void* ws, hipStream_t stream, int* handled) {
  *handled = 0;
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  if (disabled) return 0;
  if (stride == 2 && Cin == 32 && Cout == 48 && (H % 2 == 0) && (W % 2 == 0)) {
    HaloWgradParams p = {};
    p.x = x; p.dz = dz; p.part = (float*)ws; p.gs_x = gs_x; p.gs_dz = gs_dz;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
    p.tiles_per_group = N * p.tiles_x * p.tiles_y;
    const BottomSlices bs = bottom_slices(groups, p.tiles_per_group);
    p.S = bs.S; p.S0 = bs.S0; p.per = bs.per; p.groups = groups;
#ifdef GEECO_STAMPS
    if (!g_hstamps) (void)hipMalloc(&g_hstamps, 256 * 2 * 64 * 8);
    (void)hipMemset(g_hstamps, 0, 256 * 2 * 64 * 8);
    p.stamps = g_hstamps;
#endif
    constexpr int HALO_F4 = ((9 * 17 * 16 + 63) / 64) * 64;
    const size_t lds = (size_t)(2 * HALO_F4 + 2 * 4 * 16 * 12) * 16;
    static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_wgrad_kernel<32, 48>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);

      if (e != hipSuccess) {
        geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      attr_set = true;
    }
    geeco_note_kernel("conv_s2_halo_wgrad_kernel<32, 48>");
    hipLaunchKernelGGL((conv_s2_halo_wgrad_kernel<32, 48>), dim3((unsigned)bs.blocks), dim3(512), lds, stream, p);
    GEECO_LAUNCH_CHECK();
    geeco_launch_wgrad_reduce((const float*)ws, dw, db, gs_dw, gs_db, p.S, 9ll * Cin * Cout, Cout, groups, stream);
    GEECO_LAUNCH_CHECK();
    *handled = 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv2-type input gradient with an LDS halo (stride 2, CIN == 32, COUT % 16 == 0, even H/W), fused
// with the ReluGrad of the layer below:   dx[y][x][ci] = (ymask > 0) * sum_{taps} dz[oy][ox][:] . w[tap][ci][:]
// For TF SAME / stride 2 / even sizes (pad_before = 0) input row y = 2 Y' + py receives
//   py = 0: (ky = 0, oy = Y'), (ky = 2, oy = Y' - 1);   py = 1: (ky = 1, oy = Y')          (same in x),
// i.e. four parity classes with 4 / 2 / 2 / 1 taps.  A block owns 8 x 64 input pixels (4 x 32 per
// class), stages the 5 x 33 dz halo as [co/4][row][col] float4 planes and keeps the HWIO kernel
// resident as [tap][ci][15 float4] rows (b128 B-fragments straight from the TF layout; in this unfused kernel pitch 15 =
// -1 mod 16 => at most one 2-way conflict per read).  Wave = (column half, Y' row): 4 classes x 2
// ci tiles = 8 accumulator tiles, 27 (tap, 16-co block) steps of 8 MFMAs per tile.
// ------------------------------------------------------------------------------------------------
struct HaloDgradParams {
  const float* dz;
  const float* w;      // HWIO [G][9][CIN][COUT]
  const float* mask;   // [G][N][H][W][CIN] or null
  const unsigned short* fields;   // FIELDS kernels: sign fields of the mask tensor (see HaloFwdParams) instead of mask
  long long gs_fields;
  int fHp, fWp;
  float* dx;
  long long gs_dz, gs_w, gs_dx;
  int N, H, W, Ho, Wo;
  int tiles_x, tiles_y;
  long long ntiles;
  int tiles_per_group;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(512, 2) void conv_s2_halo_dgrad_kernel(const HaloDgradParams p) {
  constexpr int NT = 512;
  constexpr int COQ = COUT / 4;
  constexpr int KB = COUT / 16;
  constexpr int HR = 5, HC = 33;                       // dz halo rows / cols
  constexpr int PLANE = HR * HC;                       // 165 float4
  constexpr int SKEW = 0;                              // (the fused-bottom kernel below skews its planes; this one is off the step's path)
  constexpr int HALO_F4 = COQ * PLANE;
  constexpr int NLOAD = (HALO_F4 + NT - 1) / NT;
  constexpr int WP = 15;                               // float4 pitch of a (tap, ci) kernel row
  constexpr int W_F4 = 9 * CIN * WP;
  static_assert(CIN == 32 && COQ <= WP, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                               // 2 halo buffers

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row = wid & 3, half = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const long long per = (p.ntiles + gridDim.x - 1) / gridDim.x;
  long long tile = (long long)blockIdx.x * per;
  const long long tend = tile + per < p.ntiles ? tile + per : p.ntiles;
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

  int l_off[NLOAD], l_src[NLOAD];
  short l_hy[NLOAD], l_hx[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    int idx = tid + NT * i;
    int pix = idx / COQ, cq = idx - pix * COQ;
    int hy = pix / HC, hx = pix - hy * HC;
    l_hy[i] = (short)hy; l_hx[i] = (short)hx;
    l_off[i] = (idx < HALO_F4) ? cq * PLANE + hy * HC + hx : -1;
    l_src[i] = (hy * p.Wo + hx) * COUT + cq * 4;
  }
  f32x4 stage[NLOAD];
  auto load_halo = [&](int g_, int n_, int ty_, int tx_) {
    const int oy0 = ty_ * 4 - 1, ox0 = tx_ * 32 - 1;
    const float* zg = p.dz + (long long)g_ * p.gs_dz + (((long long)n_ * p.Ho + oy0) * p.Wo + ox0) * COUT;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      int oy = oy0 + l_hy[i], ox = ox0 + l_hx[i];
      bool v = l_off[i] >= 0 && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
      stage[i] = v ? *reinterpret_cast<const f32x4*>(zg + l_src[i]) : zero4;
    }
  };
  auto load_weights = [&](int g_) {
    const f32x4* wg = reinterpret_cast<const f32x4*>(p.w + (long long)g_ * p.gs_w);
    for (int e = tid; e < 9 * CIN * COQ; e += NT) {
      int rowi = e / COQ, c4 = e - rowi * COQ;
      sW[rowi * WP + c4] = wg[e];
    }
  };

  load_halo(g, n, ty, tx);
  load_weights(g);
  int g_w = g;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i)
    if (l_off[i] >= 0) sH[l_off[i]] = stage[i];
  __syncthreads();

  // lane r = X' column inside the wave's 16-column strip; q selects the co quad of a 16-co block
  const int a_lane = q * PLANE + SKEW * (q >> 1) + (row + 1) * HC + 16 * half + r + 1;    // + kb*4*PLANE + dy*HC + dx
  const int b_lane = r * WP + q;                                          // + (tap*CIN + 16 cit)*WP + 4 kb
  int buf = 0;
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) {
      advance(g2, n2, ty2, tx2);
      load_halo(g2, n2, ty2, tx2);
    }
    // ReluGrad mask of this wave's 4 x 2 output float4s: issued now, consumed in the epilogue
    const int yb = 2 * (ty * 4 + row), xb = 2 * (tx * 32 + 16 * half + r);
    f32x4 mk[4][2];
    bool okc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int y = yb + (c >> 1), x = xb + (c & 1);
      okc[c] = y < p.H && x < p.W;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        mk[c][t] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (p.mask && okc[c])
          mk[c][t] = *reinterpret_cast<const f32x4*>(p.mask + (long long)g * p.gs_dx +
                                                     (((long long)n * p.H + y) * p.W + x) * CIN + 16 * t + 4 * q);
      }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[c][t] = zero4;
    const f32x4* hA = sH + buf * HALO_F4 + a_lane;
    const f32x4* hB = sW + b_lane;
    f32x4* hN = sH + (buf ^ 1) * HALO_F4;

    // static schedule: 9 taps x KB blocks; tap (ky, kx) feeds class (py, px) = (ky & 1, kx & 1) with
    // source offset dy = -(ky >> 1), dx = -(kx >> 1)
    f32x4 a_cur, b_cur[2], a_nxt, b_nxt[2];
    auto frag = [&](int it, f32x4& a, f32x4 (&b)[2]) {
      const int tap = it / KB, kb = it - tap * KB;
      const int ky = tap / 3, kx = tap - ky * 3;
      a = hA[kb * (4 * PLANE + 2 * SKEW) - (ky >> 1) * HC - (kx >> 1)];
      b[0] = hB[(tap * CIN) * WP + 4 * kb];
      b[1] = hB[(tap * CIN + 16) * WP + 4 * kb];
    };
    constexpr int NIT = 9 * KB;
    frag(0, a_cur, b_cur);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + 1 < NIT) frag(it + 1, a_nxt, b_nxt);
      if (more && it >= NIT - NLOAD - 4 && it < NIT - 4) {
        const int j = it - (NIT - NLOAD - 4);
        if (l_off[j] >= 0) hN[l_off[j]] = stage[j];
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int tap = it / KB;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int c = (ky & 1) * 2 + (kx & 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_cur[t][s], a_cur[s], acc[c][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
      b_cur[0] = b_nxt[0];
      b_cur[1] = b_nxt[1];
    }
    // epilogue: class c -> pixel (yb + py, xb + px); lane owns ci = 16 t + 4 q .. +3
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!okc[c]) continue;
      const int y = yb + (c >> 1), x = xb + (c & 1);
      float* o = p.dx + (long long)g * p.gs_dx + (((long long)n * p.H + y) * p.W + x) * CIN + 4 * q;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 v = acc[c][t];
        const f32x4 m = mk[c][t];
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<f32x4*>(o + 16 * t) = v;
      }
    }
    if (!more) break;
    lds_barrier();      // next halo complete; everyone is done with this buffer (and sW)
    if (g2 != g_w) {
      load_weights(g2);
      g_w = g2;
      __syncthreads();
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
    ++tile;
  }
}

// ------------------------------------------------------------------------------------------------
// The same input gradient with the dz halo cut into 16-channel chunks (conv3: 48 <- 64 channels; any
// CIN % 16 == 0, COUT % 16 == 0 whose kernel fits LDS).  With 64 output channels the whole halo
// (5 x 33 pixels x 256 B, twice) no longer fits beside the resident kernel (9 x 48 rows x 256 B), so a
// tile is processed in COUT / 16 steps: step (tile, chunk) reads the chunk image [hy 5][hx 33][4 quads]
// (quads XOR-swizzled by (hx >> 1) & 3: the b128 reads of 16 consecutive columns are conflict free), while
// the next step's image lands in the other buffer by LDS-DMA (4 lanes fetch a pixel's 64 contiguous bytes;
// no VGPR staging).  Accumulators (4 parity classes x CIN / 16 tiles) live across the chunks of a tile.
// Why it pays: the gather GEMM re-fetches dz once per tap beyond L2 (PMC: 1.1 GB per conv3 launch for a
// 100 MB tensor) and runs at the per-CU miss rate of the vector memory path; here dz is read once.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, bool FIELDS>
__global__ __launch_bounds__(512) void conv_s2_halo_dgrad_chunked_kernel(const HaloDgradParams p) {
  constexpr int NT = 512;
  constexpr int NCH = COUT / 16;                       // chunks (steps) per tile
  constexpr int TCI = CIN / 16;                        // ci tiles per wave
  constexpr int HR = 5, HC = 33;                       // dz halo rows / cols
  // Chunk image of the dz halo, q-major: [co quad q of the chunk][pixel] with the pixel planes padded to a multiple of 16
  // granules, and kernel rows at a pitch of 2 (mod 16) granules: a ds_read_b128 is served in four groups of 16 lanes, each
  // holding every r = lane & 15 once from two neighbouring q; both pitches put the two q of a group on disjoint
  // 16-granule phases (conflict free).  The pixel-major image with an XOR swizzle and the odd row pitch of round 1 measured
  // 43 % LDS bank-conflict cycles.
  constexpr int NPIX = HR * HC;                        // 165
  constexpr int NPIXP = (NPIX + 15) / 16 * 16;         // 176
  constexpr int IMG_F4 = 4 * NPIXP;                    // 704 float4 per chunk image
  constexpr int NPIECE = (IMG_F4 + 63) / 64;           // 11 DMA pieces
  constexpr int BUF_F4 = NPIECE * 64;
  constexpr int NSLOT = (NPIECE + 7) / 8;              // pieces per wave
  constexpr int WP = COUT / 4 + 2;                     // float4 pitch of a (tap, ci) kernel row
  constexpr int W_F4 = 9 * CIN * WP;
  static_assert(CIN % 16 == 0 && COUT % 16 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                               // 2 chunk images

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int row = wid & 3, half = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const long long per = (p.ntiles + gridDim.x - 1) / gridDim.x;
  long long tile = (long long)blockIdx.x * per;
  const long long tend = tile + per < p.ntiles ? tile + per : p.ntiles;
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

  // this wave's DMA pieces: piece k = wid + 8 i covers image slots [64 k, 64 k + 64); lane -> (hy, hx, quad)
  __builtin_assume(wid >= 0 && wid < 8);
  int d_src[NSLOT];
  short d_hy[NSLOT], d_hx[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int sl = (wid + 8 * i) * 64 + lane;
    const int quad = sl / NPIXP, pix = sl - quad * NPIXP;
    const int hy = pix / HC, hx = pix - hy * HC;
    d_hy[i] = (short)((sl < IMG_F4 && pix < NPIX) ? hy : 30000);   // out-of-range marker fails the per-tile bounds test
    d_hx[i] = (short)hx;
    d_src[i] = (hy * p.Wo + hx) * COUT + quad * 4;
  }
  auto dma_chunk = [&](int buf, int g_, int n_, int ty_, int tx_, int chunk) {
    const int oy0 = ty_ * 4 - 1, ox0 = tx_ * 32 - 1;
    const float* zg = p.dz + (long long)g_ * p.gs_dz + (((long long)n_ * p.Ho + oy0) * p.Wo + ox0) * COUT + chunk * 16;
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (8 * (i + 1) <= NPIECE || wid + 8 * i < NPIECE) {      // compile-time true except in the last slot (wid < 8)
        const int oy = oy0 + d_hy[i], ox = ox0 + d_hx[i];
        const bool v = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
        const float* src = v ? zg + d_src[i] : g_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sH + buf * BUF_F4 + (wid + 8 * i) * 64), 16, 0, 0);
      }
    }
  };
  auto load_weights = [&](int g_) {
    const f32x4* wg = reinterpret_cast<const f32x4*>(p.w + (long long)g_ * p.gs_w);
    constexpr int COQ = COUT / 4;
    for (int e = tid; e < 9 * CIN * COQ; e += NT) {
      int rowi = e / COQ, c4 = e - rowi * COQ;
      sW[rowi * WP + c4] = wg[e];
    }
  };

  dma_chunk(0, g, n, ty, tx, 0);
  load_weights(g);
  int g_w = g;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // lane r = X' column inside the wave's 16-column strip; q selects the co quad of the chunk
  const int hx_lane = 16 * half + r + 1;                                  // - dx
  const int b_lane = r * WP + q;                                          // + (tap*CIN + 16 t)*WP + 4 chunk
  static_assert(NCH % 2 == 0, "the chunk loop is unrolled with the LDS buffer index = chunk & 1: a tile must take an even number of chunks");
  // Deferred output stores: the (masked) results of tile t are kept in registers and leave one float4 at a time from inside the
  // tap loops of tile t + 1 (class c in chunk c * NCH / 4, after taps 1, 4, 7, ...).  In a block all eight waves reach the
  // end of a tile together, so an epilogue of 4 x TCI stores per lane is time in which no MFMA issues: measured with the
  // stores removed, 204-212 -> 179-180 us at the bench shape.
  f32x4 pend[4][TCI];
  float* pend_o[4];
  bool pend_ok[4] = {false, false, false, false};
#pragma unroll
  for (int c = 0; c < 4; ++c) pend_o[c] = p.dx;
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) advance(g2, n2, ty2, tx2);
    // ReluGrad mask of this wave's 4 x TCI output float4s: issued now, consumed in the epilogue
    const int yb = 2 * (ty * 4 + row), xb = 2 * (tx * 32 + 16 * half + r);
    // FIELDS: one 16-bit sign field per class pixel (this lane's quad q: bit 4 t + j <-> channel 16 t + 4 q + j) instead
    // of TCI float4 of the activation itself: 4 two-byte loads per tile and lane instead of 12 sixteen-byte ones (the
    // field array is padded to whole tiles, so no bounds logic on the load)
    f32x4 mk[FIELDS ? 1 : 4][TCI];
    unsigned short mf[4];
    bool okc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int y = yb + (c >> 1), x = xb + (c & 1);
      okc[c] = y < p.H && x < p.W;
      if constexpr (FIELDS) {
        mf[c] = p.fields[(long long)g * p.gs_fields + (((long long)n * p.fHp + y) * p.fWp + x) * 4 + q];
      } else {
#pragma unroll
        for (int t = 0; t < TCI; ++t) {
          mk[c][t] = f32x4{1.f, 1.f, 1.f, 1.f};
          if (p.mask && okc[c])
            mk[c][t] = *reinterpret_cast<const f32x4*>(p.mask + (long long)g * p.gs_dx +
                                                       (((long long)n * p.H + y) * p.W + x) * CIN + 16 * t + 4 * q);
        }
      }
    }
    f32x4 acc[4][TCI];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < TCI; ++t) acc[c][t] = zero4;

    // fully unrolled: the buffer index is a compile-time constant, so the fragment addresses are loop-invariant
    // registers + immediates instead of a dozen VALU adds per chunk (VALU work is paid in MFMA time)
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const int buf = chunk & 1;
      // the next step's image lands in the other buffer while this one is consumed
      if (chunk + 1 < NCH)
        dma_chunk(buf ^ 1, g, n, ty, tx, chunk + 1);
      else if (more)
        dma_chunk(buf ^ 1, g2, n2, ty2, tx2, 0);
      const f32x4* hA = sH + buf * BUF_F4;
      const f32x4* hB = sW + b_lane + 4 * chunk;
      // static schedule: tap (ky, kx) feeds class (py, px) = (ky & 1, kx & 1) from the dz pixel at
      // (row + 1 - (ky >> 1), column - (kx >> 1))
      f32x4 a_cur, b_cur[TCI], a_nxt, b_nxt[TCI];
      auto frag = [&](int tap, f32x4& a, f32x4 (&b)[TCI]) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int hy = row + 1 - (ky >> 1), hx = hx_lane - (kx >> 1);
        a = hA[q * NPIXP + hy * HC + hx];
#pragma unroll
        for (int t = 0; t < TCI; ++t) b[t] = hB[(tap * CIN + 16 * t) * WP];
      };
      frag(0, a_cur, b_cur);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) frag(tap + 1, a_nxt, b_nxt);
        __builtin_amdgcn_sched_barrier(0);
        {
          const int ky = tap / 3, kx = tap - ky * 3;
          const int c = (ky & 1) * 2 + (kx & 1);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < TCI; ++t)
              acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_cur[t][s], a_cur[s], acc[c][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          // pending results of the previous tile: class pc's float4 number pt leaves behind tap min(3 pt + 1, 8) of its chunk
#pragma unroll
          for (int pc = 0; pc < 4; ++pc)
#pragma unroll
            for (int pt = 0; pt < TCI; ++pt)
              if (pc * NCH / 4 == chunk) {
                // (both waves of a SIMD store behind the same taps: staggering them by a tap measured 198-201 us against 193-197;
                // all of a class's stores behind tap 0, or behind taps 0, 1, 2: 201 / 196-203)
                if ((3 * pt + 1 < 8 ? 3 * pt + 1 : 8) == tap && pend_ok[pc]) stream_store<3>(pend_o[pc] + 16 * pt, pend[pc][pt]);
              }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < TCI; ++t) b_cur[t] = b_nxt[t];
      }
      if (chunk + 1 < NCH || more) dma_barrier();   // next image landed; everyone is done with this one
    }
    // epilogue: class c -> pixel (yb + py, xb + px); lane owns ci = 16 t + 4 q .. +3
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!okc[c]) continue;
      const int y = yb + (c >> 1), x = xb + (c & 1);
      float* o = p.dx + (long long)g * p.gs_dx + (((long long)n * p.H + y) * p.W + x) * CIN + 4 * q;
#pragma unroll
      for (int t = 0; t < TCI; ++t) {
        f32x4 v = acc[c][t];
        if constexpr (FIELDS) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[j] = __int_as_float(__float_as_int(v[j]) & __builtin_amdgcn_sbfe((int)mf[c], 4 * t + j, 1));
        } else {
          const f32x4 m = mk[c][t];
          v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
          v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        pend[c][t] = v;
      }
      pend_o[c] = o;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) pend_ok[c] = okc[c];
    if (!more) {        // the block's last tile: nothing left to hide behind
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (pend_ok[c]) {
#pragma unroll
          for (int t = 0; t < TCI; ++t) stream_store<3>(pend_o[c] + 16 * t, pend[c][t]);
        }
    }
    if (!more) break;
    if (g2 != g_w) {          // (the barrier above already separated everyone from the old kernel)
      load_weights(g2);
      g_w = g2;
      __syncthreads();
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    ++tile;
  }
}

template <int CIN, int COUT, bool FIELDS = false>
static int launch_dgrad_chunked(HaloDgradParams& p, hipStream_t stream) {
  constexpr int BUF_F4 = ((5 * 33 * 4 + 63) / 64) * 64;
  const size_t lds = (size_t)(9 * CIN * (COUT / 4 + 2) + 2 * BUF_F4) * 16;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_dgrad_chunked_kernel<CIN, COUT, FIELDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const long long cus = 256 - geeco_call_reserved_cus();     // data parallel: CUs left to the collective that runs beside part 2
  long long blocks = p.ntiles < cus ? p.ntiles : cus;
  geeco_note_kernel("conv_s2_halo_dgrad_chunked_kernel<%d, %d, %s>", CIN, COUT, FIELDS ? "true" : "false");
  hipLaunchKernelGGL((conv_s2_halo_dgrad_chunked_kernel<CIN, COUT, FIELDS>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv2 input gradient FUSED with conv1's filter/bias gradient (encoder bottom: conv1 4->32 s1, conv2 32->48 s2).
//   dz1 = (y1 > 0) * conv2_dgrad(dz2)          dw1[(tap, c)][co] = sum_p x[p + tap][c] dz1[p][co]     db1 = sum_p dz1[p]
// conv1 has no input gradient (its input is data), so dz1 has exactly one consumer; produced and consumed inside one
// kernel it never goes to HBM: the unfused pair writes and re-reads 805 MB per step and needs a second launch.
// Built on conv_s2_halo_dgrad_kernel (same tiles, resident kernel, dz2 halo, MFMA schedule).  Per tile the block
// also stages the 10 x 66 halo of conv1's input x (RGB padded to 4 channels) by LDS-DMA, double buffered.
// The dgrad MFMAs take the dz2 halo as the A operand (rows = 16 pixels of one parity class) and the kernel as B
// (columns = 16 conv1 channels), so a lane (r, q) ends up with dz1[pixels 4 q + 0..3][channel r]: register s of that
// accumulator IS the B operand (k = pixel, j = channel) of the filter-gradient MFMA whose k-group s takes the pixels
// {4 q + s}: no transposition, no LDS staging between the two products.  The A operand of that MFMA (rows = the 27
// (tap, RGB) columns) is read from the x halo at the same permuted pixels.  4 accumulator tiles (2 column tiles x 2
// channel tiles) live for the block's whole tile range; at the end the 8 waves are summed through LDS into one slab
// per block (wgrad_reduce_kernel adds the slabs in a fixed order).
// ------------------------------------------------------------------------------------------------
struct FusedBottomParams {
  const float* dz;      // dz2 [G][N][Ho][Wo][48]
  const float* w;       // conv2 kernel HWIO [G][9][32][48]
  const float* mask;    // y1 [G][N][H][W][32]
  const unsigned* bits; // BITS kernels: y1's ReLU sign bits [G][N][Hp][Wp] (geeco_conv1_fwd_relu_bits) instead of y1
  long long gs_bits;
  int Wp, Hp;
  const float* x;       // conv1 input [G][N][H][W][4]
  float* part;          // [G][S][9*CREAL*32 + 32]
  float* dx;            // optional: also store dz1 (null in training)
  long long gs_dz, gs_w, gs_y, gs_x;
  int N, H, W, Ho, Wo;
  int tiles_x, tiles_y, tiles_per_group, S;
  int S0, per, groups;          // slicing as HaloWgradParams (bottom_slices())
  unsigned long long* stamps;   // -DGEECO_STAMPS builds only (scripts/dev/fused_stamps.py)
};

constexpr int FB_WP = 14, FB_PLANE = 176, FB_XW = 67, FB_XPIECES = (10 * FB_XW + 63) / 64;
constexpr size_t FB_LDS_BYTES = (size_t)(9 * 32 * FB_WP + 2 * 12 * FB_PLANE + 2 * FB_XPIECES * 64) * 16;

#ifdef GEECO_STAMPS
#define FSTAMP(i)                                                                                        \
  do {                                                                                                   \
    if (lane == 0 && (wid & 3) == 0 && g == 0 && p.stamps && (i) < 64)                                   \
      p.stamps[((long long)split * 2 + (wid >> 2)) * 64 + (i)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define FSTAMP(i)
#endif

// CREAL = real input channels of conv1 (3: RGB padded to 4, the pad column is skipped; 4: RGB-D)
// BITS: the ReluGrad mask comes as one sign-bit word per pixel (2 KB per tile) instead of y1 itself (64 KB per tile,
// 805 MB per step: the kernel's largest read by far, and what its waves queue behind in the vector-memory pipe)
template <int CREAL, bool BITS>
__global__ __launch_bounds__(512) void conv2_dgrad_conv1_wgrad_kernel(const FusedBottomParams p) {
  constexpr int CIN = 32, COUT = 48;
  constexpr int NT = 512;
  constexpr int COQ = COUT / 4;
  constexpr int KB = COUT / 16;
  constexpr int HR = 5, HC = 33;                       // dz2 halo rows / cols
  // A ds_read_b128 is served in four groups of 16 lanes, each with every r = lane & 15 once from two neighbouring
  // q = lane >> 4 (MI355X_MICROARCH.md, LDS): the group is conflict free iff both q hit the same 16-granule phase, i.e.
  // the plane pitch of the dz2 halo ([co quad q][row][col]) is a multiple of 16 granules (165 -> 176; measured
  // SQ_LDS_BANK_CONFLICT 49 % of the LDS cycles with 165), and the kernel row pitch WP gives (WP r + q) mod 16 distinct
  // over a group: 14 does (even phases for one q, odd for the other), 15 left one 2-way conflict per read.
  constexpr int PLANE_USED = HR * HC;                  // 165
  constexpr int PLANE = FB_PLANE;                      // 176
  // The staging stores (ds_write_b128: 8 consecutive lanes per LDS cycle group, banks = dword address mod 32, i.e. granule
  // mod 8) put the 8 - 12 co quads of ONE halo pixel side by side: with every plane at the same phase all 8 lanes of a
  // group hit one granule slot mod 8 (8-way conflict: 64 instead of 8 LDS cycles per store, 32 such stores per tile from
  // the 8 waves inside four MFMA steps; round-2 PMC: 41.8 % of this kernel's LDS cycles were conflict cycles).  The read
  // side only needs the planes of a quad PAIR (4 kb + {0, 1}, 4 kb + {2, 3}) at one phase mod 16, so pair m is skewed by
  // SKEW * m granules inside its 176-granule slot: the stores are 2-way (16 LDS cycles, under their 13-cycle issue cost)
  // and the fragment reads stay conflict free.
  constexpr int SKEW = 2;
  static_assert(PLANE >= PLANE_USED + SKEW * (COQ / 2 - 1) && PLANE % 16 == 0, "plane pitch");
  constexpr int HALO_USED = COQ * PLANE_USED;          // 1980 granules are loaded
  constexpr int HALO_F4 = COQ * PLANE;                 // 2112 granules per buffer
  constexpr int NLOAD = (HALO_USED + NT - 1) / NT;
  constexpr int WP = FB_WP;
  constexpr int W_F4 = 9 * CIN * WP;                   // 4032
  // x halo of the 8 x 64 pixel tile (conv1: stride 1, pad 1): 10 rows x 66 pixels, one pixel = RGB0 = one granule.
  // Row pitch 67 pixels = 268 floats = 12 (mod 64): the 16 (tap, channel) offsets a ds_read_b32 spreads over its r
  // lanes then span < 32 banks in each column tile and the pixel stride between q neighbours is 32 floats.
  constexpr int XH = 10, XW = FB_XW, XUSED = 66;
  constexpr int X_F4 = XH * XW;                        // 670 granules
  constexpr int NXP = FB_XPIECES;                      // 11 DMA pieces
  constexpr int SX_F4 = NXP * 64;                      // 704 (padded)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);
  f32x4* sH = sW + W_F4;                               // 2 dz2 halo buffers
  f32x4* sX = sH + 2 * HALO_F4;                        // 2 x halo buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int row = wid & 3, half = wid >> 2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // regular blocks: `per` tiles of one encoder; the remainder block: what they leave over, encoder after encoder
  const int nreg = p.S0 * p.groups;
  const bool regular = (int)blockIdx.x < nreg;
  const int nseg = regular ? 1 : p.groups;
#pragma unroll 1
  for (int seg = 0; seg < nseg; ++seg) {
  const int g = regular ? (int)blockIdx.x / p.S0 : seg;
  const int split = regular ? (int)blockIdx.x - g * p.S0 : p.S0;
  int tile = regular ? split * p.per : p.S0 * p.per;
  const int tend = regular && tile + p.per < p.tiles_per_group ? tile + p.per : p.tiles_per_group;
  const long long slab = 9 * CREAL * 32 + 32;       // [tap][real channel][32] + bias
  float* part = p.part + ((long long)g * p.S + split) * slab;
  int n, ty, tx;
  {
    int per_img = p.tiles_x * p.tiles_y;
    n = tile / per_img;
    int rem = tile - n * per_img;
    ty = rem / p.tiles_x;
    tx = rem - ty * p.tiles_x;
  }
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    if (++tx_ == p.tiles_x) {
      tx_ = 0;
      if (++ty_ == p.tiles_y) {
        ty_ = 0;
        ++n_;
      }
    }
  };

  // dz2 halo staging (register staged, as conv_s2_halo_dgrad_kernel)
  int l_off[NLOAD], l_src[NLOAD];
  short l_hy[NLOAD], l_hx[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    int idx = tid + NT * i;
    int pix = idx / COQ, cq = idx - pix * COQ;
    int hy = pix / HC, hx = pix - hy * HC;
    // lanes beyond the halo: row marker that fails every bounds test (they fetch the zero page) and a pad granule of
    // plane 0 as their LDS slot, so that neither the load nor the store needs a predicate
    l_hy[i] = (short)(idx < HALO_USED ? hy : 30000); l_hx[i] = (short)hx;
    l_off[i] = (idx < HALO_USED) ? cq * PLANE + SKEW * (cq >> 1) + hy * HC + hx : PLANE_USED + (tid & 7);
    l_src[i] = (hy * p.Wo + hx) * COUT + cq * 4;
  }
  f32x4 stage[NLOAD];
  auto load_halo_i = [&](int i, int n_, int ty_, int tx_) {
    const int oy0 = ty_ * 4 - 1, ox0 = tx_ * 32 - 1;
    const float* zg = p.dz + (long long)g * p.gs_dz + (((long long)n_ * p.Ho + oy0) * p.Wo + ox0) * COUT;
    int oy = oy0 + l_hy[i], ox = ox0 + l_hx[i];
    bool v = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
    stage[i] = *reinterpret_cast<const f32x4*>(v ? zg + l_src[i] : g_zero_page);     // TF SAME zero padding
  };
  auto load_halo = [&](int n_, int ty_, int tx_) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) load_halo_i(i, n_, ty_, tx_);
  };
  // x halo: LDS-DMA, pieces wid and wid + 8 (11 pieces): lane -> halo pixel (hy, hx), row-major with pitch XW
  auto dma_x = [&](int n_, int ty_, int tx_, f32x4* dst) {
    const int y0 = ty_ * 8 - 1, x0 = tx_ * 64 - 1;
    const float* xg = p.x + (long long)g * p.gs_x + (long long)n_ * p.H * p.W * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = wid + 8 * i;                       // wave-uniform
      if (k < NXP) {
        const int idx = k * 64 + lane;
        const int hy = idx / XW, hx = idx - hy * XW;
        const int iy = y0 + hy, ix = x0 + hx;
        const bool v = idx < X_F4 && hx < XUSED && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = v ? xg + ((long long)iy * p.W + ix) * 4 : g_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + k * 64), 16, 0, 0);
      }
    }
  };
  {
    const f32x4* wg = reinterpret_cast<const f32x4*>(p.w + (long long)g * p.gs_w);
    for (int e = tid; e < 9 * CIN * COQ; e += NT) {
      int rowi = e / COQ, c4 = e - rowi * COQ;
      sW[rowi * WP + c4] = wg[e];
    }
  }
  if (tile < tend) {
    dma_x(n, ty, tx, sX);
    load_halo(n, ty, tx);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) sH[l_off[i]] = stage[i];
  }

  // conv1 wgrad: only the 9 * CREAL real (tap, channel) columns are computed (RGB: the 4th input channel is
  // padding): lane's columns jj = 16 tj + r = CREAL tap + c and their offsets inside the x halo
  constexpr int NCOL = 9 * CREAL;
  constexpr int NJ = (NCOL + 15) / 16;
  int xoff[NJ];
#pragma unroll
  for (int tj = 0; tj < NJ; ++tj) {
    const int jj = 16 * tj + r;
    const int tap = jj / CREAL, c = jj - tap * CREAL;
    const int ky = tap / 3, kx = tap - ky * 3;
    xoff[tj] = jj < NCOL ? ((ky * XW + kx) << 2) + c : 0;    // columns >= NCOL: any valid address, never stored
  }
  f32x4 accw[NJ][2];    // [column tile][channel tile]: lane (r, q) register k = dw1[column 16 j + 4 q + k][channel 16 t + r]
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) accw[j][t] = zero4;
  float dbl[2] = {0.f, 0.f};

  // BITS: sign words of the wave's outputs, 8 consecutive pixels x = xb .. xb + 7 of both rows (class pixel (px, k) <->
  // word px + 2 k); those of the NEXT tile are loaded during the current tile's MFMA loop (mbn)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 mb[2][2], mbn[2][2];
  // (rows and columns of the word array are padded to whole tiles and the padding is zero: no bounds logic here)
  auto bits_row = [&](int py, int n_, int ty_, int tx_) {
    const int y = 2 * (ty_ * 4 + row) + py;
    return p.bits + (long long)g * p.gs_bits + ((long long)n_ * p.Hp + y) * p.Wp + 2 * (tx_ * 32 + 16 * half + 4 * q);
  };
  if constexpr (BITS) {
    if (tile < tend) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        mb[m >> 1][m & 1] = *reinterpret_cast<const u32x4*>(bits_row(m >> 1, n, ty, tx) + 4 * (m & 1));
    }
  }
  dma_barrier();
  const int a_lane = q * PLANE + SKEW * (q >> 1) + (row + 1) * HC + 16 * half + r + 1;
  const int b_lane = r * WP + q;
  int buf = 0;
  int tcount = 0;
  for (; tile < tend; ++tile, ++tcount) {
    const bool more = tile + 1 < tend;
    int n2 = n, ty2 = ty, tx2 = tx;
    FSTAMP(tcount < 10 ? 6 * tcount + 0 : 64);
    if (more) advance(n2, ty2, tx2);
    // ReluGrad mask of this wave's outputs in the accumulator layout: class c = (py, px), pixel j = 4 q + k of the
    // wave's 16 class pixels (x = 2 (tx 32 + 16 half + j) + px), channel 16 t + r.  A wave that issues all its global
    // loads at the top of the tile sits in vector-memory back-pressure for ~4 k cycles (in-kernel timeline,
    // scripts/dev/fused_stamps.py) while its SIMD's MFMA pipe idles, so every global load of the tile - the next
    // dz2 halo, the next x halo, the 32 mask dwords - is issued from inside the MFMA loop, two per step.
    // Pixels outside the image contribute nothing to dw1: the load address is clamped, the value zeroed by okf.
    const int yb = 2 * (ty * 4 + row), xb = 2 * (tx * 32 + 16 * half + 4 * q);
    const float* mbase[2];
    float oky[2];
    int moff[2][4];
    float okx[2][4];
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int y = yb + py, yc = y < p.H ? y : p.H - 1;
      mbase[py] = p.mask + (long long)g * p.gs_y + ((long long)n * p.H + yc) * p.W * CIN;
      oky[py] = y < p.H ? 1.f : 0.f;
    }
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = xb + px + 2 * k, xc = x < p.W ? x : p.W - 1;
        moff[px][k] = xc * CIN + r;
        okx[px][k] = x < p.W ? 1.f : 0.f;
      }
    f32x4 mk[BITS ? 1 : 4][2];
    float xv[4][4][NJ];       // x halo operands of the filter-gradient MFMAs, read one step before their class starts
    const float* xt = reinterpret_cast<const float*>(sX + buf * SX_F4);
    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[c][t] = zero4;
    const f32x4* hA = sH + buf * HALO_F4 + a_lane;
    const f32x4* hB = sW + b_lane;
    f32x4* hN = sH + (buf ^ 1) * HALO_F4;
    f32x4 a_cur, b_cur[2], a_nxt, b_nxt[2];
    // Tap order: the parity classes of the input gradient complete one after the other - class 3 = (odd, odd) has the
    // centre tap only, class 2 taps 3 / 5, class 1 taps 1 / 7, class 0 the four corners - so that (BITS) the
    // filter-gradient work of a finished class (mask, 16 MFMAs, 4 per step) runs inside the loop next to the taps of
    // the following class instead of as a latency-bound tail after it.
    constexpr int ORDER[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
    auto frag = [&](int it, f32x4& a, f32x4 (&b)[2]) {
      const int tap = ORDER[it / KB], kb = it % KB;
      const int ky = tap / 3, kx = tap - ky * 3;
      a = hA[kb * (4 * PLANE + 2 * SKEW) - (ky >> 1) * HC - (kx >> 1)];
      b[0] = hB[(tap * CIN) * WP + 4 * kb];
      b[1] = hB[(tap * CIN + 16) * WP + 4 * kb];
    };
    constexpr int NIT = 9 * KB;
    f32x4 vv[2];              // dz1 of the class in flight: lane (r, q) register k = pixel 4 q + k, channel 16 t + r
    auto class_x = [&](int c) {         // class c's x operands: k-group s <-> class pixel 4 q + s
      const int py = c >> 1, px = c & 1;
      const float* xs = xt + (((2 * row + py) * XW + 2 * (16 * half + 4 * q) + px) << 2);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NJ; ++j) xv[c][s][j] = xs[((2 * s) << 2) + xoff[j]];
    };
    auto class_mask = [&](int c) {      // ReluGrad of class c's finished accumulators -> vv, bias gradient
      const int py = c >> 1, px = c & 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (BITS) {
          // channel 16 t + r <-> bit (r & 3) * 8 + (r >> 2) + 4 t of the pixel's word; bfe_i32 gives 0 / all ones
          // (pixels outside the image have zero words: they contribute nothing to dw1)
          const int wi = px + 2 * k;
          const unsigned word = mb[py][wi >> 2][wi & 3];
          const int sh = (r & 3) * 8 + (r >> 2);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int keep = __builtin_amdgcn_sbfe((int)word, sh + 4 * t, 1);
            vv[t][k] = __int_as_float(__float_as_int(acc[c][t][k]) & keep);
            dbl[t] += vv[t][k];
          }
        } else {
          const float okf = okx[px][k] * oky[py];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            vv[t][k] = mk[c][t][k] * okf > 0.f ? acc[c][t][k] : 0.f;
            dbl[t] += vv[t][k];
          }
        }
      }
      if (p.dx) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int y = yb + py, x = xb + px + 2 * k;
          if (y < p.H && x < p.W) {
            float* o = p.dx + (long long)g * p.gs_y + (((long long)n * p.H + y) * p.W + x) * CIN + r;
            o[0] = vv[0][k];
            o[16] = vv[1][k];
          }
        }
      }
    };
    auto class_mfma = [&](int c, int s) {   // k-group s of class c: k index q <-> class pixel 4 q + s
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          accw[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[c][s][j], vv[t][s], accw[j][t], 0, 0, 0);
    };
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(tcount < 10 ? 6 * tcount + 1 : 64);
    frag(0, a_cur, b_cur);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it + 1 < NIT) frag(it + 1, a_nxt, b_nxt);
      if (more && it >= NIT - NLOAD - 4 && it < NIT - 4) {
        const int j = it - (NIT - NLOAD - 4);
        hN[l_off[j]] = stage[j];
      }
      if (more && it < NLOAD) load_halo_i(it, n2, ty2, tx2);
      if (more && it == NLOAD) dma_x(n2, ty2, tx2, sX + (buf ^ 1) * SX_F4);
      if constexpr (BITS) {
        if (more && it > NLOAD && it <= NLOAD + 4) {  // the NEXT tile's sign words: 4 x 16 B per lane
          const int m = it - NLOAD - 1;
          mbn[m >> 1][m & 1] = *reinterpret_cast<const u32x4*>(bits_row(m >> 1, n2, ty2, tx2) + 4 * (m & 1));
        }
        if (it >= NIT - 4) class_x(it - (NIT - 4));
      } else {
        if (it > NLOAD && it <= NLOAD + 16) {         // two mask dwords per step: (class, pixel k) of both channel tiles
          const int m = it - NLOAD - 1, c = m >> 2, k = m & 3;
          const float* mp = mbase[c >> 1] + moff[c & 1][k];
          mk[c][0][k] = mp[0];
          mk[c][1][k] = mp[16];
        }
        if (it >= NIT - 4) class_x(it - (NIT - 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int tap = ORDER[it / KB];
        const int ky = tap / 3, kx = tap - ky * 3;
        const int c = (ky & 1) * 2 + (kx & 1);
        // D[i = pixel][j = channel]: A = dz2 halo (i = pixel r, k = co quad q), B = kernel (k, j = channel r)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b_cur[t][s], acc[c][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
      b_cur[0] = b_nxt[0];
      b_cur[1] = b_nxt[1];
    }
    FSTAMP(tcount < 10 ? 6 * tcount + 2 : 64);
    // ---- the classes still open: ReluGrad, then conv1's filter gradient straight from the accumulators ----------
#pragma unroll
    for (int c = 3; c >= 0; --c) {     // class order 3, 2, 1, 0 (the filter-gradient work of a finished class INSIDE the MFMA loop measured 484 vs 466 us: profiles/NEGATIVE_RESULTS.md)
      class_mask(c);
#pragma unroll
      for (int s = 0; s < 4; ++s) class_mfma(c, s);
    }
    if constexpr (BITS) {
      if (more) {
#pragma unroll
        for (int m = 0; m < 4; ++m) mb[m >> 1][m & 1] = mbn[m >> 1][m & 1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(tcount < 10 ? 6 * tcount + 3 : 64);
    dma_barrier();      // end of tile: next dz2 / x halos complete; everyone is done with this tile's buffers
    FSTAMP(tcount < 10 ? 6 * tcount + 4 : 64);
    n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
  }

  // ---- block reduction of conv1's gradient: [wave 8][2 NJ tiles][64 lanes] float4 (<= 48 KB) in the dz2 halo area ----
  __syncthreads();
  f32x4* sR = sH;
  constexpr int NTL = 2 * NJ;
  float* sD = reinterpret_cast<float*>(sR + 8 * NTL * 64);    // [wave 8][32 channels]
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) sR[(wid * NTL + j * 2 + t) * 64 + lane] = accw[j][t];
  // bias gradient: lane (r, q) holds the sum over its pixels for channels r and 16 + r; fold the 4 q lanes
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    dbl[t] += __shfl_xor(dbl[t], 16);
    dbl[t] += __shfl_xor(dbl[t], 32);
  }
  if (q == 0) {
    sD[wid * 32 + r] = dbl[0];
    sD[wid * 32 + 16 + r] = dbl[1];
  }
  __syncthreads();
  for (int e = tid; e < NTL * 64; e += NT) {
    const int ln = e & 63, k = e >> 6;
    f32x4 s4 = sR[(0 * NTL + k) * 64 + ln];
#pragma unroll
    for (int w = 1; w < 8; ++w) s4 += sR[(w * NTL + k) * 64 + ln];
    const int j = k >> 1, t = k & 1;
    const int co = 16 * t + (ln & 15);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int jj = 16 * j + 4 * (ln >> 4) + kk;
      if (jj < NCOL) part[jj * 32 + co] = s4[kk];     // slab layout [tap][real channel][32]: row jj
    }
  }
  if (tid < 32) {
    float s1 = 0.f;
    for (int w = 0; w < 8; ++w) s1 += sD[w * 32 + tid];
    part[NCOL * 32 + tid] = s1;
  }
  __syncthreads();     // the next segment stages into the area this one's sums were just read from
  }
}

extern "C" int64_t geeco_conv2_dgrad_conv1_wgrad_ws_bytes(int groups) {
  return (int64_t)groups * bottom_slices(groups, 0, true).S * (9 * 4 * 32 + 32) * 4;
}

template <int CREAL, bool BITS>
static hipError_t fused_bottom_attr(size_t lds) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2_dgrad_conv1_wgrad_kernel<CREAL, BITS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// y1 (ReluGrad mask = conv1's output) or y1_bits (its sign bits): exactly one is given
static int fused_bottom_impl(const float* dz2, const float* w2, const float* y1, const uint32_t* y1_bits,
                             const float* x, float* dw1, float* db1, float* dz1, int groups, int64_t gs_dz2,
                             int64_t gs_w2, int64_t gs_y1, int64_t gs_bits, int64_t gs_x, int64_t gs_dw1, int64_t gs_db1,
                             int N, int H, int W, int real_channels, void* ws, void* stream) {
  GEECO_CHECK_ARG(dz2 && w2 && (y1 || y1_bits) && x && dw1 && db1 && ws, "conv2_dgrad_conv1_wgrad: null pointer");
  GEECO_CHECK_ARG(real_channels == 3 || real_channels == 4, "conv2_dgrad_conv1_wgrad: real_channels = %d (3 or 4)",
                  real_channels);
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "conv2_dgrad_conv1_wgrad: H = %d, W = %d must be even", H, W);
  FusedBottomParams p = {};
  p.dz = dz2; p.w = w2; p.mask = y1; p.x = x; p.part = (float*)ws; p.dx = dz1;
  p.bits = y1_bits; p.gs_bits = gs_bits; p.Wp = (int)geeco_relu_bits_pitch(W); p.Hp = (int)geeco_relu_bits_rows(H);
  p.gs_dz = gs_dz2; p.gs_w = gs_w2; p.gs_y = gs_y1; p.gs_x = gs_x;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.tiles_x = cdiv(W, 64); p.tiles_y = cdiv(H, 8);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  const BottomSlices bs = bottom_slices(groups, p.tiles_per_group);
  p.S = bs.S; p.S0 = bs.S0; p.per = bs.per; p.groups = groups;
  const size_t lds = FB_LDS_BYTES;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = fused_bottom_attr<3, false>(lds);
    if (e == hipSuccess) e = fused_bottom_attr<4, false>(lds);
    if (e == hipSuccess) e = fused_bottom_attr<3, true>(lds);
    if (e == hipSuccess) e = fused_bottom_attr<4, true>(lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
#ifdef GEECO_STAMPS
  if (!g_hstamps) (void)hipMalloc(&g_hstamps, 256 * 2 * 64 * 8);
  (void)hipMemset(g_hstamps, 0, 256 * 2 * 64 * 8);
  p.stamps = g_hstamps;
#endif
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)bs.blocks);
  const bool bits = y1_bits != nullptr;
  geeco_note_kernel("conv2_dgrad_conv1_wgrad_kernel<%d, %s>", real_channels == 3 ? 3 : 4, bits ? "true" : "false");
  if (real_channels == 3 && bits)
    hipLaunchKernelGGL((conv2_dgrad_conv1_wgrad_kernel<3, true>), grid, dim3(512), lds, s, p);
  else if (real_channels == 3)
    hipLaunchKernelGGL((conv2_dgrad_conv1_wgrad_kernel<3, false>), grid, dim3(512), lds, s, p);
  else if (bits)
    hipLaunchKernelGGL((conv2_dgrad_conv1_wgrad_kernel<4, true>), grid, dim3(512), lds, s, p);
  else
    hipLaunchKernelGGL((conv2_dgrad_conv1_wgrad_kernel<4, false>), grid, dim3(512), lds, s, p);
  GEECO_LAUNCH_CHECK();
  geeco_launch_wgrad_reduce((const float*)ws, dw1, db1, gs_dw1, gs_db1, p.S, 9 * real_channels * 32, 32, groups, s);
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_conv2_dgrad_conv1_wgrad(const float* dz2, const float* w2, const float* y1, const float* x,
                                             float* dw1, float* db1, float* dz1, int groups, int64_t gs_dz2,
                                             int64_t gs_w2, int64_t gs_y1, int64_t gs_x, int64_t gs_dw1,
                                             int64_t gs_db1, int N, int H, int W, int real_channels, void* ws,
                                             void* stream) {
  GEECO_CHECK_ARG(y1, "conv2_dgrad_conv1_wgrad: null y1");
  return fused_bottom_impl(dz2, w2, y1, nullptr, x, dw1, db1, dz1, groups, gs_dz2, gs_w2, gs_y1, 0, gs_x, gs_dw1, gs_db1,
                           N, H, W, real_channels, ws, stream);
}

extern "C" int geeco_conv2_dgrad_conv1_wgrad_partial(const float* dz2, const float* w2, const float* y1, const float* x,
                                                     float* dw1, float* db1, float* dz1, int groups, int64_t gs_dz2,
                                                     int64_t gs_w2, int64_t gs_y1, int64_t gs_x, int64_t gs_dw1,
                                                     int64_t gs_db1, int N, int H, int W, int real_channels, void* ws,
                                                     void* stream, geeco_slab_reduce* pending, int reserved_cus) {
  GEECO_CHECK_ARG(pending && y1, "conv2_dgrad_conv1_wgrad_partial: null pending / y1");
  if (int e = geeco_enter_reserved_cus(reserved_cus)) return e;
  geeco_slab_reduce none = {};
  *pending = none;
  geeco_set_pending_reduce(pending);
  const int rc = fused_bottom_impl(dz2, w2, y1, nullptr, x, dw1, db1, dz1, groups, gs_dz2, gs_w2, gs_y1, 0, gs_x, gs_dw1,
                                   gs_db1, N, H, W, real_channels, ws, stream);
  geeco_set_pending_reduce(nullptr);
  geeco_leave_reserved_cus();
  return rc;
}

extern "C" int geeco_conv2_dgrad_conv1_wgrad_bits(const float* dz2, const float* w2, const uint32_t* y1_bits,
                                                  const float* x, float* dw1, float* db1, int groups, int64_t gs_dz2,
                                                  int64_t gs_w2, int64_t gs_bits, int64_t gs_x, int64_t gs_dw1,
                                                  int64_t gs_db1, int N, int H, int W, int real_channels, void* ws,
                                                  void* stream, geeco_slab_reduce* pending, int reserved_cus) {
  GEECO_CHECK_ARG(y1_bits, "conv2_dgrad_conv1_wgrad_bits: null y1_bits");
  if (int e = geeco_enter_reserved_cus(reserved_cus)) return e;
  if (pending) {
    geeco_slab_reduce none = {};
    *pending = none;
    geeco_set_pending_reduce(pending);
  }
  const int rc = fused_bottom_impl(dz2, w2, nullptr, y1_bits, x, dw1, db1, nullptr, groups, gs_dz2, gs_w2, 0, gs_bits, gs_x,
                                   gs_dw1, gs_db1, N, H, W, real_channels, ws, stream);
  if (pending) geeco_set_pending_reduce(nullptr);
  geeco_leave_reserved_cus();
  return rc;
}

// does the dispatcher below take this shape (given the HWIO kernel)?  Such layers never read the transposed copy.
int geeco_halo_dgrad_handles(int H, int W, int Cin, int Cout, int stride) {
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  static const int no_chunked = geeco_dev_getenv("GEECO_NO_HALO3") ? 1 : 0;
  if (disabled || stride != 2 || (H % 2) || (W % 2)) return 0;
  return (!no_chunked && Cin == 48 && Cout == 64) || (Cin == 32 && Cout == 48);
}

int geeco_try_halo_dgrad(const float* dz, const float* w_hwio, const float* ymask, float* dx, int groups,
                         int64_t gs_dz, int64_t gs_w, int64_t gs_dx, int N, int H, int W, int Cin, int Cout,
                         int stride, hipStream_t stream, int* handled) {
  *handled = 0;
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  if (disabled || !w_hwio) return 0;
  static const int no_chunked = geeco_dev_getenv("GEECO_NO_HALO3") ? 1 : 0;
  if (!no_chunked && stride == 2 && Cin == 48 && Cout == 64 && (H % 2 == 0) && (W % 2 == 0)) {
    HaloDgradParams p = {};
    p.dz = dz; p.w = w_hwio; p.mask = ymask; p.dx = dx;
    p.gs_dz = gs_dz; p.gs_w = gs_w; p.gs_dx = gs_dx;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = cdiv(W, 64); p.tiles_y = cdiv(H, 8);
    p.tiles_per_group = N * p.tiles_x * p.tiles_y;
    p.ntiles = (long long)groups * p.tiles_per_group;
    int rc = launch_dgrad_chunked<48, 64>(p, stream);
    if (rc) return rc;
    GEECO_LAUNCH_CHECK();
    *handled = 1;
    return 0;
  }
  if (stride == 2 && Cin == 32 && Cout == 48 && (H % 2 == 0) && (W % 2 == 0)) {
    HaloDgradParams p = {};
    p.dz = dz; p.w = w_hwio; p.mask = ymask; p.dx = dx;
    p.gs_dz = gs_dz; p.gs_w = gs_w; p.gs_dx = gs_dx;
    p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
    p.tiles_x = cdiv(W, 64); p.tiles_y = cdiv(H, 8);
    p.tiles_per_group = N * p.tiles_x * p.tiles_y;
    p.ntiles = (long long)groups * p.tiles_per_group;
    const size_t lds = (size_t)(9 * 32 * 15 + 2 * 12 * 165) * 16;
    static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_halo_dgrad_kernel<32, 48>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      attr_set = true;
    }
    long long blocks = p.ntiles < 256 ? p.ntiles : 256;
    geeco_note_kernel("conv_s2_halo_dgrad_kernel<32, 48>");
    hipLaunchKernelGGL((conv_s2_halo_dgrad_kernel<32, 48>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
    GEECO_LAUNCH_CHECK();
    *handled = 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv1-type kernels: stride 1, CIN == 4 (RGB padded, or RGB-D), COUT == 32.
// K per tap is exactly one MFMA step (4 channels), so the 9 taps are 9 MFMA k-steps.
// Forward is bound by the 128 B/pixel output stream (805 MB per step at N = 32), wgrad by reading
// it back: both keep the tiny input halo in LDS and the kernel in registers.
// ------------------------------------------------------------------------------------------------
struct Conv1FwdParams {
  const float* x;       // [G][N][H][W][4]
  const float* w;       // [G][9][w_cin][32]: w_cin = 4 (padded copy) or 3 (the RGB variable itself, channel 3 taken as zero)
  const float* bias;
  float* y;             // [G][N][H][W][32]
  unsigned* bits;       // optional [G][N][Hp][Wp]: ReLU sign bits of y (geeco_conv1_fwd_relu_bits), else null
  long long gs_x, gs_w, gs_b, gs_y, gs_bits;
  int N, H, W, tiles_x, tiles_y, relu, Wp, Hp, w_cin;
};

// PACK3 (RGB, w_cin == 3): the 27 real (tap, channel) products are packed into 7 MFMA k-steps (k = 3 tap + c = 4 s + q)
// instead of 9 steps of (R, G, B, pad): 14 MFMAs and 7 fragment reads per 16-pixel strip instead of 18 / 9.  The dropped
// terms are exact zeros (pad channel x zero weight), added in the same order before: y is bitwise unchanged.  The kernel
// is issue bound, not only HBM bound (PMC: MFMA busy 51 %, 61 % of the wave time issuing), so the MFMAs saved show.
template <bool PACK3>
__global__ __launch_bounds__(256) void conv1_halo_fwd_kernel(const Conv1FwdParams p) {
  // Persistent blocks: a block keeps the kernel fragments and bias of its encoder in registers and walks the
  // tiles t = blockIdx.x, + gridDim.x, ...; the next tile's halo is fetched into registers before the current
  // tile is computed and lands in the other LDS buffer afterwards.  One block per tile (24 576 blocks of a few
  // microseconds each) was bound by workgroup dispatch and by the exposed latency of every block's own halo
  // and kernel loads: load, MFMA and store time simply added up (ablation: 267 = 90 + 52 + 85 + 56 us).
  constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH = TH + 2;
  constexpr int NPX = HH * HW_;                    // 340 halo pixels
  constexpr int NLD = (NPX + 255) / 256;           // float4 per thread (2)
  __shared__ __attribute__((aligned(16))) float sX[2][NPX * 4];
  __shared__ __attribute__((aligned(16))) float sO[4][16 * 36];   // per wave: [pixel][32 + 4 pad]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const int per_img = p.tiles_x * p.tiles_y, ntiles = p.N * per_img;
  int t = blockIdx.x;
  if (t >= ntiles) return;
  const float* xg0 = p.x + (long long)g * p.gs_x;
  // kernel fragments: lane (r = co, q = channel) of tap tp, co tile i
  const float* wg = p.w + (long long)g * p.gs_w;
  constexpr int NS = PACK3 ? 7 : 9;
  float wf[NS][2];
  int xo[NS];          // float offset of the lane's x operand of step s inside a strip's halo window (without the pixel r)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (PACK3) {
      const int kk = 4 * s + q;                    // = 3 tap + c
      const bool v = kk < 27;
      const int tap = v ? kk / 3 : 0, c = v ? kk - tap * 3 : 0;
      const int ky = tap / 3, kx = tap - ky * 3;
      xo[s] = ((ky * HW_ + kx) << 2) + c;
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[s][i] = v ? wg[kk * 32 + i * 16 + r] : 0.f;      // w [9][3][32]
    } else {
      const int ky = s / 3, kx = s - ky * 3;
      xo[s] = ((ky * HW_ + kx) << 2) + q;
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[s][i] = q < p.w_cin ? wg[(s * p.w_cin + q) * 32 + i * 16 + r] : 0.f;
    }
  }
  f32x4 bias_r[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) bias_r[i] = *reinterpret_cast<const f32x4*>(p.bias + (long long)g * p.gs_b + i * 16 + 4 * q);

  f32x4 stage[NLD];
  auto load_halo = [&](int tt) {
    const int n = tt / per_img, rem = tt - n * per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* xg = xg0 + (long long)n * p.H * p.W * 4;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      const int hy = i / HW_, hx = i - hy * HW_;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;        // TF SAME, stride 1: pad 1 on every side
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = *reinterpret_cast<const f32x4*>(xg + ((long long)iy * p.W + ix) * 4);
      stage[k] = v;
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 256 * k;
      if (i < NPX) *reinterpret_cast<f32x4*>(&sX[buf][i * 4]) = stage[k];
    }
  };
  load_halo(t);
  store_halo(0);
  __syncthreads();
  float* so = sO[wid];
  int buf = 0;
  for (;;) {
    const int t2 = t + gridDim.x;
    const bool more = t2 < ntiles;
    if (more) load_halo(t2);
    const int n = t / per_img, rem = t - n * per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    float* yg = p.y + (long long)g * p.gs_y + (long long)n * p.H * p.W * 32;
    // wave w: rows 2w, 2w+1; 2 column halves => 4 strips of 16 pixels.  The 16 x 32 output strip is 2 KB
    // contiguous in NHWC memory: it is transposed through LDS so that each store instruction writes 1 KB
    // of consecutive bytes (lane l -> pixel l / 8 (+8), channel quad l % 8) instead of 16 separate 64 B pieces.
    unsigned myword = 0;
    const float* xt[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) xt[s] = &sX[buf][((2 * wid * HW_ + r) << 2) + xo[s]];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int oyl = 2 * wid + (st >> 1), oxl0 = 16 * (st & 1);
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      // operand address = (per-lane part, formed once per tile) + (strip part: a compile-time immediate)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float xv = xt[s][(((st >> 1) * HW_ + 16 * (st & 1)) << 2)];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i], xv, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = acc[i] + bias_r[i];
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4*>(so + r * 36 + i * 16 + 4 * q) = v;      // lane owns pixel r, channels 16 i + 4 q
      }
      // same-wave LDS round trip: the compiler's lgkmcnt wait orders the reads behind the writes
      const int oy = y0 + oyl;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int px = 8 * h + (lane >> 3), c4 = lane & 7;
        const f32x4 v = *reinterpret_cast<const f32x4*>(so + px * 36 + c4 * 4);
        const int ox = x0 + oxl0 + px;
        if (oy < p.H && ox < p.W) stream_store<0>(yg + ((long long)oy * p.W + ox) * 32 + c4 * 4, v);
        if (p.bits) {
          // sign bits of the 8 pixels this instruction stores: a compare IS a ballot (lane = 8 pixel + channel quad),
          // so byte `pixel` of the four masks holds the bits of channels 4 c4 + {0, 1, 2, 3}; every lane assembles
          // the word of pixel lane & 7 (bit (c & 3) * 8 + (c >> 2) <-> channel c) and the lanes 8 (2 st + h) + j keep
          // it: after the four strips lane L holds the word of tile pixel (row 2 wid + (L >> 5), column L & 31)
          const unsigned long long bx = __ballot(v.x > 0.f), by = __ballot(v.y > 0.f), bz = __ballot(v.z > 0.f),
                                   bw = __ballot(v.w > 0.f);
          // byte lane & 7 of each 64-bit mask, placed in byte 0 / 1 / 2 / 3: one v_perm_b32 each (a 64-bit shift by a
          // per-lane amount is quarter rate and made this HBM-bound kernel 11 % slower)
          const unsigned j = lane & 7;
          const unsigned word = __builtin_amdgcn_perm((unsigned)(bx >> 32), (unsigned)bx, 0x0c0c0c00u | j) |
                                __builtin_amdgcn_perm((unsigned)(by >> 32), (unsigned)by, 0x0c0c000cu | (j << 8)) |
                                __builtin_amdgcn_perm((unsigned)(bz >> 32), (unsigned)bz, 0x0c000c0cu | (j << 16)) |
                                __builtin_amdgcn_perm((unsigned)(bw >> 32), (unsigned)bw, 0x000c0c0cu | (j << 24));
          if ((lane >> 3) == 2 * st + h) myword = word;
        }
      }
    }
    if (p.bits) {         // one coalesced store per wave: 2 rows x 32 words
      const int oy = y0 + 2 * wid + (lane >> 5), ox = x0 + (lane & 31);
      if (oy < p.H && ox < p.W) p.bits[(long long)g * p.gs_bits + ((long long)n * p.Hp + oy) * p.Wp + ox] = myword;
    }
    if (!more) break;
    store_halo(buf ^ 1);
    lds_barrier();      // the other buffer is complete; everyone is done with this one
    buf ^= 1;
    t = t2;
  }
}

static int launch_conv1_fwd(const float* x, const float* w, const float* b, float* y, unsigned* bits, int groups,
                            int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_bits, int N, int H, int W,
                            int relu, hipStream_t stream, int w_cin = 4);

int geeco_conv1_fwd_handles(int Cin, int Cout, int stride) {
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  return !disabled && stride == 1 && Cin == 4 && Cout == 32;
}

int geeco_try_conv1_fwd(const float* x, const float* w, const float* b, float* y, int groups, int64_t gs_x,
                        int64_t gs_w, int64_t gs_b, int64_t gs_y, int N, int H, int W, int Cin, int Cout, int stride,
                        int relu, hipStream_t stream, int* handled) {
  *handled = 0;
  if (!b || !geeco_conv1_fwd_handles(Cin, Cout, stride)) return 0;
  *handled = 1;
  return launch_conv1_fwd(x, w, b, y, nullptr, groups, gs_x, gs_w, gs_b, gs_y, 0, N, H, W, relu, stream);
}

// ---- ReLU sign fields of conv2's output for conv3's input gradient (see HaloFwdParams::fields) ---------------------
extern "C" int64_t geeco_relu_fields_elems(int N, int H, int W) {    // uint16 elements per encoder; H, W of the 48-channel tensor
  return (int64_t)N * ((H + 7) / 8 * 8) * ((W + 63) / 64 * 64) * 4;
}

extern "C" int geeco_conv2_fwd_relu_fields(const float* x, const float* w, const float* b, float* y, uint16_t* fields,
                                           int groups, int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y,
                                           int64_t gs_fields, int N, int H, int W, void* stream) {
  GEECO_CHECK_ARG(x && w && b && y && fields, "conv2_fwd_relu_fields: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "conv2_fwd_relu_fields: H = %d, W = %d must be even", H, W);
  HaloFwdParams p = {};
  p.x = x; p.w = w; p.bias = b; p.y = y;
  p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_y = gs_y;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.ntiles = (long long)groups * p.tiles_per_group;
  p.relu = 1;
  p.fields = fields; p.gs_fields = gs_fields; p.fHp = (p.Ho + 7) / 8 * 8; p.fWp = (p.Wo + 63) / 64 * 64;
  return launch_s2_halo_fwd_ws<32, 48, 4>(p, (hipStream_t)stream);
}

extern "C" int geeco_conv3_dgrad_relu_fields(const float* dz, const float* w, const uint16_t* y2_fields, float* dx,
                                             int groups, int64_t gs_dz, int64_t gs_w, int64_t gs_fields, int64_t gs_dx,
                                             int N, int H, int W, void* stream, int reserved_cus) {
  GEECO_CHECK_ARG(dz && w && y2_fields && dx, "conv3_dgrad_relu_fields: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "conv3_dgrad_relu_fields: H = %d, W = %d must be even", H, W);
  HaloDgradParams p = {};
  p.dz = dz; p.w = w; p.mask = nullptr; p.dx = dx;
  p.fields = y2_fields; p.gs_fields = gs_fields; p.fHp = (H + 7) / 8 * 8; p.fWp = (W + 63) / 64 * 64;
  p.gs_dz = gs_dz; p.gs_w = gs_w; p.gs_dx = gs_dx;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.tiles_x = cdiv(W, 64); p.tiles_y = cdiv(H, 8);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.ntiles = (long long)groups * p.tiles_per_group;
  if (int e = geeco_enter_reserved_cus(reserved_cus)) return e;
  int rc = launch_dgrad_chunked<48, 64, true>(p, (hipStream_t)stream);
  geeco_leave_reserved_cus();
  if (rc) return rc;
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- ... and of conv3's output for conv4's input gradient (byte fields, see HaloFwdParams::fields8) ------------------
extern "C" int geeco_conv3_fwd_relu_fields(const float* x, const float* w, const float* b, float* y, uint8_t* fields,
                                           int groups, int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y,
                                           int64_t gs_fields, int N, int H, int W, void* stream) {
  GEECO_CHECK_ARG(x && w && b && y && fields, "conv3_fwd_relu_fields: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "conv3_fwd_relu_fields: H = %d, W = %d must be even", H, W);
  HaloFwdParams p = {};
  p.x = x; p.w = w; p.bias = b; p.y = y;
  p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_y = gs_y;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.ntiles = (long long)groups * p.tiles_per_group;
  p.relu = 1;
  p.fields8 = fields; p.gs_fields8 = gs_fields;
  return launch_s2_halo_fwd_chunked<48, 64>(p, (hipStream_t)stream);
}

extern "C" int64_t geeco_relu_bits_pitch(int W) { return (int64_t)(W + 63) / 64 * 64; }
extern "C" int64_t geeco_relu_bits_rows(int H) { return (int64_t)(H + 7) / 8 * 8; }

extern "C" int geeco_conv1_fwd_relu_bits(const float* x, const float* w, const float* b, float* y, uint32_t* bits,
                                         int groups, int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y,
                                         int64_t gs_bits, int N, int H, int W, void* stream) {
  GEECO_CHECK_ARG(x && w && b && y && bits, "conv1_fwd_relu_bits: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 1 && W >= 1, "conv1_fwd_relu_bits: bad dims");
  return launch_conv1_fwd(x, w, b, y, bits, groups, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H, W, 1, (hipStream_t)stream);
}

// ... reading the RGB model's kernel variable [G][3][3][3][32] as it is stored (x stays channel-padded to 4; the pad
// channel's kernel rows are taken as zero): no padded copy to re-derive after every optimiser step
extern "C" int geeco_conv1_fwd_relu_bits_rgb(const float* x, const float* w3, const float* b, float* y, uint32_t* bits,
                                             int groups, int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y,
                                             int64_t gs_bits, int N, int H, int W, void* stream) {
  GEECO_CHECK_ARG(x && w3 && b && y && bits, "conv1_fwd_relu_bits_rgb: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 1 && W >= 1, "conv1_fwd_relu_bits_rgb: bad dims");
  return launch_conv1_fwd(x, w3, b, y, bits, groups, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H, W, 1, (hipStream_t)stream, 3);
}

static int launch_conv1_fwd(const float* x, const float* w, const float* b, float* y, unsigned* bits, int groups,
                            int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_bits, int N, int H, int W,
                            int relu, hipStream_t stream, int w_cin) {
  Conv1FwdParams p = {};
  p.w_cin = w_cin;
  p.x = x; p.w = w; p.bias = b; p.y = y; p.gs_x = gs_x; p.gs_w = gs_w; p.gs_b = gs_b; p.gs_y = gs_y;
  p.bits = bits; p.gs_bits = gs_bits; p.Wp = (int)geeco_relu_bits_pitch(W); p.Hp = (int)geeco_relu_bits_rows(H);
  p.N = N; p.H = H; p.W = W; p.tiles_x = cdiv(W, 32); p.tiles_y = cdiv(H, 8); p.relu = relu;
  const int ntiles = N * p.tiles_x * p.tiles_y;
  static const int bpg = geeco_dev_getenv("GEECO_C1_BLOCKS") ? atoi(geeco_dev_getenv("GEECO_C1_BLOCKS")) : 768;   // blocks per encoder (256..2048 within 5 %)
  dim3 grid((unsigned)(ntiles < bpg ? ntiles : bpg), (unsigned)groups);
  geeco_note_kernel("conv1_halo_fwd_kernel");
  static const int no_pack = geeco_dev_getenv("GEECO_C1_NO_PACK3") ? 1 : 0;
  if (w_cin == 3 && !no_pack)
    hipLaunchKernelGGL(conv1_halo_fwd_kernel<true>, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(conv1_halo_fwd_kernel<false>, grid, dim3(256), 0, stream, p);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- conv1 filter/bias gradient --------------------------------------------------------------------
// dw[(tap, c)][co] = sum_pixels x[halo(pixel, tap)][c] dz[pixel][co]: MFMA row i = co (2 tiles), column
// j = (tap, c) (36 of 48 = 3 tiles), k = 4 consecutive pixels.  Persistent blocks (wave = row of a
// 4 x 16 tile) keep the 6 accumulator tiles in registers over their whole tile range.
struct Conv1WgradParams {
  const float* x;
  const float* dz;
  float* part;             // [G][S][9*4*32 + 32]
  long long gs_x, gs_dz;
  int N, H, W, tiles_x, tiles_y, tiles_per_group, S;
};

__global__ __launch_bounds__(256) void conv1_halo_wgrad_kernel(const Conv1WgradParams p) {
  constexpr int TH = 4, TW = 16, HWD = TW + 2, HH = TH + 2;
  constexpr int ZP = 48;                                   // dz row pitch (floats): 16 mod 32
  constexpr int XF = HH * HWD * 4, ZF = TH * TW * ZP;
  __shared__ __attribute__((aligned(16))) float smem[2 * (XF + ZF)];
  float* sX = smem;                 // 2 x halo
  float* sZ = smem + 2 * XF;        // 2 x dz tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y, split = blockIdx.x;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int per = (p.tiles_per_group + p.S - 1) / p.S;
  int tile = split * per;
  const int tend = tile + per < p.tiles_per_group ? tile + per : p.tiles_per_group;
  const long long slab = 9 * 4 * 32 + 32;
  float* part = p.part + ((long long)g * p.S + split) * slab;
  int n, ty, tx;
  {
    int per_img = p.tiles_x * p.tiles_y;
    n = tile / per_img;
    int rem = tile - n * per_img;
    ty = rem / p.tiles_x;
    tx = rem - ty * p.tiles_x;
  }
  // staging slots: halo (108 pixels, one float4 each) and dz tile (64 pixels x 8 float4)
  const bool xs = tid < HH * HWD;
  const int x_hy = tid / HWD, x_hx = tid - x_hy * HWD;
  int z_px[2], z_c4[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int idx = tid + 256 * i;
    z_px[i] = idx >> 3;
    z_c4[i] = idx & 7;
  }
  f32x4 xst = zero4, zst[2], dbsum[2] = {zero4, zero4};
  auto load_tile = [&](int n_, int ty_, int tx_) {
    const float* xg = p.x + (long long)g * p.gs_x + (long long)n_ * p.H * p.W * 4;
    const int iy = ty_ * TH + x_hy - 1, ix = tx_ * TW + x_hx - 1;
    xst = (xs && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
              ? *reinterpret_cast<const f32x4*>(xg + ((long long)iy * p.W + ix) * 4) : zero4;
    const float* zg = p.dz + (long long)g * p.gs_dz + (long long)n_ * p.H * p.W * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oy = ty_ * TH + (z_px[i] >> 4), ox = tx_ * TW + (z_px[i] & 15);
      zst[i] = (oy < p.H && ox < p.W) ? *reinterpret_cast<const f32x4*>(zg + ((long long)oy * p.W + ox) * 32 + z_c4[i] * 4)
                                      : zero4;
    }
  };
  auto store_tile = [&](int buf) {
    if (xs) *reinterpret_cast<f32x4*>(sX + buf * XF + tid * 4) = xst;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(sZ + buf * ZF + z_px[i] * ZP + z_c4[i] * 4) = zst[i];
      dbsum[i] += zst[i];
    }
  };
  // lane's three (tap, c) columns: jj = 16 tj + r
  int xoff[3];
  bool xval[3];
#pragma unroll
  for (int tj = 0; tj < 3; ++tj) {
    const int jj = 16 * tj + r;
    const int tap = jj >> 2, c = jj & 3;
    xval[tj] = jj < 36;
    const int ky = tap / 3, kx = tap - ky * 3;
    xoff[tj] = xval[tj] ? ((ky * HWD + kx) << 2) + c : 0;
  }
  f32x4 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zero4;

  if (tile < tend) {
    load_tile(n, ty, tx);
    store_tile(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < tend; ++tile) {
    const bool more = tile + 1 < tend;
    int n2 = n, ty2 = ty, tx2 = tx;
    if (more) {
      if (++tx2 == p.tiles_x) {
        tx2 = 0;
        if (++ty2 == p.tiles_y) {
          ty2 = 0;
          ++n2;
        }
      }
      load_tile(n2, ty2, tx2);
    }
    const float* hx = sX + buf * XF + ((wid * HWD + q) << 2);           // wave = tile row; pixel 4 s + q
    const float* hz = sZ + buf * ZF + (16 * wid + q) * ZP + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float a[2], b[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = hz[(4 * s) * ZP + 16 * i];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float v = hx[((4 * s) << 2) + xoff[j]];
        b[j] = xval[j] ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    lds_barrier();
    n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
  }
  // reduce the 4 waves through LDS: [wave][6 tiles][64 lanes] float4 = 24 KB (fits in the staging area)
  __syncthreads();
  f32x4* sR = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) sR[(wid * 6 + i * 3 + j) * 64 + lane] = acc[i][j];
  __syncthreads();
  for (int e = tid; e < 6 * 64; e += 256) {
    const int ln = e & 63, k = e >> 6;
    f32x4 s4 = sR[(0 * 6 + k) * 64 + ln];
    s4 += sR[(1 * 6 + k) * 64 + ln];
    s4 += sR[(2 * 6 + k) * 64 + ln];
    s4 += sR[(3 * 6 + k) * 64 + ln];
    const int i = k / 3, j = k - i * 3;
    const int jj = 16 * j + (ln & 15), co = 16 * i + 4 * (ln >> 4);
    if (jj < 36) *reinterpret_cast<f32x4*>(part + jj * 32 + co) = s4;
  }
  __syncthreads();
  float* sD = smem;   // [64 pixel slots][32]
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(sD + z_px[i] * 32 + z_c4[i] * 4) = dbsum[i];
  __syncthreads();
  if (tid < 32) {
    float s1 = 0.f;
    for (int px = 0; px < 64; ++px) s1 += sD[px * 32 + tid];
    part[9 * 4 * 32 + tid] = s1;
  }
}

static int conv1_wgrad_S(int groups) {
  int S = 768 / groups;
  return S < 1 ? 1 : S;
}

int64_t geeco_conv1_wgrad_ws_bytes(int groups, int Cin, int Cout, int stride) {
  if (stride == 1 && Cin == 4 && Cout == 32) return (int64_t)groups * conv1_wgrad_S(groups) * (9 * 4 * 32 + 32) * 4;
  return 0;
}

int geeco_try_conv1_wgrad(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x,
                          int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout,
                          int stride, void* ws, hipStream_t stream, int* handled) {
  *handled = 0;
  static const int disabled = geeco_dev_getenv("GEECO_NO_HALO") ? 1 : 0;
  if (disabled || !(stride == 1 && Cin == 4 && Cout == 32)) return 0;
  Conv1WgradParams p = {};
  p.x = x; p.dz = dz; p.part = (float*)ws; p.gs_x = gs_x; p.gs_dz = gs_dz;
  p.N = N; p.H = H; p.W = W; p.tiles_x = cdiv(W, 16); p.tiles_y = cdiv(H, 4);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.S = conv1_wgrad_S(groups);
  geeco_note_kernel("conv1_halo_wgrad_kernel");
  hipLaunchKernelGGL(conv1_halo_wgrad_kernel, dim3((unsigned)p.S, (unsigned)groups), dim3(256), 0, stream, p);
  GEECO_LAUNCH_CHECK();
  geeco_launch_wgrad_reduce((const float*)ws, dw, db, gs_dw, gs_db, p.S, 9 * 4 * 32, 32, groups, stream);
  GEECO_LAUNCH_CHECK();
  *handled = 1;
  return 0;
}
