// Encoder bottom FORWARD in one kernel: conv1 (3x3, stride 1, 4 -> 32, bias, ReLU) -> conv2 (3x3, stride 2, 32 -> 48, bias,
// ReLU) of the reference's conv_encoder (src/models/e2evmc/graph.py:76-85), for the three / K encoders' frames at once.
//
// Why: conv1's output y1 is the largest tensor of the step (805 MB at 96 frames of 256 x 256).  As two launches it is
// written once (conv1 forward, bound by that write stream with the MFMA pipe half idle) and read twice (conv2 forward, conv2
// filter gradient), both readers bound by what one CU ingests beside its MFMA waves (DESIGN 5.5 finding 3).  Here a
// workgroup produces the y1 halo of a conv2 tile in LDS, stores the owned part of it to HBM (the filter gradient of conv2
// still needs it) and feeds conv2's MFMAs from LDS: conv2 forward's 805 MB read and one launch disappear, and conv1's
// store / VALU work runs beside conv2's MFMAs on the same SIMD instead of beside an idle matrix pipe.
//
// Structure (12 waves, one workgroup per CU, persistent over a contiguous tile range):
//   waves 0-7   CONSUMERS = the compute waves of conv_s2_halo_fwd_ws_kernel<32, 48, 4> (conv_halo.hip): conv2 tile of 4 x 16
//               outputs, wave = (output row, K half), kernel fragments in 108 VGPRs, K halves reduced through LDS, output strip
//               transposed through LDS -> 1 KiB stores, 16-bit sign fields for conv3's input gradient.  Their A fragments
//               come from a ring of TWO y1 halo images (9 rows x 33 pixels x 32 channels, the pair-swizzled layout of that
//               kernel) which the producers fill.  The K-half-1 waves also issue the LDS-DMA of the x halo two tiles ahead
//               (11 x 35 pixels x 16 B = 7 pieces of 1 KiB; they have no other vector-memory traffic to wait behind).
//   waves 8-11  PRODUCERS = conv1 for the y1 halo of the NEXT tile: 16 strips of 16 owned pixels (8 rows x 32 columns: wave pw
//               owns rows 2 pw, 2 pw + 1) + 3 strips for the halo's 9th row and 33rd column (recomputed: they belong to the
//               neighbouring tiles, +16 % of conv1's MFMAs = +1.6 % of the pair's), packed-K MFMAs as conv1_halo_fwd_kernel
//               <PACK3>, bias + ReLU, ds_write_b128 into the halo image; the owned pixels are read back from the image in
//               memory order (same wave, no barrier) -> 1 KiB non-temporal stores of y1 + sign words by ballot / v_perm.
//   ONE barrier per tile: producers arrive when halo t + 1 is complete, consumers after their last fragment read of halo t.
// Per SIMD and tile: 2 x 108 + 70 MFMAs (9.2 k cycles) instead of 2 x 108 beside the ingest of 39 KB; the vector-memory pipe
// carries 6 KB in + 44 KB out per tile instead of 39 in + 12 out (conv2) and 6 in + 32 out (conv1).
//
// Results are BITWISE those of the two separate kernels (same MFMA sequences per output, same reduction order), which is
// how tests/test_kernels_gpu.py checks this one.
#include "geeco_common.h"        // -I geeco_amd/csrc (scripts/dev/build_dev_lib.sh)
#include "conv_bottom_fwd.h"
#include <atomic>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ float g_zero_page_bf[64];   // source of LDS-DMA lanes outside the image (TF SAME zero padding of conv1's input)

struct BottomFwdParams {
  const float* x;          // [G][N][H][W][4]
  const float* w1;         // [G][9][w_cin][32]
  const float* b1;
  float* y1;               // [G][N][H][W][32]
  unsigned* bits;          // optional [G][N][Hp][Wp] sign words of y1
  const float* w2;         // [G][9][32][48] (HWIO)
  const float* b2;
  float* y2;               // [G][N][H/2][W/2][48]
  unsigned short* fields;  // optional [G][N][fHp][fWp][4] sign fields of y2
  long long gs_x, gs_w1, gs_b1, gs_y1, gs_bits, gs_w2, gs_b2, gs_y2, gs_fields;
  int N, H, W, Ho, Wo, tiles_x, tiles_y, tiles_per_group, Hp, Wp, fHp, fWp, w_cin;
  long long ntiles;
};

// dev switches (scripts/dev/build_variant.sh): BF_PRIO 1 = producer waves at s_setprio 3, 2 = consumer waves; BF_ABL (WRONG results):
// 1 = no y1 / sign-word stores, 2 = no conv1 MFMAs, 3 = producers only keep the barriers
#ifndef BF_PRIO
#define BF_PRIO 1
#endif
#ifndef BF_ABL
#define BF_ABL 0
#endif
// who stores y1 (+ its sign words): 1 = the CONSUMER waves, from the halo image they are consuming (wave w = row w of the 8 owned
// rows: 4 pieces of 1 KiB spread over its tap loop); 0 = the producer waves right after each strip (first form: the producers,
// one wave per SIMD with dependent write -> read-back -> store chains, became the critical path: +78 us on the launch)
#ifndef BF_CSTORE
#define BF_CSTORE 2
#endif
// who forms y1's sign words when the consumers store y1: 1 = the producers from their accumulators (8 bits per lane, OR over
// the four lanes of a pixel by two wave shuffles), 0 = the consumers by ballots on the pieces they store
#ifndef BF_PBITS
#define BF_PBITS 1
#endif

#ifndef BF_NT
#define BF_NT 3      // bit 0: y1 stores non-temporal, bit 1: y2 stores
#endif
template <int SITE>
__device__ __forceinline__ void out_store(float* dst, const f32x4& v) {
  if constexpr ((BF_NT >> SITE) & 1)
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
  else
    *reinterpret_cast<f32x4*>(dst) = v;
}

template <bool PACK3>
__global__ __launch_bounds__(768) void conv1_conv2_fwd_kernel(const BottomFwdParams p) {
  constexpr int TH = 4, TW = 16, CIN = 32, COUT = 48, TI = COUT / 16;
  constexpr int ROW = 17 * 16;                      // float4 per row of a y1 halo image (17 pixel pairs x 16 slots)
  constexpr int HALO_F4 = 9 * ROW;                  // 2448 float4 = 39 168 B
  constexpr int XW = 2 * TW + 3, XH = 2 * TH + 3;   // x halo: 35 x 11 pixels of one float4
  constexpr int XPX = XW * XH;                      // 385
  constexpr int XPIECES = (XPX + 63) / 64;          // 7 LDS-DMA pieces of 1 KiB
  constexpr int X_F4 = XPIECES * 64;
  constexpr int RED_F4 = 4 * TI * 64;
  constexpr int OP = COUT / 4 + 1;                  // float4 pitch of an output pixel in the y2 store staging (odd)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sH = reinterpret_cast<f32x4*>(smem);       // 2 y1 halo images
  f32x4* sX = sH + 2 * HALO_F4;                     // 2 x halos
  f32x4* sR = sX + 2 * X_F4;                        // 2 K-half reduction buffers
  f32x4* sO = sR + 2 * RED_F4;                      // 4 strips x [16 pixels][OP]
  f32x4* sB = sO + 4 * 16 * OP;                     // conv2's bias of encoder g at sB[(g & 1) * 12 ..]: 12 VGPRs less per consumer lane

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const long long per = (p.ntiles + gridDim.x - 1) / gridDim.x;
  long long tile = (long long)blockIdx.x * per;
  const long long tend = tile + per < p.ntiles ? tile + per : p.ntiles;
  if (tile >= tend) return;
  int g, n, ty, tx;
  {
    g = (int)(tile / p.tiles_per_group);
    int rem = (int)(tile - (long long)g * p.tiles_per_group);
    const int per_img = p.tiles_x * p.tiles_y;
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

  if (wid >= 8) {
    // ===== producers: conv1 for the y1 halo of tile index i + 1 while the consumers run conv2 on tile i ==================
    const int pw = wid - 8;
    if (BF_PRIO == 1) __builtin_amdgcn_s_setprio(3);
    constexpr int NS = PACK3 ? 7 : 9;
    float wf[NS][2];
    int xo[NS];              // float offset of the lane's x operand of k-step s relative to its pixel's halo position
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (PACK3) {
        const int kk = 4 * s + q;                    // = 3 tap + c
        const bool v = kk < 27;
        const int tap = v ? kk / 3 : 0, c = v ? kk - tap * 3 : 0;
        const int ky = tap / 3, kx = tap - ky * 3;
        xo[s] = ((ky * XW + kx) << 2) + c;
      } else {
        const int ky = s / 3, kx = s - ky * 3;
        xo[s] = ((ky * XW + kx) << 2) + q;
      }
    }
    f32x4 bias_r[2];
    auto load_w1 = [&](int g_) {
      const float* wg = p.w1 + (long long)g_ * p.gs_w1;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (PACK3) {
          const int kk = 4 * s + q;
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[s][i] = kk < 27 ? wg[kk * 32 + i * 16 + r] : 0.f;     // w [9][3][32]
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) wf[s][i] = q < p.w_cin ? wg[(s * p.w_cin + q) * 32 + i * 16 + r] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) bias_r[i] = *reinterpret_cast<const f32x4*>(p.b1 + (long long)g_ * p.gs_b1 + i * 16 + 4 * q);
    };
    load_w1(g);
    int g_w = g;
    // image slot of (row, col, channel quad cq): pixel pair col >> 1, 16 slots per pair = 2 pixels x 8 quads, XOR-swizzled
    auto yidx = [&](int row, int col, int cq) {
      const int pair = col >> 1;
      return row * ROW + pair * 16 + ((((col & 1) << 3) | cq) ^ (pair & 15));
    };
    // owned strips: rows 2 pw + {0, 1}, column halves {0, 16}; halo strip: pw 0 / 1 = row 8, columns 0-15 / 16-31;
    // pw 2 = column 32 of rows 0..8 (lanes r > 8 idle); pw 3 has none
    int wi[2][2];            // [column half][i]: image slot of this lane's (pixel r, channel quads 4 i + q) in row 2 pw
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 2; ++i) wi[hh][i] = yidx(2 * pw, 16 * hh + r, 4 * i + q);
    const int hrow = pw == 2 ? (r < 8 ? r : 8) : 8;
    const int hcol = pw == 2 ? 32 : 16 * pw + r;
    const bool hlane = pw < 2 || (pw == 2 && r <= 8);
    const int hx_off = (hrow * XW + hcol) << 2;
    const int hi0 = yidx(hrow, hcol, q), hi1 = yidx(hrow, hcol, 4 + q);
    const int x_lane = (2 * pw * XW + r) << 2;

    auto conv1_tile = [&](int rb, int xb, int g_, int n_, int ty_, int tx_) {
      if (BF_ABL == 3) return;
      const float* sx = reinterpret_cast<const float*>(sX + xb * X_F4);
      f32x4* hb = sH + rb * HALO_F4;
      const int y0 = ty_ * (2 * TH), x0 = tx_ * (2 * TW);
      float* yg = p.y1 + (long long)g_ * p.gs_y1 + (long long)n_ * p.H * p.W * 32;
      unsigned myword = 0;
      const float* xt[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) xt[s] = sx + x_lane + xo[s];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int rowl = st >> 1, hh = st & 1;
        f32x4 acc[2] = {zero4, zero4};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float xv = xt[s][((rowl * XW + 16 * hh) << 2)];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (BF_ABL == 2) acc[i].x += xv * wf[s][i];
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i], xv, acc[i], 0, 0, 0);
          }
        }
        const int gy = y0 + 2 * pw + rowl;
        const bool inside = gy < p.H && x0 + 16 * hh + r < p.W;    // outside the image y1 is conv2's zero padding
        unsigned part = 0;       // this lane's 8 sign bits of pixel r: channel c = 16 i + 4 q + j <-> bit (c & 3) * 8 + (c >> 2) = 8 j + 4 i + q
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 v = acc[i] + bias_r[i];
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          hb[wi[hh][i] + rowl * ROW] = inside ? v : zero4;
          if (BF_CSTORE && BF_PBITS) {         // after the ReLU: > 0 <=> non-zero bits (as the sign fields of conv2's epilogue)
#pragma unroll
            for (int j = 0; j < 4; ++j) part |= min(__float_as_uint(v[j]), 1u) << (8 * j + 4 * i);
          }
        }
        if (BF_CSTORE && BF_PBITS && BF_ABL != 1 && BF_ABL != 4 && p.bits) {
          // the word of pixel r = OR over the four lanes q = 0..3 that hold its channel quads (lanes r, r + 16, r + 32, r + 48);
          // strip st = (row half, column half) keeps it in the lanes of quad st: lane L <-> pixel (row L >> 5, column L & 31)
          part <<= q;
          part |= __shfl_xor(part, 16, 64);
          part |= __shfl_xor(part, 32, 64);
          if (q == st) myword = part;
        }
        // read the strip back in memory order (same wave: LDS operations of a wave complete in order) and store it:
        // lane l -> pair 8 hh + 4 h + (l >> 4), slot l & 15 = pixel 8 h + (l >> 3), channel quad l & 7: 1 KiB per instruction
#pragma unroll
        for (int h = 0; h < (BF_CSTORE ? 0 : 2); ++h) {
          const int pair = 8 * hh + 4 * h + (lane >> 4);
          const f32x4 v = hb[(2 * pw + rowl) * ROW + pair * 16 + ((lane & 15) ^ (pair & 15))];
          const int gx = x0 + 16 * hh + 8 * h + (lane >> 3);
          if (BF_ABL != 1 && gy < p.H && gx < p.W) out_store<0>(yg + ((long long)gy * p.W + gx) * 32 + (lane & 7) * 4, v);
          if (BF_ABL != 1 && p.bits) {
            // a compare IS a ballot (lane = 8 pixel + channel quad): byte `pixel` of the four masks holds the bits of channels
            // 4 c4 + {0, 1, 2, 3}; every lane assembles the word of pixel lane & 7 (bit (c & 3) * 8 + (c >> 2) <-> channel c)
            // and the lanes 8 (2 st + h) + j keep it: after the four strips lane L holds the word of pixel (row L >> 5, col L & 31)
            const unsigned long long bx = __ballot(v.x > 0.f), by = __ballot(v.y > 0.f), bz = __ballot(v.z > 0.f),
                                     bw = __ballot(v.w > 0.f);
            const unsigned j = lane & 7;
            const unsigned word = __builtin_amdgcn_perm((unsigned)(bx >> 32), (unsigned)bx, 0x0c0c0c00u | j) |
                                  __builtin_amdgcn_perm((unsigned)(by >> 32), (unsigned)by, 0x0c0c000cu | (j << 8)) |
                                  __builtin_amdgcn_perm((unsigned)(bz >> 32), (unsigned)bz, 0x0c000c0cu | (j << 16)) |
                                  __builtin_amdgcn_perm((unsigned)(bw >> 32), (unsigned)bw, 0x000c0c0cu | (j << 24));
            if ((lane >> 3) == 2 * st + h) myword = word;
          }
        }
      }
      if (pw < 3) {          // wave-uniform: the halo-only strip (row 8 / column 32), into LDS only
        f32x4 acc[2] = {zero4, zero4};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float xv = sx[hx_off + xo[s]];
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i], xv, acc[i], 0, 0, 0);
        }
        const bool inside = y0 + hrow < p.H && x0 + hcol < p.W;
        if (hlane) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[i] + bias_r[i];
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            hb[i == 0 ? hi0 : hi1] = inside ? v : zero4;
          }
        }
      }
      if ((!BF_CSTORE || BF_PBITS) && BF_ABL != 1 && BF_ABL != 4 && p.bits) {           // one coalesced store per wave: 2 rows x 32 words
        const int gy = y0 + 2 * pw + (lane >> 5), gx = x0 + (lane & 31);
        if (gy < p.H && gx < p.W) p.bits[(long long)g_ * p.gs_bits + ((long long)n_ * p.Hp + gy) * p.Wp + gx] = myword;
      }
    };

    asm volatile("s_barrier" ::: "memory");                        // (P0) x halo of the first tile has landed
    conv1_tile(0, 0, g, n, ty, tx);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (P1) y1 halo 0 complete; x halo 1 landed
    int i = 0;
    for (;;) {
      const bool more = tile + 1 < tend;
      if (more) {
        advance(g, n, ty, tx);
        if (g != g_w) {
          load_w1(g);
          g_w = g;
        }
        conv1_tile((i + 1) & 1, (i + 1) & 1, g, n, ty, tx);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile barrier
      if (!more) break;
      ++tile;
      ++i;
    }
    return;
  }

  // ===== consumers: conv2 on the halo ring =====================================================================================
  if (BF_PRIO == 2) __builtin_amdgcn_s_setprio(3);
  const int strip = wid & 3, khalf = (wid >> 2) & 1;
  // x halo DMA (K-half-1 waves): wave 4 + j issues pieces j and j + 4; slot sl = 64 piece + lane = halo pixel (sl / 35, sl % 35)
  const int xw = wid - 4;
  auto dma_x = [&](int xb, int g_, int n_, int ty_, int tx_) {
    const int iy0 = ty_ * (2 * TH) - 1, ix0 = tx_ * (2 * TW) - 1;     // conv1: TF SAME, stride 1: pad 1 on every side
    const float* xg = p.x + (long long)g_ * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * 4;
    int ln = lane;
    asm volatile("" : "+v"(ln));                        // keeps the per-lane halo coordinates out of loop-invariant registers
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (xw + 4 * k < XPIECES) {                       // wave-uniform
        const int sl = (xw + 4 * k) * 64 + ln;          // formed per tile: held across the tile loop they spill the compute waves
        const int hy = sl / XW, hx = sl - hy * XW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool v = sl < XPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = v ? xg + (hy * p.W + hx) * 4 : g_zero_page_bf;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sX + xb * X_F4 + (xw + 4 * k) * 64), 16, 0, 0);
      }
    }
  };
  if (khalf) {
    dma_x(0, g, n, ty, tx);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_barrier" ::: "memory");                 // (P0)
  if (khalf && tile + 1 < tend) {
    int g1 = g, n1 = n, ty1 = ty, tx1 = tx;
    advance(g1, n1, ty1, tx1);
    dma_x(1, g1, n1, ty1, tx1);
  }
  int g_w = g;
  const int cq_lane = khalf * 4 + q;       // this wave sums channels [16 khalf, 16 khalf + 16)
  f32x4 wreg[9][TI];
  // kernel fragments straight from the HWIO kernel: lane (r, q) of co tile i holds w[tap][4 cq_lane + s][16 i + r], s = 0..3
  auto load_wreg = [&](int g_) {
    int lofs = (4 * cq_lane) * COUT + r;
    asm volatile("" : "+v"(lofs));            // encoder changes are rare: no registers held for them across the tile loop
    const float* wg = p.w2 + (long long)g_ * p.gs_w2 + lofs;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float* w0 = wg + tap * CIN * COUT + 16 * i;
        wreg[tap][i] = f32x4{w0[0], w0[COUT], w0[2 * COUT], w0[3 * COUT]};
      }
    // bias -> LDS (wave 0; two slots by encoder parity: the epilogues of the previous encoder's last tile may still be reading theirs)
    int lb = lane;
    asm volatile("" : "+v"(lb));
    if (wid == 0 && lb < COUT / 4)
      sB[(g_ & 1) * (COUT / 4) + lb] = *reinterpret_cast<const f32x4*>(p.b2 + (long long)g_ * p.gs_b2 + 4 * lb);
  };
  load_wreg(g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");                 // (P1)
  int buf = 0;
  for (;;) {
    const bool more = tile + 1 < tend;
    int g2 = g, n2 = n, ty2 = ty, tx2 = tx;
    if (more) advance(g2, n2, ty2, tx2);
    if (khalf && tile + 2 < tend) {                         // x halo of tile + 2 -> the x buffer tile + 1's producers are not reading
      int g3 = g2, n3 = n2, ty3 = ty2, tx3 = tx2;
      advance(g3, n3, ty3, tx3);
      dma_x(buf, g3, n3, ty3, tx3);
    }
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
    // y1 of THIS tile's halo image -> HBM (the producers fill the image and write the sign words): wave w stores row w of the 8
    // owned rows, 32 pixels = 4 KiB of consecutive NHWC bytes, as 4 pieces of 8 pixels (lane l -> pair 4 h + (l >> 4), slot l & 15
    // = pixel 8 h + (l >> 3), channel quad l & 7).  BF_CSTORE 1: read before tap 2 h, stored after it; 2: all four behind the tap loop
    f32x4 yv = zero4;
    const int gy1 = ty * (2 * TH) + wid;
    int ln = lane;
    asm volatile("" : "+v"(ln));           // per-tile address arithmetic of the stores below: held across the tile loop it spills
    auto y1_read_to = [&](int h, f32x4& dst) {
      const int pair = 4 * h + (ln >> 4);
      dst = sH[buf * HALO_F4 + wid * ROW + pair * 16 + ((ln & 15) ^ (pair & 15))];
    };
    [[maybe_unused]] unsigned myword = 0;
    auto y1_store_from = [&](int h, const f32x4& src) {
      const int gx = tx * (2 * TW) + 8 * h + (ln >> 3);
      float* yg = p.y1 + (long long)g * p.gs_y1 + (((long long)n * p.H + gy1) * p.W + gx) * 32 + (ln & 7) * 4;
      if (BF_ABL != 1 && gy1 < p.H && gx < p.W) out_store<0>(yg, src);
      if (!BF_PBITS && BF_ABL != 1 && BF_ABL != 4 && p.bits) {
        // a compare IS a ballot (lane = 8 pixel + channel quad): byte `pixel` of the four masks holds the bits of channels
        // 4 c4 + {0, 1, 2, 3}; every lane assembles the word of pixel lane & 7 (bit (c & 3) * 8 + (c >> 2) <-> channel c) and
        // the lanes 8 h + j keep it: after the four pieces lane L < 32 holds the word of pixel (row w, column L)
        const unsigned long long bx = __ballot(src.x > 0.f), by = __ballot(src.y > 0.f), bz = __ballot(src.z > 0.f),
                                 bw = __ballot(src.w > 0.f);
        const unsigned j = ln & 7;
        const unsigned word = __builtin_amdgcn_perm((unsigned)(bx >> 32), (unsigned)bx, 0x0c0c0c00u | j) |
                              __builtin_amdgcn_perm((unsigned)(by >> 32), (unsigned)by, 0x0c0c000cu | (j << 8)) |
                              __builtin_amdgcn_perm((unsigned)(bz >> 32), (unsigned)bz, 0x0c000c0cu | (j << 16)) |
                              __builtin_amdgcn_perm((unsigned)(bw >> 32), (unsigned)bw, 0x000c0c0cu | (j << 24));
        if ((ln >> 3) == h) myword = word;
      }
    };
    auto y1_read = [&](int h) { y1_read_to(h, yv); };
    auto y1_store = [&](int h) { y1_store_from(h, yv); };
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) frag(tap + 1, a_nxt);
      if (BF_CSTORE == 1 && tap < 8 && (tap & 1) == 0) y1_read(tap >> 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch read ABOVE this group's MFMAs
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TI; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][i][s], a_cur[s], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (BF_CSTORE == 1 && tap < 8 && (tap & 1) == 0) {
        y1_store(tap >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      a_cur = a_nxt;
    }
    if (BF_CSTORE == 2) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        y1_read(h);
        y1_store(h);
      }
    }
    if (BF_CSTORE == 3) {       // two pieces in flight: the second read's latency hides behind the first store
      f32x4 yw;
#pragma unroll
      for (int h = 0; h < 4; h += 2) {
        y1_read_to(h, yv);
        y1_read_to(h + 1, yw);
        y1_store_from(h, yv);
        y1_store_from(h + 1, yw);
      }
    }
    if (BF_CSTORE && !BF_PBITS && BF_ABL != 1 && BF_ABL != 4 && p.bits && ln < 32) {
      const int gx = tx * (2 * TW) + ln;
      if (gy1 < p.H && gx < p.W) p.bits[(long long)g * p.gs_bits + ((long long)n * p.Hp + gy1) * p.Wp + gx] = myword;
    }
    f32x4* red = sR + (int)(tile & 1) * RED_F4;
    if (khalf == 1) {
#pragma unroll
      for (int i = 0; i < TI; ++i) red[(strip * TI + i) * 64 + lane] = acc[i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the x halo issued at the top of this tile has landed
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile barrier: partial sums visible; halo `buf` free; halo of tile + 1 complete
    if (khalf == 0) {
      // epilogue: lane owns pixel (ty*4 + strip, tx*16 + r), channels 16 i + 4 q .. +3; transposed through LDS so that every
      // store instruction writes 1 KiB of consecutive bytes
      f32x4* so = sO + strip * 16 * OP;
      unsigned field = 0;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        f32x4 v = acc[i] + red[(strip * TI + i) * 64 + lane] + sB[(g & 1) * (COUT / 4) + 4 * i + q];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        so[r * OP + 4 * i + q] = v;
        if (p.fields) {
#pragma unroll
          for (int j = 0; j < 4; ++j) field |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
        }
      }
      if (p.fields) {
        const int fy = ty * TH + strip, fx = tx * TW + (ln & 15);
        if (fy < p.Ho && fx < p.Wo)
          p.fields[(long long)g * p.gs_fields + (((long long)n * p.fHp + fy) * p.fWp + fx) * 4 + (ln >> 4)] = (unsigned short)field;
      }
      const int oy = ty * TH + strip;
      float* yo = p.y2 + (long long)g * p.gs_y2 + (((long long)n * p.Ho + oy) * p.Wo + tx * TW) * COUT;
      constexpr int C4 = COUT / 4;
#pragma unroll
      for (int jj = 0; jj < TI; ++jj) {          // 16 * C4 float4 = TI x 64 lanes
        const int m = ln + 64 * jj;
        const int px = m / C4, c4 = m - px * C4;
        const f32x4 v = so[px * OP + c4];
        if (oy < p.Ho && tx * TW + px < p.Wo) out_store<1>(yo + m * 4, v);
      }
    }
    if (!more) break;
    if (g2 != g_w) {             // the range crosses into the next encoder: new kernel fragments
      load_wreg(g2);
      g_w = g2;
    }
    g = g2; n = n2; ty = ty2; tx = tx2;
    buf ^= 1;
    ++tile;
  }
}

extern "C" int geeco_conv1_conv2_fwd(const float* x, const float* w1, const float* b1, float* y1, uint32_t* bits,
                                     const float* w2, const float* b2, float* y2, uint16_t* fields, int groups, int64_t gs_x,
                                     int64_t gs_w1, int64_t gs_b1, int64_t gs_y1, int64_t gs_bits, int64_t gs_w2, int64_t gs_b2,
                                     int64_t gs_y2, int64_t gs_fields, int N, int H, int W, int w_cin, void* stream) {
  GEECO_CHECK_ARG(x && w1 && b1 && y1 && w2 && b2 && y2, "conv1_conv2_fwd: null pointer");
  GEECO_CHECK_ARG(groups >= 1 && N >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                  "conv1_conv2_fwd: H = %d, W = %d must be even", H, W);
  GEECO_CHECK_ARG(w_cin == 3 || w_cin == 4, "conv1_conv2_fwd: conv1 kernel with %d input channels (3 or 4)", w_cin);
  GEECO_CHECK_ARG((long long)H * W * 4 < (1ll << 31), "conv1_conv2_fwd: frame too large for 32-bit halo offsets");
  BottomFwdParams p = {};
  p.x = x; p.w1 = w1; p.b1 = b1; p.y1 = y1; p.bits = bits; p.w2 = w2; p.b2 = b2; p.y2 = y2; p.fields = fields;
  p.gs_x = gs_x; p.gs_w1 = gs_w1; p.gs_b1 = gs_b1; p.gs_y1 = gs_y1; p.gs_bits = gs_bits; p.gs_w2 = gs_w2; p.gs_b2 = gs_b2;
  p.gs_y2 = gs_y2; p.gs_fields = gs_fields;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2;
  p.tiles_x = cdiv(p.Wo, 16); p.tiles_y = cdiv(p.Ho, 4);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.ntiles = (long long)groups * p.tiles_per_group;
  p.Hp = (int)geeco_relu_bits_rows(H); p.Wp = (int)geeco_relu_bits_pitch(W);
  p.fHp = (p.Ho + 7) / 8 * 8; p.fWp = (p.Wo + 63) / 64 * 64;
  p.w_cin = w_cin;
  const size_t lds = (size_t)(2 * 9 * 17 * 16 + 2 * 7 * 64 + 2 * 4 * 3 * 64 + 4 * 16 * 13 + 2 * 12) * 16;
  static std::atomic<bool> attr_set{false};   // idempotent attribute calls: racing threads at worst repeat them
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_conv2_fwd_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_conv2_fwd_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const long long blocks = p.ntiles < 256 ? p.ntiles : 256;
  if (w_cin == 3) {
    geeco_note_kernel("conv1_conv2_fwd_kernel<true>");
    hipLaunchKernelGGL(conv1_conv2_fwd_kernel<true>, dim3((unsigned)blocks), dim3(768), lds, (hipStream_t)stream, p);
  } else {
    geeco_note_kernel("conv1_conv2_fwd_kernel<false>");
    hipLaunchKernelGGL(conv1_conv2_fwd_kernel<false>, dim3((unsigned)blocks), dim3(768), lds, (hipStream_t)stream, p);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}
