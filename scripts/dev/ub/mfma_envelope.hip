// Microbenchmark (round 6): the attainable f32-MFMA rate of a kernel SHAPED LIKE THE PRODUCT'S big convolutions, as a function of
// the bytes it has to move -- the "envelope" the per-kernel roofline fractions should be read against.  Round 5's loop
// (mfma_power.hip: every compute wave issues its own global loads, one ds_read_b128 per MFMA) reached 101 TFLOP/s beside
// 3.2 TB/s while conv2's forward reaches 115 at the same traffic: a loop the product beats bounds nothing.  This one has the
// product's structure (conv_s2_halo_fwd_ws_kernel, csrc/conv_halo.hip) and NONE of its address arithmetic, bounds tests,
// bias / ReLU / sign words or tile bookkeeping:
//   * one block per CU (256 blocks), 8 compute waves (two per SIMD) + LW loader waves;
//   * the loaders do nothing but LDS-DMA (global_load_lds, 1 KiB per instruction) IN pieces per tile into a ring of three LDS
//     images, two tiles ahead; ONE barrier per tile;
//   * a compute wave runs 108 x v_mfma_f32_16x16x4_f32 per tile (9 taps x 4 k-steps x 3 output tiles: conv2's count) with the
//     B operands ("kernel fragments") resident in 108 VGPRs and ONE ds_read_b128 of the A operand per RD MFMAs (12 in conv2
//     forward, 4 / 8 in the other kernels), prefetched one group ahead;
//   * stores: OUT pieces of 1 KiB per tile, issued either by the K-half-0 compute waves behind their MFMA loop, after the
//     K-half-1 partner's partial sums went through LDS (STORE = 1: the product's arrangement), or by the loader waves
//     (STORE = 2: the compute waves never touch the vector-memory pipe -- the most favourable arrangement there is), or not at
//     all (STORE = 0).
// Further switches (template parameters; what each measured: profiles/r06/ub_mfma_envelope.txt, findings 29-30, 35-36):
//   GEOM  1 conv2 forward's addresses (9 runs of 4.1 KiB per tile, 32 KiB apart; 4 x 3 KiB out), 2 + its epilogue instruction for instruction,
//         3 + pair-swizzled DMA order, 4 loaders without the edge clamp, 5 a 2 x 32 tile instead of 4 x 16, 6 paced DMA issue (SLEEP);
//   STORE 3 split epilogue (both waves of a SIMD finalise half of the strip), 4 deferred epilogue (tile t - 1 finished inside tile t's MFMA
//         loop), 5 the partner wave sleeps behind the barrier (SLEEP);
//   SLEEP loaders (or, STORE 5, the K-half-1 waves) sleep SLEEP x 64 cycles behind the tile barrier;  NMF MFMAs per wave and tile (108 = conv2;
//         fewer: how much a CU ingests);  LW loader waves;  `tiles` 6 000 (steady state) or 96 (the product's launch length, with in-kernel
//         timelines of the K-half-0 / K-half-1 waves as scripts/dev/halo_stamps.py prints them for the product).
// Prints TFLOP/s (algorithmic = executed here), the HBM traffic it ran beside (reads + writes) and the mean shader clock.
//   build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o mfma_envelope mfma_envelope.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// in-kernel timeline (as conv_halo.hip's HSTAMP): [block][K half][64] s_memtime of lane 0 of waves 0 and 4, six stamps per tile for the first ten tiles
__device__ unsigned long long g_stamps[256 * 2 * 64];
#define ESTAMP(i) do { if (NMF == 107 + 1 && GEOM >= 2 && STORE != 2 && STORE != 5 && lane == 0 && (wid & 3) == 0 && (i) < 64) g_stamps[(blockIdx.x * 2 + (wid >> 2)) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// IN: 1 KiB DMA pieces per tile and block; OUT: 1 KiB store pieces per tile and block (multiple of 4); RD: MFMAs per ds_read_b128;
// STORE: 0 none, 1 by the K-half-0 compute waves (with the K-half reduction through LDS), 2 by the loader waves
template <int IN, int OUT, int RD, int STORE, int LW, int GEOM = 0, int SLEEP = 0, int NMF = 108>
__global__ __launch_bounds__(512 + 64 * LW) void env(const f32x4* __restrict__ big, long long big_f4, f32x4* __restrict__ outbuf, long long out_f4,
                                                     const float* __restrict__ wsrc, float* sink, unsigned long long* clk, int tiles, unsigned* fields) {
  constexpr int NBUF = 3;
  constexpr int IMG_F4 = (IN > 16 ? IN : 16) * 64;                 // (at least 16 KiB, so that the A-operand reads have somewhere to wander)
  constexpr int NSLOT = (IN + LW - 1) / LW;                  // DMA pieces per loader wave and tile
  constexpr int OSLOT_L = (OUT + LW - 1) / LW;               // store pieces per loader wave and tile (STORE == 2)
  constexpr int OSLOT_C = OUT / 4;                           // store pieces per K-half-0 compute wave and tile (STORE == 1)
  constexpr int NGRP = NMF / RD;                             // A-operand reads per wave and tile (NMF = 108: conv2's MFMA count; fewer: how much can a CU ingest?)
  static_assert(NMF % RD == 0 && RD % 4 == 0, "RD divides 108 and is a multiple of the 4 k-steps of one read");
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  f32x4* sH = lds;                                           // NBUF images
  f32x4* sR = lds + NBUF * IMG_F4;                           // 2 x (4 strips x 3 x 64) partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every block streams through its own 1/256 of the buffers
  const long long in_span = big_f4 / gridDim.x, out_span = out_f4 / gridDim.x;
  const f32x4* in0 = big + (long long)blockIdx.x * in_span;
  f32x4* out0 = outbuf + (long long)blockIdx.x * out_span;
  // the images start out holding numbers, whatever IN is
  for (int i = tid; i < NBUF * IMG_F4 + 2 * 768 + 4 * 16 * 13; i += blockDim.x) lds[i] = f32x4{wsrc[i & 4095], wsrc[(i + 7) & 4095], wsrc[(i + 13) & 4095], wsrc[(i + 29) & 4095]};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (wid >= 8) {
    // ===== loader waves ==============================================================================================
    const int lw = wid - 8;
    long long ipos = 0, opos = 0;
    f32x4 junk = {1.f, 2.f, 3.f, (float)lane};
    // GEOM: the pieces of a tile come from where conv2's forward reads them: image n = [256][256][32 floats] (2^19 float4), tile
    // (ty, tx) = output rows 4 ty .. +3, columns 16 tx .. +15 = input rows 8 ty .. +8, 33 pixels from column 32 tx: 9 runs of
    // 4 224 bytes, 32 KiB apart; a block walks tx, then ty, then on to the next image (the product's order)
    int d_src[NSLOT > 0 ? NSLOT : 1];
    if (GEOM) {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        int sl = (lw + LW * i) * 64 + lane;
        sl = sl < 9 * 272 ? sl : 9 * 272 - 1;
        const int row = sl / 272, rem = sl - row * 272;
        if (GEOM == 5) {      // what a 2 x 32 output tile would read: 5 runs of 8 KiB (64 of its 65 pixels), 32 KiB apart
          const int sl5 = (lw + LW * i) * 64 + lane;
          d_src[i] = ((sl5 >> 9) * 256 + ((sl5 & 511) >> 3)) * 8 + (sl5 & 7);
        } else if (GEOM == 3) {      // the product's pair-swizzled image: slot u of pixel pair `pair` holds element u ^ (pair & 15) of the pair's 16 float4
          const int pair = rem >> 4, u = (rem & 15) ^ (pair & 15);
          d_src[i] = (row * 256 + 2 * pair + (u >> 3)) * 8 + (u & 7);
        } else
        d_src[i] = (row * 256 + (rem >> 3)) * 8 + (rem & 7);
      }
    }
    int gn = blockIdx.x, gty = 0, gtx = 0;
    auto dma = [&](int buf) {
      if (GEOM) {
        const f32x4* xg = GEOM == 5 ? big + (((long long)gn << 19) + ((long long)(gty * 4) * 256 + gtx * 64) * 8)
                                    : big + (((long long)gn << 19) + ((long long)(gty * 8) * 256 + gtx * 32) * 8);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i)
          if (lw + LW * i < IN) {
            // (the last column's 33rd pixel and the last row band's 9th row fall outside the image: clamped to the image's last float4)
            long long off = d_src[i];
            if (GEOM == 5) {
              const long long lim5 = (1ll << 19) - 1 - ((long long)(gty * 4) * 256 + gtx * 64) * 8;
              off = off < lim5 ? off : lim5;
            } else if (GEOM != 4) {
              const long long lim = (1ll << 19) - 1 - ((long long)(gty * 8) * 256 + gtx * 32) * 8;
              off = off < lim ? off : lim;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)(xg + off), (lptr_t)(sH + buf * IMG_F4 + (lw + LW * i) * 64), 16, 0, 0);
            if (GEOM == 6 && SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);      // PACED issue: SLEEP x 64 cycles between a loader's DMA pieces
          }
        if (GEOM == 5) { if (++gtx == 4) { gtx = 0; if (++gty == 64) { gty = 0; gn = (gn + 1) & 255; } } return; }
        if (++gtx == 8) { gtx = 0; if (++gty == 32) { gty = 0; gn = (gn + 1) & 255; } }
        return;
      }
#pragma unroll
      for (int i = 0; i < NSLOT; ++i)
        if (lw + LW * i < IN) {
          __builtin_amdgcn_global_load_lds((gptr_t)(in0 + ipos + (lw + LW * i) * 64 + lane), (lptr_t)(sH + buf * IMG_F4 + (lw + LW * i) * 64), 16, 0, 0);
        }
      ipos += IN * 64;
      if (ipos + IN * 64 > in_span) ipos = 0;
    };
    constexpr int MINE = NSLOT;                               // (waves with one piece fewer wait for one piece less; kept simple: IN % LW == 0 in the sweep)
    if (IN > 0) { dma(0); dma(1); wait_vm<MINE>(); }
    asm volatile("s_barrier" ::: "memory");
    int slot = 2;
    for (int t = 0; t < tiles; ++t) {
      // SLEEP: the loaders hold their DMA back by SLEEP x 64 cycles behind the tile barrier (the image they fetch is needed two tiles
      // from now; the K-half-0 waves' stores of the tile just finished enter the CU's vector-memory queue first)
      if (SLEEP > 0 && STORE != 5 && GEOM != 6) __builtin_amdgcn_s_sleep(SLEEP);
      if (IN > 0) { dma(slot); slot = slot + 1 == NBUF ? 0 : slot + 1; }
      if (STORE == 2) {
#pragma unroll
        for (int j = 0; j < OSLOT_L; ++j)
          if (lw + LW * j < OUT) __builtin_nontemporal_store(junk, out0 + opos + (lw + LW * j) * 64 + lane);
        opos += OUT * 64;
        if (opos + OUT * 64 > out_span) opos = 0;
        // the next tile's image must have landed: everything but this tile's DMA pieces and stores may stay in flight
        if (IN > 0) wait_vm<MINE + OSLOT_L>();
      } else if (IN > 0) {
        wait_vm<MINE>();
      }
      asm volatile("s_barrier" ::: "memory");
    }
    wait_vm<0>();
    if (lane == 0 && wid == 8) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    return;
  }
  // ===== compute waves ===============================================================================================
  const int strip = wid & 3, khalf = wid >> 2;
  f32x4 wreg[9][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int o = ((tap * 3 + i) * 64 + lane + 97 * wid) * 4;
      wreg[tap][i] = f32x4{wsrc[o & 4095], wsrc[(o + 1) & 4095], wsrc[(o + 2) & 4095], wsrc[(o + 3) & 4095]};
    }
  f32x4 keep = {0.f, 0.f, 0.f, 0.f};
  long long opos = 0;
  int cn = blockIdx.x, cty = 0, ctx = 0;
  [[maybe_unused]] int pn = 0, pty = 0, ptx = 0;
  [[maybe_unused]] f32x4 prev[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  [[maybe_unused]] unsigned dfield = 0;
  asm volatile("s_barrier" ::: "memory");
  int buf = 0;
  for (int t = 0; t < tiles; ++t) {
    ESTAMP(t < 10 ? 6 * t + 0 : 64);
    ESTAMP(t < 10 ? 6 * t + 1 : 64);
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const f32x4* hA = sH + buf * IMG_F4;
    // group g reads its A operand at a position that changes with group, strip and lane (conflict-free: consecutive lanes, consecutive
    // float4); the group's part of the offset is a compile-time constant, like the tap offsets of the product's halo image
    const f32x4* hW = hA + (strip * 2 + khalf) * 64 + lane;
    auto frag = [&](int g) { return hW[((g * 512) % (IMG_F4 - 511)) & ~63]; };
    f32x4 a_cur = frag(0), a_nxt = a_cur;
    // STORE == 4: DEFERRED epilogue -- the K-half-0 wave keeps tile t - 1's accumulators (`prev`) and finishes that tile from INSIDE
    // tile t's MFMA loop, a piece per group: both waves of a SIMD enter their MFMA loops right behind the barrier (the matrix pipe
    // never waits for an epilogue), and a store that is held behind MFMAs (finding 13) holds nothing up
    [[maybe_unused]] f32x4 dv[3];
    [[maybe_unused]] f32x4* dso = sR + 2 * 768 + strip * 16 * 13;
    [[maybe_unused]] const long long dpix = ((long long)(pn & 127) * 128 + pty * 4 + strip) * 128 + ptx * 16;
    auto deferred = [&](int g) {
      if (STORE != 4 || khalf != 0 || t == 0) return;
      const int r = lane & 15, q = lane >> 4;
      const f32x4* red = sR + ((t - 1) & 1) * 768;
      if (g >= 1 && g <= 3) {                      // one output tile per group: partner's partial sums, bias, ReLU, into the staging
        const int i = g - 1;
        f32x4 v = prev[i] + red[(strip * 3 + i) * 64 + lane] + wreg[i][0];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        dso[r * 13 + 4 * i + q] = v;
#pragma unroll
        for (int j = 0; j < 4; ++j) dfield |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
      }
      if (g == 4) {
        reinterpret_cast<unsigned short*>(fields)[(dpix + r) * 4 + q] = (unsigned short)dfield;
        dfield = 0;
      }
      if (g >= 5 && g <= 7) {                      // one 1 KiB store per group
        const int m = lane + 64 * (g - 5);
        const int px = m / 12, c4 = m - px * 12;
        __builtin_nontemporal_store(dso[px * 13 + c4], outbuf + dpix * 12 + m);
      }
    };
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
      if (g + 1 < NGRP) a_nxt = frag(g + 1);
      deferred(g);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < RD; ++m) {
        const int mm = g * RD + m;                            // 0 .. 107: tap = mm / 12, k-step = (mm / 3) & 3, output tile = mm % 3
        acc[mm % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[mm / 12][mm % 3][(mm / 3) & 3], a_cur[RD == 12 ? ((mm / 3) & 3) : (m & 3)], acc[mm % 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
    }
    ESTAMP(t < 10 ? 6 * t + 2 : 64);
    if (STORE == 4) {
      f32x4* red = sR + (t & 1) * 768;
      if (khalf == 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) red[(strip * 3 + i) * 64 + lane] = acc[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) prev[i] = acc[i];
        pn = cn; pty = cty; ptx = ctx;
      }
      ESTAMP(t < 10 ? 6 * t + 3 : 64);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ESTAMP(t < 10 ? 6 * t + 4 : 64);
    } else if (STORE == 3) {
      // SPLIT EPILOGUE: both waves of a SIMD finish the tile together and each finalises HALF of the strip (pixels r < 8: the K-half-0
      // wave, r >= 8: its partner): a lane sends its partial sums of the pixels it does not own through LDS and receives the partner's for
      // the ones it owns; both waves then run bias / ReLU / sign words / transposition / stores at the same time -- no store of one wave
      // sits behind the other wave's MFMA stream (finding 13), and the two MFMA loops that follow run INTERLEAVED on the SIMD
      f32x4* red = sR + (t & 1) * 768;
      const int r = lane & 15, q = lane >> 4;
      const bool mine = (r < 8) == (khalf == 0);
      if (!mine) {
#pragma unroll
        for (int i = 0; i < 3; ++i) red[(strip * 3 + i) * 64 + lane] = acc[i];
      }
      ESTAMP(t < 10 ? 6 * t + 3 : 64);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ESTAMP(t < 10 ? 6 * t + 4 : 64);
      f32x4* so = sR + 2 * 768 + (strip * 2 + khalf) * 8 * 13;
      const long long pix = ((long long)(cn & 127) * 128 + cty * 4 + strip) * 128 + ctx * 16 + 8 * khalf;
      if (mine) {
        unsigned field = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          f32x4 v = acc[i] + red[(strip * 3 + i) * 64 + lane] + wreg[i][0];
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          so[(r & 7) * 13 + 4 * i + q] = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) field |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
        }
        reinterpret_cast<unsigned short*>(fields)[(pix + (r & 7)) * 4 + q] = (unsigned short)field;
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {          // the wave's 8 pixels x 192 bytes = 1.5 KiB of consecutive bytes
        const int m = lane + 64 * jj;
        if (m < 96) {
          const int px = m / 12, c4 = m - px * 12;
          __builtin_nontemporal_store(so[px * 13 + c4], outbuf + pix * 12 + m);
        }
      }
    } else if (STORE == 1 || STORE == 5) {
      f32x4* red = sR + (t & 1) * 768;
      if (khalf == 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) red[(strip * 3 + i) * 64 + lane] = acc[i];
      }
      ESTAMP(t < 10 ? 6 * t + 3 : 64);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ESTAMP(t < 10 ? 6 * t + 4 : 64);
      // STORE == 5: the K-half-1 wave gives its partner's stores an MFMA-free window before it starts the next tile's loop
      if (STORE == 5 && khalf == 1 && SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
      if (khalf == 0 && GEOM >= 2) {
        // the product's epilogue, instruction for instruction (conv_s2_halo_fwd_ws_kernel): bias, ReLU, the 16-bit sign word of the
        // lane's 12 outputs (stored as a short), the strip transposed through LDS so that every store writes 1 KiB of consecutive bytes
        f32x4* so = sR + 2 * 768 + strip * 16 * 13;
        const int r = lane & 15, q = lane >> 4;
        unsigned field = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          f32x4 v = acc[i] + red[(strip * 3 + i) * 64 + lane] + wreg[i][0];
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          so[r * 13 + 4 * i + q] = v;
#pragma unroll
          for (int j = 0; j < 4; ++j) field |= min(__float_as_uint(v[j]), 1u) << (4 * i + j);
        }
        const long long pix = GEOM == 5 ? ((long long)(cn & 127) * 128 + cty * 2 + (strip >> 1)) * 128 + ctx * 32 + (strip & 1) * 16
                                        : ((long long)(cn & 127) * 128 + cty * 4 + strip) * 128 + ctx * 16;
        reinterpret_cast<unsigned short*>(fields)[(pix + r) * 4 + q] = (unsigned short)field;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          const int m = lane + 64 * jj;
          const int px = m / 12, c4 = m - px * 12;
          __builtin_nontemporal_store(so[px * 13 + c4], outbuf + pix * 12 + m);
        }
      } else if (khalf == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const f32x4 v = acc[i] + red[(strip * 3 + i) * 64 + lane];
          if (GEOM) {      // output image n = [128][128][48 floats]: the wave's strip = 16 pixels x 192 bytes = 3 KiB of one row
            if (i < OSLOT_C) __builtin_nontemporal_store(v, outbuf + ((((long long)(cn & 127) * 128 + cty * 4 + strip) * 128 + ctx * 16) * 12 + i * 64 + lane));
            else keep += v;
          } else if (i < OSLOT_C) __builtin_nontemporal_store(v, out0 + opos + (strip * OSLOT_C + i) * 64 + lane);
          else keep += v;
        }
        opos += OUT * 64;
        if (opos + OUT * 64 > out_span) opos = 0;
      }
    } else {
      keep += acc[0] + acc[1] + acc[2];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    ESTAMP(t < 10 ? 6 * t + 5 : 64);
    if (GEOM == 5) { if (++ctx == 4) { ctx = 0; if (++cty == 64) { cty = 0; ++cn; } } }
    else if (++ctx == 8) { ctx = 0; if (++cty == 32) { cty = 0; ++cn; } }
    buf = buf + 1 == NBUF ? 0 : buf + 1;
  }
  sink[blockIdx.x * 512 + tid] = keep.x + keep.y + keep.z + keep.w;
}

template <int IN, int OUT, int RD, int STORE, int LW, int GEOM = 0, int SLEEP = 0, int NMF = 108>
void run(const f32x4* big, long long big_f4, f32x4* outbuf, long long out_f4, const float* wsrc, float* sink, unsigned long long* clk, int tiles) {
  static unsigned* fields = nullptr;
  if (!fields) hipMalloc(&fields, 128ll * 128 * 128 * 8);
  constexpr int IMG_F4 = (IN > 16 ? IN : 16) * 64;
  const size_t ldsb = (size_t)(3 * IMG_F4 + 2 * 768 + 4 * 16 * 13) * 16;
  auto kern = env<IN, OUT, RD, STORE, LW, GEOM, SLEEP, NMF>;
  if (NMF != 108) printf("%d MFMAs per wave and tile: ", NMF);
  if (SLEEP && GEOM == 6) printf("%d x 64 cycles between a loader's DMA pieces: ", SLEEP);
  else if (SLEEP) printf("%s sleep %d x 64 cycles behind the barrier: ", STORE == 5 ? "K-half-1 waves" : "loaders", SLEEP);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512 + 64 * LW), ldsb, 0, big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles, fields);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512 + 64 * LW), ldsb, 0, big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles, fields);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  unsigned long long h[512]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  const double tf = (double)NMF * 2048.0 * 8 * 256 * tiles / (ms * 1e-3) / 1e12;
  const double rd = (double)IN * 1024 * 256 * tiles / (ms * 1e-3) / 1e12, wr = (STORE ? (double)OUT : 0.0) * 1024 * 256 * tiles / (ms * 1e-3) / 1e12;
  if (GEOM == 2 && tiles == 96) {
    printf("  (stores: %s)\n", STORE == 3 ? "both K halves" : STORE == 4 ? "K half 0, deferred into the next tile's MFMA loop" : "K half 0");
    static unsigned long long hs[256 * 2 * 64];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof(hs));
    const char* names[5] = {"(advance)", "MFMA loop (9 steps)", "partial sums -> LDS", "barrier", "epilogue"};
    for (int kh = 0; kh < 2; ++kh) {
      printf("  timeline, K half %d (wave %d), tiles 2..8 of every block, cycles:", kh, 4 * kh);
      for (int i = 0; i < 5; ++i) {
        double sum = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int t = 2; t < 9; ++t) { sum += (double)(hs[(b * 2 + kh) * 64 + 6 * t + i + 1] - hs[(b * 2 + kh) * 64 + 6 * t + i]); ++n; }
        printf("  %s %.0f", names[i], sum / n);
      }
      double per = 0; int n = 0;
      for (int b = 0; b < 256; ++b) for (int t = 2; t < 8; ++t) { per += (double)(hs[(b * 2 + kh) * 64 + 6 * (t + 1)] - hs[(b * 2 + kh) * 64 + 6 * t]); ++n; }
      printf("  | tile period %.0f\n", per / n);
    }
  }
  if (LW != 4) printf("%d loader waves: ", LW);
  printf("%sin %2d KiB out %2d KiB per tile | 1 ds_read per %2d MFMAs | stores %-14s | %6.1f TFLOP/s = %4.1f %% | %5.2f + %4.2f = %5.2f TB/s | %.3f GHz | tile %5.0f cycles\n",
         GEOM == 6 ? "conv2 geometry + its epilogue, DMA pieces paced: " : GEOM == 5 ? "conv2 as 2 x 32 tiles (5 runs of 8 KiB) + its epilogue: " : GEOM == 4 ? "conv2 geometry + its epilogue, loaders without the edge clamp: " : GEOM == 3 ? "conv2 geometry + its epilogue + pair-swizzled DMA: " : GEOM == 2 ? "conv2 geometry + its epilogue: " : GEOM ? "conv2 geometry: " : "", IN, STORE ? OUT : 0, RD, STORE == 0 ? "none" : STORE == 1 ? "compute waves" : STORE == 3 ? "both K halves" : STORE == 4 ? "deferred" : STORE == 5 ? "K half 0, partner sleeps" : "loader waves", tf, tf / 157.3 * 100, rd, wr, rd + wr,
         cyc / real * 0.1, cyc / 256 / tiles);
  fflush(stdout);
}

int main() {
  float* wsrc; float* sink; unsigned long long* clk;
  hipMalloc(&wsrc, 4096 * 4); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
  float h[4096];
  srand(11);
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(wsrc, h, sizeof(h), hipMemcpyHostToDevice);
  const long long big_f4 = (2ll << 30) / 16, out_f4 = (1ll << 30) / 16;
  f32x4 *big, *outbuf;
  hipMalloc(&big, big_f4 * 16 + (1 << 20)); hipMemset(big, 0x3c, big_f4 * 16 + (1 << 20));      // (+ 1 MiB: the unclamped form reads 4 KiB past the last image)
  //      // 0x3c3c3c3c = 0.0115 as a float: finite operands
  hipMalloc(&outbuf, out_f4 * 16);
  const int tiles = 6000;       // ~25 ms per launch: long enough for the clock governor
#define RUN(IN, OUT, RD, STORE) run<IN, OUT, RD, STORE, 4>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles)
  for (int rep = 0; rep < 2; ++rep) {
    printf("---- pass %d\n", rep + 1);
    // no memory traffic at all: the MFMA stream beside its own LDS reads and the tile barrier
    RUN(0, 0, 12, 0); RUN(0, 0, 4, 0);
    // the three traffic points of the product's big kernels (conv2 dgrad + conv1 wgrad ~1.1, conv3 forward ~2.3, conv2 forward / wgrad ~3.2 TB/s)
    RUN(12, 4, 12, 2); RUN(12, 4, 12, 1); RUN(12, 4, 4, 1);
    RUN(28, 8, 12, 2); RUN(28, 8, 12, 1); RUN(28, 8, 4, 1);
    RUN(40, 12, 12, 0); RUN(40, 12, 12, 2); RUN(40, 12, 12, 1); RUN(40, 12, 4, 1);
    // (more than 40 KiB per tile does not fit a ring of three images in 160 KB of LDS)
    // the same bytes read and written WHERE conv2's forward reads and writes them (9 runs of 4.1 KiB per tile, 32 KiB apart; 4 x 3 KiB out)
    run<40, 12, 12, 0, 4, 1>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 1>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    // ... and with the product's 96 tiles per block (24 576 tiles over 256 blocks): what launch, prologue and drain cost a 0.35 ms kernel
    run<40, 12, 12, 1, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, 96);
    run<40, 12, 12, 1, 4, 3>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 3>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, 96);
    // the split epilogue (both waves of a SIMD finish together, each stores half of the strip)
    run<40, 12, 12, 3, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 3, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, 96);
    // the deferred epilogue (the K-half-0 wave finishes tile t - 1 from inside tile t's MFMA loop)
    run<40, 12, 12, 4, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 4, 4, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, 96);
    // the partner wave holds its MFMA loop back so that the K-half-0 wave's stores go out at once
    run<40, 12, 12, 5, 4, 2, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 5, 4, 2, 4>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 5, 4, 2, 6>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 5, 4, 2, 8>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 5, 4, 2, 12>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    // more / fewer loader waves for the same 40 pieces
    run<40, 12, 12, 1, 8, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 2, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    // paced DMA issue
    run<40, 12, 12, 1, 4, 6, 1>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 6, 2>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 6, 4>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 6, 8>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 4>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 5>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    // how many bytes per clock can a CU take in through LDS-DMA when the MFMA stream is short?  (conv2's filter gradient needs 51 KiB per
    // 6.9 k cycles of MFMA = 7.4 B/clk to be matrix-bound)
    run<40, 12, 12, 0, 4, 0, 0, 72>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 0, 4, 0, 0, 36>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 0, 4, 0, 0, 12>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 0, 4, 1, 0, 12>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 2, 4>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 2, 8>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 2, 16>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 2, 32>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 4, 16>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
    run<40, 12, 12, 1, 4, 0, 16>(big, big_f4, outbuf, out_f4, wsrc, sink, clk, tiles);
  }
  return 0;
}
