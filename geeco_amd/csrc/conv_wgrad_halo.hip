// Filter/bias gradient of the stride-2 middle layers (conv3 .. conv6 of the encoder) with the input halo and
// the dz tile staged ONCE per block in LDS.
//
// Autodiff of reference src/models/e2evmc/graph.py:86-105 (conv3..conv6: 3x3, stride 2, TF SAME) taken by
// tf.train.AdamOptimizer.minimize (src/models/e2evmc/estimator.py:243-244):
//
//   dw[tap][ci][co] = sum_pixels x[2 oy + ky][2 ox + kx][ci] * dz[oy][ox][co]          db[co] = sum dz
//
// Why: the generic kernel of conv_wgrad.hip gathers the x rows of every (tap, 64 k-rows) tile again from L2
// (conv3: 1.38 GB L2->LDS for 403 MB of operands; PMC FETCH 1.6-4.8x the algorithmic bytes; 47 % of the MFMA
// peak).  Here a block owns a (CIB input channels) x (64 output channels) block of dw for ALL nine taps and walks a
// contiguous range of TH x TW output-pixel tiles: per tile the (2 TH + 1) x (2 TW + 1) input halo of its CIB
// channels and the TH x TW x 64 dz tile land in LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no
// ds_write), double buffered behind the previous tile's MFMAs, and all nine taps read their fragments from there:
// x is fetched once per co block, dz once per ci block, 47 FLOP per staged byte instead of 16.
//
// Wave roles: wave = (ci tile of 16, COT co tiles of 16); all waves walk the same pixels, so there is no
// intra-block reduction: 9 x COT accumulator tiles per wave live in registers for the block's whole tile range and
// are written once, to the block's part of split slab `s` ([G][S][9*Cin*Cout + Cout], summed in a fixed order by
// wgrad_reduce_kernel: bitwise reproducible).  MFMA v_mfma_f32_16x16x4_f32: k = 4 consecutive output pixels of a
// row, A = dz fragment (row i = co), B = x fragment (column j = ci) => each lane owns 4 consecutive co of one
// (tap, ci) row = one 16-byte store into the HWIO slab.
//
// LDS images (float4 granules, each DMA piece = 64 consecutive granules = 1 KiB):
//   x   [row][halo column][XPQ granules]: XPQ = 16 with the granule index XOR-ed by ((column >> 1) & 1) << 2 for
//       CIB = 64, XPQ = 14 (pitch 56 floats = 8 mod 16) unswizzled for CIB = 48: the ds_read_b32 of a B fragment
//       (16 channels x the 4 columns 2 (4 s + q) + kx) then covers 2 x 16 distinct banks per 32-lane group;
//   dz  [pixel][16 granules], granule index XOR-ed by (pixel & 1) << 2 (A fragment: 16 channels x 4 pixels).
#include "geeco_common.h"
#include <stdlib.h>

static __device__ float g_zero_page[64];   // source of the DMA lanes that fall outside the image (TF SAME zero padding) or on pad granules

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


struct WgradHaloParams {
  const float* x;
  const float* dz;
  const float* zero;         // g_zero_page (its address as a kernel argument: no GOT load inside the tile loop)
  float* part;               // [G][S][9*Cin*Cout + Cout]
  long long gs_x, gs_dz;
  int N, H, W, Ho, Wo, Cin, Cout;
  int tiles_x, tiles_y, tiles_per_group;
  int S;                     // pixel slices (slabs) per group
  int n_cib, n_cob;          // ci / co blocks
  int n_sigma;               // G * n_cib * S  (slices dealt round-robin to the XCDs)
};

void geeco_launch_wgrad_reduce(const float* part, float* dw, float* db, long long gs_dw, long long gs_db, int S,
                               long long KC, int Cout, int groups, hipStream_t s);

// CIB = 16 NCI input channels per block; COB = 16 NCO COT output channels per block (64, 96 or 128); NW = NCI * NCO waves.
// Wide co blocks (COT = 3, 4) halve the number of blocks that fetch the same x halo: a 64-channel co block needs
// 47 KB of DMA per 9.2k cycles of MFMA work per SIMD (conv4: 5.1 B/clk/CU - the ingest limit next to MFMA waves), a
// 128-channel one 55 KB per 18.4k.
// NBUF = 2: halo / dz double buffered, one block per CU;  NBUF = 1: single buffers (half the LDS), the host launches
// TWO blocks per CU (MINW waves per SIMD in total) which fill each other's DMA waits and barrier stalls.
template <int NCI, int NCO, int COT, int TH, int TW, int XPQ, bool SWZ, int NBUF, int MINW>
__global__ __launch_bounds__(64 * NCI * NCO, MINW) void conv_s2_wgrad_lds_kernel(const WgradHaloParams p) {
  constexpr int NW = NCI * NCO, NT = 64 * NW;
  constexpr int CIB = 16 * NCI, CQ = CIB / 4;
  constexpr int HY = 2 * TH + 1, HX = 2 * TW + 1;
  constexpr int X_USED = HY * HX * XPQ;                  // float4 granules of the halo image
  constexpr int NXP = (X_USED + 63) / 64;                // its 1 KiB DMA pieces
  constexpr int X_F4 = NXP * 64;
  constexpr int COB = 16 * NCO * COT, ZQ = COB / 4;      // co block and its float4 granules per pixel
  constexpr int Z_F4 = TH * TW * ZQ;                     // dz tile: [pixel][ZQ granules]
  constexpr int NZP = Z_F4 / 64;
  constexpr int KG = TW / 4;                             // k-groups (4 consecutive pixels) per tile row
  static_assert((ZQ == 16 || ZQ == 24 || ZQ == 32) && TW % 4 == 0 && Z_F4 % 64 == 0, "co block = 64, 96 or 128 channels");
  static_assert(XPQ >= CQ && (SWZ ? (XPQ == 16 && CQ == 16) : (XPQ % 4 == 2)), "x pixel pitch must be 8 (mod 16) floats or swizzled");
  constexpr int NDB = (Z_F4 + NT - 1) / NT;              // dz granules per thread for the bias gradient
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sX = reinterpret_cast<f32x4*>(smem);            // NBUF halo buffers
  f32x4* sZ = sX + NBUF * X_F4;                          // NBUF dz tiles

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int cit = wid % NCI, cog = wid / NCI;            // this wave's ci tile and group of COT co tiles
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- which (group, ci block, pixel slice, co block) is this block?  Slices sigma = (g, cib, s) are dealt
  // round-robin to the 8 XCDs (blocks b and b + 8 share an XCD) and the co blocks of one slice are neighbours on
  // ITS XCD, so the x halo they all read is served by one L2. ---------------------------------------------------------
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int cob = j % p.n_cob;
  const int sigma = (j / p.n_cob) * 8 + xcd;
  if (sigma >= p.n_sigma) return;                         // whole block leaves together (before any barrier)
  const int split = sigma % p.S;
  const int cib = (sigma / p.S) % p.n_cib;
  const int g = sigma / (p.S * p.n_cib);
  const int ci0 = cib * CIB, co0 = cob * COB;
  const int Cin = p.Cin, Cout = p.Cout;

  const int per = (p.tiles_per_group + p.S - 1) / p.S;
  int tile = split * per;
  const int tend = tile + per < p.tiles_per_group ? tile + per : p.tiles_per_group;
  const long long slab = 9ll * Cin * Cout + Cout;
  float* part = p.part + ((long long)g * p.S + split) * slab;

  int n, ty, tx;
  {
    const int per_img = p.tiles_x * p.tiles_y;
    n = tile / per_img;
    const int rem = tile - n * per_img;
    ty = rem / p.tiles_x;
    tx = rem - ty * p.tiles_x;
  }

  // ---- this wave's DMA pieces: halo pieces kx = wid + NW i (i < NSX, granules [64 kx, +64) of the x image) and dz pieces
  // kz = wid + NW j (j < NSZ), numbered separately so that a slot has ONE role for every wave and only the last slot of each kind
  // needs a (wave-uniform) range test (as conv_dgrad_lds.hip) -----------------------------------------------------------------
  constexpr int NSX = (NXP + NW - 1) / NW, NSZ = (NZP + NW - 1) / NW;
  __builtin_assume(wid >= 0 && wid < NW);
  int x_src[NSX], z_src[NSZ];
  short x_a[NSX], x_b[NSX], z_a[NSZ], z_b[NSZ];       // halo: (row, column); dz: (tile row, tile column); 30000 = never valid
#pragma unroll
  for (int i = 0; i < NSX; ++i) {
    const int sl = (wid + NW * i) * 64 + lane;
    const int rw = sl / (HX * XPQ), rem = sl - rw * (HX * XPQ);
    const int hx = rem / XPQ, qs = rem - hx * XPQ;
    const int quad = SWZ ? (qs ^ (((hx >> 1) & 1) << 2)) : qs;
    const bool ok = sl < X_USED && quad < CQ;
    x_a[i] = (short)(ok ? rw : 30000);
    x_b[i] = (short)hx;
    x_src[i] = (rw * p.W + hx) * Cin + ci0 + quad * 4;
  }
#pragma unroll
  for (int j = 0; j < NSZ; ++j) {
    const int f = (wid + NW * j) * 64 + lane;
    const int px = f / ZQ, qs = f - px * ZQ;
    const int zr = px / TW, zc = px - zr * TW;
    const int quad = qs ^ ((zc & 1) << 2);
    z_a[j] = (short)zr;
    z_b[j] = (short)zc;
    z_src[j] = (zr * p.Wo + zc) * Cout + co0 + quad * 4;
  }
  // The DMA pieces of tile (n_, ty_, tx_) into buffer buf.  Everything that depends on the tile only - the two tile base
  // addresses, the distances to the image edges, the zero page's offset from either base - is formed ONCE per tile in scalar
  // registers; a piece is then two compares, a 64-bit select and an add per lane.  (Until round 4 every piece recomputed its
  // 64-bit base from the kernel arguments and fetched the zero page's address through the GOT behind an s_waitcnt lgkmcnt(0):
  // ~80 instructions and an SMEM round trip per piece, issued by all waves at the same point of the tile.)
  auto dma_tile = [&](int buf, int n_, int ty_, int tx_) {
    const int iy0 = ty_ * TH * 2, ix0 = tx_ * TW * 2;
    const float* xt = p.x + (long long)g * p.gs_x + (((long long)n_ * p.H + iy0) * p.W + ix0) * Cin;
    const float* zt = p.dz + (long long)g * p.gs_dz + (((long long)n_ * p.Ho + ty_ * TH) * p.Wo + tx_ * TW) * Cout;
    const long long zero_x = p.zero - xt, zero_z = p.zero - zt;            // element offsets of the zero page from the bases
    const int hy = p.H - iy0, hx_ = p.W - ix0, zy = p.Ho - ty_ * TH, zx = p.Wo - tx_ * TW;
#pragma unroll
    for (int i = 0; i < NSX; ++i)
      if (NW * (i + 1) <= NXP || wid + NW * i < NXP) {      // compile-time true except in the last slot
        const bool v = x_a[i] < hy && x_b[i] < hx_;
        const long long off = v ? (long long)x_src[i] : zero_x;
        __builtin_amdgcn_global_load_lds((gptr_t)(xt + off), (lptr_t)(sX + buf * X_F4 + (wid + NW * i) * 64), 16, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < NSZ; ++j)
      if (NW * (j + 1) <= NZP || wid + NW * j < NZP) {
        const bool v = z_a[j] < zy && z_b[j] < zx;
        const long long off = v ? (long long)z_src[j] : zero_z;
        __builtin_amdgcn_global_load_lds((gptr_t)(zt + off), (lptr_t)(sZ + buf * Z_F4 + (wid + NW * j) * 64), 16, 0, 0);
      }
  };
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    if (++tx_ == p.tiles_x) {
      tx_ = 0;
      if (++ty_ == p.tiles_y) {
        ty_ = 0;
        ++n_;
      }
    }
  };

  f32x4 dbsum[NDB];
#pragma unroll
  for (int k = 0; k < NDB; ++k) dbsum[k] = zero4;
  f32x4 acc[9][COT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < COT; ++i) acc[t][i] = zero4;

  if (tile < tend) dma_tile(0, n, ty, tx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- per-lane fragment offsets (floats) ------------------------------------------------------------------------
  // B (x): channel ci0 + 16 cit + r of halo column 2 (4 s + q) + kx;  A (dz): channel co0 + 16 (COT cog + i) + r of pixel 4 s + q
  int xe[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int col = 2 * q + kx;
    const int quad = cit * 4 + (r >> 2);
    const int slot = SWZ ? (quad ^ (((col >> 1) & 1) << 2)) : quad;    // (8 s + col) >> 1 has the parity of col >> 1
    xe[kx] = (col * XPQ + slot) * 4 + (r & 3);
  }
  int ze[COT];
#pragma unroll
  for (int i = 0; i < COT; ++i) {
    const int quad = (cog * COT + i) * 4 + (r >> 2);
    ze[i] = (q * ZQ + (quad ^ ((q & 1) << 2))) * 4 + (r & 3);           // pixel 4 s + q has the parity of q
  }

  int buf = 0;
  for (; tile < tend; ++tile) {
    const bool more = tile + 1 < tend;
    int n2 = n, ty2 = ty, tx2 = tx;
    // (Issuing the DMA of half of the waves mid-tile, which pays on the input-gradient twin of this kernel, measured
    // 0.4 % slower here.)
    if (more) {
      advance(n2, ty2, tx2);
      if (NBUF == 2) dma_tile(buf ^ 1, n2, ty2, tx2);     // lands behind this tile's MFMAs
    }
    if (cib == 0) {                                       // only the first ci block's slab carries the bias gradient (the others' sums
                                                          // were 8 VALU adds + 2 LDS reads per tile for nothing: -2...-3 us per step)
#pragma unroll
      for (int k = 0; k < NDB; ++k)                       // bias gradient: NDB dz granules per thread and tile
        if (tid + k * NT < Z_F4) dbsum[k] += sZ[buf * Z_F4 + tid + k * NT];
    }
    const float* hx = reinterpret_cast<const float*>(sX + buf * X_F4);
    const float* hz = reinterpret_cast<const float*>(sZ + buf * Z_F4);
    // software pipeline over the TH * KG k-groups: the fragments of group u + 1 are read while group u's MFMAs run
    float a[2][COT], b[2][9];
    auto load_frags = [&](int u, int set) {
      const int oyl = u / KG, s = u - oyl * KG;
#pragma unroll
      for (int i = 0; i < COT; ++i) a[set][i] = hz[(oyl * TW + 4 * s) * (ZQ * 4) + ze[i]];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        b[set][t] = hx[((2 * oyl + ky) * HX + 8 * s) * XPQ * 4 + xe[kx]];
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int u = 0; u < TH * KG; ++u) {
      // NB pinning this order with sched_barrier (reads of group u + 1 strictly before the MFMAs of group u) was
      // measured 5 % SLOWER: the compiler then waits with lgkmcnt(0) in front of every MFMA group anyway
      if (u + 1 < TH * KG) load_frags(u + 1, (u + 1) & 1);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < COT; ++i)
          acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 1][i], b[u & 1][t], acc[t][i], 0, 0, 0);
    }
    if (NBUF == 2) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      buf ^= 1;
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done reading the images
      if (more) {
        dma_tile(0, n2, ty2, tx2);                        // the CU's other block computes meanwhile
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    n = n2; ty = ty2; tx = tx2;
  }

  // ---- epilogue: lane owns row (tap, ci = ci0 + 16 cit + r), co = co0 + 16 (COT cog + i) + 4 q .. +3 ------------------
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < COT; ++i)
      *reinterpret_cast<f32x4*>(part + ((long long)t * Cin + ci0 + 16 * cit + r) * Cout + co0 + 16 * (cog * COT + i) + 4 * q) =
          acc[t][i];
  // ---- bias gradient (blocks of the first ci block only): per-thread granule sums -> LDS -> fixed-order column sums -----
  if (cib == 0) {
    f32x4* sB = sX;                                       // the main loop ended with a barrier: LDS is free
#pragma unroll
    for (int k = 0; k < NDB; ++k)
      if (tid + k * NT < Z_F4) sB[tid + k * NT] = dbsum[k];
    __syncthreads();
    if (tid < COB) {
      const float* bf = reinterpret_cast<const float*>(sB);
      float s1 = 0.f;
#pragma unroll 4
      for (int px = 0; px < TH * TW; ++px)
        s1 += bf[(px * ZQ + ((tid >> 2) ^ (((px % TW) & 1) << 2))) * 4 + (tid & 3)];
      part[9ll * Cin * Cout + co0 + tid] = s1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
struct WgradHaloPlan {
  int variant;       // 0 = not handled; 1 = CIB 48 (conv3 type), TW 16; 4 = CIB 48, TW 8; 2 = CIB 64, TW 16; 3 = CIB 64, TW 8;
                     // 5 = CIB 32 x COB 128, TW 8
  int TH, TW, n_cib, n_cob, S;
  int bpc;           // blocks per CU: 1 = double-buffered images, 2 = single-buffered
};

static WgradHaloPlan wgrad_halo_plan(int groups, int N, int H, int W, int Cin, int Cout, int stride) {
  WgradHaloPlan pl = {};
  static const int disabled = (geeco_dev_getenv("GEECO_NO_HALO") || geeco_dev_getenv("GEECO_NO_WGRAD_LDS")) ? 1 : 0;
  if (disabled || stride != 2 || (H & 1) || (W & 1) || Cout % 64 != 0) return pl;
  const int Ho = H / 2, Wo = W / 2;
  if (Wo < 8 || Ho < 2) return pl;                        // the tiny top layers stay with the gather kernel
  // tile shape: 2 x 16 or 4 x 8 output pixels (same LDS); the squarer one has 7 % less halo ((9 x 17) / (8 x 16) = 1.20
  // input pixels fetched per input pixel used, against (5 x 33) / (4 x 32) = 1.29)
  // (measured, bench shapes: conv3 209.8 -> 203.5 us, conv4 142.3 -> 139.0, conv5 116.5 -> 113.9)
  static const int sq_env = geeco_dev_getenv("GEECO_WGRAD_SQUARE") ? atoi(geeco_dev_getenv("GEECO_WGRAD_SQUARE")) : 1;
  const bool wide = Wo >= 16 && !(sq_env && Ho >= 4);
  if (Cin == 48 && Wo >= 16) {
    pl.variant = wide ? 1 : 4; pl.TH = wide ? 2 : 4; pl.TW = wide ? 16 : 8; pl.n_cib = 1;
  } else if (Cin % 64 == 0) {
    pl.variant = wide ? 2 : 3;
    pl.TH = wide ? 2 : 4; pl.TW = wide ? 16 : 8;
    pl.n_cib = Cin / 64;
  } else {
    return pl;
  }
  pl.n_cob = Cout / 64;
  const long long tiles = (long long)N * cdiv(Ho, pl.TH) * cdiv(Wo, pl.TW);
  if (tiles >= (1ll << 30) || (long long)H * W * Cin >= (1ll << 31) || (long long)Ho * Wo * Cout >= (1ll << 31)) {
    pl.variant = 0;                                       // 32-bit tile counters / in-frame offsets
    return pl;
  }
  // One block per CU (its LDS images take > 80 KB).  Blocks are dealt round-robin to the 8 XCDs and the n_cob co
  // blocks of a slice sit on one XCD: a launch must not put more than 32 blocks on any XCD, or that XCD runs two
  // rounds while the others idle (measured on conv5: 33 blocks on four XCDs took 200 us instead of 100).
  static const int bpc_env = geeco_dev_getenv("GEECO_WGRAD_BPC") ? atoi(geeco_dev_getenv("GEECO_WGRAD_BPC")) : 0;
  pl.bpc = bpc_env == 1 || bpc_env == 2 ? bpc_env : 1;
  // 32 x 128 blocks of dw instead of 64 x 64 (same slab bytes): 35.6 KB of DMA per tile instead of 47 KB for the same
  // MFMA work - the 64 x 64 blocks sit at the CU's ingest limit (5.1 B/clk next to MFMA waves)
  static const int cob128 = geeco_dev_getenv("GEECO_WGRAD_NO_COB128") ? 0 : 1;
  if (cob128 && pl.bpc == 1 && pl.variant == 3 && Cout % 128 == 0) {
    pl.variant = 5;
    pl.n_cib = Cin / 32;
    pl.n_cob = Cout / 128;
  }
  // 64 x 96 blocks where 128 does not divide Cout (conv5: 192 = 2 x 96): 256 blocks instead of 240 and 27 MFMAs per 12
  // fragment reads instead of 18 per 11, against 1.5x the slab bytes: 114.4 -> 111.6 us, the step -2.5 us
  static const int cob96 = geeco_dev_getenv("GEECO_WGRAD_NO_COB96") ? 0 : 1;
  if (cob96 && pl.bpc == 1 && pl.variant == 3 && Cout % 96 == 0) {
    pl.variant = 6;
    pl.n_cob = Cout / 96;
  }
  int S = (8 * (32 * pl.bpc / pl.n_cob)) / (groups * pl.n_cib);
  if (S < 1) S = 1;
  if (S > tiles) S = (int)tiles;
  pl.S = S;
  return pl;
}

int64_t geeco_wgrad_lds_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride) {
  const WgradHaloPlan pl = wgrad_halo_plan(groups, N, H, W, Cin, Cout, stride);
  if (!pl.variant) return 0;
  return (int64_t)groups * pl.S * (9ll * Cin * Cout + Cout) * 4;
}

template <int NCI, int NCO, int COT, int TH, int TW, int XPQ, bool SWZ, int NBUF, int MINW>
static int launch_wgrad_lds(const WgradHaloParams& p, int blocks, hipStream_t stream) {
  constexpr int X_F4 = ((2 * TH + 1) * (2 * TW + 1) * XPQ + 63) / 64 * 64;
  constexpr size_t lds = (size_t)(NBUF * X_F4 + NBUF * TH * TW * 4 * NCO * COT) * 16;
  static_assert(lds * (3 - NBUF) <= 160 * 1024, "LDS budget (two blocks per CU when single-buffered)");
  static int attr_state = 0;          // 0 = not set; set once (idempotent: racing threads set the same value)
  if (__atomic_load_n(&attr_state, __ATOMIC_ACQUIRE) == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_wgrad_lds_kernel<NCI, NCO, COT, TH, TW, XPQ, SWZ, NBUF, MINW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    __atomic_store_n(&attr_state, 1, __ATOMIC_RELEASE);
  }
  geeco_note_kernel("conv_s2_wgrad_lds_kernel<%d, %d, %d, %d, %d, %d, %s, %d, %d>", NCI, NCO, COT, TH, TW, XPQ, SWZ ? "true" : "false", NBUF, MINW);
  hipLaunchKernelGGL((conv_s2_wgrad_lds_kernel<NCI, NCO, COT, TH, TW, XPQ, SWZ, NBUF, MINW>), dim3((unsigned)blocks), dim3(64 * NCI * NCO), lds,
                     stream, p);
  return 0;
}

int geeco_try_wgrad_lds(const float* x, const float* dz, float* dw, float* db, int groups, int64_t gs_x, int64_t gs_dz,
                        int64_t gs_dw, int64_t gs_db, int N, int H, int W, int Cin, int Cout, int stride, void* ws,
                        hipStream_t stream, int* handled) {
  *handled = 0;
  const WgradHaloPlan pl = wgrad_halo_plan(groups, N, H, W, Cin, Cout, stride);
  if (!pl.variant) return 0;
  WgradHaloParams p = {};
  p.x = x; p.dz = dz; p.part = (float*)ws; p.gs_x = gs_x; p.gs_dz = gs_dz;
  {
    static const float* zero_page = nullptr;      // resolved once (idempotent: racing threads store the same address)
    const float* z = __atomic_load_n(&zero_page, __ATOMIC_ACQUIRE);
    if (!z) {
      void* sym = nullptr;
      hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_zero_page));
      if (e != hipSuccess || !sym) {
        geeco_set_error("hipGetSymbolAddress(g_zero_page) failed: %s", hipGetErrorString(e));
        return (int)(e != hipSuccess ? e : hipErrorInvalidSymbol);
      }
      z = (const float*)sym;
      __atomic_store_n(&zero_page, z, __ATOMIC_RELEASE);
    }
    p.zero = z;
  }
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.Cin = Cin; p.Cout = Cout;
  p.tiles_x = cdiv(p.Wo, pl.TW); p.tiles_y = cdiv(p.Ho, pl.TH);
  p.tiles_per_group = N * p.tiles_x * p.tiles_y;
  p.S = pl.S; p.n_cib = pl.n_cib; p.n_cob = pl.n_cob;
  p.n_sigma = groups * pl.n_cib * pl.S;
  const int blocks = 8 * cdiv(p.n_sigma, 8) * pl.n_cob;
  int rc = 0;
  if (pl.bpc == 1) {
    switch (pl.variant) {
      case 1: rc = launch_wgrad_lds<3, 4, 1, 2, 16, 14, false, 2, 3>(p, blocks, stream); break;
      case 4: rc = launch_wgrad_lds<3, 4, 1, 4, 8, 14, false, 2, 3>(p, blocks, stream); break;
      case 2: rc = launch_wgrad_lds<4, 2, 2, 2, 16, 16, true, 2, 2>(p, blocks, stream); break;
      case 5: rc = launch_wgrad_lds<2, 4, 2, 4, 8, 10, false, 2, 2>(p, blocks, stream); break;
      case 6: rc = launch_wgrad_lds<4, 2, 3, 4, 8, 16, true, 2, 2>(p, blocks, stream); break;
      default: rc = launch_wgrad_lds<4, 2, 2, 4, 8, 16, true, 2, 2>(p, blocks, stream); break;
    }
  } else {
#ifdef GEECO_DEV_KERNELS      // GEECO_WGRAD_BPC=2: single-buffered images, two blocks per CU
    switch (pl.variant) {
      case 1: rc = launch_wgrad_lds<3, 4, 1, 2, 16, 14, false, 1, 6>(p, blocks, stream); break;
      case 4: rc = launch_wgrad_lds<3, 4, 1, 4, 8, 14, false, 1, 6>(p, blocks, stream); break;
      case 2: rc = launch_wgrad_lds<4, 2, 2, 2, 16, 16, true, 1, 4>(p, blocks, stream); break;
      default: rc = launch_wgrad_lds<4, 2, 2, 4, 8, 16, true, 1, 4>(p, blocks, stream); break;
    }
#else
    return 0;                 // (unreachable: the product plan never asks for two blocks per CU)
#endif
  }
  if (rc) return rc;
  GEECO_LAUNCH_CHECK();
  geeco_launch_wgrad_reduce((const float*)ws, dw, db, gs_dw, gs_db, pl.S, 9ll * Cin * Cout, Cout, groups, stream);
  GEECO_LAUNCH_CHECK();
  *handled = 1;
  return 0;
}
