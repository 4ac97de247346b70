// Input gradient of the stride-2 middle layers (conv4 .. conv6 of the encoder) with the weights and the dz halo streamed
// through LDS by LDS-DMA in 16-output-channel chunks, fused with the ReluGrad of the layer below.
//
// Autodiff of reference src/models/e2evmc/graph.py:91-105 (3x3, stride 2, TF SAME, even sizes => pad_before = 0) taken
// by tf.train.AdamOptimizer.minimize (src/models/e2evmc/estimator.py:243-244):
//
//   dx[2Y+py][2X+px][ci] = (y[2Y+py][2X+px][ci] > 0) * sum_{taps of class (py,px)} sum_co dz[Y+dy][X+dx][co] w[ky][kx][ci][co]
//   py = 0: (ky = 0, dy = 0), (ky = 2, dy = -1);   py = 1: (ky = 1, dy = 0)     (same in x): classes with 4 / 2 / 2 / 1 taps.
//
// Why: the gather GEMM (conv_gemm.hip) serves these layers at 46-61 % (conv6: 33 %) of the MFMA peak: every block gathers
// its dz rows again per tap from L2, the four parity classes have short, unequal K loops and blocks are prologue
// dominated.  Here a 512-thread block owns 8 groups of 16 class pixels (a 16 x 32 input-pixel tile for wide images, two
// whole 16 x 16 frames for conv6) x 64 input channels and loops over Cout in chunks of 16: per chunk the 9 x 64 x 16
// weight slab (read straight from the HWIO kernel: 64 contiguous bytes per (tap, ci) row) and the dz halo of the tile
// land in LDS by LDS-DMA, double buffered behind the previous chunk's MFMAs; the stream of (tile, chunk) steps is
// prefetched across tile boundaries.  All four classes of a pixel group are accumulated by ONE wave (16 accumulator
// tiles), so the nine taps share four dz fragments and the class imbalance disappears.
//
// MFMA v_mfma_f32_16x16x4_f32, row i = ci, column j = pixel, k = 4 consecutive co: both operands are ds_read_b128 of
// naturally contiguous data (weights: 16 ci rows x 64 B; dz: 16 pixels x 64 B), each feeding 4 MFMAs; a lane ends up
// with 4 consecutive NHWC channels of one pixel = one 16-byte masked store.
#include "geeco_common.h"
#include <atomic>
#include <stdlib.h>

static __device__ float g_zero_page[64];   // source of the DMA lanes that fall outside the image

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct DgradLdsParams {
  const float* dz;       // [G][N][Ho][Wo][Cout]
  const float* w;        // HWIO [G][9][Cin][Cout]
  const float* mask;     // y of the layer below [G][N][H][W][Cin] (ReluGrad) or null
  const unsigned char* fields;   // instead of mask: its sign fields [G][N][H][W][Cin / 8] (byte (T >> 1) * 4 + q, bit 4 (T & 1) + j <-> channel 16 T + 4 q + j)
  long long gs_fields;
  float* dx;             // [G][N][H][W][Cin]
  long long gs_dz, gs_w, gs_dx;
  int N, H, W, Ho, Wo, Cin, Cout;
  int n_cib;             // Cin / (16 NCIT)
  int tiles_y, tiles_x;  // tiles per frame (wide images) -- 1 x 1 for the two-frame tiles
  int tiles_per_group;   // tiles of one encoder
  int items;             // G * n_cib * tiles_per_group
  int stagger;           // waves 4-7 issue their DMA pieces mid-chunk
};

__device__ __forceinline__ bool getenv_stagger(const DgradLdsParams& p) { return p.stagger != 0; }

// A group = PR x PC = 16 class pixels; a tile = 8 groups stacked in y (FR frames per tile: 1 for wide images, 2 when a
// frame holds only 4 groups).  NCIT = ci tiles of 16 per block: 4 (64 channels, 94 KB of LDS, one block per CU) or 2 (32
// channels, 57 KB, TWO blocks per CU: twice as many, half as heavy items when 64-channel items cannot fill the chip evenly).
// Blocks take items blockIdx.x, blockIdx.x + gridDim.x, ...: with two blocks per CU, blocks b and b + 256 tend to share a
// CU, so a CU's total stays balanced when the item count is not a multiple of the grid.
// NW = waves (= pixel groups) per block: 8, or 4 for the smallest layer (one 16 x 16 frame per tile, 47 KB of LDS, three
// blocks per CU) so that its few tiles still spread over the whole chip.
template <int PR, int PC, int FR, int NCIT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : (NCIT == 4 ? 2 : 4)) void conv_s2_dgrad_lds_kernel(const DgradLdsParams p) {
  static_assert(PR * PC == 16, "16 pixels per MFMA column group");
  constexpr int GPF = NW / FR;                      // groups per frame of the tile
  constexpr int IR = GPF * PR + 1, IC = PC + 1;     // dz halo image of one frame: rows -1 .. GPF*PR-1, cols -1 .. PC-1
  // LDS images in 16-byte granules (4 co).  A ds_read_b128 is served in four groups of 16 lanes, each holding every
  // r = lane & 15 exactly once (from two different q = lane >> 4): the 16 lanes hit 16 distinct bank quads iff the granule
  // index is  (multiple of 16) * q + r-dependent part with 16 consecutive values.  Hence q-major images:
  //   weights [tap][ci tile][q][16 ci rows],  dz [q][pixel] planes of NPIXP (multiple of 16) pixels.
  // (The natural [row][q] order gives a 2-way conflict on every read: measured SQ_LDS_BANK_CONFLICT 50 %.)
  constexpr int NPIX = FR * IR * IC;
  constexpr int NPIXP = (NPIX + 15) / 16 * 16;
  constexpr int Z_F4 = 4 * NPIXP;
  constexpr int NZP = (Z_F4 + 63) / 64;
  constexpr int ZP_F4 = NZP * 64;
  constexpr int CIB = 16 * NCIT;
  constexpr int W_F4 = 9 * NCIT * 64;               // [tap][ci tile][q 4][ci row 16]
  constexpr int NWP = 9 * NCIT;                     // one piece per (tap, ci tile)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sW = reinterpret_cast<f32x4*>(smem);       // 2 weight chunks
  f32x4* sZ = sW + 2 * W_F4;                        // 2 dz halo chunks

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int Cin = p.Cin, Cout = p.Cout;
  const int nch = Cout >> 4;

  int item = blockIdx.x;
  const int item_step = gridDim.x;
  if (item >= p.items) return;

  // ---- DMA pieces of this wave.  Weight pieces kw = wid + NW i (i < NSW) and dz pieces kz = wid + NW j (j < NSZ) are numbered
  // separately, so that a slot has ONE role for every wave and only the last slot of each kind needs a (wave-uniform) range test:
  // with the combined numbering of rounds 2-4 every slot carried two scalar branches and the pointer of each piece travelled
  // through v_mov chains between their arms (94 branches, 118 v_mov per 72 MFMAs in the chunk loop).
  constexpr int NSW = (NWP + NW - 1) / NW, NSZ = (NZP + NW - 1) / NW;
  __builtin_assume(wid >= 0 && wid < NW);
  unsigned w_off[NSW];               // byte offset of this lane's granule from the chunk's weight base (uniform base + 32-bit lane offset)
  int d_off[NSZ];
  short d_a[NSZ], d_b[NSZ];          // dz pieces: frame-local row / col (-1 based) ; d_f: frame of the tile
  signed char d_f[NSZ];
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int k = wid + NW * i;       // piece k = (tap, ci tile): lane = q * 16 + ci row
    const int tap = k / NCIT, cit = k - tap * NCIT;
    w_off[i] = (unsigned)(((tap * Cin + cit * 16 + (lane & 15)) * Cout + (lane >> 4) * 4) * 4);
  }
#pragma unroll
  for (int j = 0; j < NSZ; ++j) {
    const int k = wid + NW * j;
    const int sl = k * 64 + lane;                   // granule index inside the dz image: q * NPIXP + pixel
    const int qq = sl / NPIXP, px = sl - qq * NPIXP;
    const int f = px / (IR * IC), rem = px - f * (IR * IC);
    const int ir = rem / IC, ic = rem - ir * IC;
    const bool ok = k < NZP && sl < Z_F4 && px < NPIX;
    d_f[j] = (signed char)(ok ? f : 100);
    d_a[j] = (short)(ir - 1);
    d_b[j] = (short)(ic - 1);
    d_off[j] = ((ir - 1) * p.Wo + (ic - 1)) * Cout + qq * 4;
  }
  // item -> (encoder g, ci block, frame n0, class-pixel origin Y0, X0)
  auto decode = [&](int it, int& g_, int& cib_, int& n0_, int& y0_, int& x0_) {
    const int per_g = p.n_cib * p.tiles_per_group;
    g_ = it / per_g;
    int rem = it - g_ * per_g;
    cib_ = rem / p.tiles_per_group;
    const int t = rem - cib_ * p.tiles_per_group;
    const int tpf = p.tiles_y * p.tiles_x;
    const int fi = t / tpf, tt = t - fi * tpf;
    n0_ = fi * FR;
    const int ty = tt / p.tiles_x;
    y0_ = ty * (GPF * PR);
    x0_ = (tt - ty * p.tiles_x) * PC;
  };
  // Sources of this wave's pieces for the NEXT chunk to fetch, set once per item and advanced by 16 channels per chunk: the
  // weights as ONE uniform base (scalar add per chunk; lane offsets never change), the dz pieces as per-lane pointers (bounds
  // tests, frame offsets: ~12 VALU instructions per piece and item; lanes outside the image stay on the zero page, step 0).
  const float* w_next = nullptr;
  const float* d_ptr[NSZ];
  int d_step[NSZ];
  auto setup_item = [&](int g_, int cib_, int n0_, int y0_, int x0_) {
    w_next = p.w + (long long)g_ * p.gs_w + (long long)cib_ * CIB * Cout;
    const float* zg = p.dz + (long long)g_ * p.gs_dz + (((long long)n0_ * p.Ho + y0_) * p.Wo + x0_) * Cout;
#pragma unroll
    for (int j = 0; j < NSZ; ++j) {
      const int f = d_f[j];
      const bool v = n0_ + f < p.N && (unsigned)(y0_ + d_a[j]) < (unsigned)p.Ho && (unsigned)(x0_ + d_b[j]) < (unsigned)p.Wo;
      d_ptr[j] = v ? zg + (long long)f * p.Ho * p.Wo * Cout + d_off[j] : g_zero_page;
      d_step[j] = v ? 16 : 0;
    }
  };
  auto dma_next = [&](int buf) {                    // the chunk the sources point at -> LDS buffer buf; then one chunk further
    const char* wb = reinterpret_cast<const char*>(w_next);
#pragma unroll
    for (int i = 0; i < NSW; ++i)
      if (NW * (i + 1) <= NWP || wid + NW * i < NWP)      // compile-time true except in the last slot
        __builtin_amdgcn_global_load_lds((gptr_t)(wb + w_off[i]), (lptr_t)(sW + buf * W_F4 + (wid + NW * i) * 64), 16, 0, 0);
    w_next += 16;
#pragma unroll
    for (int j = 0; j < NSZ; ++j)
      if (NW * (j + 1) <= NZP || wid + NW * j < NZP) {
        __builtin_amdgcn_global_load_lds((gptr_t)d_ptr[j], (lptr_t)(sZ + buf * ZP_F4 + (wid + NW * j) * 64), 16, 0, 0);
        d_ptr[j] += d_step[j];
      }
  };

  // ---- this wave's pixel group and fragment offsets (in float4 granules) ----------------------------------------------
  const int gf = wid / GPF, gyl = wid - gf * GPF;   // frame of the tile, group row inside the frame
  const int pr = r / PC, pc = r - pr * PC;
  const int zpix = (gf * IR + gyl * PR + pr + 1) * IC + pc + 1;       // image pixel of lane r at shift (0, 0)
  int zoff[4];                                      // shifts (dy, dx) = (0,0), (0,-1), (-1,0), (-1,-1)
  zoff[0] = q * NPIXP + zpix; zoff[1] = zoff[0] - 1; zoff[2] = zoff[0] - IC; zoff[3] = zoff[0] - IC - 1;
  const int woff = q * 16 + r;                      // + (tap * 4 + cit) * 64

  int g, cib, n0, y0, x0;
  decode(item, g, cib, n0, y0, x0);
  setup_item(g, cib, n0, y0, x0);
  dma_next(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const bool late = wid >= NW / 2 && getenv_stagger(p);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4][NCIT];                               // [class py * 2 + px][ci tile]
  int buf = 0;
  while (true) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < NCIT; ++t) acc[c][t] = zero4;
    int gn = g, cibn = cib, n0n = n0, y0n = y0, x0n = x0;
    const bool more_items = item + item_step < p.items;
    if (more_items) decode(item + item_step, gn, cibn, n0n, y0n, x0n);
    for (int ch = 0; ch < nch; ++ch) {
      // prefetch the next (item, chunk) step: the stream crosses tile boundaries.  Waves 0-3 issue their DMA pieces at
      // the head of the chunk, waves 4-7 (their SIMD partners) in the middle of the tap loop: an LDS-DMA instruction
      // holds its wave for 100-200 cycles, and partners that stall at the same time leave the SIMD's MFMA pipe idle.
      auto prefetch = [&]() {
        if (ch + 1 < nch) {
          dma_next(buf ^ 1);
        } else if (more_items) {
          setup_item(gn, cibn, n0n, y0n, x0n);
          dma_next(buf ^ 1);
        }
      };
      if (!late) prefetch();
      const f32x4* zb = sZ + buf * ZP_F4;
      const f32x4* wb = sW + buf * W_F4 + woff;
      f32x4 bz[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bz[s] = zb[zoff[s]];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int cls = (ky == 1 ? 2 : 0) + (kx == 1 ? 1 : 0);
          const int sh = (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0);
          const int tap = ky * 3 + kx;
          if (tap == 4 && late) prefetch();
          f32x4 a[NCIT];
#pragma unroll
          for (int t = 0; t < NCIT; ++t) a[t] = wb[(tap * NCIT + t) * 64];
          // k component outer, ci tile inner: an accumulator is touched again only after 3 other MFMAs (the dependent
          // latency of v_mfma_f32_16x16x4_f32 is 40 cycles against 32 of issue)
#pragma unroll
          for (int t = 0; t < NCIT; ++t) acc[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, bz[sh].x, acc[cls][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NCIT; ++t) acc[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, bz[sh].y, acc[cls][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NCIT; ++t) acc[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, bz[sh].z, acc[cls][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NCIT; ++t) acc[cls][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, bz[sh].w, acc[cls][t], 0, 0, 0);
        }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      buf ^= 1;
    }
    // ---- epilogue: ReluGrad mask + 16-byte stores; lane = pixel r of the group, channels ci0 + 16 t + 4 q .. +3 -----
    {
      const int n = n0 + gf;
      if (n < p.N) {
        const int Y = y0 + gyl * PR + pr, X = x0 + pc;
        const long long gbase = (long long)g * p.gs_dx;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int py = c >> 1, px = c & 1;
          const long long pix = ((long long)n * p.H + 2 * Y + py) * p.W + 2 * X + px;
          const long long o = gbase + pix * Cin + cib * CIB + 4 * q;
          // sign fields: one byte per pair of channel tiles (this lane's quad q) instead of NCIT float4 of the activation
          unsigned fld[(NCIT + 1) / 2];
          if (p.fields) {
            const unsigned char* fp = p.fields + (long long)g * p.gs_fields + pix * (Cin >> 3) + ((cib * NCIT) >> 1) * 4 + q;
#pragma unroll
            for (int u = 0; u < (NCIT + 1) / 2; ++u) fld[u] = fp[4 * u];
          }
#pragma unroll
          for (int t = 0; t < NCIT; ++t) {
            f32x4 v = acc[c][t];
            if (p.fields) {
              const int T = cib * NCIT + t;      // (NCIT even: T & 1 == t & 1, the byte index is u = t >> 1)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                v[j] = __int_as_float(__float_as_int(v[j]) & __builtin_amdgcn_sbfe((int)fld[t >> 1], 4 * (T & 1) + j, 1));
            } else if (p.mask) {
              const f32x4 m = *reinterpret_cast<const f32x4*>(p.mask + o + 16 * t);
              v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
              v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<f32x4*>(p.dx + o + 16 * t) = v;
          }
        }
      }
    }
    if (!more_items) break;
    item += item_step;
    g = gn; cib = cibn; n0 = n0n; y0 = y0n; x0 = x0n;
  }
}

// ------------------------------------------------------------------------------------------------------------------
template <int PR, int PC, int FR, int NCIT, int NW = 8>
static int launch_dgrad_lds(const DgradLdsParams& p, int blocks, hipStream_t stream) {
  constexpr int IR = (NW / FR) * PR + 1, IC = PC + 1;
  constexpr int ZP_F4 = (4 * ((FR * IR * IC + 15) / 16 * 16) + 63) / 64 * 64;
  constexpr size_t lds = (size_t)(2 * 9 * NCIT * 64 + 2 * ZP_F4) * 16;
  static_assert(lds * (NW == 4 ? 3 : (NCIT == 4 ? 1 : 2)) <= 160 * 1024, "LDS budget");
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: racing threads at worst repeat it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_dgrad_lds_kernel<PR, PC, FR, NCIT, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      geeco_set_error("hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  geeco_note_kernel("conv_s2_dgrad_lds_kernel<%d, %d, %d, %d, %d>", PR, PC, FR, NCIT, NW);
  hipLaunchKernelGGL((conv_s2_dgrad_lds_kernel<PR, PC, FR, NCIT, NW>), dim3((unsigned)blocks), dim3(64 * NW), lds, stream, p);
  return 0;
}

// does the dispatcher below take this shape (given the HWIO kernel)?  Such layers never read the transposed copy.
int geeco_dgrad_lds_handles(int H, int W, int Cin, int Cout, int stride) {
  static const int disabled = (geeco_dev_getenv("GEECO_NO_HALO") || geeco_dev_getenv("GEECO_NO_DGRAD_LDS")) ? 1 : 0;
  if (disabled || stride != 2 || (H & 1) || (W & 1) || Cin % 64 != 0 || Cout % 16 != 0 || Cout < 32) return 0;
  const int Ho = H / 2, Wo = W / 2;
  if (!((Ho % 8 == 0 && Wo % 16 == 0) || (Ho == 8 && Wo == 8))) return 0;
  if ((long long)H * W * Cin >= (1ll << 31) || (long long)Ho * Wo * Cout >= (1ll << 31) || 9ll * Cin * Cout >= (1ll << 30)) return 0;      // (weight granules are addressed by 32-bit BYTE offsets)
  return 1;
}

static int dgrad_lds_impl(const float* dz, const float* w_hwio, const float* ymask, const unsigned char* fields,
                          int64_t gs_fields, float* dx, int groups, int64_t gs_dz, int64_t gs_w, int64_t gs_dx, int N, int H,
                          int W, int Cin, int Cout, int stride, hipStream_t stream, int* handled);

int geeco_try_dgrad_lds(const float* dz, const float* w_hwio, const float* ymask, float* dx, int groups, int64_t gs_dz,
                        int64_t gs_w, int64_t gs_dx, int N, int H, int W, int Cin, int Cout, int stride,
                        hipStream_t stream, int* handled) {
  return dgrad_lds_impl(dz, w_hwio, ymask, nullptr, 0, dx, groups, gs_dz, gs_w, gs_dx, N, H, W, Cin, Cout, stride, stream,
                        handled);
}

// shapes geeco_conv3x3_dgrad_relu_fields serves (the LDS-staged kernel's, with whole 32-channel tile pairs)
extern "C" int geeco_conv3x3_dgrad_relu_fields_supported(int H, int W, int Cin, int Cout, int stride) {
  return geeco_dgrad_lds_handles(H, W, Cin, Cout, stride) && Cin % 32 == 0;
}

// geeco_conv3x3_dgrad for the shapes this file serves, with the ReluGrad mask given as sign fields
extern "C" int geeco_conv3x3_dgrad_relu_fields(const float* dz, const float* w, const uint8_t* y_fields, float* dx, int groups,
                                               int64_t gs_dz, int64_t gs_w, int64_t gs_fields, int64_t gs_dx, int N, int H,
                                               int W, int Cin, int Cout, int stride, void* stream) {
  GEECO_CHECK_ARG(dz && w && y_fields && dx, "conv3x3_dgrad_relu_fields: null pointer");
  GEECO_CHECK_ARG(Cin % 32 == 0, "conv3x3_dgrad_relu_fields: Cin = %d (whole tile pairs)", Cin);
  int handled = 0;
  const int rc = dgrad_lds_impl(dz, w, nullptr, y_fields, gs_fields, dx, groups, gs_dz, gs_w, gs_dx, N, H, W, Cin, Cout, stride,
                                (hipStream_t)stream, &handled);
  if (rc) return rc;
  GEECO_CHECK_ARG(handled, "conv3x3_dgrad_relu_fields: shape %dx%d, %d -> %d channels, stride %d is not served by the "
                           "LDS-staged input-gradient kernel", H, W, Cin, Cout, stride);
  return 0;
}

static int dgrad_lds_impl(const float* dz, const float* w_hwio, const float* ymask, const unsigned char* fields,
                          int64_t gs_fields, float* dx, int groups, int64_t gs_dz, int64_t gs_w, int64_t gs_dx, int N, int H,
                          int W, int Cin, int Cout, int stride, hipStream_t stream, int* handled) {
  *handled = 0;
  static const int disabled = (geeco_dev_getenv("GEECO_NO_HALO") || geeco_dev_getenv("GEECO_NO_DGRAD_LDS")) ? 1 : 0;
  if (disabled || !w_hwio || stride != 2 || (H & 1) || (W & 1) || Cin % 64 != 0 || Cout % 16 != 0 || Cout < 32) return 0;
  const int Ho = H / 2, Wo = W / 2;
  int variant = 0;
  if (Ho % 8 == 0 && Wo % 16 == 0) variant = 1;          // 8 groups of 1 x 16 class pixels: a 16 x 32 input-pixel tile
  else if (Ho == 8 && Wo == 8) variant = 2;               // 2 x 8 groups: a tile = two whole frames
  if (!variant) return 0;
  if ((long long)H * W * Cin >= (1ll << 31) || (long long)Ho * Wo * Cout >= (1ll << 31) || 9ll * Cin * Cout >= (1ll << 30)) return 0;      // (weight granules are addressed by 32-bit BYTE offsets)
  DgradLdsParams p = {};
  p.dz = dz; p.w = w_hwio; p.mask = ymask; p.fields = fields; p.gs_fields = gs_fields; p.dx = dx; p.gs_dz = gs_dz; p.gs_w = gs_w; p.gs_dx = gs_dx;
  p.N = N; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Cin = Cin; p.Cout = Cout;
  static const int no_stagger = geeco_dev_getenv("GEECO_DGRAD_NO_STAGGER") ? 1 : 0;
  p.stagger = !no_stagger;
  long long tiles;
  if (variant == 1) {
    p.tiles_y = Ho / 8; p.tiles_x = Wo / 16;
    tiles = (long long)N * p.tiles_y * p.tiles_x;
  } else {
    p.tiles_y = 1; p.tiles_x = 1;
    tiles = (N + 1) / 2;
  }
  static const int no_small = geeco_dev_getenv("GEECO_DGRAD_NO_SMALL") ? 1 : 0;
  if (variant == 2 && !no_small) {
    // 8 x 8 class pixels per frame: one-frame tiles of 4 groups, 32-channel items, 256-thread blocks, three per CU.  The
    // two-frame / 8-wave form has only groups * (Cin / 64) * N / 2 items (conv6 of the bench: 144 for 256 CUs).
    p.tiles_per_group = N;
    p.n_cib = Cin / 32;
    const long long items = (long long)groups * p.n_cib * N;
    if (items >= (1ll << 30)) return 0;
    p.items = (int)items;
    const int blocks = p.items < 768 ? p.items : 768;
    int rcs = launch_dgrad_lds<2, 8, 1, 2, 4>(p, blocks, stream);
    if (rcs) return rcs;
    GEECO_LAUNCH_CHECK();
    *handled = 1;
    return 0;
  }
  p.tiles_per_group = (int)tiles;
  // 64-channel items (one block per CU) or 32-channel items (two blocks per CU): whichever spreads the launch more evenly
  // over the 256 CUs; cost of an item in CU-time: 1 resp. 1/2.  Ties go to the 64-channel form (more reuse per staged byte).
  const long long items64 = (long long)groups * (Cin / 64) * tiles;
  if (items64 * 2 >= (1ll << 30)) return 0;
  static const int force_ncit = geeco_dev_getenv("GEECO_DGRAD_NCIT") ? atoi(geeco_dev_getenv("GEECO_DGRAD_NCIT")) : 0;
  const double span64 = (double)((items64 + 255) / 256), span32 = 0.5 * (double)((2 * items64 + 255) / 256);
  const bool use32 = force_ncit == 2 || (force_ncit != 4 && span32 < span64);
  int rc;
  if (use32) {
    p.n_cib = Cin / 32;
    p.items = (int)(2 * items64);
    const int blocks = p.items < 512 ? p.items : 512;
#ifdef GEECO_DEV_KERNELS      // variant 2 only gets here with GEECO_DGRAD_NO_SMALL (two-frame tiles of the 8 x 8 layers)
    rc = variant == 1 ? launch_dgrad_lds<1, 16, 1, 2>(p, blocks, stream) : launch_dgrad_lds<2, 8, 2, 2>(p, blocks, stream);
#else
    rc = launch_dgrad_lds<1, 16, 1, 2>(p, blocks, stream);
#endif
  } else {
    p.n_cib = Cin / 64;
    p.items = (int)items64;
    const int blocks = p.items < 256 ? p.items : 256;
#ifdef GEECO_DEV_KERNELS
    rc = variant == 1 ? launch_dgrad_lds<1, 16, 1, 4>(p, blocks, stream) : launch_dgrad_lds<2, 8, 2, 4>(p, blocks, stream);
#else
    rc = launch_dgrad_lds<1, 16, 1, 4>(p, blocks, stream);
#endif
  }
  if (rc) return rc;
  GEECO_LAUNCH_CHECK();
  *handled = 1;
  return 0;
}
