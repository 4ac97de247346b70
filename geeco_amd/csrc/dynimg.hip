// Dynamic image (temporal rank pooling) + per-sample min/max normalisation, HBM-bound streaming.
//
// Replaces reference src/models/e2evmc/graph.py:30-55 (dynimg) with coefficients from :17-28:
//   D[n] = sum_t alpha_t X[n][t];  out[n] = (D[n] - min D[n]) / (max D[n] - min D[n] + 1e-6)
// Pass 1 streams the K frames once (16 B per lane), writes D (channel-padded) and per-block
// min/max partials; pass 2 folds the partials (one wave reduce) and normalises in place.
// Algorithmic bytes: N*K*HW*C*4 read + N*HW*Cpad*4 written (+ one extra read/write of D).
#include "geeco_common.h"
#include <stdlib.h>
#include <vector>

#define DYN_MAXK 64

struct DynParams {
  const float* frames;
  const float* frames2;
  const float* depth;      // optional 4th channel kept in its own tensor ([N][K][HW] / [N][HW]): RGB-D without packing
  const float* depth2;
  long long dsample_stride, dframe_stride;
  long long sample_stride, frame_stride;
  int N, K;
  long long HW;
  int C, Cpad;
  float* out;
  float* last;   // optional [N][HW][4]: the LAST frame of the stack channel-padded (conv1's input of the current frame)
  float* part;   // [N][nblk][2]
  int nblk;
  float alpha[DYN_MAXK];
  // DIFF (geeco_goal_dynimgs_fwd): the pair image alpha2[0] * last frame + alpha2[1] * target (graph.py:397-400) from the same pass:
  // the last frame is in registers anyway, so the pair image costs one read of the target instead of a launch that reads both
  const float* tgt;       // [N][HW][3]
  const float* tgt_depth; // [N][HW] (DEPTH)
  float* diff_out;        // [N][HW][4]
  float* part2;           // [N][nblk][2]
  float alpha2[2];
  // U8 (geeco_goal_dynimgs_u8_fwd): the RGB frames are the recorder's uint8 values, still in the episode's resident frames:
  // win[n] = address of the first frame of window n ([K][HW][3] bytes, consecutive frames), tgt_u8[n] = its target frame
  const unsigned char* const* win;
  const unsigned char* const* tgt_u8;
};

__device__ __forceinline__ const float* dyn_frame_ptr(const DynParams& p, int n, int t) {
  if (p.frames2 && t == 1) return p.frames2 + (long long)n * p.HW * p.C;
  return p.frames + (long long)n * p.sample_stride + (long long)t * p.frame_stride;
}

__device__ __forceinline__ void block_minmax_store(float mn, float mx, float* dst) {
  __shared__ float smn[4], smx[4];
  mn = wave_reduce_min(mn);
  mx = wave_reduce_max(mx);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smn[wid] = mn;
    smx[wid] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dst[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    dst[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

// float(u8) / 255.0f (_parse_v4, geeco_gym.py:312) without the division sequence: one Newton correction of a * (1/255) is the
// correctly rounded quotient for every a in 0..255 (tests/test_kernels_gpu.py::test_u8_unit_conversion_is_the_division checks all
// 256 values bitwise against the division of geeco_gather_windows).
__device__ __forceinline__ float u8_unit(float a) {
  const float r = 1.0f / 255.0f;
  const float q = a * r;
  const float e = __builtin_fmaf(-255.0f, q, a);
  return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ f32x4 u8x4_unit(unsigned int b) {
  return f32x4{u8_unit((float)(b & 255u)), u8_unit((float)((b >> 8) & 255u)), u8_unit((float)((b >> 16) & 255u)),
               u8_unit((float)(b >> 24))};
}

// 4 pixels = 12 bytes = three dwords of a uint8 RGB frame -> the three float4 the fp32 path loads
__device__ __forceinline__ void load_u8_unit(const unsigned char* frame, long long u, f32x4& v0, f32x4& v1, f32x4& v2) {
  // (the frame address comes out of a table in memory: say that it is global memory, or the compiler emits flat loads)
  typedef const __attribute__((address_space(1))) unsigned int* gptr;
  gptr s = (gptr)(reinterpret_cast<const unsigned int*>(frame) + u * 3);
  // non-temporal: the window is read once; what should stay in L2 / the memory-side cache are the three images this kernel writes
  // for conv1 (measured, same box: uint8 input stage 73 -> 58 us, fp32 131-136 -> 111-116 us in the step)
  const unsigned int b0 = __builtin_nontemporal_load(s), b1 = __builtin_nontemporal_load(s + 1), b2 = __builtin_nontemporal_load(s + 2);
  v0 = u8x4_unit(b0);
  v1 = u8x4_unit(b1);
  v2 = u8x4_unit(b2);
}

// C == 3, Cpad == 4, HW % 4 == 0: one thread = 4 pixels = 3 float4 in, 4 float4 out.  DEPTH: a 4th channel comes from its
// own tensor (one more float4 = the depth of the 4 pixels per frame): rgb || depth (estimator.py:169,172) is formed in
// registers instead of packing all N * K frames to 4 channels first (1.07 GB read + 1.43 GB written per step at K = 32).
// (The single-image form: per-timestep pair images of the sequence models, standalone dynimg calls.  The goal model's input
// stage is dynimg_goal_onepass_kernel below.)
template <bool DEPTH>
__global__ __launch_bounds__(256) void dynimg_wsum3_kernel(const DynParams p) {
  const int n = blockIdx.y;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;   // 4-pixel unit
  const long long U = p.HW >> 2;
  float mn = INFINITY, mx = -INFINITY;
  if (u < U) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll 4
    for (int t = 0; t < p.K; ++t) {
      const float w = p.alpha[t];
      const f32x4* src = reinterpret_cast<const f32x4*>(dyn_frame_ptr(p, n, t)) + u * 3;
      const f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
      a0 += w * v0;
      a1 += w * v1;
      a2 += w * v2;
      if (DEPTH) {
        const float* dp = (p.depth2 && t == 1) ? p.depth2 + (long long)n * p.HW
                                               : p.depth + (long long)n * p.dsample_stride + (long long)t * p.dframe_stride;
        a3 += w * reinterpret_cast<const f32x4*>(dp)[u];
      }
    }
    float e[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
    float d4[4] = {a3.x, a3.y, a3.z, a3.w};
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      mn = fminf(mn, e[i]);
      mx = fmaxf(mx, e[i]);
    }
    if (DEPTH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mn = fminf(mn, d4[i]);
        mx = fmaxf(mx, d4[i]);
      }
    }
    f32x4* dst = reinterpret_cast<f32x4*>(p.out + ((long long)n * p.HW + u * 4) * 4);
    dst[0] = f32x4{e[0], e[1], e[2], DEPTH ? d4[0] : 0.f};
    dst[1] = f32x4{e[3], e[4], e[5], DEPTH ? d4[1] : 0.f};
    dst[2] = f32x4{e[6], e[7], e[8], DEPTH ? d4[2] : 0.f};
    dst[3] = f32x4{e[9], e[10], e[11], DEPTH ? d4[3] : 0.f};
  }
  block_minmax_store(mn, mx, p.part + ((long long)n * p.nblk + blockIdx.x) * 2);
}

// ------------------------------------------------------------------------------------------------------------------
// The goal model's input stage in ONE pass (round 5; rounds 3-4: the K-frame pass + a normalisation launch that re-read and
// re-wrote both images: 134 MB of the stage's 662 MB).  A thread keeps its pixels of BOTH images in registers across the
// per-sample min / max, then normalises and stores them once:
//   * thread = UPT units of 4 pixels (units u, u + THREADS, ...: consecutive lanes read consecutive 48 B), block = THREADS
//     threads; a sample is covered by bps = ceil(HW / 4 / (THREADS * UPT)) blocks with consecutive block indices;
//   * block min / max of the two images -> ONE lane publishes them as write-through (sc1) stores into the block's slot of the
//     sample's partial array, waits for those stores, and counts the block in with an agent-scope atomic add (memory side); the
//     same wave polls the sample's counter (sc1 loads) until its bps blocks are in, then fetches the bps slots with sc1 loads
//     and reduces them -- every hand-off byte is written through and read past this XCD's L2 (per-XCD L2s are not coherent), so
//     no L2 write-back / invalidate is needed; critical path = one store, one atomic, one load latency;
//   * a block waits only for the OTHER BLOCKS OF ITS SAMPLE (not a grid barrier): they run the same K-frame pass and arrive
//     together; blocks of a sample have consecutive indices and workgroups start in index order, so every sample ahead of a
//     partially started one is complete or fully resident: the wait cannot deadlock however many blocks fit on the chip.
//     In-order start of workgroups is how the dispatcher of every CDNA part behaves, NOT a documented guarantee (CU masking, a
//     partitioned device or a co-resident persistent kernel could starve a sample's last blocks).  So the wait is BOUNDED
//     (g_wait_polls polls, seconds) and an expired wait is LOUD: the block counts itself into the sample's sticky `timeouts`
//     word and normalises with NaN, and geeco_goal_dynimgs_timeouts() (which the host calls wherever it synchronises anyway)
//     reports the count.  The word is what makes it loud: the NaN images are visible as such (endpoints), but conv1's ReLU
//     -- max(x, 0) returns 0 for a NaN -- would let a finite loss come out of them.  Nothing ever continues on stale min / max.  A workspace that has seen a timeout stays poisoned (its counters are no longer zero
//     between calls) until the caller zero-fills it again;
//   * ordering of the hand-off, at the hardware level (the C++ model has no word for "write-through store"): the slot stores
//     and the counter add are agent-scope atomics = sc1 accesses that complete at the memory side, past the non-coherent
//     per-XCD L2s; the producer drains its slot stores (s_waitcnt vmcnt(0)) BEFORE it issues the add; on the consumer side
//     every lane of the polling wave takes the final count from lane 0 through readfirstlane and the ADDRESS of its slot loads
//     is computed from that count (+ count >> 31, i.e. + 0), so the slot loads are issued behind the poll that saw the full
//     count by data dependence -- for all 64 lanes, for the compiler and for the wave -- not by reconvergence and without a
//     fence (a workgroup-scope acquire fence here measured +2...3 us in the step: it drains the wave's prefetched frame loads
//     of the NEXT sample, which the hand-off does not depend on).  Agent-scope release / acquire (buffer_wbl2 / buffer_inv
//     sc1 per block) would write back and invalidate an XCD's whole L2 for four floats that never live in it;
//   * the LAST block of a sample to leave zeroes the sample's two counters again: every call finds and leaves them zero (the
//     slots need no reset: every block rewrites its own before it counts itself in).
// The arithmetic per pixel is that of dynimg_wsum3_kernel + dynimg_norm_kernel (same sums in the same order, (D - min) / range
// with the IEEE division): bitwise the same images.
// ------------------------------------------------------------------------------------------------------------------
struct DynCtl {           // per sample: the two counters, on a 64-byte line of their own; zero between calls.  Behind the N
  unsigned arrive, depart;      // control blocks: N x bps slots of {min, max of the buffer image, min, max of the pair image}
  unsigned timeouts;            // sticky: blocks of this sample whose wait expired (never reset by the kernels)
  unsigned pad[13];
};

// polls of the sample counter before a block gives up (s_sleep 4 + one sc1 load each: 2^22 polls are seconds; the blocks of a
// sample arrive within microseconds of each other).  geeco_goal_dynimgs_set_wait_polls: tests set 0, so that every block that is
// not the last of its sample to arrive reports a timeout.
static unsigned g_wait_polls = 1u << 22;

// the wait itself: lane 0 polls, every lane of the wave gets the final count
__device__ __forceinline__ unsigned dyn_wait_for_sample(DynCtl* c, unsigned got, unsigned bps, unsigned polls, bool lane0) {
  if (lane0) {
    for (unsigned spin = 0; got < bps && spin < polls; ++spin) {
      __builtin_amdgcn_s_sleep(4);
      got = __hip_atomic_load(&c->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (got < bps) __hip_atomic_fetch_add(&c->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __builtin_amdgcn_readfirstlane(got);
}

// a wave-uniform address as such (two SGPRs): loads from it + a 32-bit per-lane offset take the scalar-base form and need one VGPR
// of address instead of a 64-bit pair per load (which the compiler precomputes per frame of the unrolled ring and spills)
__device__ __forceinline__ const char* dyn_uniform(const void* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

template <bool DEPTH, bool U8, int THREADS, int UPT>
__global__ __launch_bounds__(THREADS) void dynimg_goal_onepass_kernel(const DynParams p, DynCtl* ctl, int bps, unsigned polls) {
  constexpr int NW = THREADS / 64;
  const int n = blockIdx.x / bps, b = blockIdx.x - n * bps;
  const int tid = threadIdx.x;
  [[maybe_unused]] const unsigned char* wbase = U8 ? p.win[n] : nullptr;
  const long long U = p.HW >> 2;             // units of 4 pixels per frame
  const long long ub = (long long)b * (THREADS * UPT);      // first unit of this block
  // Two register layouts of a thread's 12 RGB floats per unit slot j (the depths, one float4 per unit, always belong to unit
  // ub + j * THREADS + tid):
  //   unit layout (U8 source): x[j][c] = c-th float4 of unit ub + j * THREADS + tid (one dwordx3 of bytes per lane and frame);
  //   flat layout (fp32 source): x[j][c] = float4 number (j * 3 + c) * THREADS + tid of the block's stretch of the frame, so
  //     every load instruction of a wave reads 1 KiB contiguous (the unit layout reads 16 of every 48 bytes per instruction:
  //     three times the cache-line requests; measured on the first form of this kernel).  Sums, products and min / max do not
  //     care which pixel a float belongs to; the three arrays that are stored per pixel (current frame, both images) go through
  //     an LDS transposition (to_units) once, after the frame loop.
  f32x4 A[UPT][4], D[UPT][4];      // buffer image / pair image: [slot][three RGB float4, the 4 depths]
#pragma unroll
  for (int j = 0; j < UPT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) A[j][q] = D[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float w0 = p.alpha2[0], w1 = p.alpha2[1];
  // Threads past the end of a ragged last block read the frame's last unit / float4 again and drop what they computed: the frame
  // loop has no branch.
  // (per-lane positions as 32-bit BYTE offsets from wave-uniform frame addresses: one VGPR each, and the loads take the
  // scalar-base form; a frame is at most HW * 16 bytes, checked by the launcher to stay below 2^31)
  unsigned uc[UPT], fo[UPT][3];      // unit index (clamped); byte offset of the slot's c-th float4 in the frame (flat layout)
  bool live[UPT], flive[UPT][3];
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    const long long u = ub + (long long)j * THREADS + tid;
    live[j] = u < U;
    uc[j] = (unsigned)(live[j] ? u : U - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long f = ub * 3 + (long long)(j * 3 + c) * THREADS + tid;
      flive[j][c] = U8 ? live[j] : f < 3 * U;
      fo[j][c] = (unsigned)((f < 3 * U ? f : 3 * U - 1) * 16);
    }
  }
  auto ld4 = [](const f32x4* q0) {      // (global address space said explicitly: a pointer rebuilt from integers would load "flat")
    typedef const __attribute__((address_space(1))) f32x4* gptr;
    gptr q = (gptr)q0;
    return __builtin_nontemporal_load(q);
  };
  auto load = [&](int t, int j, f32x4& v0, f32x4& v1, f32x4& v2, f32x4& v3) {
    if (U8) {
      load_u8_unit(wbase + (long long)t * p.HW * 3, uc[j], v0, v1, v2);
    } else {
      const char* src = dyn_uniform(p.frames + (long long)n * p.sample_stride + (long long)t * p.frame_stride);
      v0 = ld4(reinterpret_cast<const f32x4*>(src + fo[j][0])); v1 = ld4(reinterpret_cast<const f32x4*>(src + fo[j][1]));
      v2 = ld4(reinterpret_cast<const f32x4*>(src + fo[j][2]));
    }
    if (DEPTH) v3 = ld4(reinterpret_cast<const f32x4*>(dyn_uniform(p.depth + (long long)n * p.dsample_stride + (long long)t * p.dframe_stride) + uc[j] * 16u));
  };
  // Frames 0 .. K-2, software-pipelined: a ring of UNR frames of staging registers; a frame's registers are refilled with the
  // frame UNR ahead as soon as its products are taken, so a wave always has ~UNR * UPT * 3 loads in flight.
  constexpr int UNR = DEPTH ? 2 : 3;      // (the pair image's registers are not live yet in this loop)
  const int KM = p.K - 1;
  const int groups = KM / UNR;
  if (groups > 0) {
    f32x4 v[UNR][UPT][4];
#pragma unroll
    for (int k = 0; k < UNR; ++k)
#pragma unroll
      for (int j = 0; j < UPT; ++j) load(k, j, v[k][j][0], v[k][j][1], v[k][j][2], v[k][j][3]);
    for (int g = 0; g + 1 < groups; ++g) {      // steady state: no condition inside
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const float w = p.alpha[g * UNR + k];
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
          A[j][0] += w * v[k][j][0]; A[j][1] += w * v[k][j][1]; A[j][2] += w * v[k][j][2];
          if (DEPTH) A[j][3] += w * v[k][j][3];
        }
#pragma unroll
        for (int j = 0; j < UPT; ++j) load((g + 1) * UNR + k, j, v[k][j][0], v[k][j][1], v[k][j][2], v[k][j][3]);
        __builtin_amdgcn_sched_barrier(0);      // keep this order: (consume frame k, refill its registers), next k -- the scheduler
      }                                         // otherwise sinks all refills behind the last wait of the round
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const float w = p.alpha[(groups - 1) * UNR + k];
#pragma unroll
      for (int j = 0; j < UPT; ++j) {
        A[j][0] += w * v[k][j][0]; A[j][1] += w * v[k][j][1]; A[j][2] += w * v[k][j][2];
        if (DEPTH) A[j][3] += w * v[k][j][3];
      }
    }
  }
  for (int t = groups * UNR; t < KM; ++t) {     // K - 1 not a multiple of UNR: the remaining frames one by one
    const float w = p.alpha[t];
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      f32x4 v0, v1, v2, v3;
      load(t, j, v0, v1, v2, v3);
      A[j][0] += w * v0; A[j][1] += w * v1; A[j][2] += w * v2;
      if (DEPTH) A[j][3] += w * v3;
    }
  }
  // flat layout -> unit layout of one slot's three float4 through LDS (fp32 source only; every thread of the block takes part)
  __shared__ f32x4 tbuf[U8 ? 1 : THREADS * 3];
  auto to_units = [&](f32x4& x0, f32x4& x1, f32x4& x2) {
    if (U8) return;
    __syncthreads();
    tbuf[tid] = x0; tbuf[THREADS + tid] = x1; tbuf[2 * THREADS + tid] = x2;
    __syncthreads();
    x0 = tbuf[3 * tid]; x1 = tbuf[3 * tid + 1]; x2 = tbuf[3 * tid + 2];
  };
  {   // the window's last frame = the current frame: also the ConvEncoder's input and the first term of the pair image
    const float w = p.alpha[p.K - 1];
    f32x4 c[UPT][4], g[UPT][4];
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      c[j][3] = g[j][3] = f32x4{0.f, 0.f, 0.f, 0.f};
      load(p.K - 1, j, c[j][0], c[j][1], c[j][2], c[j][3]);
      if (U8) {
        load_u8_unit(p.tgt_u8[n], uc[j], g[j][0], g[j][1], g[j][2]);
      } else {
        const char* ts = dyn_uniform(p.tgt + (long long)n * p.HW * 3);
        g[j][0] = ld4(reinterpret_cast<const f32x4*>(ts + fo[j][0])); g[j][1] = ld4(reinterpret_cast<const f32x4*>(ts + fo[j][1]));
        g[j][2] = ld4(reinterpret_cast<const f32x4*>(ts + fo[j][2]));
      }
      if (DEPTH) g[j][3] = ld4(reinterpret_cast<const f32x4*>(dyn_uniform(p.tgt_depth + (long long)n * p.HW) + uc[j] * 16u));
    }
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      A[j][0] += w * c[j][0]; A[j][1] += w * c[j][1]; A[j][2] += w * c[j][2];
      if (DEPTH) A[j][3] += w * c[j][3];
      // the pair image, summed in the order of the two-frame pass: 0 + alpha2[0] * current, + alpha2[1] * target
      D[j][0] += w0 * c[j][0]; D[j][1] += w0 * c[j][1]; D[j][2] += w0 * c[j][2];
      D[j][0] += w1 * g[j][0]; D[j][1] += w1 * g[j][1]; D[j][2] += w1 * g[j][2];
      if (DEPTH) {
        D[j][3] += w0 * c[j][3];
        D[j][3] += w1 * g[j][3];
      }
    }
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
      to_units(c[j][0], c[j][1], c[j][2]);
      if (live[j]) {
        const f32x4 v0 = c[j][0], v1 = c[j][1], v2 = c[j][2], v3 = c[j][3];
        f32x4* lo = reinterpret_cast<f32x4*>(p.last + ((long long)n * p.HW + (long long)uc[j] * 4) * 4);
        lo[0] = f32x4{v0.x, v0.y, v0.z, v3.x};
        lo[1] = f32x4{v0.w, v1.x, v1.y, v3.y};
        lo[2] = f32x4{v1.z, v1.w, v2.x, v3.z};
        lo[3] = f32x4{v2.y, v2.z, v2.w, v3.w};
      }
    }
  }
  // ---- per-sample min / max of both images --------------------------------------------------------------------------
  float mn1 = INFINITY, mx1 = -INFINITY, mn2 = INFINITY, mx2 = -INFINITY;
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
#pragma unroll
    for (int q = 0; q < (DEPTH ? 4 : 3); ++q) {
      if (!(q < 3 ? flive[j][q] : live[j])) continue;
      const f32x4 a = A[j][q], d = D[j][q];
      mn1 = fminf(fminf(mn1, fminf(a.x, a.y)), fminf(a.z, a.w));
      mx1 = fmaxf(fmaxf(mx1, fmaxf(a.x, a.y)), fmaxf(a.z, a.w));
      mn2 = fminf(fminf(mn2, fminf(d.x, d.y)), fminf(d.z, d.w));
      mx2 = fmaxf(fmaxf(mx2, fmaxf(d.x, d.y)), fmaxf(d.z, d.w));
    }
  }
  __shared__ float red[NW][4];
  __shared__ float s_norm[4];      // min1, range1, min2, range2
  mn1 = wave_reduce_min(mn1); mx1 = wave_reduce_max(mx1);
  mn2 = wave_reduce_min(mn2); mx2 = wave_reduce_max(mx2);
  const int wid = tid >> 6;
  if ((tid & 63) == 0) {
    red[wid][0] = mn1; red[wid][1] = mx1; red[wid][2] = mn2; red[wid][3] = mx2;
  }
  __syncthreads();
  if (wid == 0) {
    // wave 0: lane 0 folds the waves' partials and publishes the block's four numbers as write-through stores into its slot of
    // the sample's partial array, waits for those stores, then counts the block in (agent-scope atomic add, memory side).  Then
    // it polls the sample's counter; once all bps blocks are in, lanes 0..bps-1 fetch the slots (sc1 loads: served past this
    // XCD's L2) and a wave reduction gives the sample's min / max -- one store, one atomic and one load latency on the critical
    // path (the first form used four returning atomic max + four read-backs: ~15 us of latency).
    DynCtl* c = ctl + n;
    f32x4* slots = reinterpret_cast<f32x4*>(ctl + p.N) + (long long)n * bps;
    const int lane = tid;
    unsigned got = 0;
    if (lane == 0) {
      for (int i = 1; i < NW; ++i) {
        mn1 = fminf(mn1, red[i][0]); mx1 = fmaxf(mx1, red[i][1]);
        mn2 = fminf(mn2, red[i][2]); mx2 = fmaxf(mx2, red[i][3]);
      }
      float* sp = reinterpret_cast<float*>(slots + b);
      __hip_atomic_store(sp + 0, mn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (agent-scope relaxed = sc1 write-through stores)
      __hip_atomic_store(sp + 1, mx1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp + 2, mn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp + 3, mx2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      got = __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    }
    // the other blocks of this sample run the same pass over the same number of frames: they are at most a few us behind
    got = dyn_wait_for_sample(c, got, (unsigned)bps, polls, lane == 0);
    const bool expired = got < (unsigned)bps;
    slots += got >> 31;      // + 0 (a count never has bit 31 set): the slot loads below carry an ADDRESS dependency on the final count
    f32x4 q = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = lane; i < bps; i += 64) {
      const float* sp = reinterpret_cast<const float*>(slots + i);
      const float q0 = __hip_atomic_load(sp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float q1 = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float q2 = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float q3 = __hip_atomic_load(sp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q.x = fminf(q.x, q0); q.y = fmaxf(q.y, q1); q.z = fminf(q.z, q2); q.w = fmaxf(q.w, q3);
    }
    const float a1 = wave_reduce_min(q.x), b1 = wave_reduce_max(q.y), a2 = wave_reduce_min(q.z), b2 = wave_reduce_max(q.w);
    if (lane == 0) {
      s_norm[0] = a1; s_norm[1] = expired ? NAN : b1 - a1 + 1e-6f;      // graph.py:49 (expired wait: NaN images, never stale ones)
      s_norm[2] = a2; s_norm[3] = expired ? NAN : b2 - a2 + 1e-6f;
      // leave: the last block out zeroes the two counters for the next call (every block has read the slots by then)
      if (__hip_atomic_fetch_add(&c->depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)bps - 1u) {
        __hip_atomic_exchange(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_exchange(&c->depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  const float m1 = s_norm[0], r1 = s_norm[1], m2 = s_norm[2], r2 = s_norm[3];
  // ---- normalise in registers, store once ---------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < UPT; ++j) {
    to_units(A[j][0], A[j][1], A[j][2]);
    to_units(D[j][0], D[j][1], D[j][2]);
    if (!live[j]) continue;
    const long long u = (long long)uc[j];
    float e[12], f[12], e4[4] = {0.f, 0.f, 0.f, 0.f}, f4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      e[4 * q + 0] = (A[j][q].x - m1) / r1; e[4 * q + 1] = (A[j][q].y - m1) / r1;
      e[4 * q + 2] = (A[j][q].z - m1) / r1; e[4 * q + 3] = (A[j][q].w - m1) / r1;
      f[4 * q + 0] = (D[j][q].x - m2) / r2; f[4 * q + 1] = (D[j][q].y - m2) / r2;
      f[4 * q + 2] = (D[j][q].z - m2) / r2; f[4 * q + 3] = (D[j][q].w - m2) / r2;
    }
    if (DEPTH) {
      e4[0] = (A[j][3].x - m1) / r1; e4[1] = (A[j][3].y - m1) / r1; e4[2] = (A[j][3].z - m1) / r1; e4[3] = (A[j][3].w - m1) / r1;
      f4[0] = (D[j][3].x - m2) / r2; f4[1] = (D[j][3].y - m2) / r2; f4[2] = (D[j][3].z - m2) / r2; f4[3] = (D[j][3].w - m2) / r2;
    }
    f32x4* dst = reinterpret_cast<f32x4*>(p.out + ((long long)n * p.HW + u * 4) * 4);
    dst[0] = f32x4{e[0], e[1], e[2], e4[0]};
    dst[1] = f32x4{e[3], e[4], e[5], e4[1]};
    dst[2] = f32x4{e[6], e[7], e[8], e4[2]};
    dst[3] = f32x4{e[9], e[10], e[11], e4[3]};
    f32x4* dd = reinterpret_cast<f32x4*>(p.diff_out + ((long long)n * p.HW + u * 4) * 4);
    dd[0] = f32x4{f[0], f[1], f[2], f4[0]};
    dd[1] = f32x4{f[3], f[4], f[5], f4[1]};
    dd[2] = f32x4{f[6], f[7], f[8], f4[2]};
    dd[3] = f32x4{f[9], f[10], f[11], f4[3]};
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same stage with TWO samples per block, one after the other (round 5, second form): with one sample per block every
// block of the chip reads for ~85 us and then stores for ~20 us, all in step -- the 100 MB of stores never overlap the 430 MB
// of loads.  Here a group of bps consecutive blocks owns samples pi and pi + half; a block runs the frame loop of its chunk of
// sample pi, publishes its min / max, and normalises + stores that sample's images from INSIDE the frame loop of sample
// pi + half (half-way through: every block of the group has long finished the first sample), so the first half of the stores
// rides beside the second half of the loads.  One unit (4 pixels) per thread and sample; everything else as above.
// ------------------------------------------------------------------------------------------------------------------
template <bool DEPTH, bool U8, int THREADS>
__global__ __launch_bounds__(THREADS) void dynimg_goal_onepass2_kernel(const DynParams p, DynCtl* ctl, int bps, int half, unsigned polls) {
  constexpr int NW = THREADS / 64;
  const int pi = blockIdx.x / bps, b = blockIdx.x - pi * bps;
  const int tid = threadIdx.x, wid = tid >> 6;
  const long long U = p.HW >> 2;
  const long long ub = (long long)b * THREADS;
  unsigned uc, fo[3];
  bool live, flive[3];
  {
    const long long u = ub + tid;
    live = u < U;
    uc = (unsigned)(live ? u : U - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long long f = ub * 3 + (long long)c * THREADS + tid;
      flive[c] = U8 ? live : f < 3 * U;
      fo[c] = (unsigned)((f < 3 * U ? f : 3 * U - 1) * 16);
    }
  }
  const float w0 = p.alpha2[0], w1 = p.alpha2[1];
  __shared__ f32x4 tbuf[U8 ? 1 : THREADS * 3];
  __shared__ float red[NW][4];
  __shared__ float s_norm[4];
  auto ld4 = [](const f32x4* q0) {
    typedef const __attribute__((address_space(1))) f32x4* gptr;
    return __builtin_nontemporal_load((gptr)q0);
  };
  auto to_units = [&](f32x4& x0, f32x4& x1, f32x4& x2) {
    if (U8) return;
    __syncthreads();
    tbuf[tid] = x0; tbuf[THREADS + tid] = x1; tbuf[2 * THREADS + tid] = x2;
    __syncthreads();
    x0 = tbuf[3 * tid]; x1 = tbuf[3 * tid + 1]; x2 = tbuf[3 * tid + 2];
  };
  // ---- frame loop + last frame of sample n: A = buffer image, D = pair image (flat layout for fp32 sources); `mid()` is called
  // once, about half-way through the frames
  auto accumulate = [&](int n, f32x4 (&A)[4], f32x4 (&D)[4], auto&& mid) {
    [[maybe_unused]] const unsigned char* wbase = U8 ? p.win[n] : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) A[q] = D[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int t, f32x4& v0, f32x4& v1, f32x4& v2, f32x4& v3) {
      if (U8) {
        load_u8_unit(wbase + (long long)t * p.HW * 3, uc, v0, v1, v2);
      } else {
        const char* src = dyn_uniform(p.frames + (long long)n * p.sample_stride + (long long)t * p.frame_stride);
        v0 = ld4(reinterpret_cast<const f32x4*>(src + fo[0])); v1 = ld4(reinterpret_cast<const f32x4*>(src + fo[1]));
        v2 = ld4(reinterpret_cast<const f32x4*>(src + fo[2]));
      }
      if (DEPTH) v3 = ld4(reinterpret_cast<const f32x4*>(dyn_uniform(p.depth + (long long)n * p.dsample_stride + (long long)t * p.dframe_stride) + uc * 16u));
    };
    constexpr int UNR = DEPTH ? 3 : 4;      // frames in flight: 12 loads of 16 B per lane (the first sample's images stay live meanwhile)
    const int KM = p.K - 1;
    const int groups = KM / UNR;
    const int gmid = groups >> 1;
    bool called = false;
    if (groups > 0) {
      f32x4 v[UNR][4];
#pragma unroll
      for (int k = 0; k < UNR; ++k) load(k, v[k][0], v[k][1], v[k][2], v[k][3]);
      for (int g = 0; g + 1 < groups; ++g) {
        if (g == gmid) {
          mid();
          called = true;
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
          const float w = p.alpha[g * UNR + k];
          A[0] += w * v[k][0]; A[1] += w * v[k][1]; A[2] += w * v[k][2];
          if (DEPTH) A[3] += w * v[k][3];
          load((g + 1) * UNR + k, v[k][0], v[k][1], v[k][2], v[k][3]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const float w = p.alpha[(groups - 1) * UNR + k];
        A[0] += w * v[k][0]; A[1] += w * v[k][1]; A[2] += w * v[k][2];
        if (DEPTH) A[3] += w * v[k][3];
      }
    }
    if (!called) mid();
    for (int t = groups * UNR; t < KM; ++t) {
      const float w = p.alpha[t];
      f32x4 v0, v1, v2, v3;
      load(t, v0, v1, v2, v3);
      A[0] += w * v0; A[1] += w * v1; A[2] += w * v2;
      if (DEPTH) A[3] += w * v3;
    }
    const float w = p.alpha[p.K - 1];
    f32x4 c[4], g[4];
    c[3] = g[3] = f32x4{0.f, 0.f, 0.f, 0.f};
    load(p.K - 1, c[0], c[1], c[2], c[3]);
    if (U8) {
      load_u8_unit(p.tgt_u8[n], uc, g[0], g[1], g[2]);
    } else {
      const char* ts = dyn_uniform(p.tgt + (long long)n * p.HW * 3);
      g[0] = ld4(reinterpret_cast<const f32x4*>(ts + fo[0])); g[1] = ld4(reinterpret_cast<const f32x4*>(ts + fo[1]));
      g[2] = ld4(reinterpret_cast<const f32x4*>(ts + fo[2]));
    }
    if (DEPTH) g[3] = ld4(reinterpret_cast<const f32x4*>(dyn_uniform(p.tgt_depth + (long long)n * p.HW) + uc * 16u));
    A[0] += w * c[0]; A[1] += w * c[1]; A[2] += w * c[2];
    if (DEPTH) A[3] += w * c[3];
    D[0] += w0 * c[0]; D[1] += w0 * c[1]; D[2] += w0 * c[2];
    D[0] += w1 * g[0]; D[1] += w1 * g[1]; D[2] += w1 * g[2];
    if (DEPTH) {
      D[3] += w0 * c[3];
      D[3] += w1 * g[3];
    }
    to_units(c[0], c[1], c[2]);
    if (live) {
      f32x4* lo = reinterpret_cast<f32x4*>(p.last + ((long long)n * p.HW + (long long)uc * 4) * 4);
      lo[0] = f32x4{c[0].x, c[0].y, c[0].z, c[3].x};
      lo[1] = f32x4{c[0].w, c[1].x, c[1].y, c[3].y};
      lo[2] = f32x4{c[1].z, c[1].w, c[2].x, c[3].z};
      lo[3] = f32x4{c[2].y, c[2].z, c[2].w, c[3].w};
    }
  };
  // ---- block min / max of sample n -> its slot; count the block in
  auto publish = [&](int n, const f32x4 (&A)[4], const f32x4 (&D)[4]) {
    float mn1 = INFINITY, mx1 = -INFINITY, mn2 = INFINITY, mx2 = -INFINITY;
#pragma unroll
    for (int q = 0; q < (DEPTH ? 4 : 3); ++q) {
      if (!(q < 3 ? flive[q] : live)) continue;
      const f32x4 a = A[q], d = D[q];
      mn1 = fminf(fminf(mn1, fminf(a.x, a.y)), fminf(a.z, a.w));
      mx1 = fmaxf(fmaxf(mx1, fmaxf(a.x, a.y)), fmaxf(a.z, a.w));
      mn2 = fminf(fminf(mn2, fminf(d.x, d.y)), fminf(d.z, d.w));
      mx2 = fmaxf(fmaxf(mx2, fmaxf(d.x, d.y)), fmaxf(d.z, d.w));
    }
    mn1 = wave_reduce_min(mn1); mx1 = wave_reduce_max(mx1);
    mn2 = wave_reduce_min(mn2); mx2 = wave_reduce_max(mx2);
    __syncthreads();                     // (red may still be read by the previous sample's publish)
    if ((tid & 63) == 0) {
      red[wid][0] = mn1; red[wid][1] = mx1; red[wid][2] = mn2; red[wid][3] = mx2;
    }
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < NW; ++i) {
        mn1 = fminf(mn1, red[i][0]); mx1 = fmaxf(mx1, red[i][1]);
        mn2 = fminf(mn2, red[i][2]); mx2 = fmaxf(mx2, red[i][3]);
      }
      float* sp = reinterpret_cast<float*>(reinterpret_cast<f32x4*>(ctl + p.N) + (long long)n * bps + b);
      __hip_atomic_store(sp + 0, mn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp + 1, mx1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp + 2, mn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sp + 3, mx2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&ctl[n].arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // ---- wait for sample n's blocks, fold their slots, normalise this block's part in registers and store it
  auto finish = [&](int n, f32x4 (&A)[4], f32x4 (&D)[4]) {
    __syncthreads();                     // (s_norm may still be read by the previous sample's finish)
    if (wid == 0) {
      DynCtl* c = ctl + n;
      const f32x4* slots = reinterpret_cast<const f32x4*>(ctl + p.N) + (long long)n * bps;
      unsigned got = 0;
      if (tid == 0) got = __hip_atomic_load(&c->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      got = dyn_wait_for_sample(c, got, (unsigned)bps, polls, tid == 0);
      const bool expired = got < (unsigned)bps;
      slots += got >> 31;      // + 0: address dependency of the slot loads on the final count (see dyn_wait_for_sample)
      f32x4 q = {INFINITY, -INFINITY, INFINITY, -INFINITY};
      for (int i = tid; i < bps; i += 64) {
        const float* sp = reinterpret_cast<const float*>(slots + i);
        const float q0 = __hip_atomic_load(sp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float q1 = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float q2 = __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float q3 = __hip_atomic_load(sp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q.x = fminf(q.x, q0); q.y = fmaxf(q.y, q1); q.z = fminf(q.z, q2); q.w = fmaxf(q.w, q3);
      }
      const float a1 = wave_reduce_min(q.x), b1 = wave_reduce_max(q.y), a2 = wave_reduce_min(q.z), b2 = wave_reduce_max(q.w);
      if (tid == 0) {
        s_norm[0] = a1; s_norm[1] = expired ? NAN : b1 - a1 + 1e-6f;      // graph.py:49 (expired wait: NaN images, never stale ones)
        s_norm[2] = a2; s_norm[3] = expired ? NAN : b2 - a2 + 1e-6f;
        if (__hip_atomic_fetch_add(&c->depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)bps - 1u) {
          __hip_atomic_exchange(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_exchange(&c->depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();
    // one image at a time (normalise in place, transpose, store): half the live registers of doing both together
    auto emit = [&](f32x4 (&X)[4], float m, float r, float* out) {
#pragma unroll
      for (int q = 0; q < (DEPTH ? 4 : 3); ++q) {
        X[q].x = (X[q].x - m) / r; X[q].y = (X[q].y - m) / r; X[q].z = (X[q].z - m) / r; X[q].w = (X[q].w - m) / r;
      }
      to_units(X[0], X[1], X[2]);
      if (!live) return;
      const f32x4 z = DEPTH ? X[3] : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4* dst = reinterpret_cast<f32x4*>(out + ((long long)n * p.HW + (long long)uc * 4) * 4);
      dst[0] = f32x4{X[0].x, X[0].y, X[0].z, z.x};
      dst[1] = f32x4{X[0].w, X[1].x, X[1].y, z.y};
      dst[2] = f32x4{X[1].z, X[1].w, X[2].x, z.z};
      dst[3] = f32x4{X[2].y, X[2].z, X[2].w, z.w};
    };
    emit(A, s_norm[0], s_norm[1], p.out);
    emit(D, s_norm[2], s_norm[3], p.diff_out);
  };
  const int nA = pi, nB = pi + half;
  f32x4 AA[4], DA[4];
  accumulate(nA, AA, DA, [] {});
  publish(nA, AA, DA);
  if (nB < p.N) {
    f32x4 AB[4], DB[4];
    accumulate(nB, AB, DB, [&] { finish(nA, AA, DA); });
    publish(nB, AB, DB);
    finish(nB, AB, DB);
  } else {
    finish(nA, AA, DA);
  }
}

// Generic: one thread = one pixel, C <= Cpad <= 8 channels (C == 4: one float4 per frame).
__global__ __launch_bounds__(256) void dynimg_wsum_generic_kernel(const DynParams p) {
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  if (px < p.HW) {
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    if (p.C == 4) {
#pragma unroll 4
      for (int t = 0; t < p.K; ++t) {
        f32x4 v = reinterpret_cast<const f32x4*>(dyn_frame_ptr(p, n, t))[px];
        const float w = p.alpha[t];
        acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w;
      }
    } else {
      for (int t = 0; t < p.K; ++t) {
        const float* src = dyn_frame_ptr(p, n, t) + px * p.C;
        const float w = p.alpha[t];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < p.C) acc[c] += w * src[c];
      }
    }
    float* dst = p.out + ((long long)n * p.HW + px) * p.Cpad;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < p.C) {
        mn = fminf(mn, acc[c]);
        mx = fmaxf(mx, acc[c]);
      }
      if (c < p.Cpad) dst[c] = c < p.C ? acc[c] : 0.f;
    }
  }
  block_minmax_store(mn, mx, p.part + ((long long)n * p.nblk + blockIdx.x) * 2);
}

// Pass 2: fold partials, normalise in place.  One thread = 4 consecutive floats of out.
__global__ __launch_bounds__(256) void dynimg_norm_kernel(float* out, const float* part, int nblk, long long HW,
                                                          int C, int Cpad) {
  const int n = blockIdx.y;
  __shared__ float s_mn, s_rng;
  {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) {
      mn = fminf(mn, part[((long long)n * nblk + i) * 2]);
      mx = fmaxf(mx, part[((long long)n * nblk + i) * 2 + 1]);
    }
    __shared__ float smn[4], smx[4];
    mn = wave_reduce_min(mn);
    mx = wave_reduce_max(mx);
    if ((threadIdx.x & 63) == 0) {
      smn[threadIdx.x >> 6] = mn;
      smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
      float b = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
      s_mn = a;
      s_rng = b - a + 1e-6f;   // graph.py:49
    }
    __syncthreads();
  }
  const float mn = s_mn, rng = s_rng;
  const long long total = HW * Cpad;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  float* o = out + (long long)n * total + i4;
  if ((total & 3) == 0) {
    f32x4 v = *reinterpret_cast<f32x4*>(o);
    float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = (int)((i4 + k) % Cpad);
      e[k] = c < C ? (e[k] - mn) / rng : 0.f;
    }
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int k = 0; k < 4 && i4 + k < total; ++k) {
      int c = (int)((i4 + k) % Cpad);
      o[k] = c < C ? (o[k] - mn) / rng : 0.f;
    }
  }
}

extern "C" void geeco_dynimg_alpha(int K, float* alpha) {
  // graph.py:17-28: float32 harmonic numbers summed in ascending order
  auto H = [](int t) {
    float h = 0.f;
    for (int i = 1; i <= t; ++i) h = h + 1.0f / (float)i;
    return h;
  };
  const float HT = H(K);
  for (int t = 1; t <= K; ++t) alpha[t - 1] = (float)(2 * (K - t + 1)) - (float)(K + 1) * (HT - H(t - 1));
}

static int dyn_nblk(long long HW, int C) {
  if (C == 3 && (HW & 3) == 0) return (int)cdiv64(HW >> 2, 256);
  return (int)cdiv64(HW, 256);
}

extern "C" int64_t geeco_dynimg_ws_bytes(int N, int64_t hwc) {
  // upper bound over both kernels: one partial pair per 256 pixels
  return (int64_t)N * (cdiv64(hwc, 256) + 1) * 2 * 4;
}

static int dynimg_fwd_impl(const float* frames, const float* frames2, int64_t sample_stride, int64_t frame_stride,
                           const float* alpha_host, int N, int K, int64_t HW, int C, int Cpad, float* out, float* last,
                           void* ws, void* stream);

extern "C" int geeco_dynimg_fwd(const float* frames, const float* frames2, int64_t sample_stride,
                                int64_t frame_stride, const float* alpha_host, int N, int K, int64_t HW, int C,
                                int Cpad, float* out, void* ws, void* stream) {
  return dynimg_fwd_impl(frames, frames2, sample_stride, frame_stride, alpha_host, N, K, HW, C, Cpad, out, nullptr, ws,
                         stream);
}

static int dynimg_fwd_impl(const float* frames, const float* frames2, int64_t sample_stride, int64_t frame_stride,
                           const float* alpha_host, int N, int K, int64_t HW, int C, int Cpad, float* out, float* last,
                           void* ws, void* stream) {
  GEECO_CHECK_ARG(frames && alpha_host && out && ws, "dynimg_fwd: null pointer");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "dynimg_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 1 && C >= 1 && C <= Cpad && Cpad <= 8, "dynimg_fwd: bad dims");
  GEECO_CHECK_ARG(!frames2 || K == 2, "dynimg_fwd: frames2 only with K == 2");
  DynParams p = {};
  p.frames = frames; p.frames2 = frames2; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.N = N; p.K = K; p.HW = HW; p.C = C; p.Cpad = Cpad; p.out = out; p.last = last; p.part = (float*)ws;
  p.nblk = dyn_nblk(HW, C);
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)p.nblk, (unsigned)N);
  const bool aligned = (sample_stride % 4 == 0) && (frame_stride % 4 == 0);
  if (C == 3 && Cpad == 4 && (HW & 3) == 0 && aligned)
    hipLaunchKernelGGL(dynimg_wsum3_kernel<false>, grid, dim3(256), 0, s, p);
  else {
    GEECO_CHECK_ARG(C != 4 || aligned, "dynimg_fwd: C == 4 needs 16-byte aligned frames");
    p.nblk = (int)cdiv64(HW, 256);
    grid.x = p.nblk;
    hipLaunchKernelGGL(dynimg_wsum_generic_kernel, grid, dim3(256), 0, s, p);
  }
  GEECO_LAUNCH_CHECK();
  dim3 g2((unsigned)cdiv64(HW * Cpad, 1024), (unsigned)N);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, out, (const float*)ws, p.nblk, (long long)HW, C, Cpad);
  GEECO_LAUNCH_CHECK();
  return 0;
}

static int dynimg_rgbd_impl(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                            const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                            const float* alpha_host, int N, int K, int64_t HW, float* out, float* last, void* ws,
                            void* stream);

extern "C" int geeco_dynimg_rgbd_fwd(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                                     const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                                     const float* alpha_host, int N, int K, int64_t HW, float* out, void* ws, void* stream) {
  return dynimg_rgbd_impl(rgb, rgb2, sample_stride, frame_stride, depth, depth2, dsample_stride, dframe_stride, alpha_host, N,
                          K, HW, out, nullptr, ws, stream);
}

static int dynimg_rgbd_impl(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                            const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                            const float* alpha_host, int N, int K, int64_t HW, float* out, float* last, void* ws,
                            void* stream) {
  GEECO_CHECK_ARG(rgb && depth && alpha_host && out && ws, "dynimg_rgbd_fwd: null pointer");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "dynimg_rgbd_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "dynimg_rgbd_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG((!rgb2) == (!depth2) && (!rgb2 || K == 2), "dynimg_rgbd_fwd: rgb2 / depth2 come together, with K == 2");
  GEECO_CHECK_ARG(sample_stride % 4 == 0 && frame_stride % 4 == 0 && dsample_stride % 4 == 0 && dframe_stride % 4 == 0,
                  "dynimg_rgbd_fwd: 16-byte aligned frames");
  DynParams p = {};
  p.frames = rgb; p.frames2 = rgb2; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.depth = depth; p.depth2 = depth2; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW; p.C = 3; p.Cpad = 4; p.out = out; p.last = last; p.part = (float*)ws;
  p.nblk = dyn_nblk(HW, 3);
  for (int t = 0; t < K; ++t) p.alpha[t] = alpha_host[t];
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dynimg_wsum3_kernel<true>, dim3((unsigned)p.nblk, (unsigned)N), dim3(256), 0, s, p);
  GEECO_LAUNCH_CHECK();
  dim3 g2((unsigned)cdiv64(HW * 4, 1024), (unsigned)N);
  hipLaunchKernelGGL(dynimg_norm_kernel, g2, dim3(256), 0, s, out, (const float*)ws, p.nblk, (long long)HW, 4, 4);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// The goal model's three conv1 inputs (graph.py:386-401) in ONE launch (round 5; round 4: two, round 3: three, before: five):
// one pass over the window computes the buffer image and the pair image of (current frame, target) in registers, writes the
// current frame's padded copy, and normalises both images before their only store (dynimg_goal_onepass_kernel).
// ws = geeco_goal_dynimgs_ws_bytes(N, HW) bytes, ZERO-FILLED once by the caller; every call leaves its counters zero.
extern "C" int64_t geeco_goal_dynimgs_ws_bytes(int N, int64_t HW) {
  if (N <= 0 || HW <= 0) return 0;
  return (int64_t)N * ((int64_t)sizeof(DynCtl) + cdiv64(HW >> 2, 256) * 16);      // counters + the most slots a sample can have
}

// Blocks whose wait for the other blocks of their sample has EVER expired on this workspace, summed over the N samples (sticky
// until the caller zero-fills ws again; such samples' images are NaN, see above).  0 = every image that came out of this
// workspace was normalised with its sample's true min / max.  Copies N x 64 bytes to the host and SYNCHRONISES the stream: for
// the places where the host waits for the device anyway (loss read-out, end of an epoch), not for the step.
extern "C" int geeco_goal_dynimgs_timeouts(const void* ws, int N, void* stream, int64_t* count_host) {
  GEECO_CHECK_ARG(ws && count_host && N >= 1, "goal_dynimgs_timeouts: null pointer / N=%d", N);
  std::vector<DynCtl> host((size_t)N);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(host.data(), ws, (size_t)N * sizeof(DynCtl), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    geeco_set_error("goal_dynimgs_timeouts: %s", hipGetErrorString(e));
    return (int)e;
  }
  int64_t n = 0;
  for (const DynCtl& c : host) n += c.timeouts;
  *count_host = n;
  return 0;
}

// Polls a block spends waiting for its sample's blocks before it reports a timeout (process-wide; default 2^22, i.e. seconds).
// Returns the previous value.  0 makes every block that is not the last of its sample to arrive report at once: how the tests
// provoke the error path deterministically.
extern "C" unsigned geeco_goal_dynimgs_set_wait_polls(unsigned polls) {
  const unsigned old = g_wait_polls;
  g_wait_polls = polls;
  return old;
}

template <bool DEPTH, bool U8>
static void goal_onepass_dispatch(const DynParams& p, DynCtl* ctl, hipStream_t s) {
  const long long U = p.HW >> 2;
  // 1024-thread blocks of 2 units per thread (8 pixels: 48 / 64 accumulator registers of both images) when that still gives
  // the chip about a block per CU; otherwise 256-thread blocks of one unit (small batches: the predictor's N = 1)
  const long long bps_big = cdiv64(U, 2048);
  if ((long long)p.N * bps_big >= 192) {
    if constexpr (!U8) {
      // fp32 windows: two samples per block, the first one's stores inside the second one's frame loop (same box, in the step:
      // 105.7-108.0 us against 110.9-114.7 for one sample per block)
      const int half = (p.N + 1) / 2, bps2 = (int)cdiv64(U, 1024);
      geeco_note_kernel("dynimg_goal_onepass2_kernel<%s, %s, 1024>", DEPTH ? "true" : "false", U8 ? "true" : "false");
      hipLaunchKernelGGL((dynimg_goal_onepass2_kernel<DEPTH, U8, 1024>), dim3((unsigned)(half * bps2)), dim3(1024), 0, s, p, ctl, bps2, half, g_wait_polls);
    } else {
      // uint8 frames: a quarter of the bytes and twelve conversions per pixel: the load phase is short and the one-sample form
      // with two units per thread keeps more of it in flight (58.6 us alone against 63.8)
      geeco_note_kernel("dynimg_goal_onepass_kernel<%s, %s, 1024, 2>", DEPTH ? "true" : "false", U8 ? "true" : "false");
      hipLaunchKernelGGL((dynimg_goal_onepass_kernel<DEPTH, U8, 1024, 2>), dim3((unsigned)(p.N * bps_big)), dim3(1024), 0, s, p, ctl, (int)bps_big, g_wait_polls);
    }
  } else {
    const long long bps = cdiv64(U, 256);
    geeco_note_kernel("dynimg_goal_onepass_kernel<%s, %s, 256, 1>", DEPTH ? "true" : "false", U8 ? "true" : "false");
    hipLaunchKernelGGL((dynimg_goal_onepass_kernel<DEPTH, U8, 256, 1>), dim3((unsigned)(p.N * bps)), dim3(256), 0, s, p, ctl, (int)bps, g_wait_polls);
  }
}

static int goal_dynimgs_launch(DynParams& p, const float* alpha_host, const float* alpha2_host, bool u8, float* cur_out,
                               float* buf_out, float* diff_out, void* ws, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GEECO_CHECK_ARG((long long)p.N * cdiv64(p.HW >> 2, 256) < (1ll << 31) && p.HW * 16 < (1ll << 31),
                  "goal_dynimgs: %d samples of %lld pixels exceed the grid / the 32-bit in-frame offsets", p.N, p.HW);
  p.C = 3; p.Cpad = 4; p.out = buf_out; p.last = cur_out;
  for (int t = 0; t < p.K; ++t) p.alpha[t] = alpha_host[t];
  p.diff_out = diff_out;
  p.alpha2[0] = alpha2_host[0]; p.alpha2[1] = alpha2_host[1];
  DynCtl* ctl = (DynCtl*)ws;
  const bool depth = p.depth != nullptr;
  if (u8) {
    if (depth) goal_onepass_dispatch<true, true>(p, ctl, s);
    else goal_onepass_dispatch<false, true>(p, ctl, s);
  } else {
    if (depth) goal_onepass_dispatch<true, false>(p, ctl, s);
    else goal_onepass_dispatch<false, false>(p, ctl, s);
  }
  GEECO_LAUNCH_CHECK();
  return 0;
}

extern "C" int geeco_goal_dynimgs_fwd(const float* rgb, int64_t sample_stride, int64_t frame_stride, const float* tgt_rgb,
                                      const float* depth, int64_t dsample_stride, int64_t dframe_stride,
                                      const float* tgt_depth, const float* alpha_host, const float* alpha2_host, int N, int K,
                                      int64_t HW, float* cur_out, float* buf_out, float* diff_out, void* ws, void* stream) {
  GEECO_CHECK_ARG(rgb && tgt_rgb && alpha_host && alpha2_host && cur_out && buf_out && diff_out && ws,
                  "goal_dynimgs_fwd: null pointer");
  GEECO_CHECK_ARG((!depth) == (!tgt_depth), "goal_dynimgs_fwd: depth and tgt_depth come together");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "goal_dynimgs_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "goal_dynimgs_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG(sample_stride % 4 == 0 && frame_stride % 4 == 0 && dsample_stride % 4 == 0 && dframe_stride % 4 == 0,
                  "goal_dynimgs_fwd: 16-byte aligned frames");
  DynParams p = {};
  p.frames = rgb; p.sample_stride = sample_stride; p.frame_stride = frame_stride;
  p.depth = depth; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW;
  p.tgt = tgt_rgb; p.tgt_depth = tgt_depth;
  return goal_dynimgs_launch(p, alpha_host, alpha2_host, false, cur_out, buf_out, diff_out, ws, stream);
}

// The same input stage fed from the episodes' resident uint8 frames (the data path's "next" row: the window of
// _window_v3, geeco_gym.py:615-631, and the / 255 of _parse_v4, :312, happen inside the load): win_ptrs_dev / tgt_ptrs_dev are
// DEVICE arrays of N addresses (window n = K consecutive [HW][3] uint8 frames starting at win_ptrs_dev[n]; its target frame
// at tgt_ptrs_dev[n]), so a captured graph keeps replaying while the host repoints the tables between steps.  Depth (float32)
// stays a dense [N][K][HW] / [N][HW] tensor.  Outputs are bitwise those of geeco_gather_windows + geeco_goal_dynimgs_fwd.
extern "C" int geeco_goal_dynimgs_u8_fwd(const void* const* win_ptrs_dev, const void* const* tgt_ptrs_dev, const float* depth,
                                         int64_t dsample_stride, int64_t dframe_stride, const float* tgt_depth,
                                         const float* alpha_host, const float* alpha2_host, int N, int K, int64_t HW,
                                         float* cur_out, float* buf_out, float* diff_out, void* ws, void* stream) {
  GEECO_CHECK_ARG(win_ptrs_dev && tgt_ptrs_dev && alpha_host && alpha2_host && cur_out && buf_out && diff_out && ws,
                  "goal_dynimgs_u8_fwd: null pointer");
  GEECO_CHECK_ARG((!depth) == (!tgt_depth), "goal_dynimgs_u8_fwd: depth and tgt_depth come together");
  GEECO_CHECK_ARG(K >= 1 && K <= DYN_MAXK, "goal_dynimgs_u8_fwd: K=%d outside 1..%d", K, DYN_MAXK);
  GEECO_CHECK_ARG(N >= 1 && HW >= 4 && (HW & 3) == 0, "goal_dynimgs_u8_fwd: HW=%lld must be a multiple of 4", (long long)HW);
  GEECO_CHECK_ARG(dsample_stride % 4 == 0 && dframe_stride % 4 == 0, "goal_dynimgs_u8_fwd: 16-byte aligned depth frames");
  DynParams p = {};
  p.win = reinterpret_cast<const unsigned char* const*>(win_ptrs_dev);
  p.tgt_u8 = reinterpret_cast<const unsigned char* const*>(tgt_ptrs_dev);
  p.depth = depth; p.dsample_stride = dsample_stride; p.dframe_stride = dframe_stride;
  p.N = N; p.K = K; p.HW = HW;
  p.tgt_depth = tgt_depth;
  return goal_dynimgs_launch(p, alpha_host, alpha2_host, true, cur_out, buf_out, diff_out, ws, stream);
}

// ---- pixel packing: [n][HW][C1] (+ [n][HW][C2]) -> [n][HW][Cpad] ---------------------------------
__global__ __launch_bounds__(256) void pack_pixels_kernel(const float* src, long long s1, const float* src2,
                                                          long long s2, long long HW, int C1, int C2, int Cpad,
                                                          float* dst) {
  const int n = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const float* a = src + (long long)n * s1 + px * C1;
  const float* b = src2 ? src2 + (long long)n * s2 + px * C2 : nullptr;
  float* o = dst + ((long long)n * HW + px) * Cpad;
  if (Cpad == 4) {
    float e[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) e[c] = c < C1 ? a[c] : (b && c - C1 < C2 ? b[c - C1] : 0.f);
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int c = 0; c < Cpad; ++c) o[c] = c < C1 ? a[c] : (b && c - C1 < C2 ? b[c - C1] : 0.f);
  }
}

extern "C" int geeco_pack_pixels(const float* src, int64_t src_sample_stride, const float* src2,
                                 int64_t src2_sample_stride, int N, int64_t HW, int C1, int C2, int Cpad,
                                 float* dst, void* stream) {
  GEECO_CHECK_ARG(src && dst, "pack_pixels: null pointer");
  GEECO_CHECK_ARG(N >= 1 && HW >= 1 && C1 >= 1 && C1 + (src2 ? C2 : 0) <= Cpad, "pack_pixels: bad dims");
  dim3 grid((unsigned)cdiv64(HW, 256), (unsigned)N);
  hipLaunchKernelGGL(pack_pixels_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (long long)src_sample_stride,
                     src2, (long long)src2_sample_stride, (long long)HW, C1, C2, Cpad, dst);
  GEECO_LAUNCH_CHECK();
  return 0;
}

// ---- on-device window builder ---------------------------------------------------------------------
// The reference materialises every K-frame window of an episode on the host (_window_v3,
// src/data/geeco_gym.py:615-631) and feeds 12.6 MB per sample over PCIe.  Here an episode's frames
// are uploaded ONCE (RGB as the uint8 values the recorder stored, data_recorder / tfrecord.py:73-74)
// and each batch's windows are gathered in HBM:  out[n][k][:] = conv(src[starts[n] + k][:]),
// conv(u8) = float(u8) / 255.0f  (the division of _parse_v4, geeco_gym.py:312, bit-exact).
template <typename T>
__global__ __launch_bounds__(256) void gather_windows_kernel(const T* __restrict__ src, const int* __restrict__ starts,
                                                             int K, long long frame_elems, float divisor,
                                                             float* __restrict__ out) {
  const int n = blockIdx.z, k = blockIdx.y;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= frame_elems) return;
  const T* s = src + (long long)(starts[n] + k) * frame_elems + i4;
  float* o = out + ((long long)n * K + k) * frame_elems + i4;
  if (i4 + 4 <= frame_elems) {
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = divisor != 1.f ? (float)s[j] / divisor : (float)s[j];
    *reinterpret_cast<f32x4*>(o) = f32x4{e[0], e[1], e[2], e[3]};
  } else {
    for (int j = 0; i4 + j < frame_elems; ++j) o[j] = divisor != 1.f ? (float)s[j] / divisor : (float)s[j];
  }
}

extern "C" int geeco_gather_windows(const void* src, int src_is_u8, const int* starts_dev, int N, int K,
                                    int64_t frame_elems, float divisor, float* out, void* stream) {
  GEECO_CHECK_ARG(src && starts_dev && out, "gather_windows: null pointer");
  GEECO_CHECK_ARG(N >= 1 && K >= 1 && frame_elems >= 4 && frame_elems % 4 == 0, "gather_windows: bad dims");
  GEECO_CHECK_ARG(divisor != 0.f, "gather_windows: divisor == 0");
  dim3 grid((unsigned)cdiv64(frame_elems, 1024), (unsigned)K, (unsigned)N);
  if (src_is_u8)
    hipLaunchKernelGGL(gather_windows_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, starts_dev, K, (long long)frame_elems, divisor, out);
  else
    hipLaunchKernelGGL(gather_windows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src,
                       starts_dev, K, (long long)frame_elems, divisor, out);
  GEECO_LAUNCH_CHECK();
  return 0;
}
