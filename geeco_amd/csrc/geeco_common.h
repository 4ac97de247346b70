// Shared helpers for the gfx950 kernels (wave = 64 lanes, MFMA f32 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/geeco_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void geeco_set_error(const char* fmt, ...);
// records the name of the kernel a dispatcher is about to launch (no-op unless a trace was begun on this thread)
void geeco_note_kernel(const char* fmt, ...);
// conv_wgrad.hip: while set (per thread), geeco_launch_wgrad_reduce records the slab sum there instead of launching it
void geeco_set_pending_reduce(geeco_slab_reduce* p);

// Development switches (kernel-variant A/B, forced tile shapes; scripts/dev/SWITCHES.md lists them).  They exist only in the
// DEVELOPMENT build of the library (scripts/dev/build_dev_lib.sh: -DGEECO_DEV_KERNELS -> libgeeco_hip_dev.so), where the
// environment is consulted when GEECO_DEV=1 is set.  In the product library (csrc/build.sh) geeco_dev_getenv is the constant
// nullptr: every fork below folds to the measured-best path at compile time, the kernels only a switch could select are not
// compiled in (their launch code sits under #ifdef GEECO_DEV_KERNELS), and no GEECO_* variable is ever read.
#ifdef GEECO_DEV_KERNELS
const char* geeco_dev_getenv(const char* name);
#else
static inline constexpr const char* geeco_dev_getenv(const char*) { return nullptr; }
#endif

// CUs the persistent bottom-of-the-backward kernels leave free: the `reserved_cus` argument of the entry point being served on this
// thread (errors.cpp; 0 outside such a call)
int geeco_call_reserved_cus(void);
int geeco_enter_reserved_cus(int k);     // GEECO_EINVAL outside 0..128
void geeco_leave_reserved_cus(void);

#define GEECO_CHECK_ARG(cond, ...)              \
  do {                                          \
    if (!(cond)) {                              \
      geeco_set_error(__VA_ARGS__);             \
      return GEECO_EINVAL;                      \
    }                                           \
  } while (0)

#define GEECO_LAUNCH_CHECK()                                         \
  do {                                                               \
    hipError_t e_ = hipGetLastError();                               \
    if (e_ != hipSuccess) {                                          \
      geeco_set_error("launch failed: %s", hipGetErrorString(e_));   \
      return (int)e_;                                                \
    }                                                                \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// TF 'SAME' padding: out = ceil(in/s); pad_total = max((out-1)s + k - in, 0); before = total/2.
static inline void same_pad(int size, int k, int s, int* out, int* before) {
  int o = (size + s - 1) / s;
  int tot = (o - 1) * s + k - size;
  if (tot < 0) tot = 0;
  *out = o;
  *before = tot / 2;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
