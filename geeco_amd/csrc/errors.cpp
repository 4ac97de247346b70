#include "geeco_common.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void geeco_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#ifdef GEECO_DEV_KERNELS
const char* geeco_dev_getenv(const char* name) {
  static const bool dev = [] {
    const char* v = getenv("GEECO_DEV");
    return v && v[0] && strcmp(v, "0") != 0;
  }();
  return dev ? getenv(name) : nullptr;
}
#endif
// 1: this library was built with the development kernel variants and reads GEECO_* switches under GEECO_DEV=1; 0: product library
extern "C" int geeco_has_dev_kernels(void) {
#ifdef GEECO_DEV_KERNELS
  return 1;
#else
  return 0;
#endif
}

// CUs the persistent bottom-of-the-backward kernels leave free: an ARGUMENT of the entry points that launch them (round 4; a
// process-wide setting before).  The dispatchers below those entry points read it from here; thread-local and set only for the
// duration of one call, like the pending slab sum of conv_wgrad.hip: no state survives a call.
static thread_local int g_call_reserved_cus = 0;
int geeco_call_reserved_cus(void) { return g_call_reserved_cus; }
int geeco_enter_reserved_cus(int k) {
  if (k < 0 || k > 128) {
    geeco_set_error("reserved_cus: %d outside 0..128", k);
    return GEECO_EINVAL;
  }
  g_call_reserved_cus = k;
  return 0;
}
void geeco_leave_reserved_cus(void) { g_call_reserved_cus = 0; }

extern "C" const char* geeco_last_error(void) { return g_err; }
extern "C" int geeco_abi_version(void) { return GEECO_ABI_VERSION; }

// ---- dispatch trace (diagnostics for bench.py / tests): which kernels a call launched ----------------
static thread_local bool g_trace_on = false;
static thread_local char g_trace[1024] = "";

void geeco_note_kernel(const char* fmt, ...) {
  if (!g_trace_on) return;
  size_t n = strlen(g_trace);
  if (n + 2 >= sizeof(g_trace)) return;
  if (n) g_trace[n++] = ';';
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_trace + n, sizeof(g_trace) - n, fmt, ap);
  va_end(ap);
}

extern "C" void geeco_debug_kernel_trace_begin(void) {
  g_trace_on = true;
  g_trace[0] = 0;
}
extern "C" const char* geeco_debug_kernel_trace_end(void) {
  g_trace_on = false;
  return g_trace;
}
