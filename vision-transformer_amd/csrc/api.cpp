// Version and thread-local error reporting of libvitm_gfx950.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>
#include <atomic>

#include "../../include/vitm_gfx950.h"

static thread_local char g_err[512] = "";

void vitm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vitm_version(void) { return "vitm_gfx950 0.1.0 (gfx950, bf16 MFMA)"; }
extern "C" const char* vitm_last_error(void) { return g_err; }

int vitm_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int n = cache[dev & 63].load(std::memory_order_relaxed);
  if (n <= 0) {
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cache[dev & 63].store(n, std::memory_order_relaxed);
  }
  return n;
}
