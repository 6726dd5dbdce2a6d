// Library-level entry points: version, error text, device count.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void ca_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ca_version(void) { return 100; }
extern "C" const char* ca_last_error(void) { return g_err; }
extern "C" int ca_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- CA_DEBUG_API: a stand-in for a resident collective kernel (tests / tools only) -------------------------------
// `blocks` workgroups of `threads` threads, each holding `lds_bytes` of LDS, that do nothing for `ms` milliseconds
// (s_memrealtime, 100 MHz; s_sleep between polls: no memory traffic, no vector work).  Launched on a side stream it
// occupies CUs the way an RCCL ring kernel does during the backward of an N > 1 run, so the effect of lost CUs on the
// persistent GEMM launches can be measured on one GPU (tools/r05_hog_gemm.py).
__global__ void ca_cu_hog_kernel(unsigned long long ticks) {
  extern __shared__ char hog_lds[];
  if (threadIdx.x == 0) hog_lds[0] = 1;  // (the allocation is what matters)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int ca_debug_cu_hog(int32_t blocks, int32_t threads, int32_t lds_bytes, double ms, void* stream) {
  CA_CHECK_ARG(blocks > 0 && blocks <= 4096 && threads >= 64 && threads <= 1024 && (threads % 64) == 0 && lds_bytes >= 0 &&
                   lds_bytes <= 160 * 1024 && ms > 0.0 && ms <= 2000.0,
               "ca_debug_cu_hog: bad argument");
  static int attr = 0;
  if (lds_bytes > attr) {
    hipFuncSetAttribute((const void*)ca_cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = 160 * 1024;
  }
  hipLaunchKernelGGL(ca_cu_hog_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)(ms * 1e5));
  CA_CHECK_LAUNCH("ca_debug_cu_hog");
  return CA_OK;
}
