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
