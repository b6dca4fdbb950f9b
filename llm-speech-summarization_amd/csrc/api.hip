// api.hip — error reporting and library identity for libspeechllm.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void sl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sl_last_error(void) { return g_err; }

extern "C" int sl_version(void) { return 2; }

extern "C" int sl_device_arch(char* buf, int n) {
  SL_CHECK_ARG(buf != nullptr && n > 0, "sl_device_arch: bad buffer");
  int dev = 0;
  SL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SL_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, (size_t)n, "%s", prop.gcnArchName);
  return 0;
}
