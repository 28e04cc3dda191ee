// libsdmi: error reporting + version (C ABI, see include/sdmi.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sdmi.h"

static thread_local char g_err[512] = "";

void sdmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int sdmi_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sdmi_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SDMI_ELAUNCH;
  }
  return SDMI_OK;
}

// Dynamic-LDS opt-in of a kernel beyond the 64 KB default: once per (call site, device), safe against
// racing host threads (the attribute is per device function per device; setting it twice is harmless),
// and a failure is reported instead of surfacing later as a failed launch.
int sdmi_optin_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return SDMI_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sdmi_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", what,
                   bytes, dev, hipGetErrorString(e));
    return SDMI_ELAUNCH;
  }
  done.fetch_or(bit, std::memory_order_release);
  return SDMI_OK;
}

extern "C" int sdmi_version(void) { return SDMI_VERSION; }
extern "C" const char* sdmi_last_error(void) { return g_err; }
