// libsdmi: error reporting + version (C ABI, see include/sdmi.h).
#include <hip/hip_runtime.h>
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

extern "C" int sdmi_version(void) { return SDMI_VERSION; }
extern "C" const char* sdmi_last_error(void) { return g_err; }
