// C-ABI plumbing shared by the entry points of libdelora_hip.so: version, thread-local error text.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

thread_local char g_dl_err[256] = {0};

int dl_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_dl_err, sizeof(g_dl_err), fmt, ap);
  va_end(ap);
  return code;
}

// Launch errors only: never synchronises (the call stays asynchronous and graph-capturable).
int dl_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dl_fail(DL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DL_OK;
}

extern "C" int dl_abi_version(void) { return DL_ABI_VERSION; }
extern "C" const char* dl_last_error(void) { return g_dl_err; }
