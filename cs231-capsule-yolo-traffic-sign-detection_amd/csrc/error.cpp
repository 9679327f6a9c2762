// Thread-local error text for the C-ABI (include/capsyolo_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "capsyolo_hip.h"

static thread_local char g_err[512] = "";

int cy_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code ? code : CY_EINVAL;
}

extern "C" const char* capsyolo_last_error(void) { return g_err; }
extern "C" int capsyolo_abi_version(void) { return 5; }
