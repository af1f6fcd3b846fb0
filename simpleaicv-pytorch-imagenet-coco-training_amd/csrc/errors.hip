#include "sacv_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void sacv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sacv_last_error(void) { return g_err; }
extern "C" int sacv_abi_version(void) { return 2; }
