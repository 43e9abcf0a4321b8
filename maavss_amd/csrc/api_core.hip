// Error state + version entry points of the C-ABI (include/maavss.h).
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void maavss_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
const char* maavss_last_error(void) { return g_err; }
int maavss_version(void) { return 100; }
const char* maavss_arch(void) { return "gfx950"; }
}
