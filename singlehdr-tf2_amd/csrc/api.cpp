// libshdr: error state and version (host only).
#include <stdarg.h>
#include <stdio.h>

#include "shdr_internal.h"

namespace shdr {
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
}  // namespace shdr

extern "C" const char* shdr_last_error(void) { return shdr::g_last_error; }
extern "C" const char* shdr_version(void) { return "libshdr 0.1 gfx950"; }
