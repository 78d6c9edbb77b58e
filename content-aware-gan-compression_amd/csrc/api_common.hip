// ABI-level helpers: version, arch, thread-local error string.
#include "common.h"
#include <string.h>

namespace cagc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cagc

extern "C" int cagc_abi_version(void) { return CAGC_ABI_VERSION; }
extern "C" const char* cagc_last_error(void) { return cagc::g_err; }
extern "C" const char* cagc_arch(void) { return "gfx950"; }
