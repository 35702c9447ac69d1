// Library-level entry points: ABI version and the per-thread error string.
#include "common.h"

#include <string.h>

namespace ldm {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ldm

extern "C" int ldm_abi_version(void) { return LDM_ABI_VERSION; }
extern "C" const char* ldm_last_error(void) { return ldm::g_err; }
