// Library-level entry points: ABI version and the per-thread error string.
#include "common.h"

#include <stdlib.h>
#include <string.h>

#ifndef LDM_WT_STORES_DEFAULT
#define LDM_WT_STORES_DEFAULT false
#endif

namespace ldm {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool wt_stores_enabled() {
  static const bool on = [] {
    const char* e = getenv("LDM_WT_STORES");
    return e != nullptr ? e[0] != '0' : LDM_WT_STORES_DEFAULT;
  }();
  return on;
}
}  // namespace ldm

extern "C" int ldm_abi_version(void) { return LDM_ABI_VERSION; }
extern "C" const char* ldm_last_error(void) { return ldm::g_err; }
