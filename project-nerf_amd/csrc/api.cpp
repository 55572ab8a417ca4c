// Error channel and ABI version of libnerf_hip.so (see include/nerf_hip.h).
#include "common.h"

namespace nerf {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace nerf

extern "C" const char* nerf_last_error(void) { return nerf::g_err; }
extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }
