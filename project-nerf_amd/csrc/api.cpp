// Error channel, ABI version, development options and per-device launch facts of libnerf_hip.so
// (see include/nerf_hip.h).
#include <stdlib.h>
#include <string.h>
#include <mutex>
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

// ---- development options: the environment is read once, when the first entry point asks ----
struct OptionSlot { const char* name; const char* env; int Options::*field; };
static const OptionSlot kSlots[] = {
    {"chain_legacy", "NERF_CHAIN_LEGACY", &Options::chain_legacy},
    {"fwd_cycles", "NERF_FWD_CYCLES", &Options::fwd_cycles},
    {"wgrad_overhead", "NERF_WGRAD_OVH", &Options::wgrad_overhead},
    {"wgrad_bw_x16", "NERF_WGRAD_BW", &Options::wgrad_bw_x16},
    {"wgrad_fixed", "NERF_WGRAD_FIXED", &Options::wgrad_fixed},
    {"wgrad_debug", "NERF_WGRAD_DEBUG", &Options::wgrad_debug},
    {"wgrad_small_span", "NERF_WGRAD_SMALL_SPAN", &Options::wgrad_small_span},
    {"wgrad_small_cap", "NERF_WGRAD_SMALL_CAP", &Options::wgrad_small_cap},
    {"wgrad_only", "NERF_WGRAD_ONLY", &Options::wgrad_only},
    {"hash_bwd_only_level", "NERF_HASH_BWD_ONLY_LEVEL", &Options::hash_bwd_only_level},
    {"hash_bwd_atomic", "NERF_HASH_BWD_ATOMIC", &Options::hash_bwd_atomic},
    {"wgrad_atomic", "NERF_WGRAD_ATOMIC", &Options::wgrad_atomic},
    {"wgrad_k16", "NERF_WGRAD_K16", &Options::wgrad_k16},
    {"wgrad_big_only", "NERF_WGRAD_BIG_ONLY", &Options::wgrad_big_only},
    {"infer_shape32", "NERF_INFER_SHAPE32", &Options::infer_shape32},
    {"infer64", "NERF_INFER64", &Options::infer64},
    {"stash_fp8", "NERF_STASH_FP8", &Options::stash_fp8},
    {"chain_grid", "NERF_CHAIN_GRID", &Options::chain_grid},
    {"wgrad_grid", "NERF_WGRAD_GRID", &Options::wgrad_grid},
    {"hash_fwd_lds_kb", "NERF_HASH_FWD_LDS_KB", &Options::hash_fwd_lds_kb},
    {"hash_xcd", "NERF_HASH_XCD", &Options::hash_xcd},
    {"composite_wgs_per_cu", "NERF_COMPOSITE_WGS", &Options::composite_wgs_per_cu},
    {"deterministic", "NERF_DETERMINISTIC", &Options::deterministic},
    {"tv_blocks", "NERF_TV_BLOCKS", &Options::tv_blocks},
};

Options& options() {
  static Options o;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const OptionSlot& s : kSlots)
      if (const char* v = getenv(s.env)) o.*(s.field) = (*v == 0) ? 1 : atoi(v);
  });
  return o;
}

// ---- per-device facts: CU count and which kernels already carry their dynamic-LDS attribute
// (hipFuncSetAttribute applies to the current device only) ----
namespace {
constexpr int kMaxDevices = 64, kMaxKernels = 64;
struct DeviceFacts {
  int n_cu = 0;
  int n_kernels = 0;
  const void* kernels[kMaxKernels];
};
DeviceFacts g_dev[kMaxDevices];
std::mutex g_dev_mutex;
}  // namespace

int device_cu_count(int* n_cu) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(NERF_ELAUNCH, "cannot query the current device");
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  if (g_dev[dev].n_cu == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(NERF_ELAUNCH, "cannot query device %d", dev);
    g_dev[dev].n_cu = prop.multiProcessorCount;
  }
  *n_cu = g_dev[dev].n_cu;
  return NERF_OK;
}

int ensure_dynamic_lds(const void* kernel, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return fail(NERF_ELAUNCH, "cannot query the current device");
  std::lock_guard<std::mutex> lock(g_dev_mutex);
  DeviceFacts& d = g_dev[dev];
  for (int i = 0; i < d.n_kernels; ++i)
    if (d.kernels[i] == kernel) return NERF_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return fail(NERF_ELAUNCH, "%s: cannot raise the dynamic LDS limit to %d on device %d", what, bytes, dev);
  if (d.n_kernels < kMaxKernels) d.kernels[d.n_kernels++] = kernel;
  return NERF_OK;
}
}  // namespace nerf

extern "C" const char* nerf_last_error(void) { return nerf::g_err; }
extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }

extern "C" int nerf_set_option(const char* name, int value) {
  NERF_REQUIRE(name != nullptr, "nerf_set_option: NULL name");
  for (const nerf::OptionSlot& s : nerf::kSlots)
    if (strcmp(s.name, name) == 0) {
      nerf::options().*(s.field) = value;
      return NERF_OK;
    }
  return nerf::fail(NERF_EINVAL, "nerf_set_option: unknown option '%s'", name);
}

extern "C" int nerf_get_option(const char* name, int* value) {
  NERF_REQUIRE(name != nullptr && value != nullptr, "nerf_get_option: NULL argument");
  for (const nerf::OptionSlot& s : nerf::kSlots)
    if (strcmp(s.name, name) == 0) {
      *value = nerf::options().*(s.field);
      return NERF_OK;
    }
  return nerf::fail(NERF_EINVAL, "nerf_get_option: unknown option '%s'", name);
}
