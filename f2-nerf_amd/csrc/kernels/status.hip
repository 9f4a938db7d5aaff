// status.hip -- ABI version and status strings of libf2nerf_hip.so.
#include "common.hiph"

#include <atomic>

namespace
{
std::atomic<int> g_options[F2N_OPT_COUNT];  // zero-initialised: every route at its default

bool option_value_ok(int key, int value)
{
  switch (key) {
    case F2N_OPT_SHADE_FWD:
    case F2N_OPT_SHADE_BWD:
    case F2N_OPT_BWD_COMBINE:
    case F2N_OPT_BWD_PHASES: return value == 0 || value == 1;
    case F2N_OPT_MARCH: return value >= 0 && value <= 2;
    case F2N_OPT_SHADE_VARIANT: return value >= 0 && value <= 3;
    case F2N_OPT_RAYTILE: return value == 0 || value == 16 || value == 32;
    case F2N_OPT_HASH_BWD: return value >= 0 && value <= 2;
    case F2N_OPT_RAYTILE_WALK: return value >= 0 && value <= 3;
    default: return false;
  }
}
}  // namespace

extern "C" int f2n_set_option(int key, int value)
{
  if (key < 0 || key >= F2N_OPT_COUNT || !option_value_ok(key, value)) return F2N_E_INVALID_ARG;
  return g_options[key].exchange(value, std::memory_order_relaxed);
}

extern "C" int f2n_get_option(int key)
{
  if (key < 0 || key >= F2N_OPT_COUNT) return F2N_E_INVALID_ARG;
  return g_options[key].load(std::memory_order_relaxed);
}

extern "C" int f2n_abi_version(void) { return F2N_ABI_VERSION; }

extern "C" const char * f2n_status_string(int status)
{
  switch (status) {
    case F2N_OK: return "ok";
    case F2N_E_INVALID_ARG: return "invalid argument";
    case F2N_E_LAUNCH: return "kernel launch failed";
    case F2N_E_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}
