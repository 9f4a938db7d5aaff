// status.hip -- ABI version and status strings of libf2nerf_hip.so.
#include "common.hiph"

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
