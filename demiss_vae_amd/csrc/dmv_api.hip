// Version and error strings of the C ABI.
#include "dmv_common.h"

extern "C" int dmv_version(void) { return DMV_VERSION; }

extern "C" const char* dmv_strerror(int rc) {
  switch (rc) {
    case DMV_OK: return "ok";
    case DMV_E_BADARG: return "dmv: bad argument (null pointer, negative size, or inconsistent dimensions / pitches)";
    case DMV_E_UNSUPPORTED: return "dmv: unsupported value (enum out of range, hidden > 256, nb > 8, or more than 64 mixture components)";
    case DMV_E_WORKSPACE: return "dmv: workspace too small";
    default: break;
  }
  if (rc > 0) return hipGetErrorString((hipError_t)rc);
  return "dmv: unknown error";
}
