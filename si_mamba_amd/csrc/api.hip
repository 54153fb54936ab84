// Version / error-string entry points of the C ABI.
#include <hip/hip_runtime.h>

#include "../../include/simamba.h"

extern "C" int simamba_abi_version(void) { return SIMAMBA_ABI_VERSION; }

extern "C" const char* simamba_strerror(int rc) {
  switch (rc) {
    case SIMAMBA_OK: return "ok";
    case SIMAMBA_E_NULLPTR: return "simamba: required pointer is NULL";
    case SIMAMBA_E_SHAPE: return "simamba: bad shape (negative size, dim <= 0 or batch > 65535)";
    case SIMAMBA_E_DTYPE: return "simamba: io_dtype must be SIMAMBA_F32 or SIMAMBA_BF16";
    case SIMAMBA_E_DSTATE: return "simamba: dstate must be in [1,16]";
    case SIMAMBA_E_WIDTH: return "simamba: conv width must be in [2,4]";
    case SIMAMBA_E_WORKSPACE: return "simamba: workspace missing or too small";
    case SIMAMBA_E_GROUPS: return "simamba: need 2 <= G <= 128, knn + 1 <= min(G, 32), k (+1) <= G, 1 <= F <= 64";
    case SIMAMBA_E_ALIGN: return "simamba: pointer not aligned";
    case SIMAMBA_E_VARIANT: return "simamba: scan variant unknown or not applicable to this shape";
    default: break;
  }
  if (rc > 0) return hipGetErrorString(static_cast<hipError_t>(rc));
  return "simamba: unknown error code";
}
