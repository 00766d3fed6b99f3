// Status strings + ABI version of libfsf_hip.
#include "common.h"

extern "C" const char* fsf_status_string(int status) {
  switch (status) {
    case FSF_OK: return "ok";
    case FSF_ERR_INVALID_ARG: return "invalid argument";
    case FSF_ERR_WORKSPACE: return "workspace too small";
    case FSF_ERR_KEY_RANGE: return "row key does not fit the packed 64-bit key / value outside the given bounds";
    case FSF_ERR_HIP: return "HIP runtime error";
    case FSF_ERR_CAPACITY: return "output capacity exceeded";
    case FSF_ERR_UNSUPPORTED: return "unsupported shape";
    default: return "unknown status";
  }
}

extern "C" int fsf_abi_version(void) { return FSF_ABI_VERSION; }
