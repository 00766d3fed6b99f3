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

// ---- process-wide switches (include/fsf_hip.h: fsf_set_option)
#include <stdlib.h>
namespace fsf {
std::atomic<int64_t> g_opt_pool_brute{[] {
  const char* e = getenv("FSF_POOL_BRUTE");  // read once, when the library is loaded
  return (int64_t)(e ? atoll(e) : 0);
}()};
std::atomic<int64_t> g_host_waits{0};
}  // namespace fsf

extern "C" int fsf_set_option(int32_t option, int64_t value) {
  if (option == FSF_OPT_POOL_BRUTE) {
    fsf::g_opt_pool_brute.store(value, std::memory_order_relaxed);
    return FSF_OK;
  }
  return FSF_ERR_INVALID_ARG;
}

extern "C" int64_t fsf_get_option(int32_t option) {
  if (option == FSF_OPT_POOL_BRUTE) return fsf::g_opt_pool_brute.load(std::memory_order_relaxed);
  if (option == FSF_OPT_HOST_WAITS) return fsf::g_host_waits.load(std::memory_order_relaxed);
  return -1;
}
