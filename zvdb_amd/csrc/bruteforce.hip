// bruteforce.hip -- placeholder until the MFMA kernel lands
#include "common.h"
namespace zvdb {
uint64_t bruteforce_workspace_bytes(const zvdb_hip_index *, uint32_t, uint32_t, uint32_t) { return 256; }
int launch_bruteforce(zvdb_hip_index *, const float *, uint32_t, uint32_t, uint32_t, uint64_t, void *,
                      uint64_t, uint64_t *, float *, uint32_t *, hipStream_t) {
    set_error("brute force not built yet");
    return ZVDB_ERR_UNSUPPORTED;
}
}  // namespace zvdb
