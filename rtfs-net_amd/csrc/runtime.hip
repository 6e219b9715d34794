// Host-side state shared by the launchers: everything here is per device and thread-safe, so the C ABI stays re-entrant
// (include/rtfs_amd.h: "stateless apart from per-device immutable configuration").
#include "common.h"
#include <mutex>
#include <vector>

namespace {
struct LdsEntry {
    const void* kernel;
    int device;
    size_t bytes;
};
std::mutex g_lds_mu;
std::vector<LdsEntry> g_lds;
}  // namespace

int rtfs_set_max_lds(const void* kernel, size_t bytes) {
    if (bytes > 160 * 1024) return RTFS_ERR_SHAPE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RTFS_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_lds_mu);
    for (auto& e : g_lds)
        if (e.kernel == kernel && e.device == dev) {
            if (bytes <= e.bytes) return RTFS_OK;
            if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return RTFS_ERR_LAUNCH;
            e.bytes = bytes;
            return RTFS_OK;
        }
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return RTFS_ERR_LAUNCH;
    g_lds.push_back({kernel, dev, bytes});
    return RTFS_OK;
}
