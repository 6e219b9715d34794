// Host-side state shared by the launchers: everything here is per device and thread-safe, so the C ABI stays re-entrant
// (include/rtfs_amd.h: "stateless apart from per-device immutable configuration").
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>

static std::atomic<unsigned long long> g_launches{0};
void rtfs_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" unsigned long long rtfs_debug_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

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


// ---- side streams: slots 1.. for the batch parts of the separator call (rtfs_separator_forward_f32), slot 15 for the video side of the CAF
// (api.hip separator_part).  (Slot 0 carried a block's step-14 statistics pass until it moved into step 3's launch.)
// One (stream, fork event, join event) triple per (device, caller stream, slot), created on first use and kept for the life of the process.
namespace {
struct SideEntry {
    int device;
    hipStream_t owner;
    int slot;
    RtfsSide side;
};
std::mutex g_side_mu;
std::vector<SideEntry> g_side;
}  // namespace

int rtfs_side_stream(hipStream_t owner, int slot, RtfsSide* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RTFS_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (auto& e : g_side)
        if (e.device == dev && e.owner == owner && e.slot == slot) {
            *out = e.side;
            return RTFS_OK;
        }
    RtfsSide s{};
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(s.stream);
        return RTFS_ERR_LAUNCH;
    }
    if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {  // nothing half-built is left behind
        (void)hipEventDestroy(s.fork);
        (void)hipStreamDestroy(s.stream);
        return RTFS_ERR_LAUNCH;
    }
    g_side.push_back({dev, owner, slot, s});
    *out = s;
    return RTFS_OK;
}
