// Helpers shared by the training-side kernel files (k_train_*.hip).
#pragma once
#include "common.h"
#include "kernels.h"

namespace {
// dynamic LDS above the default limit needs the attribute once per kernel
template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return RTFS_ERR_SHAPE;
    return bytes > 48 * 1024 ? rtfs_set_max_lds((const void*)kernel, bytes) : RTFS_OK;
}
// grid of 256-thread workgroups for a grid-stride loop over n elements
inline unsigned grid_for(size_t n, unsigned cap = 8192) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
// Workgroup b of a launch runs on XCD b % 8, each with its own L2: kernels whose neighbouring workgroups re-read each other's rows
// (depthwise windows) index their work by xcd_block(), which hands XCD k the k-th contiguous eighth of the logical workgroups - with the
// plain order every XCD pulls the whole tensor through the fabric (measured 78 us for a 33 MB depthwise pass).  gridDim.x % 8 == 0.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nb) { return (bid & 7u) * (nb >> 3) + (bid >> 3); }
inline unsigned grid8(unsigned g) { return (g + 7u) / 8u * 8u; }
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * sigmoidf_(2.0f * x) - 1.0f; }
}  // namespace
