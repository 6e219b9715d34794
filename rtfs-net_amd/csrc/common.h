// Shared device/host helpers for the RTFS-Net gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTFS_OK 0
#define RTFS_ERR_SHAPE (-1)
#define RTFS_ERR_WORKSPACE (-2)
#define RTFS_ERR_LAUNCH (-3)
#define RTFS_ERR_ARG (-4)

#define RTFS_EPS 1e-5f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

void rtfs_count_launch();  // runtime.hip: process-wide count of kernel launches (rtfs_debug_launch_count; a regression guard for the small-batch path)
static inline int rtfs_launch_status() {
    rtfs_count_launch();
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RTFS_OK : RTFS_ERR_LAUNCH;
}

// Raise a kernel's dynamic-LDS limit to `bytes` on the CURRENT device if it is not already that high.  The attribute is per device
// (and the library may be driven from several host threads / devices in one process): the cache is keyed by (device, kernel) and
// guarded by a mutex (runtime.hip).
int rtfs_set_max_lds(const void* kernel, size_t bytes);
struct RtfsSide {
    hipStream_t stream;
    hipEvent_t fork, join;
};
int rtfs_side_stream(hipStream_t owner, int slot, RtfsSide* out);

#define RTFS_RETURN_IF(cond, code) \
    do {                           \
        if (cond) return (code);   \
    } while (0)

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide (sum, sum of squares) in double, then ONE pair of f64 atomics per block.
// `red` = LDS scratch of >= 2 * (blockDim.x / 64) doubles.
__device__ __forceinline__ void block_stats_atomic(float s, float ss, double* red, double* dst) {
    double ds = wave_sum_d((double)s), dss = wave_sum_d((double)ss);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[2 * w] = ds;
        red[2 * w + 1] = dss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int i = 0; i < nw; ++i) {
            a += red[2 * i];
            b += red[2 * i + 1];
        }
        atomicAdd(dst, a);
        atomicAdd(dst + 1, b);
    }
}

// same, for two float partial sums whose destination is a pair of doubles (dst[0] += s, dst[1] += ss)
__device__ __forceinline__ void block_stats_atomic_pair(float s, float ss, double* red, double* dst) { block_stats_atomic(s, ss, red, dst); }

// GroupNorm(1,C) affine folded to y = x*scale + shift from accumulated (sum, sumsq).
__device__ __forceinline__ void gln_fold(const double* st, double inv_count, float gamma, float beta, float& scale,
                                         float& shift) {
    // the cancellation-prone part (E[x^2] - mean^2, and mean * scale in the shift) stays in f64; the reciprocal square
    // root runs in f32 (an f64 rsqrt + division costs ~80 instructions in every consumer's prologue)
    const double mean = st[0] * inv_count;
    double var = st[1] * inv_count - mean * mean;
    var = var < 0 ? 0 : var;
    const float rstd = 1.0f / sqrtf((float)(var + (double)RTFS_EPS));
    scale = gamma * rstd;
    shift = (float)((double)beta - mean * (double)scale);
}

// v_exp_f32 + v_rcp_f32 (1 ulp each); saturates correctly: exp2(+inf) -> rcp(inf) = 0
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float preluf_(float x, float a) { return x >= 0.f ? x : a * x; }

// legacy 'nearest' source index: floor(dst * in / out)
__device__ __forceinline__ int nearest_src(int dst, int n_in, int n_out) {
    int s = (int)(((long long)dst * n_in) / n_out);
    return s < n_in - 1 ? s : n_in - 1;
}
