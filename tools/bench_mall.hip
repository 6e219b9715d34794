// Diagnostic micro-benchmark (not part of the product), round 3: does the traversal ORDER of a consumer relative to its producer matter?
// The memory-side cache (256 MB) sits behind the L2s; a 265 MB tensor written front-to-back and then read front-to-back finds its head
// evicted by its own tail, read back-to-front it should find the tail resident.
//   pass: y[i] = 1.5 x[i] + 1 over n floats (16 bytes per lane, block b handles a contiguous chunk), ping-pong x <-> y
//   policy 0: every pass ascending; policy 1: passes alternate ascending / descending
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bmall tools/bench_mall.hip ; run: tools/_bmall [MB per tensor]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool WR>
__global__ __launch_bounds__(256) void pass(const v4f* __restrict__ x, v4f* __restrict__ y, size_t n4, int rev, float* sink) {
    size_t b = blockIdx.x;
    if (rev) b = gridDim.x - 1 - b;
    const size_t i0 = b * 1024 + threadIdx.x;
    v4f v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = x[min(i0 + 256 * j, n4 - 1)];
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const v4f o = v[j] * 1.5f + 1.0f;
        if (WR) {
            if (i0 + 256 * j < n4) y[i0 + 256 * j] = o;
        } else
            acc += o[0] + o[1] + o[2] + o[3];
    }
    if (!WR && acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 265;
    const size_t n4 = mb * 1000000 / 16;
    v4f *x, *y;
    float* sink;
    CK(hipMalloc(&x, n4 * 16));
    CK(hipMalloc(&y, n4 * 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, n4 * 16));
    CK(hipMemset(y, 0, n4 * 16));
    const unsigned grid = (unsigned)((n4 + 1023) / 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int policy = 0; policy < 2; ++policy) {
        // read + write ping-pong
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 40; ++it) {
                const int rev = policy ? (it & 1) : 0;
                hipLaunchKernelGGL(pass<true>, dim3(grid), dim3(256), 0, 0, (it & 1) ? y : x, (it & 1) ? x : y, n4, rev, sink);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%zu MB  rw ping-pong  policy %d: %.1f us per pass = %.2f TB/s (read + write)\n", mb, policy, ms * 25, 2.0 * n4 * 16 / (ms / 40 * 1e-3) / 1e12);
        }
        // write then read-only (stats pass after a producer)
        for (int rep = 0; rep < 2; ++rep) {
            float tw = 0, tr = 0;
            for (int it = 0; it < 20; ++it) {
                float ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(pass<true>, dim3(grid), dim3(256), 0, 0, x, y, n4, 0, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                tw += ms;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(pass<false>, dim3(grid), dim3(256), 0, 0, y, x, n4, policy, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                tr += ms;
            }
            if (rep) printf("%zu MB  producer %.1f us, read-only consumer (policy %d) %.1f us = %.2f TB/s\n", mb, tw * 50, policy, tr * 50, n4 * 16 / (tr / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
