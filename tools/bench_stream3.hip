// Diagnostic micro-benchmark (not part of the product), round 3, second pass.  bench_stream2 showed that loads in flight are NOT what
// separates a 4.7 TB/s persistent copy from the documented 6.3 TB/s one-shot copy (more in flight was slower).  This one separates
// the remaining suspects: grid shape (one-shot / static stride / dynamic tile counter), power-of-two strides, read/write base
// alignment, occupancy - first on a flat copy, then on the pointwise kernels' row-walk pattern (lane (r, h): pixels 2r, 2r+1 of
// channel row 2j + h; a wave-instruction = two 256-byte segments in two rows P = 251*129 floats apart).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bs3 tools/bench_stream3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: one-shot (block = one piece of U*4 KB); 1: static stride over pieces; 2: dynamic counter
template <int U, int MODE>
__global__ __launch_bounds__(256) void copyp(const v4f* __restrict__ x, v4f* __restrict__ y, size_t npieces, unsigned* __restrict__ ctr) {
    extern __shared__ unsigned char smem[];
    unsigned* nxt = reinterpret_cast<unsigned*>(smem);
    size_t piece = blockIdx.x;
    while (piece < npieces) {
        const size_t base = piece * (U * 256) + threadIdx.x;
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x *= 1.5f; y[base + u * 256] = v[u]; }
        if (MODE == 0) break;
        if (MODE == 1) piece += gridDim.x;
        if (MODE == 2) {
            __syncthreads();
            if (threadIdx.x == 0) *nxt = atomicAdd(ctr, 1u) + gridDim.x;
            __syncthreads();
            piece = *nxt;
        }
    }
}

// row-walk pipeline, see header.  stage = QR registers (f32x2) per stream = 2*QR channel rows; DEPTH stages ahead.
template <int NT_, int QR, int DEPTH, int MF, int MODE>
__global__ __launch_bounds__(NT_) void rows_pipe(const float* __restrict__ R, const float* __restrict__ A, float* __restrict__ OUT, int C, int P,
                                                 int ntiles, int tps, float* __restrict__ sink, unsigned* __restrict__ ctr) {
    extern __shared__ unsigned char smem[];
    unsigned* nxt = reinterpret_cast<unsigned*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r_ = lane & 31, h = lane >> 5;
    const int NS = C / (2 * QR);
    half8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 2, 2, 3, 3, 4, 4};
    f32x16 acc = {0};
    int tile = blockIdx.x;
    while (tile < ntiles) {
        const int b = tile / tps;
        const int p0 = min((tile - b * tps) * (NT_ / 64 * 64) + wave * 64 + 2 * r_, P - 2);
        const float* __restrict__ rs = R + (size_t)b * C * P + p0 + (size_t)h * P;
        const float* __restrict__ as = A + (size_t)b * C * P + p0 + (size_t)h * P;
        float* __restrict__ os = OUT + (size_t)b * C * P + p0 + (size_t)h * P;
        f32x2u r[DEPTH + 1][QR], a[DEPTH + 1][QR];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < QR; ++j) {
                r[d][j] = *reinterpret_cast<const f32x2u*>(rs + (unsigned)((d * QR + j) * 2 * P));
                a[d][j] = *reinterpret_cast<const f32x2u*>(as + (unsigned)((d * QR + j) * 2 * P));
            }
        for (int s0 = 0; s0 < NS; s0 += DEPTH + 1) {
#pragma unroll
            for (int k = 0; k <= DEPTH; ++k) {
                const int s = s0 + k;
                const int sp = min(s + DEPTH, NS - 1);
                const int bufp = (k + DEPTH) % (DEPTH + 1);
                if (DEPTH > 0) {
#pragma unroll
                    for (int j = 0; j < QR; ++j) {
                        r[bufp][j] = *reinterpret_cast<const f32x2u*>(rs + (unsigned)((sp * QR + j) * 2 * P));
                        a[bufp][j] = *reinterpret_cast<const f32x2u*>(as + (unsigned)((sp * QR + j) * 2 * P));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < QR; ++j) {
                        r[0][j] = *reinterpret_cast<const f32x2u*>(rs + (unsigned)((s * QR + j) * 2 * P));
                        a[0][j] = *reinterpret_cast<const f32x2u*>(as + (unsigned)((s * QR + j) * 2 * P));
                    }
                }
                if (s < NS) {
#pragma unroll
                    for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < QR; ++j)
                        *reinterpret_cast<f32x2u*>(os + (unsigned)((s * QR + j) * 2 * P)) = r[k][j] * 1.5f + a[k][j];
                }
            }
        }
        if (MODE == 0) break;
        if (MODE == 1) tile += gridDim.x;
        if (MODE == 2) {
            __syncthreads();
            if (threadIdx.x == 0) *nxt = atomicAdd(ctr, 1u) + gridDim.x;
            __syncthreads();
            tile = *nxt;
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

int main(int argc, char** argv) {
    const size_t n = (size_t)256 * 1024 * 1024;  // floats: 1 GiB per tensor
    float *x, *y, *z, *out;
    unsigned* ctr;
    CK(hipMalloc(&x, n * 4 + (8 << 20)));
    CK(hipMalloc(&y, n * 4 + (8 << 20)));
    CK(hipMalloc(&z, n * 4 + (8 << 20)));
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&ctr, 4096));
    CK(hipMemset(x, 0, n * 4));
    CK(hipMemset(y, 0, n * 4));
    CK(hipMemset(z, 0, n * 4));
    printf("x %p y %p z %p\n", x, y, z);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0);
        const int R = 10;
        for (int i = 0; i < R; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipError_t e = hipGetLastError();
        printf("%-72s %8.1f us  %6.2f TB/s %s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
        fflush(stdout);
    };
    const size_t n4 = n / 4;
    const double cb = 2.0 * n * 4;
#define COPY(U, MODE, G, LDS, YOFF) { const size_t np = n4 / (U * 256); const int g = MODE == 0 ? (int)np : G; \
        hipFuncSetAttribute((const void*)copyp<U, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
        char nm[128]; snprintf(nm, 128, "copy U=%d %s grid=%d lds=%dK yoff=%d", U, MODE == 0 ? "one-shot" : MODE == 1 ? "static" : "dynamic", g, LDS, YOFF); \
        timeit(nm, cb, [&] { if (MODE == 2) hipMemsetAsync(ctr, 0, 4, 0); \
            hipLaunchKernelGGL((copyp<U, MODE>), dim3(g), dim3(256), LDS * 1024 + 16, 0, (const v4f*)x, (v4f*)(y + YOFF), np, ctr); }); }
    COPY(1, 0, 0, 0, 0) COPY(2, 0, 0, 0, 0) COPY(4, 0, 0, 0, 0) COPY(8, 0, 0, 0, 0)
    COPY(1, 0, 0, 20, 0) COPY(1, 0, 0, 40, 0) COPY(4, 0, 0, 40, 0) COPY(4, 0, 0, 64, 0)
    COPY(1, 0, 0, 0, 1024) COPY(1, 0, 0, 0, 263168)
    COPY(1, 1, 2048, 0, 0) COPY(1, 1, 2040, 0, 0) COPY(1, 1, 1920, 0, 0) COPY(1, 1, 4080, 0, 0) COPY(1, 1, 2040, 0, 263168)
    COPY(4, 1, 2040, 0, 0) COPY(4, 1, 1016, 0, 0) COPY(4, 1, 504, 0, 0)
    COPY(1, 2, 2048, 0, 0) COPY(4, 2, 2048, 0, 0) COPY(4, 2, 1024, 0, 0) COPY(8, 2, 1024, 0, 0) COPY(8, 2, 512, 0, 0)

    {
        const int C = 256, P = 251 * 129, B = 32;
        const double rb = 3.0 * B * C * (double)P * 4;
#define ROWS(NT_, WGPC, QR, DEPTH, MF, MODE) { \
            const int tps = (P + NT_ - 1) / NT_, nt = tps * B; \
            const size_t lds = WGPC == 1 ? 100 * 1024 : (WGPC == 2 ? 70 * 1024 : (WGPC == 4 ? 36 * 1024 : 16)); \
            const int g = MODE == 0 ? nt : 256 * (WGPC ? WGPC : 8); \
            hipFuncSetAttribute((const void*)rows_pipe<NT_, QR, DEPTH, MF, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            char nm[128]; snprintf(nm, 128, "rows thr=%d wg/cu=%d QR=%d depth=%d mfma=%d %s grid=%d", NT_, WGPC, QR, DEPTH, MF, MODE == 0 ? "one-shot" : MODE == 1 ? "static" : "dynamic", g); \
            timeit(nm, rb, [&] { if (MODE == 2) hipMemsetAsync(ctr, 0, 4, 0); \
                hipLaunchKernelGGL((rows_pipe<NT_, QR, DEPTH, MF, MODE>), dim3(g), dim3(NT_), lds, 0, x, z, y, C, P, nt, tps, out, ctr); }); }
        // free occupancy (wg/cu = 0 -> no LDS limit), small register footprint: what the pattern itself can do
        ROWS(256, 0, 4, 0, 0, 0)
        ROWS(256, 0, 8, 0, 0, 0)
        ROWS(256, 0, 4, 1, 0, 0)
        ROWS(256, 0, 4, 0, 0, 1)
        ROWS(256, 0, 8, 0, 0, 1)
        ROWS(256, 0, 4, 0, 0, 2)
        ROWS(64, 0, 4, 0, 0, 0)
        ROWS(64, 0, 8, 0, 0, 0)
        ROWS(128, 0, 8, 0, 0, 0)
        ROWS(512, 0, 4, 0, 0, 0)
        // occupancy-limited like the real kernels
        ROWS(512, 1, 4, 0, 0, 1)
        ROWS(512, 1, 4, 0, 12, 1)
        ROWS(512, 1, 4, 1, 12, 1)
        ROWS(512, 1, 8, 1, 24, 1)
        ROWS(512, 1, 4, 0, 12, 2)
        ROWS(512, 1, 8, 1, 24, 2)
        ROWS(256, 2, 8, 0, 0, 1)
        ROWS(256, 2, 8, 1, 0, 1)
        ROWS(256, 2, 8, 1, 24, 1)
        ROWS(256, 2, 8, 1, 24, 2)
        ROWS(256, 2, 16, 1, 48, 1)
        ROWS(256, 4, 8, 0, 0, 1)
        ROWS(256, 4, 8, 1, 0, 1)
        ROWS(256, 4, 8, 1, 24, 1)
        ROWS(256, 4, 8, 1, 24, 2)
        ROWS(256, 4, 8, 1, 24, 0)
    }
    return 0;
}
