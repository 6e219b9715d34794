// Diagnostic micro-benchmark (not part of the product), round 3: why do the streaming kernels sit at 4.2-4.7 TB/s when a
// float4 copy is documented at 6.29 TB/s?  Hypothesis: bytes in flight per CU (memory-level parallelism), not access width.
//   part 1: flat float4 copy with 1 / 2 / 4 / 8 loads in flight per lane, plain vs nontemporal, persistent vs one-shot grid
//   part 2: the pointwise kernels' row-walk pattern (a wave owns 64 adjacent pixels = one 8-byte access per lane and channel
//           row, rows P = 251*129 floats apart) as a software pipeline: stage = QR rows of TWO input streams + one output
//           stream (the block-boundary kernel's epilogue), DEPTH stages of loads in flight, NW waves per CU, optional MFMA
//           ballast per stage so the wave is as busy as the real kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bs2 tools/bench_stream2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy4(const v4f* __restrict__ x, v4f* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u].x *= 1.5f;
            if (NT) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) { v4f v = x[i]; v.x *= 1.5f; y[i] = v; }
}
// contiguous chunk per workgroup (U KB-sized pieces in flight per wave)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy4_chunk(const v4f* __restrict__ x, v4f* __restrict__ y, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
    for (size_t i = b0 + threadIdx.x; i < b1; i += U * 256) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t k = i + u * 256 < b1 ? i + u * 256 : b1 - 1; v[u] = NT ? __builtin_nontemporal_load(x + k) : x[k]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u].x *= 1.5f;
            if (i + u * 256 < b1) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
        }
    }
}
template <int U>
__global__ __launch_bounds__(256) void read4(const v4f* __restrict__ x, float* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 123.456f) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void write4(v4f* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v4f{1.f, 2.f, 3.f, (float)u};
    }
}

// ---- part 2: row-walk pipeline.  Tensors R (C rows), A (C rows) -> OUT (C rows) per sample; wave = 64 px, lane = 2 adjacent px.
// stage = QR rows (channels) of R and A; DEPTH = stages of loads issued ahead of the one being consumed.
template <int NT_, int QR, int DEPTH, int MF, bool NTS>
__global__ __launch_bounds__(NT_) void rows_pipe(const float* __restrict__ R, const float* __restrict__ A, float* __restrict__ OUT, int C, int P,
                                                 int ntiles, int tps, float* __restrict__ sink) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NS = C / QR;
    half8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 2, 2, 3, 3, 4, 4};
    f32x16 acc = {0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tps;
        const int p0 = min((tile - b * tps) * (NT_ / 64 * 64) + wave * 64 + 2 * lane, P - 2);
        const float* __restrict__ rs = R + (size_t)b * C * P + p0;
        const float* __restrict__ as = A + (size_t)b * C * P + p0;
        float* __restrict__ os = OUT + (size_t)b * C * P + p0;
        f32x2u r[DEPTH + 1][QR], a[DEPTH + 1][QR];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < QR; ++j) {
                r[d][j] = *reinterpret_cast<const f32x2u*>(rs + (unsigned)((d * QR + j) * P));
                a[d][j] = *reinterpret_cast<const f32x2u*>(as + (unsigned)((d * QR + j) * P));
            }
        // stages in groups of DEPTH+1 so every buffer index is static
        for (int s0 = 0; s0 < NS; s0 += DEPTH + 1) {
#pragma unroll
            for (int k = 0; k <= DEPTH; ++k) {
                const int s = s0 + k;
                const int sp = min(s + DEPTH, NS - 1);  // prefetch stage (clamped: re-reads the last stage at the end)
                constexpr int dummy = 0; (void)dummy;
                const int bufp = (k + DEPTH) % (DEPTH + 1);
#pragma unroll
                for (int j = 0; j < QR; ++j) {
                    r[bufp][j] = *reinterpret_cast<const f32x2u*>(rs + (unsigned)((sp * QR + j) * P));
                    a[bufp][j] = *reinterpret_cast<const f32x2u*>(as + (unsigned)((sp * QR + j) * P));
                }
                if (s < NS) {
#pragma unroll
                    for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < QR; ++j) {
                        const f32x2u y = r[k][j] * 1.5f + a[k][j];
                        if (NTS) __builtin_nontemporal_store(y, reinterpret_cast<f32x2u*>(os + (unsigned)((s * QR + j) * P)));
                        else *reinterpret_cast<f32x2u*>(os + (unsigned)((s * QR + j) * P)) = y;
                    }
                }
            }
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

int main(int argc, char** argv) {
    const size_t n = (size_t)256 * 1024 * 1024;  // floats: 1 GiB per tensor
    float *x, *y, *z, *out;
    CK(hipMalloc(&x, n * 4 + 4096));
    CK(hipMalloc(&y, n * 4 + 4096));
    CK(hipMalloc(&z, n * 4 + 4096));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(x, 0, n * 4));
    CK(hipMemset(y, 0, n * 4));
    CK(hipMemset(z, 0, n * 4));
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
        printf("%-64s %8.1f us  %6.2f TB/s %s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
        fflush(stdout);
    };
    const size_t n4 = n / 4;
    const double cb = 2.0 * n * 4;
#define COPY(U, NT, G) { char nm[96]; snprintf(nm, 96, "copy4 U=%d %s grid=%d", U, NT ? "nt" : "plain", G); \
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy4<U, NT>), dim3(G), dim3(256), 0, 0, (const v4f*)x, (v4f*)y, n4); }); }
    COPY(1, false, 2048) COPY(1, false, 4096) COPY(1, false, 262144)
    COPY(2, false, 2048) COPY(4, false, 2048) COPY(8, false, 2048) COPY(4, false, 1024) COPY(8, false, 1024) COPY(8, false, 512)
    COPY(4, false, 65536) COPY(4, true, 2048) COPY(8, true, 2048) COPY(8, true, 1024)
#define COPYC(U, NT, G) { char nm[96]; snprintf(nm, 96, "copy4_chunk U=%d %s grid=%d", U, NT ? "nt" : "plain", G); \
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy4_chunk<U, NT>), dim3(G), dim3(256), 0, 0, (const v4f*)x, (v4f*)y, n4); }); }
    COPYC(4, false, 2048) COPYC(8, false, 2048) COPYC(8, true, 2048) COPYC(8, false, 16384)
    timeit("read4 U=1 grid=2048", n * 4.0, [&] { hipLaunchKernelGGL((read4<1>), dim3(2048), dim3(256), 0, 0, (const v4f*)x, out, n4); });
    timeit("read4 U=4 grid=2048", n * 4.0, [&] { hipLaunchKernelGGL((read4<4>), dim3(2048), dim3(256), 0, 0, (const v4f*)x, out, n4); });
    timeit("read4 U=8 grid=2048", n * 4.0, [&] { hipLaunchKernelGGL((read4<8>), dim3(2048), dim3(256), 0, 0, (const v4f*)x, out, n4); });
    timeit("write4 U=1 grid=2048", n * 4.0, [&] { hipLaunchKernelGGL((write4<1>), dim3(2048), dim3(256), 0, 0, (v4f*)y, n4); });
    timeit("write4 U=4 grid=2048", n * 4.0, [&] { hipLaunchKernelGGL((write4<4>), dim3(2048), dim3(256), 0, 0, (v4f*)y, n4); });

    // part 2: (B, 256, P) tensors, B = 32 -> 1.06 GB each (x = residual, z = a1, y = out)
    {
        const int C = 256, P = 251 * 129, B = 32;
        const double rb = 3.0 * B * C * (double)P * 4;
#define ROWS(NT_, WGPC, QR, DEPTH, MF, NTS) { \
            const int tps = (P + NT_ - 1) / NT_, nt = tps * B; \
            const size_t lds = WGPC == 1 ? 100 * 1024 : (WGPC == 2 ? 70 * 1024 : 36 * 1024); \
            hipFuncSetAttribute((const void*)rows_pipe<NT_, QR, DEPTH, MF, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            char nm[128]; snprintf(nm, 128, "rows thr=%d wg/cu=%d QR=%d depth=%d mfma/stage=%d %s", NT_, WGPC, QR, DEPTH, MF, NTS ? "nt-store" : ""); \
            timeit(nm, rb, [&] { hipLaunchKernelGGL((rows_pipe<NT_, QR, DEPTH, MF, NTS>), dim3(256 * WGPC), dim3(NT_), lds, 0, x, z, y, C, P, nt, tps, out); }); }
        // the b2b kernel today: 8 waves per CU, 4+4 loads in flight per stage, nothing ahead (depth 0 is not expressible: depth 1 with QR 4 ~ today)
        ROWS(512, 1, 4, 1, 0, false)
        ROWS(512, 1, 4, 1, 12, false)
        ROWS(512, 1, 8, 1, 0, false)
        ROWS(512, 1, 8, 1, 24, false)
        ROWS(512, 1, 8, 2, 24, false)
        ROWS(512, 1, 16, 1, 48, false)
        ROWS(512, 1, 16, 1, 48, true)
        ROWS(256, 1, 16, 1, 48, false)
        ROWS(256, 1, 16, 2, 48, false)
        ROWS(256, 1, 32, 1, 96, false)
        ROWS(256, 2, 8, 1, 24, false)
        ROWS(256, 2, 16, 1, 48, false)
        ROWS(256, 2, 16, 1, 0, false)
        ROWS(256, 2, 16, 1, 48, true)
        ROWS(256, 4, 8, 1, 24, false)
        ROWS(256, 4, 8, 2, 24, false)
        ROWS(256, 4, 4, 2, 12, false)
        ROWS(1024, 1, 4, 1, 12, false)
        ROWS(1024, 1, 8, 1, 24, false)
    }
    return 0;
}
