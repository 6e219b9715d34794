// Diagnostic micro-benchmark (not part of the product), round 3, fifth pass: does the row PITCH matter (TLB reach / DRAM page locality)?
// Same row-walk kernels as bench_stream5.hip, 256 rows per tile, 2 reads + 1 write, ~1 GB per tensor, row pitch 8 KB ... 512 KB.
// (fourth pass follows)  bench_stream4: row-pattern READS run at 6.0-6.4 TB/s,
// row-pattern WRITES at 3.4-4.2, a flat aligned write at 4.6, and read + write times ADD (the DRAM bus is half duplex).  What do writes
// want?  Suspects: partial 128-byte lines at segment / tile edges (rows start at odd multiples of 4 bytes: pitch P = 251*129), lines
// shared by workgroups on different XCDs (L2s cannot merge them), nontemporal hints.
//   PITCH: row pitch in floats (32379 = real, 32384 = padded to 128 bytes);  XCD: adjacent tiles of a row on one XCD
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bs5 tools/bench_stream5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int S> struct VT { typedef float type __attribute__((ext_vector_type(S), aligned(4))); };
template <> struct VT<1> { typedef float type; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool NT>
__global__ __launch_bounds__(256) void flat_write(float* __restrict__ y, size_t n4, int off, int xcd) {
    size_t b = blockIdx.x;
    if (xcd) b = (size_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const size_t i = b * 256 + threadIdx.x;
    if (i < n4) {
        v4u* p = reinterpret_cast<v4u*>(y + off + 4 * i);
        if (NT) __builtin_nontemporal_store(v4u{1.f, 2.f, 3.f, 4.f}, p); else *p = v4u{1.f, 2.f, 3.f, 4.f};
    }
}

template <int NT_, int S, bool SPLIT, int NRD, bool WR, int QR, bool XCD, int NTS>
__global__ __launch_bounds__(NT_) void rows(const float* __restrict__ R, const float* __restrict__ A, float* __restrict__ OUT, int C, int P, int PITCH,
                                            int ntiles, int tps, float* __restrict__ sink) {
    typedef typename VT<S>::type vt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = SPLIT ? lane & 31 : lane, h = SPLIT ? lane >> 5 : 0;
    constexpr int WPX = (SPLIT ? 32 : 64) * S;
    constexpr int RS = SPLIT ? 2 : 1;
    const int NS = C / (RS * QR);
    float accum = 0.f;
    int tile = blockIdx.x;
    if (XCD) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // grid is a multiple of 8 (padded); tiles >= ntiles idle
    if (tile >= ntiles) return;
    const int b = tile / tps;
    const int p0 = min((tile - b * tps) * (NT_ / 64 * WPX) + wave * WPX + S * r_, P - S);
    const size_t base = (size_t)b * C * PITCH + p0 + (size_t)h * PITCH;
    const float* __restrict__ rs = R + base;
    const float* __restrict__ as = A + base;
    float* __restrict__ os = OUT + base;
    for (int s = 0; s < NS; ++s) {
        vt r[QR], a[QR];
#pragma unroll
        for (int j = 0; j < QR; ++j) {
            const unsigned o = (unsigned)((s * QR + j) * RS * PITCH);
            if (NRD >= 1) r[j] = *reinterpret_cast<const vt*>(rs + o); else r[j] = vt(1.0f);
            if (NRD >= 2) a[j] = *reinterpret_cast<const vt*>(as + o); else a[j] = vt(2.0f);
        }
#pragma unroll
        for (int j = 0; j < QR; ++j) {
            const unsigned o = (unsigned)((s * QR + j) * RS * PITCH);
            const vt yv = r[j] * 1.5f + a[j];
            if (WR) { if (NTS) __builtin_nontemporal_store(yv, reinterpret_cast<vt*>(os + o)); else *reinterpret_cast<vt*>(os + o) = yv; }
            else accum += reinterpret_cast<const float*>(&yv)[0];
        }
    }
    if (accum == 123.456f) sink[0] = accum;
}

int main() {
    const size_t n = (size_t)256 * 1024 * 1024 + (4 << 20);
    float *x, *y, *z, *out;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&y, n * 4));
    CK(hipMalloc(&z, n * 4));
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
        printf("%-86s %8.1f us  %6.2f TB/s %s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
        fflush(stdout);
    };
    const int C = 256;
    for (int P : {2048, 8192, 32384, 131072}) {
        const int B = (int)(((size_t)256 * 1024 * 1024) / ((size_t)C * P));
        const double tb = (double)B * C * P * 4;
#define ROWS(NT_, S, SPLIT, NRD, WR, QR, XCD, NTS, PITCH) { \
        const int tpx = NT_ / 64 * (SPLIT ? 32 : 64) * S; const int tps = (P + tpx - 1) / tpx, nt = tps * B; \
        const int g = (nt + 7) / 8 * 8; \
        char nm[160]; snprintf(nm, 160, "P=%d B=%d thr=%d S=%d %s R=%d W=%d QR=%d (%d B/row/tile)", P, B, NT_, S, SPLIT ? "2rows" : "1row ", NRD, WR, QR, tpx * 4); \
        timeit(nm, tb * (NRD + WR), [&] { hipLaunchKernelGGL((rows<NT_, S, SPLIT, NRD, WR, QR, XCD, NTS>), dim3(g), dim3(NT_), 0, 0, x, z, y, C, P, PITCH, nt, tps, out); }); }
        ROWS(256, 2, true, 2, true, 8, false, 0, P)
        ROWS(256, 2, true, 1, false, 8, false, 0, P)
        ROWS(256, 2, true, 0, true, 8, false, 0, P)
        ROWS(256, 4, false, 2, true, 8, false, 0, P)
#undef ROWS
    }
    return 0;
}
