// Diagnostic micro-benchmark (not part of the product), round 3, third pass: the row-walk pattern tops out at ~4.7 TB/s whatever the
// grid shape, occupancy or prefetch depth (bench_stream3) while a flat one-shot copy runs at 6.3.  Which property of the pattern costs it?
//   streams: read-only / two reads / write-only / copy / two reads + write
//   segment shape of one wave-instruction: S pixels per lane, SPLIT = lane halves in two different rows (the MFMA B-fragment order)
//   TPB: tile pixels per workgroup
// Tensors (B = 32, C = 256, P = 251*129) f32; a tile walks all C rows of its pixels.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bs4 tools/bench_stream4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int S> struct VT { typedef float type __attribute__((ext_vector_type(S), aligned(4))); };
template <> struct VT<1> { typedef float type; };

// NRD read streams (0, 1, 2), WR: write stream.  QR row-steps per stage (each lane: QR registers of S floats per stream).
template <int NT_, int S, bool SPLIT, int NRD, bool WR, int QR, int MODE>
__global__ __launch_bounds__(NT_) void rows(const float* __restrict__ R, const float* __restrict__ A, float* __restrict__ OUT, int C, int P, int ntiles,
                                            int tps, float* __restrict__ sink) {
    typedef typename VT<S>::type vt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_ = SPLIT ? lane & 31 : lane, h = SPLIT ? lane >> 5 : 0;
    constexpr int WPX = (SPLIT ? 32 : 64) * S;   // pixels per wave
    constexpr int RS = SPLIT ? 2 : 1;            // rows per wave-instruction
    const int NS = C / (RS * QR);
    float accum = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += (MODE == 0 ? ntiles : gridDim.x)) {
        const int b = tile / tps;
        const int p0 = min((tile - b * tps) * (NT_ / 64 * WPX) + wave * WPX + S * r_, P - S);
        const size_t base = (size_t)b * C * P + p0 + (size_t)h * P;
        const float* __restrict__ rs = R + base;
        const float* __restrict__ as = A + base;
        float* __restrict__ os = OUT + base;
        for (int s = 0; s < NS; ++s) {
            vt r[QR], a[QR];
#pragma unroll
            for (int j = 0; j < QR; ++j) {
                const unsigned o = (unsigned)((s * QR + j) * RS * P);
                if (NRD >= 1) r[j] = *reinterpret_cast<const vt*>(rs + o); else r[j] = vt(1.0f);
                if (NRD >= 2) a[j] = *reinterpret_cast<const vt*>(as + o); else a[j] = vt(2.0f);
            }
#pragma unroll
            for (int j = 0; j < QR; ++j) {
                const unsigned o = (unsigned)((s * QR + j) * RS * P);
                const vt yv = r[j] * 1.5f + a[j];
                if (WR) *reinterpret_cast<vt*>(os + o) = yv;
                else if (S == 1) accum += *reinterpret_cast<const float*>(&yv); else accum += reinterpret_cast<const float*>(&yv)[0] + reinterpret_cast<const float*>(&yv)[S - 1];
            }
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
        printf("%-78s %8.1f us  %6.2f TB/s %s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
        fflush(stdout);
    };
    const int C = 256, P = 251 * 129, B = 32;
    const double tb = (double)B * C * P * 4;
#define ROWS(NT_, S, SPLIT, NRD, WR, QR, MODE) { \
        const int tpx = NT_ / 64 * (SPLIT ? 32 : 64) * S; const int tps = (P + tpx - 1) / tpx, nt = tps * B; \
        const int g = MODE == 0 ? nt : 2048; \
        char nm[128]; snprintf(nm, 128, "thr=%d S=%d %s reads=%d write=%d QR=%d %s (tile %d px = %d B/row)", NT_, S, SPLIT ? "2 rows/instr" : "1 row/instr ", NRD, WR, QR, MODE == 0 ? "one-shot" : "static", tpx, tpx * 4); \
        timeit(nm, tb * (NRD + WR), [&] { hipLaunchKernelGGL((rows<NT_, S, SPLIT, NRD, WR, QR, MODE>), dim3(g), dim3(NT_), 0, 0, x, z, y, C, P, nt, tps, out); }); }
    // streams, real shape (S = 2, split)
    ROWS(256, 2, true, 1, false, 8, 0)
    ROWS(256, 2, true, 2, false, 8, 0)
    ROWS(256, 2, true, 0, true, 8, 0)
    ROWS(256, 2, true, 1, true, 8, 0)
    ROWS(256, 2, true, 2, true, 8, 0)
    // segment shapes, copy (1 read + 1 write) and 2 reads + write
    ROWS(256, 1, true, 1, true, 8, 0)
    ROWS(256, 1, false, 1, true, 8, 0)
    ROWS(256, 2, false, 1, true, 8, 0)
    ROWS(256, 4, true, 1, true, 8, 0)
    ROWS(256, 4, false, 1, true, 8, 0)
    ROWS(256, 4, false, 1, true, 4, 0)
    ROWS(256, 2, false, 2, true, 8, 0)
    ROWS(256, 4, true, 2, true, 8, 0)
    ROWS(256, 4, false, 2, true, 8, 0)
    ROWS(256, 4, false, 2, true, 4, 0)
    ROWS(128, 4, false, 2, true, 8, 0)
    ROWS(64, 4, false, 2, true, 8, 0)
    ROWS(512, 4, false, 2, true, 8, 0)
    ROWS(1024, 4, false, 2, true, 4, 0)
    ROWS(256, 4, false, 2, true, 8, 1)
    // read-only shapes
    ROWS(256, 4, false, 1, false, 8, 0)
    ROWS(256, 4, false, 2, false, 8, 0)
    ROWS(256, 4, false, 0, true, 8, 0)
    return 0;
}
