// Diagnostic micro-benchmark (not part of the product), round 3: the row-walking access pattern of the depthwise kernels (k_dw.hip) against
// the flat pass of bench_mall.hip.  A wave owns one channel plane's band of TH rows (516-byte rows, 4-byte aligned, 8 bytes per lane) and walks it
// with RQ rows in flight, re-requesting rows either one per trip (BS = 1, what dw1p_body does) or in bursts of BS consecutive rows every BS trips
// (does the DRAM side reward 2-4 KB contiguous requests from one stream over 512-byte ones from 4096 interleaved streams?).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_brows tools/bench_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

// planes: NP planes of H rows x W floats (contiguous rows, plane pitch PS floats).  grid = NP / 4 * bands (XCD-contiguous), 4 waves = 4 planes
template <int RQ, int BS, bool WR>
__global__ __launch_bounds__(256, 4) void walk(const float* __restrict__ x, float* __restrict__ y, int H, int W, int PS, int TH, int nbands, int nblk, float* sink) {
    int id = blockIdx.x;
    {
        const int n8 = gridDim.x >> 3;
        id = (id & 7) * n8 + (id >> 3);
    }
    if (id >= nblk) return;
    const int band = id % nbands, pg = id / nbands;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int plane = pg * 4 + wave;
    const int r0 = band * TH, r1 = min(r0 + TH, H);
    const float* xp = x + (size_t)plane * PS + 2 * lane;
    float* yp = y + (size_t)plane * PS + 2 * lane;
    f32x2u q[RQ];
    auto ld = [&](int t) { return *reinterpret_cast<const f32x2u*>(xp + (size_t)min(t, H - 1) * W); };
#pragma unroll
    for (int k = 0; k < RQ; ++k) q[k] = ld(r0 + k);
    float acc = 0.f;
    for (int t = r0; t < r1; t += RQ) {
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
            const f32x2u v = q[k];
            if (BS == 1) q[k] = ld(t + RQ + k);
            f32x2u o = v * 1.5f + 1.0f;
            // some dependent work per row, as the depthwise rows have (~40 VALU instructions)
#pragma unroll
            for (int j = 0; j < 12; ++j) o = o * 1.0001f + 0.5f;
            if (t + k < r1) {
                if (WR) *reinterpret_cast<f32x2u*>(yp + (size_t)(t + k) * W) = o;
                else acc += o.x + o.y;
            }
            if (BS > 1 && (k % BS) == BS - 1) {
#pragma unroll
                for (int j = 0; j < BS; ++j) q[k - BS + 1 + j] = ld(t + RQ + k - BS + 1 + j);
            }
        }
    }
    if (!WR && acc == 123.456f) *sink = acc;
}


// The same walk with a wave = one whole 129-column row (64 lanes x 2 columns + column 128 as an extra dword of lane 63) and the OUTPUT staged
// through a per-wave LDS ring: rows are appended at their byte position in the plane (516 bytes each), and every completed 512-byte-ALIGNED
// chunk is stored with one 8-byte-per-lane instruction - every store instruction writes four whole 128-byte lines.
template <int RQ, int FEAT = 0>
__global__ __launch_bounds__(256, 4) void walk_staged(const float* __restrict__ x, float* __restrict__ y, int H, int PS, int TH, int nbands, int nblk, const float* __restrict__ wts = nullptr, double* __restrict__ stats = nullptr) {
    constexpr int W = 129, RING = 2048;
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[4][RING];
    int id = blockIdx.x;
    {
        const int n8 = gridDim.x >> 3;
        id = (id & 7) * n8 + (id >> 3);
    }
    if (id >= nblk) return;
    const int band = id % nbands, pg = id / nbands;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* ring = ring_all[wave];
    const int plane = pg * 4 + wave;
    const int r0 = band * TH, r1 = min(r0 + TH, H);
    const float* xp = x + (size_t)plane * PS + 2 * lane;
    const float* xe = x + (size_t)plane * PS + 128;
    char* yb = reinterpret_cast<char*>(y + (size_t)plane * PS);
    f32x2u q[RQ];
    float qe[RQ];
    auto ld = [&](int t) { return *reinterpret_cast<const f32x2u*>(xp + (size_t)min(max(t, 0), H - 1) * W); };
    auto lde = [&](int t) { return xe[(size_t)min(max(t, 0), H - 1) * W]; };
    f32x2u hsum = {0.f, 0.f};
    if (FEAT & 1) {  // the three halo rows a 4x4 window needs around the band
        const f32x2u h0 = ld(r0 - 1), h1 = ld(r1), h2 = ld(r1 + 1);
        hsum = h0 + h1 + h2;
    }
#pragma unroll
    for (int k = 0; k < RQ; ++k) { q[k] = ld(r0 + k); qe[k] = lde(r0 + k); }
    float wsum = 0.f;
    if (FEAT & 4) {  // per-workgroup set-up: weight loads + ~100 dependent instructions
        const int c = plane & 63;
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = wts[c * 16 + i];
#pragma unroll
        for (int rep = 0; rep < 6; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) wsum = fmaf(w[i], wsum + (float)lane, w[(i + rep) & 15]);
    }
    float st = 0.f;
    const unsigned s0 = (unsigned)r0 * 516u, s1 = (unsigned)r1 * 516u;
    unsigned chunk = s0 >> 9;  // next 512-byte chunk to flush
    auto flush = [&](unsigned c) {  // chunk c is complete in the ring (or is the band's last, partial one)
        const unsigned cb = c << 9;
        const unsigned la = (cb & (RING - 1)) + 8u * lane;
        const f32x2u v = *reinterpret_cast<const f32x2u*>(ring + la);
        const unsigned b0 = cb + 8u * lane;
        if (cb >= s0 && cb + 512u <= s1) *reinterpret_cast<f32x2u*>(yb + b0) = v;  // uniform: the whole chunk is this band's
        else {
            if (b0 >= s0 && b0 + 4 <= s1) *reinterpret_cast<float*>(yb + b0) = v.x;
            if (b0 + 4 >= s0 && b0 + 8 <= s1) *reinterpret_cast<float*>(yb + b0 + 4) = v.y;
        }
    };
    for (int t = r0; t < r1; t += RQ) {
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
            const f32x2u v = q[k];
            const float ve = qe[k];
            q[k] = ld(t + RQ + k);
            qe[k] = lde(t + RQ + k);
            f32x2u o = (FEAT & 16) ? v : v * 1.5f + 1.0f;
            float oe = (FEAT & 16) ? ve : ve * 1.5f + 1.0f;
            if (FEAT & 16) {  // values preserved (random data stays random over the ping-pong); the same number of dependent operations
                f32x2u tt = o;
#pragma unroll
                for (int j = 0; j < 12; ++j) tt = tt * 1.0001f + 0.5f;
                if (tt.x == 123.456f) o.x = tt.y;
            } else {
#pragma unroll
                for (int j = 0; j < ((FEAT & 2) ? 56 : 12); ++j) o = o * 1.0001f + 0.5f;
            }
            if (FEAT & 1) o += hsum * 1e-30f;
            if (FEAT & 4) o += wsum * 1e-30f;
            if (FEAT & 8) st += o.x + o.y;
            if (t + k < r1) {
                const unsigned pos = (unsigned)(t + k) * 516u;
                *reinterpret_cast<float*>(ring + ((pos + 8u * lane) & (RING - 1))) = o.x;
                *reinterpret_cast<float*>(ring + ((pos + 8u * lane + 4u) & (RING - 1))) = o.y;
                if (lane == 63) *reinterpret_cast<float*>(ring + ((pos + 512u) & (RING - 1))) = oe;
                while (((chunk + 1) << 9) <= pos + 516u) { flush(chunk); ++chunk; }  // uniform
            }
        }
    }
    if ((chunk << 9) < s1) flush(chunk);
    if (FEAT & 8) {  // per-workgroup statistics: LDS fold + two f64 atomics onto the sample's pair
        __shared__ double red[8];
        double d = (double)st;
        for (int o = 32; o; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) red[wave] = d;
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(stats + 2 * (plane >> 6), red[0] + red[1] + red[2] + red[3]);
            atomicAdd(stats + 2 * (plane >> 6) + 1, red[0] * red[1]);
        }
    }
}

template <int FEAT>
int run_staged(const float* x, float* y, int NP, int H, int PS, int TH, const float* wts, double* stats) {
    const int nbands = (H + TH - 1) / TH, nblk = NP / 4 * nbands;
    const unsigned grid = (unsigned)((nblk + 7) / 8 * 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((walk_staged<8, FEAT>), dim3(grid), dim3(256), 0, 0, (it & 1) ? y : x, (it & 1) ? const_cast<float*>(x) : y, H, PS, TH, nbands, nblk, wts, stats);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)NP * H * 516 * 2;
        if (rep) printf("rw  staged, features %2d      TH %3d: %.1f us = %.2f TB/s\n", FEAT, TH, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}

template <int RQ, int BS, bool WR>
int run(const char* name, const float* x, float* y, int NP, int H, int W, int PS, int TH, float* sink) {
    const int nbands = (H + TH - 1) / TH, nblk = NP / 4 * nbands;
    const unsigned grid = (unsigned)((nblk + 7) / 8 * 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL((walk<RQ, BS, WR>), dim3(grid), dim3(256), 0, 0, (it & 1) ? y : x, (it & 1) ? const_cast<float*>(x) : y, H, W, PS, TH, nbands, nblk, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)NP * H * 512 * (WR ? 2 : 1);
        if (rep) printf("%-28s TH %3d: %.1f us = %.2f TB/s\n", name, TH, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int NP = 2048, H = 251, W = argc > 1 ? atoi(argv[1]) : 129, PS = argc > 2 ? atoi(argv[2]) : 32384;
    printf("W = %d, plane pitch %d\n", W, PS);
    float *x, *y, *sink;
    CK(hipMalloc(&x, (size_t)NP * PS * 4 + 4096));
    CK(hipMalloc(&y, (size_t)NP * PS * 4 + 4096));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, (size_t)NP * PS * 4));
    CK(hipMemset(y, 0, (size_t)NP * PS * 4));
    if (W == 129) {
        // correctness of the staged store: y = f(x) must equal the direct kernel's output
        float* y2;
        CK(hipMalloc(&y2, (size_t)NP * PS * 4 + 4096));
        CK(hipMemset(y2, 0, (size_t)NP * PS * 4));
        std::vector<float> hx((size_t)NP * PS);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        const int TH = 24, nbands = (H + TH - 1) / TH, nblk = NP / 4 * nbands;
        hipLaunchKernelGGL((walk_staged<8>), dim3((nblk + 7) / 8 * 8), dim3(256), 0, 0, x, y2, H, PS, TH, nbands, nblk);
        std::vector<float> h2(hx.size());
        CK(hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int p = 0; p < NP; p += 97)
            for (int i = 0; i < H * W; ++i) {
                float o = hx[(size_t)p * PS + i] * 1.5f + 1.0f;
                if (i % W != 128) for (int j = 0; j < 12; ++j) o = o * 1.0001f + 0.5f;
                if (fabsf(o - h2[(size_t)p * PS + i]) > 1e-4f * fabsf(o)) ++bad;
            }
        printf("staged store check: %zu mismatches\n", bad);
        CK(hipMemset(x, 0, (size_t)NP * PS * 4));
        float* wts;
        double* stats;
        CK(hipMalloc(&wts, 64 * 16 * 4));
        CK(hipMemset(wts, 0, 64 * 16 * 4));
        CK(hipMalloc(&stats, 64 * 16));
        CK(hipMemset(stats, 0, 64 * 16));
        // which of the product kernel's extras costs what (bit 0 halo rows, 1 heavy arithmetic, 2 per-workgroup set-up, 3 statistics atomics)
        if (run_staged<16>(x, y, NP, H, PS, 24, wts, stats)) return 1;  // copy of zeros
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u + 12345u) % 65521u) * 3.1e-5f - 1.0f;
        CK(hipMemcpy(y, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        if (run_staged<16>(x, y, NP, H, PS, 24, wts, stats)) return 1;  // copy of random data
        CK(hipMemset(x, 0, (size_t)NP * PS * 4));
        CK(hipMemset(y, 0, (size_t)NP * PS * 4));
        if (run_staged<0>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<1>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<2>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<4>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<8>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<15>(x, y, NP, H, PS, 24, wts, stats)) return 1;
        if (run_staged<0>(x, y, NP, H, PS, 24, wts, stats)) return 1;
    }
    for (int TH : {24}) {
        if (run<8, 1, true>("rw  RQ 8  one row per trip", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<8, 4, true>("rw  RQ 8  bursts of 4", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<8, 8, true>("rw  RQ 8  bursts of 8", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<16, 1, true>("rw  RQ 16 one row per trip", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<16, 4, true>("rw  RQ 16 bursts of 4", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<16, 8, true>("rw  RQ 16 bursts of 8", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<8, 1, false>("r   RQ 8  one row per trip", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<8, 4, false>("r   RQ 8  bursts of 4", x, y, NP, H, W, PS, TH, sink)) return 1;
        if (run<16, 8, false>("r   RQ 16 bursts of 8", x, y, NP, H, W, PS, TH, sink)) return 1;
    }
    return 0;
}
