// Diagnostic micro-benchmark (not part of the product): streaming-read rates of a (B,64,251,129) fp32 tensor under
// the access patterns of the depthwise kernels.   hipcc --offload-arch=gfx950 -O3 -o /tmp/bs tools/bench_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void v0_flat(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void v1_flat4(const float4* __restrict__ x, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
// thread = (channel, column pair), walks TH rows; U rows of loads issued per trip
template <int U, bool WORK, int EDGE = 0>
__global__ __launch_bounds__(256) void v2_rows(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int TH) {
    const int NP = (W + 1) / 2;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < C * NP;
    const int c = live ? g / NP : 0, p = live ? g - c * NP : 0;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * TH, r1 = min(r0 + TH, H);
    const float* xs = x + (size_t)b * C * H * W;
    const unsigned o0 = (unsigned)c * H * W + 2 * p, o1 = (unsigned)c * H * W + min(2 * p + 1, W - 1);
    f32x2 acc = {0.f, 0.f};
    f32x2 w[16];
    for (int i = 0; i < 16; ++i) w[i] = f32x2{0.01f * i, 0.02f * i + threadIdx.x * 1e-6f};
    const int lane = threadIdx.x & 63;
    const unsigned e0 = (unsigned)c * H * W + (lane == 0 ? max(2 * p - 1, 0) : min(2 * p + 2, W - 1)), e1 = (unsigned)c * H * W + min(2 * p + 3, W - 1);
    float q[U][4];
    auto ld = [&](int t, float (&d)[4]) {
        const float* rp = xs + (size_t)min(t, H - 1) * W;
        d[0] = rp[o0]; d[1] = rp[o1]; d[2] = 0.f; d[3] = 0.f;
        if (EDGE == 1) { if (lane == 0 || lane == 63) d[2] = rp[e0]; if (lane == 63) d[3] = rp[e1]; }
        if (EDGE == 2) { d[2] = rp[lane == 0 || lane == 63 ? e0 : o0]; d[3] = rp[lane == 63 ? e1 : o1]; }  // unmasked: redundant addresses
    };
    for (int k = 0; k < U; ++k) ld(r0 + k, q[k]);
    for (int t = r0; t < r1; t += U) {
        float v[U][4];
        for (int k = 0; k < U; ++k) for (int j = 0; j < 4; ++j) v[k][j] = q[k][j];
        for (int k = 0; k < U; ++k) ld(t + U + k, q[k]);
        for (int k = 0; k < U; ++k) {
            if (WORK) {
                const float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k][1]), 0x138, 0xf, 0xf, false));
                const float r_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k][0]), 0x130, 0xf, 0xf, false));
                const float r2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k][1]), 0x130, 0xf, 0xf, false));
                float vv[5] = {l, v[k][0], v[k][1], r_, r2};
                if (EDGE) { if (lane == 0) vv[0] = v[k][2]; if (lane == 63) { vv[3] = v[k][2]; vv[4] = v[k][3]; } }
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) acc = f32x2{vv[j], vv[j + 1]} * w[i * 4 + j] + acc;
            } else {
                acc += f32x2{v[k][0], v[k][1]};
            }
        }
    }
    if (acc.x + acc.y == 123.456f) out[0] = acc.x;
}

// 16-byte loads from a base that is only 4-byte aligned (row pitch odd): does the hardware take them, at what rate?
__global__ __launch_bounds__(256) void v3_flat4_unaligned(const float* __restrict__ x, float* __restrict__ out, size_t n4) {
    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4 v = *reinterpret_cast<const f4*>(x + 1 + 4 * i);
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void v4_flat2_unaligned(const float* __restrict__ x, float* __restrict__ out, size_t n2) {
    typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const f2 v = *reinterpret_cast<const f2*>(x + 1 + 2 * i);
        s += v.x + v.y;
    }
    if (s == 123.456f) out[0] = s;
}
// correctness of the unaligned vector load
__global__ void v3_check(const float* __restrict__ x, float* __restrict__ out) {
    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
    const f4 v = *reinterpret_cast<const f4*>(x + 1 + 4 * threadIdx.x);
    out[threadIdx.x] = v.x + 10.f * v.y + 100.f * v.z + 1000.f * v.w;
}

__global__ __launch_bounds__(256) void c0_copy(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i] * 1.5f;
}
__global__ __launch_bounds__(256) void c1_copy4(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = x[i]; v.x *= 1.5f; y[i] = v; }
}
__global__ __launch_bounds__(256) void c2_copy2u(const float* __restrict__ x, float* __restrict__ y, size_t n2) {
    typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        f2 v = *reinterpret_cast<const f2*>(x + 1 + 2 * i); v.x *= 1.5f; *reinterpret_cast<f2*>(y + 1 + 2 * i) = v;
    }
}
// the pointwise kernels' pattern: a wave owns 32*S consecutive pixels and walks the channel rows (pitch P, odd) of one sample
template <int S>
__global__ __launch_bounds__(256) void c3_rows(const float* __restrict__ x, float* __restrict__ y, int C, int P, int ntiles, int tps) {
    typedef float fv __attribute__((ext_vector_type(S), aligned(4)));
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tps;
        const int p0 = min((tile - b * tps) * 128 * S + wave * 32 * S + S * r, P - S);
        const float* xs = x + (size_t)b * C * P + p0;
        float* ys = y + (size_t)b * C * P + p0;
#pragma unroll 2
        for (int ks = 0; ks < C; ks += 16) {
            fv v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const fv*>(xs + (size_t)(ks + 8 * h + j) * P);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] *= 1.5f; *reinterpret_cast<fv*>(ys + (size_t)(ks + 8 * h + j) * P) = v[j]; }
        }
    }
}

int main() {
    const int SCALE = 4;
    const int B = 32, C = 64, H = 251, W = 129;
    const size_t n = (size_t)B * C * H * W;
    const size_t nbig = n * SCALE;
    float *x, *out;
    CK(hipMalloc(&x, nbig * 4 + 64));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, nbig * 4 + 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int R = 20;
        for (int i = 0; i < R; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.1f us  %6.2f TB/s\n", name, ms / R * 1e3, n * 4.0 / (ms / R * 1e-3) / 1e12);
    };
    timeit("flat dword, 2048 WGs", [&] { hipLaunchKernelGGL(v0_flat, dim3(2048), dim3(256), 0, 0, x, out, n); });
    timeit("flat dword, 8192 WGs", [&] { hipLaunchKernelGGL(v0_flat, dim3(8192), dim3(256), 0, 0, x, out, n); });
    timeit("flat dwordx4, 2048 WGs", [&] { hipLaunchKernelGGL(v1_flat4, dim3(2048), dim3(256), 0, 0, (const float4*)x, out, n / 4); });
    timeit("flat dwordx4 +4B misaligned, 2048 WGs", [&] { hipLaunchKernelGGL(v3_flat4_unaligned, dim3(2048), dim3(256), 0, 0, x, out, n / 4 - 1); });
    timeit("flat dwordx2 +4B misaligned, 2048 WGs", [&] { hipLaunchKernelGGL(v4_flat2_unaligned, dim3(2048), dim3(256), 0, 0, x, out, n / 2 - 1); });
    {
        std::vector<float> hx(1024);
        for (int i = 0; i < 1024; ++i) hx[i] = (float)(i % 7);
        hipMemcpy(x, hx.data(), 4096, hipMemcpyHostToDevice);
        float* o2; hipMalloc(&o2, 256);
        hipLaunchKernelGGL(v3_check, dim3(1), dim3(64), 0, 0, x, o2);
        float ho[64]; hipMemcpy(ho, o2, 256, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) { const float e = hx[1 + 4 * t] + 10.f * hx[2 + 4 * t] + 100.f * hx[3 + 4 * t] + 1000.f * hx[4 + 4 * t]; bad += ho[t] != e; }
        printf("unaligned dwordx4 load check: %s\n", bad ? "WRONG" : "ok");
    }
    float* y;
    CK(hipMalloc(&y, nbig * 4 + 64));
    size_t nn = n;
    auto timeit2 = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int R = 20;
        for (int i = 0; i < R; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.1f us  %6.2f TB/s (read + write)\n", name, ms / R * 1e3, 2 * nn * 4.0 / (ms / R * 1e-3) / 1e12);
    };
    timeit2("copy dword", [&] { hipLaunchKernelGGL(c0_copy, dim3(4096), dim3(256), 0, 0, x, y, n); });
    timeit2("copy dwordx4 aligned", [&] { hipLaunchKernelGGL(c1_copy4, dim3(4096), dim3(256), 0, 0, (const float4*)x, (float4*)y, n / 4); });
    timeit2("copy dwordx2 +4B misaligned", [&] { hipLaunchKernelGGL(c2_copy2u, dim3(4096), dim3(256), 0, 0, x, y, n / 2 - 1); });
    nn = nbig;
    timeit2("copy dword 1 GB", [&] { hipLaunchKernelGGL(c0_copy, dim3(4096), dim3(256), 0, 0, x, y, nbig); });
    timeit2("copy dwordx4 aligned 1 GB", [&] { hipLaunchKernelGGL(c1_copy4, dim3(4096), dim3(256), 0, 0, (const float4*)x, (float4*)y, nbig / 4); });
    {
        const int Cc = 64 * 4, P = 251 * 129, Bc = 8 * SCALE;  // (8, 256, P): same bytes as (32, 64, P); SCALE 4 = the real tensors
        int tps = (P + 127) / 128, nt = tps * Bc;
        timeit2("rows copy S=1 (256 ch rows, pitch odd)", [&] { hipLaunchKernelGGL((c3_rows<1>), dim3(512), dim3(256), 0, 0, x, y, Cc, P, nt, tps); });
        tps = (P + 255) / 256; nt = tps * Bc;
        timeit2("rows copy S=2", [&] { hipLaunchKernelGGL((c3_rows<2>), dim3(512), dim3(256), 0, 0, x, y, Cc, P, nt, tps); });
        tps = (P + 511) / 512; nt = tps * Bc;
        timeit2("rows copy S=4", [&] { hipLaunchKernelGGL((c3_rows<4>), dim3(512), dim3(256), 0, 0, x, y, Cc, P, nt, tps); });
    }
    const int NP = (W + 1) / 2, gx = (C * NP + 255) / 256;
    for (int TH : {64}) {
        char nm[64];
        const dim3 grid(gx, (H + TH - 1) / TH, B);
        snprintf(nm, 64, "rows U=2 sum TH=%d", TH);  timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<2, false>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=4 sum TH=%d", TH);  timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<4, false>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=8 sum TH=%d", TH);  timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<8, false>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=4 conv TH=%d", TH); timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<4, true>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=8 conv TH=%d", TH); timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<8, true>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=4 conv edge-masked TH=%d", TH); timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<4, true, 1>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
        snprintf(nm, 64, "rows U=4 conv edge-unmasked TH=%d", TH); timeit(nm, [&] { hipLaunchKernelGGL((v2_rows<4, true, 2>), grid, dim3(256), 0, 0, x, out, C, H, W, TH); });
    }
    return 0;
}
