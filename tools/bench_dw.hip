// Diagnostic harness (not part of the product): times the depthwise kernels of rtfs-net_amd/csrc/k_dw.hip in isolation at the
// bench shape (B=32, 64 ch, 251 x 129 full resolution, 125 x 64 low resolution).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irtfs-net_amd/csrc -o tools/_bdw tools/bench_dw.hip
#include "../rtfs-net_amd/csrc/k_dw.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int C = 64, H = 251, W = 129, Hg = 125, Wg = 64;
    const int TH = argc > 1 ? atoi(argv[1]) : 64;
    const int B = argc > 2 ? atoi(argv[2]) : 32;
    const size_t n = (size_t)B * C * H * W, ng = (size_t)B * C * Hg * Wg;
    float *x, *y, *add, *gate, *emb, *par;
    double* st;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&add, n * 4));
    CK(hipMalloc(&gate, ng * 4)); CK(hipMalloc(&emb, ng * 4)); CK(hipMalloc(&par, 64 * 64 * 4)); CK(hipMalloc(&st, 64 * 8 * 8));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(add, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gate, h.data(), ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(emb, h.data(), ng * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(par, h.data(), 64 * 64 * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(64 * 8);
    for (int i = 0; i < 64 * 8; i += 2) { hs[i] = 10.0; hs[i + 1] = 1e6; }
    CK(hipMemcpy(st, hs.data(), 64 * 8 * 8, hipMemcpyHostToDevice));
    DwArgs a;
    a.x = x; a.w[0] = par; a.w[1] = par + 1024; a.bias[0] = nullptr; a.out[0] = y; a.out[1] = y;
    a.stats_out[0] = st + 256; a.stats_out[1] = st + 320;
    a.in_stats = st; a.in_inv_count = 1.0 / ((double)C * H * W); a.in_gamma = par + 2048; a.in_beta = par + 2112;
    a.C = C; a.H = H; a.W = W; a.TH = TH; a.Hg = Hg; a.Wg = Wg;
    a.loc_stats = st; a.loc_inv_count = a.in_inv_count; a.loc_gamma = par + 2048; a.loc_beta = par + 2112;
    a.gate = gate; a.gate_stats = st; a.gate_gamma = par + 2048; a.gate_beta = par + 2112;
    a.emb = emb; a.emb_stats = st; a.emb_gamma = par + 2048; a.emb_beta = par + 2112; a.g_inv_count = 1.0 / ((double)C * Hg * Wg);
    a.add_stats = st; a.add_inv_count = a.in_inv_count; a.add_gamma = par + 2048; a.add_beta = par + 2112;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timeit = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0);
        const int R = 20;
        for (int i = 0; i < R; ++i) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("B=%2d TH=%3d %-34s %8.1f us  %6.2f TB/s\n", B, TH, name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12);
    };
    const double T = n * 4.0;
    timeit("stats   <1,false,1>", T, [&] { launch_dw_s1(a, 1, false, 1, B, 0); });
    timeit("stats   <1,true,1>", T, [&] { launch_dw_s1(a, 1, true, 1, B, 0); });
    timeit("conv    <1,false,0>", 2 * T, [&] { launch_dw_s1(a, 1, false, 0, B, 0); });
    a.addend = nullptr;
    timeit("apply   <1,true,2>", 2 * T, [&] { launch_dw_s1(a, 1, true, 2, B, 0); });
    a.addend = add;
    timeit("apply+  <1,false,2>", 3 * T, [&] { launch_dw_s1(a, 1, false, 2, B, 0); });
    DwArgs s = a;
    s.out[0] = y; s.out[1] = y + ng; s.bias[0] = par + 3000;
    timeit("s2+pool", T + 2.0 * ng * 4, [&] { launch_dw_s2_pool(s, B, 0); });
    return 0;
}
