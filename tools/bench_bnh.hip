// Diagnostic harness (not part of the product): the fused bottleneck + block head kernel of rtfs-net_amd/csrc/k_bnh.hip alone at the bench
// shape (B=32, P=32379, padded rows), timed, and - built with -DBNH_STAMP - with s_memtime stamps at the phase boundaries of workgroup 0's first tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DBNH_STAMP] -Irtfs-net_amd/csrc -o tools/_bbnh tools/bench_bnh.hip rtfs-net_amd/csrc/runtime.hip
#include "../rtfs-net_amd/csrc/k_stft.hip"
#include "../rtfs-net_amd/csrc/k_bnh.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int P = 32379, cs = (P + 63) / 64 * 64;
    const int T = 251, F = 129;
    float *a0, *a1, *res, *xe, *par, *wenc;
    void* encimg;
    double* st;
    unsigned* ctr;
    CK(hipMalloc(&a0, (size_t)B * 2 * P * 4 + 64)); CK(hipMalloc(&wenc, 256 * 18 * 4)); CK(hipMalloc(&encimg, 32768)); CK(hipMalloc(&a1, (size_t)B * 256 * cs * 4)); CK(hipMalloc(&res, (size_t)B * 256 * cs * 4));
    CK(hipMalloc(&xe, (size_t)B * 64 * cs * 4)); CK(hipMalloc(&par, 1 << 20)); CK(hipMalloc(&st, B * 16)); CK(hipMalloc(&ctr, 256));
    {
        std::vector<float> h((size_t)64 * cs);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
        for (int i = 0; i < B; ++i) CK(hipMemcpy(a0 + (size_t)i * 2 * P, h.data(), (size_t)2 * P * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(wenc, h.data(), 256 * 18 * 4, hipMemcpyHostToDevice));
        CK(hipMemset(par, 0, 1 << 20));  // all-zero weight images: the timing does not depend on the values
        CK(hipMemset(st, 0, B * 16));
        if (launch_enc_stats(a0, wenc, st, encimg, EncPadJobs(), B, T, F, 0)) return 1;
    }
    BnHeadArgs a;
    a.spec = a0; a.enc_img = encimg; a.T = T; a.F = F; a.a1 = a1; a.res = res; a.xenc = xe;
    a.stats = st; a.inv_count = 1.0 / (256.0 * P); a.gamma = par + 16384; a.beta = par + 16384;
    a.w16 = par + 32768; a.bias = par + 16384; a.gw = par + 16384; a.gb = par + 16384; a.slope = par + 16384; a.w2_16 = par; a.bp = par + 16384;
    a.P = P; a.cs = cs; a.tile_ctr = ctr;
    if (argc > 2) a.stagger = atoi(argv[2]);
    if (argc > 3) a.res = nullptr;  // the product's form since the first boundary forms residual_0 itself
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    {  // the statistics pass alone
        float bst = 1e9f;
        for (int i = 0; i < 8; ++i) {
            CK(hipMemsetAsync(st, 0, B * 16, 0));
            (void)hipEventRecord(e0);
            if (launch_enc_stats(a0, wenc, st, encimg, EncPadJobs(), B, T, F, 0)) return 1;
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            bst = ms < bst ? ms : bst;
        }
        printf("B=%d enc_stats best %.1f us\n", B, bst * 1e3);
    }
    float best = 1e9f, sum = 0;
    const int R = 12;
    for (int i = 0; i < 3 + R; ++i) {
        CK(hipMemsetAsync(ctr, 0, 256, 0));
        (void)hipEventRecord(e0);
        if (launch_bn_head(a, B, 0) != RTFS_OK) { printf("launch failed\n"); return 1; }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)B * (256 + (a.res ? 256 : 0) + 64) * cs * 4;
    printf("B=%d bn_head  avg %.1f us  best %.1f us  %.2f TB/s (avg)\n", B, sum / R * 1e3, best * 1e3, bytes / (sum / R * 1e-3) / 1e12);
#ifdef BNH_STAMP
    unsigned h[16 * 32];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(bnh_stamps), sizeof(h)));
    printf("cycles per phase, workgroup 0 (tile: start->first chunk | chunks 0..7 | output tiles 0..7 | x_enc store | barrier | total)\n");
    for (int t = 0; t < 12; ++t) {
        const unsigned* s = h + t * 32;
        printf("tile %2d: x %6u |", t, s[1] - s[0]);
        for (int k = 0; k < 8; ++k) printf(" %5u", s[2 + k] - s[1 + k]);
        printf(" |");
        for (int g = 0; g < 8; ++g) printf(" %5u", s[10 + g] - s[9 + g]);
        printf(" | st %5u | bar %5u | total %6u", s[18] - s[17], s[19] - s[18], s[19] - s[0]);
        printf(" || chunk 2: barrier+gemm1 %u | m0..7:", s[20] - s[3]);
        for (int m = 0; m < 7; ++m) printf(" %u", s[21 + m] - s[20 + m]);
        printf(" %u\n", s[4] - s[27]);
    }
#endif
    return 0;
}
