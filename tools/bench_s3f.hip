// Diagnostic harness (not part of the product): the fused tail + S3 + taps kernel of rtfs-net_amd/csrc/k_s3f.hip alone at the bench shape
// (B=32, P=32379, padded rows), timed, and - built with -DS3F_STAMP - with s_memtime stamps at the phase boundaries of workgroup 0's first tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DS3F_STAMP] -Irtfs-net_amd/csrc -o tools/_bs3f tools/bench_s3f.hip rtfs-net_amd/csrc/runtime.hip
#include "../rtfs-net_amd/csrc/k_stft.hip"
#include "../rtfs-net_amd/csrc/k_s3f.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int P = 32379, cs = (P + 63) / 64 * 64;
    const int T = 251, F = 129;
    float *x, *res, *a0, *z, *par, *wenc;
    void* encimg;
    double* st;
    unsigned* ctr;
    CK(hipMalloc(&x, (size_t)B * 64 * cs * 4)); CK(hipMalloc(&res, (size_t)B * 256 * cs * 4)); CK(hipMalloc(&a0, (size_t)B * 2 * P * 4 + 64)); CK(hipMalloc(&wenc, 256 * 18 * 4)); CK(hipMalloc(&encimg, 32768));
    CK(hipMalloc(&z, (size_t)B * 18 * cs * 4)); CK(hipMalloc(&par, 1 << 20)); CK(hipMalloc(&st, B * 16)); CK(hipMalloc(&ctr, 256));
    {
        std::vector<float> h((size_t)64 * cs);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
        for (int i = 0; i < B; ++i) CK(hipMemcpy(x + (size_t)i * 64 * cs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < B * 4; ++i) {
            CK(hipMemcpy(res + (size_t)i * 64 * cs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemset(par, 0, 1 << 20));  // all-zero weight images: the timing does not depend on the values
        for (int i = 0; i < B; ++i) CK(hipMemcpy(a0 + (size_t)i * 2 * P, h.data(), (size_t)2 * P * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(wenc, h.data(), 256 * 18 * 4, hipMemcpyHostToDevice));
        CK(hipMemset(st, 0, B * 16));
        if (launch_enc_stats(a0, wenc, st, encimg, EncPadJobs(), B, T, F, 0)) return 1;  // (the all-zero weight image serves as its own padded form)
    }
    TailS3Args a;
    a.x = x; a.res = res; a.spec = a0; a.enc_img = encimg; a.T = T; a.F = F; a.z = z;
    a.w1_16 = par; a.b1 = par + 16384; a.w16 = par + 32768; a.bias = par + 16384; a.slope = par + 16384; a.w16b = par + 131072;
    a.stats = st; a.inv_count = 1.0 / (256.0 * P); a.P = P; a.cs = cs; a.cout_live = 18; a.tile_ctr = ctr;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    const int R = 12;
    for (int i = 0; i < 3 + R; ++i) {
        CK(hipMemsetAsync(ctr, 0, 256, 0));
        (void)hipEventRecord(e0);
        if (launch_tail_s3t(a, B, 0) != RTFS_OK) { printf("launch failed\n"); return 1; }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)B * (64 + 256 + 18) * cs * 4;
    printf("B=%d tail_s3t  avg %.1f us  best %.1f us  %.2f TB/s (avg)\n", B, sum / R * 1e3, best * 1e3, bytes / (sum / R * 1e-3) / 1e12);
#ifdef S3F_STAMP
    unsigned h[16 * 32];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(s3f_stamps), sizeof(h)));
    printf("cycles per phase, workgroup 0 (tile: start->x | chunks 0..7 | groups 0..7 | store | barrier | total)\n");
    for (int t = 0; t < 12; ++t) {
        const unsigned* s = h + t * 32;
        printf("tile %2d: x %6u |", t, s[1] - s[0]);
        for (int k = 0; k < 8; ++k) printf(" %5u", s[2 + k] - s[1 + k]);
        printf(" |");
        for (int g = 0; g < 8; ++g) printf(" %5u", s[10 + g] - s[9 + g]);
        printf(" | st %5u | bar %5u | total %6u || chunk 3: gemm1 %u valu+R %u barrier %u mfma %u\n", s[18] - s[17], s[19] - s[18], s[19] - s[0], s[20] - s[4], s[21] - s[20], s[22] - s[21], s[5] - s[22]);
    }
#endif
    return 0;
}
