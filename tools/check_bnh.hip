// Diagnostic (not part of the product): bn_head_kernel's a1 (encoder conv rebuilt on the fly) against enc_conv_kernel -> pwr_kernel<BN>,
// and enc_stats_kernel's statistics against enc_conv_kernel's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Irtfs-net_amd/csrc -o tools/_cbnh tools/check_bnh.hip rtfs-net_amd/csrc/runtime.hip
#include "../rtfs-net_amd/csrc/k_stft.hip"
#include "../rtfs-net_amd/csrc/k_pwr.hip"
#include "../rtfs-net_amd/csrc/k_bnh.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = 2, T = argc > 1 ? atoi(argv[1]) : 33, F = 129, P = T * F, cs = (P + 63) / 64 * 64;
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> spec((size_t)B * 2 * P), wenc(256 * 18), wbn(256 * 256), vec(256 * 8);
    for (auto& v : spec) v = nd(g) * 3.f;
    for (auto& v : wenc) v = nd(g) * 0.3f;
    for (auto& v : wbn) v = nd(g) / 16.f;
    if (argc > 2) for (int i = 0; i < 256 * 256; ++i) wbn[i] = (i / 256 == i % 256) ? 1.f : 0.f;  // identity bottleneck: a1 = ReLU(gLN(a0)) + bias
    for (auto& v : vec) v = 0.5f + 0.3f * nd(g);
    // bottleneck image [8 chunks][hi|lo][256 co][32 k] of 256 * W[co][k]
    std::vector<_Float16> img((size_t)8 * 2 * 256 * 32);
    for (int c = 0; c < 8; ++c)
        for (int co = 0; co < 256; ++co)
            for (int k = 0; k < 32; ++k) {
                const float v = 256.f * wbn[co * 256 + c * 32 + k];
                const _Float16 hi = (_Float16)v;
                img[((size_t)(c * 2 + 0) * 256 + co) * 32 + k] = hi;
                img[((size_t)(c * 2 + 1) * 256 + co) * 32 + k] = (_Float16)(v - (float)hi);
            }
    // the head kernel fetches its weight chunks by LDS-DMA from an image whose rows are padded to 72 bytes
    std::vector<_Float16> imgp((size_t)8 * 2 * 256 * 36, (_Float16)0.f);
    for (size_t row = 0; row < (size_t)8 * 2 * 256; ++row)
        for (int k = 0; k < 32; ++k) imgp[row * 36 + k] = img[row * 32 + k];
    float *dspec, *dwenc, *dvec, *a0, *a1r, *a1, *res, *xe, *zero;
    void *dimg, *dimgp, *encimg;
    double *st, *st2;
    CK(hipMalloc(&dspec, spec.size() * 4)); CK(hipMalloc(&dwenc, wenc.size() * 4)); CK(hipMalloc(&dvec, vec.size() * 4));
    CK(hipMalloc(&a0, (size_t)B * 256 * cs * 4)); CK(hipMalloc(&a1r, (size_t)B * 256 * cs * 4)); CK(hipMalloc(&a1, (size_t)B * 256 * cs * 4));
    CK(hipMalloc(&res, (size_t)B * 256 * cs * 4)); CK(hipMalloc(&xe, (size_t)B * 64 * cs * 4)); CK(hipMalloc(&zero, 1 << 18));
    CK(hipMalloc(&dimg, img.size() * 2)); CK(hipMalloc(&dimgp, imgp.size() * 2)); CK(hipMemcpy(dimgp, imgp.data(), imgp.size() * 2, hipMemcpyHostToDevice)); CK(hipMalloc(&encimg, 32768)); CK(hipMalloc(&st, 64 * 8)); CK(hipMalloc(&st2, 64 * 8));
    CK(hipMemcpy(dspec, spec.data(), spec.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwenc, wenc.data(), wenc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dvec, vec.data(), vec.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(zero, 0, 1 << 18)); CK(hipMemset(st, 0, 512)); CK(hipMemset(st2, 0, 512));
    CK(hipMemset(a0, 0, (size_t)B * 256 * cs * 4));
    if (launch_enc_conv(dspec, dwenc, a0, st, B, 256, T, F, cs, (size_t)256 * cs, 0)) return 1;
    if (launch_enc_stats(dspec, dwenc, st2, encimg, EncPadJobs(), B, T, F, 0)) return 1;
    if (argc > 3) {  // unit statistics, unit gamma, zero beta / bias: a1 = ReLU(a0)
        double fake[4] = {0, 256.0 * P, 0, 256.0 * P};
        CK(hipMemcpy(st, fake, 32, hipMemcpyHostToDevice)); CK(hipMemcpy(st2, fake, 32, hipMemcpyHostToDevice));
        for (int i = 0; i < 256; ++i) { vec[i] = 0; vec[256 + i] = 1; vec[512 + i] = 0; }
        CK(hipMemcpy(dvec, vec.data(), vec.size() * 4, hipMemcpyHostToDevice));
    }
    PwArgs pa;
    pa.x = a0; pa.bias = dvec; pa.out = a1r; pa.stats = st; pa.inv_count = 1.0 / (256.0 * P); pa.gamma = dvec + 256; pa.beta = dvec + 512; pa.P = P; pa.cs = cs; pa.w16 = dimg;
    if (launch_pwr_audio_bn(pa, B, 0)) return 1;
    BnHeadArgs f;
    f.spec = dspec; f.enc_img = encimg; f.T = T; f.F = F; f.a1 = a1; f.res = res; f.xenc = xe; f.stats = st2; f.inv_count = pa.inv_count;
    f.gamma = dvec + 256; f.beta = dvec + 512; f.w16 = dimgp; f.bias = dvec; f.gw = dvec + 768; f.gb = dvec + 1024; f.slope = dvec + 1280; f.w2_16 = zero; f.bp = zero;
    f.P = P; f.cs = cs;
    if (launch_bn_head(f, B, 0)) { printf("bn_head does not qualify\n"); return 1; }
    CK(hipDeviceSynchronize());
    double h1[4], h2[4];
    CK(hipMemcpy(h1, st, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2, st2, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) printf("stat[%d] conv %.9g  gram %.9g  rel %.2e\n", i, h1[i], h2[i], fabs(h1[i] - h2[i]) / fabs(h1[i]));
    std::vector<float> r((size_t)B * 256 * cs), n(r.size());
    CK(hipMemcpy(r.data(), a1r, r.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(n.data(), a1, n.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, ref = 0;
    size_t worst = 0;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < 256; ++c)
            for (int p = 0; p < P; ++p) {
                const size_t i = ((size_t)b * 256 + c) * cs + p;
                const double d = fabs((double)r[i] - n[i]);
                if (d > mx) { mx = d; worst = i; }
                ref = fmax(ref, fabs((double)r[i]));
            }
    const int wb = worst / ((size_t)256 * cs), wc = (worst / cs) % 256, wp = worst % cs;
    printf("a1: max abs err %.3e (max |ref| %.3e) at b %d c %d p %d (t %d f %d): ref %.6f new %.6f\n", mx, ref, wb, wc, wp, wp / F, wp % F, r[worst], n[worst]);
    // error map by pixel class
    double eb[4] = {0, 0, 0, 0};
    for (int p = 0; p < P; ++p) {
        const int t = p / F, ff = p % F;
        const int cls = (t == 0 || t == T - 1 ? 1 : 0) + (ff == 0 || ff == F - 1 ? 2 : 0);
        for (int c = 0; c < 256; ++c) eb[cls] = fmax(eb[cls], fabs((double)r[(size_t)c * cs + p] - n[(size_t)c * cs + p]));
    }
    printf("max err interior %.3e  t-border %.3e  f-border %.3e  corner %.3e\n", eb[0], eb[1], eb[2], eb[3]);
    double ep[2] = {0, 0}, ec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int bad = 0;
    for (int p = 0; p < P; ++p)
        for (int c = 0; c < 256; ++c) {
            const double d = fabs((double)r[(size_t)c * cs + p] - n[(size_t)c * cs + p]);
            ep[p & 1] = fmax(ep[p & 1], d);
            ec[c >> 5] = fmax(ec[c >> 5], d);
            if (d > 0.3 && bad < 40 && (c == 13 || c == 40)) { printf("  bad: c %3d p %5d (t %2d f %3d) ref %.5f new %.5f\n", c, p, p / F, p % F, r[(size_t)c * cs + p], n[(size_t)c * cs + p]); ++bad; }
        }
    printf("by pixel parity: %.3e %.3e; by channel tile:", ep[0], ep[1]);
    for (int i = 0; i < 8; ++i) printf(" %.2e", ec[i]);
    printf("\n");
    return 0;
}
