// Pointwise (1x1) convolution family on the f32 matrix cores (v_mfma_f32_32x32x2_f32):
//   Y[b][co][p] = epilogue( bias[co] + sum_ci Wt[ci][co] * prologue(X[b][ci][p]) )
// over (B, C, P) tensors with the pixel index p = t*F + f contiguous.  One kernel template
// serves every 1x1 conv on the path; the reference modules it replaces are
//   audio bottleneck  ConvNormAct(gLN -> ReLU -> 1x1)          tdavnet.py:59, conv_layers.py:65-129
//   block gateway+projection, residual_conv                    separators/tdanet.py:29-57,106-107,129
//   S^3 mask head PReLU -> 1x1 -> ReLU -> complex multiply     TDAVNet/mask_generator.py:49-59,67-99
//   decoder ConvTranspose2d 256->2 3x3 as 18 per-tap 1x1 maps  TDAVNet/decoder.py:96-104
// Tiling: 256 threads = 4 waves; the co x pixel block tile is cut into 32x32 MFMA tiles, the K
// (input-channel) loop runs in chunks of 32 staged through LDS: X chunk [32][PT] (prologue applied
// while staging) and W chunk [32][COUT] (weights are stored transposed, [ci][co], so both MFMA
// operands are read from LDS with consecutive lanes on consecutive banks).
#include "common.h"
#include "kernels.h"

enum { PRO_NONE = 0, PRO_GLN_RELU = 1, PRO_GATEWAY = 2, PRO_PRELU = 3 };
enum { EPI_BIAS = 0, EPI_BIAS_RES = 1, EPI_S3 = 2, EPI_TAPS = 3 };

template <int CIN, int COUT, int PT, int WAVES_M, int PRO, int EPI>
__global__ __launch_bounds__(256) void pw_kernel(PwArgs a) {
    constexpr int KC = 32;
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WM = COUT / 32 / WAVES_M;  // co tiles per wave, strided by WAVES_M
    constexpr int WN = PT / 32 / WAVES_N;    // pixel tiles per wave
    static_assert(WM >= 1 && WN >= 1 && CIN % KC == 0, "tile config");
    __shared__ float Xs[KC][PT];
    __shared__ float Ws[KC][COUT];
    __shared__ float sc[PRO == PRO_GLN_RELU || PRO == PRO_GATEWAY ? CIN : 1];
    __shared__ float sh[PRO == PRO_GLN_RELU || PRO == PRO_GATEWAY ? CIN : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * PT;
    const int P = a.P;
    const size_t xb = (size_t)b * CIN * P;

    if (PRO == PRO_GLN_RELU) {
        for (int c = tid; c < CIN; c += 256) gln_fold(a.stats + 2 * b, a.inv_count, a.gamma[c], a.beta[c], sc[c], sh[c]);
    } else if (PRO == PRO_GATEWAY) {
        for (int c = tid; c < CIN; c += 256) {
            sc[c] = a.gw[c];
            sh[c] = a.gb[c];
        }
    }
    const float slope = (PRO == PRO_GATEWAY || PRO == PRO_PRELU) ? a.slope[0] : 0.f;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    for (int c0 = 0; c0 < CIN; c0 += KC) {
        __syncthreads();  // previous chunk fully consumed (also orders sc/sh on the first pass)
        // ---- stage X chunk with the prologue applied
        for (int idx = tid; idx < KC * PT; idx += 256) {
            const int k = idx / PT, pp = idx % PT;
            const int p = p0 + pp, ci = c0 + k;
            float v = 0.f;
            if (p < P) {
                const size_t off = xb + (size_t)ci * P + p;
                v = a.x[off];
                if (PRO == PRO_GLN_RELU) {
                    v = fmaxf(fmaf(v, sc[ci], sh[ci]), 0.f);
                } else if (PRO == PRO_GATEWAY) {
                    if (a.x2) v += a.x2[off];
                    v = preluf_(fmaf(v, sc[ci], sh[ci]), slope);
                    a.res_out[off] = v;
                } else if (PRO == PRO_PRELU) {
                    v = preluf_(v, slope);
                }
            }
            Xs[k][pp] = v;
        }
        // ---- stage W chunk (rows of the transposed weight are contiguous)
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.wt + (size_t)c0 * COUT);
            f32x4* dst = reinterpret_cast<f32x4*>(&Ws[0][0]);
            for (int idx = tid; idx < KC * COUT / 4; idx += 256) dst[idx] = src[idx];
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < KC; kk += 2) {
            const int k = kk + (lane >> 5);
            float av[WM], bv[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) av[m] = Ws[k][(wm + m * WAVES_M) * 32 + (lane & 31)];
#pragma unroll
            for (int n = 0; n < WN; ++n) bv[n] = Xs[k][(wn * WN + n) * 32 + (lane & 31)];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co)
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int p = p0 + (wn * WN + n) * 32 + (lane & 31);
        if (p >= P) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (EPI == EPI_S3) {
                static_assert(EPI != EPI_S3 || (COUT == 256 && WAVES_M == 4 && WM == 2), "S3 pairs (c, c+128) in one lane");
                const int c = wm * 32 + row;  // 0..127: real-part channel; c+128: imaginary part
                const float mr = fmaxf(acc[0][n][r] + a.bias[c], 0.f);
                const float mi = fmaxf(acc[WM - 1][n][r] + a.bias[c + 128], 0.f);
                const size_t o = ((size_t)b * COUT + c) * P + p;
                const float er = a.aux[o], ei = a.aux[o + (size_t)128 * P];
                a.out[o] = er * mr - ei * mi;
                a.out[o + (size_t)128 * P] = er * mi + ei * mr;
            } else {
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    const int co = (wm + m * WAVES_M) * 32 + row;
                    if (EPI == EPI_TAPS) {
                        if (co < a.cout_live) a.out[((size_t)b * a.cout_live + co) * P + p] = acc[m][n][r];
                    } else {
                        const size_t o = ((size_t)b * COUT + co) * P + p;
                        float v = acc[m][n][r] + a.bias[co];
                        if (EPI == EPI_BIAS_RES) v += a.aux[o];
                        a.out[o] = v;
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int PT, int WAVES_M, int PRO, int EPI>
static int launch_pw_t(const PwArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((pw_kernel<CIN, COUT, PT, WAVES_M, PRO, EPI>), dim3(cdiv(a.P, PT), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_pw_audio_bn(const PwArgs& a, int B, hipStream_t st) { return launch_pw_t<256, 256, 64, 4, PRO_GLN_RELU, EPI_BIAS>(a, B, st); }
int launch_pw_gateway_proj(const PwArgs& a, int B, hipStream_t st) { return launch_pw_t<256, 64, 128, 2, PRO_GATEWAY, EPI_BIAS>(a, B, st); }
int launch_pw_residual(const PwArgs& a, int B, hipStream_t st) { return launch_pw_t<64, 256, 64, 4, PRO_NONE, EPI_BIAS_RES>(a, B, st); }
int launch_pw_s3(const PwArgs& a, int B, hipStream_t st) { return launch_pw_t<256, 256, 64, 4, PRO_PRELU, EPI_S3>(a, B, st); }
int launch_pw_dec_taps(const PwArgs& a, int B, hipStream_t st) { return launch_pw_t<256, 32, 256, 1, PRO_NONE, EPI_TAPS>(a, B, st); }
