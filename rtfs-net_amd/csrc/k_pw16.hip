// Pointwise (1x1) convolution family on the f16 matrix cores with 3-product split precision:
//   x = xh + xl, w*256 = wh + wl (each part an f16);  x.w ~= (xh.wh + xh.wl + xl.wh) / 256,  f32 accumulate.
// Dropping xl.wl leaves a relative error of ~2^-22 per product (measured end to end: 6e-7, the f32 noise
// floor), at 16/3 = 5.3x the rate of v_mfma_f32_32x32x2_f32.
// Same math / modules as k_pw.hip (the exact-f32 version kept for A/B):
//   Y[b][co][p] = epilogue( bias[co] + sum_ci W[co][ci] * prologue(X[b][ci][p]) ).
// Tiling: 256 threads = 4 waves; K chunks of 32 input channels.  X is converted + transposed while staging:
// LDS images Xh/Xl [pixel][32 ci (+8 pad)] and Wh/Wl [co][32 ci (+8 pad)], so both MFMA operands are
// 16-byte ds_read_b128 fragments (8 consecutive k of one row) on conflict-free 80-byte row strides.
#include "common.h"
#include "kernels.h"


enum { PRO_NONE = 0, PRO_GLN_RELU = 1, PRO_GATEWAY = 2, PRO_PRELU = 3 };
enum { EPI_BIAS = 0, EPI_BIAS_RES = 1, EPI_S3 = 2, EPI_TAPS = 3 };

#define KC 32
#define LDH 40  // padded row length in halfs (80 B)

template <int CIN, int COUT, int PT, int WAVES_M, int PRO, int EPI, int NT = 256>
__device__ __forceinline__ void pw16_body(const PwArgs& a, const float* __restrict__ X, const float* __restrict__ X2,
                                          float* __restrict__ RES, const float* __restrict__ AUX, float* __restrict__ OUT) {
    constexpr int WAVES_N = (NT / 64) / WAVES_M;
    constexpr int WM = COUT / 32 / WAVES_M;
    constexpr int WN = PT / 32 / WAVES_N;
    static_assert(WM >= 1 && WN >= 1 && CIN % KC == 0, "tile config");
    __shared__ __attribute__((aligned(16))) _Float16 Xh[PT * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Xl[PT * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Wh[COUT * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Wl[COUT * LDH];
    __shared__ float sc[PRO == PRO_GLN_RELU || PRO == PRO_GATEWAY ? CIN : 1];
    __shared__ float sh[PRO == PRO_GLN_RELU || PRO == PRO_GATEWAY ? CIN : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * PT;
    const int P = a.P;
    const size_t xb = (size_t)b * CIN * P;

    if (PRO == PRO_GLN_RELU) {
        for (int c = tid; c < CIN; c += NT) gln_fold(a.stats + 2 * b, a.inv_count, a.gamma[c], a.beta[c], sc[c], sh[c]);
    } else if (PRO == PRO_GATEWAY) {
        for (int c = tid; c < CIN; c += NT) {
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
            for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.f;

    for (int c0 = 0; c0 < CIN; c0 += KC) {
        __syncthreads();
        // ---- stage X: each task = (pixel, group of 8 input channels) -> two 16-byte LDS rows pieces
        for (int task = tid; task < PT * 4; task += NT) {
            const int pp = task % PT, gq = task / PT;
            const int p = p0 + pp;
            const bool live = p < P;
            const int pcl = live ? p : P - 1;  // loads are unconditional (clamped): no branch + wait per load
            half8 vh, vl;
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = X[xb + (size_t)(c0 + gq * 8 + i) * P + pcl];
            if (PRO == PRO_GATEWAY && X2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) vv[i] += X2[xb + (size_t)(c0 + gq * 8 + i) * P + pcl];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = c0 + gq * 8 + i;
                float v = vv[i];
                if (PRO == PRO_GLN_RELU) {
                    v = fmaxf(fmaf(v, sc[ci], sh[ci]), 0.f);
                } else if (PRO == PRO_GATEWAY) {
                    v = preluf_(fmaf(v, sc[ci], sh[ci]), slope);
                    if (live) RES[xb + (size_t)ci * P + p] = v;
                } else if (PRO == PRO_PRELU) {
                    v = preluf_(v, slope);
                }
                const _Float16 hi = (_Float16)v;
                vh[i] = hi;
                vl[i] = (_Float16)(v - (float)hi);
            }
            *reinterpret_cast<half8*>(&Xh[pp * LDH + gq * 8]) = vh;
            *reinterpret_cast<half8*>(&Xl[pp * LDH + gq * 8]) = vl;
        }
        // ---- stage W: global image [chunk][hi|lo][co][32] halfs -> padded LDS rows
        {
            const half8* src = reinterpret_cast<const half8*>(a.w16) + (size_t)(c0 / KC) * 2 * COUT * 4;
            for (int i = tid; i < COUT * 4; i += NT) {
                const int co = i >> 2, part = i & 3;
                *reinterpret_cast<half8*>(&Wh[co * LDH + part * 8]) = src[i];
                *reinterpret_cast<half8*>(&Wl[co * LDH + part * 8]) = src[COUT * 4 + i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC; ks += 16) {
            half8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int row = (wm + m * WAVES_M) * 32 + r;
                ah[m] = *reinterpret_cast<const half8*>(&Wh[row * LDH + ks + 8 * h]);
                al[m] = *reinterpret_cast<const half8*>(&Wl[row * LDH + ks + 8 * h]);
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int col = (wn * WN + n) * 32 + r;
                bh[n] = *reinterpret_cast<const half8*>(&Xh[col * LDH + ks + 8 * h]);
                bl[n] = *reinterpret_cast<const half8*>(&Xl[col * LDH + ks + 8 * h]);
            }
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                }
        }
    }

    // ---- epilogue.  C/D layout: col = lane&31 (pixel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (co)
    constexpr float WINV = 1.0f / 256.0f;  // weights are stored pre-scaled by 2^8
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int p = p0 + (wn * WN + n) * 32 + r;
        if (p >= P) continue;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            if (EPI == EPI_S3) {
                static_assert(EPI != EPI_S3 || (COUT == 256 && WAVES_M == 4 && WM == 2), "S3 pairs (c, c+128) in one lane");
                const int c = wm * 32 + row;
                const float mr = fmaxf(fmaf(acc[0][n][q], WINV, a.bias[c]), 0.f);
                const float mi = fmaxf(fmaf(acc[WM - 1][n][q], WINV, a.bias[c + 128]), 0.f);
                const size_t o = ((size_t)b * COUT + c) * P + p;
                const float er = AUX[o], ei = AUX[o + (size_t)128 * P];
                OUT[o] = er * mr - ei * mi;
                OUT[o + (size_t)128 * P] = er * mi + ei * mr;
            } else {
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    const int co = (wm + m * WAVES_M) * 32 + row;
                    if (EPI == EPI_TAPS) {
                        if (co < a.cout_live) OUT[((size_t)b * a.cout_live + co) * P + p] = acc[m][n][q] * WINV;
                    } else {
                        const size_t o = ((size_t)b * COUT + co) * P + p;
                        float v = fmaf(acc[m][n][q], WINV, a.bias[co]);
                        if (EPI == EPI_BIAS_RES) v += AUX[o];
                        OUT[o] = v;
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int PT, int WAVES_M, int PRO, int EPI, int NT>
__global__ __launch_bounds__(NT) void pw16_kernel(PwArgs a) {
    pw16_body<CIN, COUT, PT, WAVES_M, PRO, EPI, NT>(a, a.x, a.x2, a.res_out, a.aux, a.out);
}

template <int CIN, int COUT, int PT, int WAVES_M, int PRO, int EPI, int NT = 256>
static int launch_pw16_t(const PwArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((pw16_kernel<CIN, COUT, PT, WAVES_M, PRO, EPI, NT>), dim3(cdiv(a.P, PT), B), dim3(NT), 0, st, a);
    return rtfs_launch_status();
}

int launch_pw16_audio_bn(const PwArgs& a, int B, hipStream_t st) { return launch_pw16_t<256, 256, 128, 4, PRO_GLN_RELU, EPI_BIAS, 512>(a, B, st); }
int launch_pw16_gateway_proj(const PwArgs& a, int B, hipStream_t st) { return launch_pw16_t<256, 64, 128, 2, PRO_GATEWAY, EPI_BIAS>(a, B, st); }
int launch_pw16_residual(const PwArgs& a, int B, hipStream_t st) { return launch_pw16_t<64, 256, 64, 4, PRO_NONE, EPI_BIAS_RES>(a, B, st); }
int launch_pw16_s3(const PwArgs& a, int B, hipStream_t st) { return launch_pw16_t<256, 256, 128, 4, PRO_PRELU, EPI_S3, 512>(a, B, st); }
int launch_pw16_dec_taps(const PwArgs& a, int B, hipStream_t st) { return launch_pw16_t<256, 32, 256, 1, PRO_NONE, EPI_TAPS>(a, B, st); }

// ---------------------------------------------------------------- MFMA f16 fragment-layout self test
// D(32x32) = A(32x16) B(16x32) with the layouts this file assumes: lane (r = l&31, h = l>>5) holds
// A[r][8h+j], B[8h+j][r] (j = 0..7) and D[(q&3) + 8(q>>2) + 4h][r] in accumulator register q.
__global__ void mfma_f16_selftest_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    half8 av, bv;
    for (int j = 0; j < 8; ++j) {
        av[j] = (_Float16)A[r * 16 + 8 * h + j];
        bv[j] = (_Float16)B[(8 * h + j) * 32 + r];
    }
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int q = 0; q < 16; ++q) D[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
    // D[1024 ...]: f32 -> f16 -> f32 round trip of A (shows whether the conversion keeps f16 subnormals)
    for (int j = 0; j < 8; ++j) D[1024 + r * 16 + 8 * h + j] = (float)av[j];
}

int launch_mfma_f16_selftest(const float* A, const float* B, float* D, hipStream_t st) {
    hipLaunchKernelGGL(mfma_f16_selftest_kernel, dim3(1), dim3(64), 0, st, A, B, D);
    return rtfs_launch_status();
}
