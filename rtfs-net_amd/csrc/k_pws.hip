// Streaming pointwise convolutions for the 256<->64 channel 1x1 convs of the RTFS block (gateway + projection,
// residual_conv; reference separators/tdanet.py:29-57,106-107,129).  These are HBM-bound (2 KB of activations per
// pixel against 6 f16 MFMAs per 16 channels), so the kernel is organised around the memory stream:
//   * the whole f16x3 weight image (64 KB) is loaded into LDS once per workgroup and stays resident;
//   * workgroups are persistent and walk 128-pixel (64 for COUT 256) tiles; inside a tile every wave owns 32 pixels and needs no
//     barrier: each lane loads the 8 input channels of its MFMA B fragment straight from global memory
//     (32 consecutive pixels per channel row = full 128-byte segments), applies the prologue, splits to f16 hi/lo
//     in registers and feeds the matrix cores; each activation is read by exactly one lane;
//   * the gateway variant writes PReLU(dw1x1(x [+ x_res])) through to `res_out` from the same registers.
// Math and split-precision scheme as k_pw16.hip.
#include "common.h"
#include "kernels.h"

enum { PRO_NONE = 0, PRO_GATEWAY = 2 };
enum { EPI_BIAS = 0, EPI_BIAS_RES = 1 };

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF>
__device__ __forceinline__ void pws_body(const PwArgs& a, int ntiles, int tiles_per_sample, const float* __restrict__ X,
                                         const float* __restrict__ X2, float* __restrict__ RES, const float* __restrict__ AUX,
                                         float* __restrict__ OUT) {
    constexpr int LDW = CIN + 8;       // padded weight row (halfs): 16-byte rows shifted by 4 banks -> conflict-free b128
    constexpr int MTW = COUT / 32 > 4 ? 4 : COUT / 32;  // co tiles per wave (<= 64 accumulator registers)
    constexpr int CSPLIT = COUT / 32 / MTW;             // waves sharing one 32-pixel tile (each re-reads its x)
    constexpr int PTB = 32 * (4 / CSPLIT);              // pixels per workgroup tile
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Wh = reinterpret_cast<_Float16*>(smem);
    _Float16* Wl = Wh + COUT * LDW;
    float* gsc = reinterpret_cast<float*>(Wl + COUT * LDW);  // gateway scale / bias per input channel
    float* gsh = gsc + CIN;
    float* cks = gsh + CIN;  // CAF: folded key / value embeddings (dw 1x1 . eval BatchNorm), layers/fusion.py:205-226
    float* ckb = cks + (CAF ? CIN : 0);
    float* cvs = ckb + (CAF ? CIN : 0);
    float* cvb = cvs + (CAF ? CIN : 0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // ---- resident weights: global image [CIN/32][hi|lo][COUT][32] halfs -> LDS [hi|lo][COUT][LDW]
    {
        const half8* src = reinterpret_cast<const half8*>(a.w16);
        for (int i = tid; i < (CIN / 32) * 2 * COUT * 4; i += 256) {
            const int pc = i & 3, co = (i >> 2) % COUT, part = (i / (4 * COUT)) & 1, chunk = i / (8 * COUT);
            *reinterpret_cast<half8*>((part ? Wl : Wh) + co * LDW + chunk * 32 + pc * 8) = src[i];
        }
        if (PRO == PRO_GATEWAY)
            for (int c = tid; c < CIN; c += 256) {
                gsc[c] = a.gw[c];
                gsh[c] = a.gb[c];
                if (CAF) {
                    const float sk = a.caf_bn_key[c] / sqrtf(a.caf_bn_key[3 * CIN + c] + RTFS_EPS);
                    const float sv = a.caf_bn_val[c] / sqrtf(a.caf_bn_val[3 * CIN + c] + RTFS_EPS);
                    cks[c] = a.caf_w_key[c] * sk;
                    ckb[c] = a.caf_bn_key[CIN + c] - a.caf_bn_key[2 * CIN + c] * sk;
                    cvs[c] = a.caf_w_val[c] * sv;
                    cvb[c] = a.caf_bn_val[CIN + c] - a.caf_bn_val[2 * CIN + c] * sv;
                }
            }
    }
    const float slope = PRO == PRO_GATEWAY ? a.slope[0] : 0.f;
    __syncthreads();

    const int P = a.P;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_sample;
        const int p = (tile - b * tiles_per_sample) * PTB + (wave / CSPLIT) * 32 + r;
        const int m0 = (wave % CSPLIT) * MTW;
        const bool live = p < P;
        const size_t xb = (size_t)b * CIN * P + (live ? p : P - 1);
        size_t cafb = 0;
        if (CAF) {  // nearest up-sampling of the video-side terms: tv = floor(t * Tv / T)
            const int t = (live ? p : P - 1) / a.caf_F;
            cafb = (size_t)b * CIN * a.caf_Tv + nearest_src(t, a.caf_Tv, a.caf_T);
        }
        f32x16 acc[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < CIN; ks += 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t off = xb + (size_t)(ks + 8 * h + j) * P;
                v[j] = X[off];  // dead lanes read a valid (clamped) pixel; nothing of theirs is stored
                if (CAF) {
                    const int ci = ks + 8 * h + j;
                    const size_t co_ = cafb + (size_t)ci * a.caf_Tv;
                    v[j] = fmaf(fmaxf(fmaf(v[j], cks[ci], ckb[ci]), 0.f), a.caf_r[co_], a.caf_att[co_] * fmaf(v[j], cvs[ci], cvb[ci]));
                }
                if (PRO == PRO_GATEWAY && HAS_X2) v[j] += X2[off];
            }
            if (PRO == PRO_GATEWAY) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(gsc + ks + 8 * h), s1 = *reinterpret_cast<const f32x4*>(gsc + ks + 8 * h + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(gsh + ks + 8 * h), t1 = *reinterpret_cast<const f32x4*>(gsh + ks + 8 * h + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = preluf_(fmaf(v[j], j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? t0[j & 3] : t1[j & 3]), slope);
                    if (live) RES[xb + (size_t)(ks + 8 * h + j) * P] = v[j];
                }
            }
            half8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 hi = (_Float16)v[j];
                bh[j] = hi;
                bl[j] = (_Float16)(v[j] - (float)hi);
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const half8 ah = *reinterpret_cast<const half8*>(Wh + ((m0 + m) * 32 + r) * LDW + ks + 8 * h);
                const half8 al = *reinterpret_cast<const half8*>(Wl + ((m0 + m) * 32 + r) * LDW + ks + 8 * h);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m], 0, 0, 0);
            }
        }
        if (live) {
            const size_t ob = (size_t)b * COUT * P + p;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                float res[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = (m0 + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    res[q] = 0.f;
                    if (EPI == EPI_BIAS_RES) res[q] = AUX[ob + (size_t)co * P];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = (m0 + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    OUT[ob + (size_t)co * P] = fmaf(acc[m][q], WINV, a.bias[co]) + res[q];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the 16-load / 16-store groups apart (register pressure)
            }
        }
    }
}

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF>
__global__ __launch_bounds__(256, 2) void pws_kernel(PwArgs a, int ntiles, int tiles_per_sample) {
    pws_body<CIN, COUT, PRO, EPI, HAS_X2, CAF>(a, ntiles, tiles_per_sample, a.x, a.x2, a.res_out, a.aux, a.out);
}

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF = false>
static int launch_pws_t(const PwArgs& a, int B, hipStream_t st) {
    const size_t lds = (size_t)2 * COUT * (CIN + 8) * 2 + (size_t)(CAF ? 6 : 2) * CIN * 4;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)pws_kernel<CIN, COUT, PRO, EPI, HAS_X2, CAF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RTFS_ERR_LAUNCH;
        configured = true;
    }
    constexpr int PTB = 32 * (4 / (COUT / 32 > 4 ? COUT / 32 / 4 : 1));
    const int tps = cdiv(a.P, PTB), ntiles = tps * B;
    const int grid = ntiles < 512 ? ntiles : 512;  // 2 resident workgroups per CU
    hipLaunchKernelGGL((pws_kernel<CIN, COUT, PRO, EPI, HAS_X2, CAF>), dim3(grid), dim3(256), lds, st, a, ntiles, tps);
    return rtfs_launch_status();
}

int launch_pws_gateway_proj(const PwArgs& a, int B, hipStream_t st) {
    if (a.caf_r) return a.x2 ? launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, true, true>(a, B, st) : RTFS_ERR_ARG;
    return a.x2 ? launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, true>(a, B, st) : launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, false>(a, B, st);
}
int launch_pws_residual(const PwArgs& a, int B, hipStream_t st) { return launch_pws_t<64, 256, PRO_NONE, EPI_BIAS_RES, false>(a, B, st); }
