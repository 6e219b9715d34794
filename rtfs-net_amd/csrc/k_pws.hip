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

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF, int S>
__device__ __forceinline__ void pws_body(const PwArgs& a, int ntiles, int tiles_per_sample, const float* __restrict__ X,
                                         const float* __restrict__ X2, float* __restrict__ RES, const float* __restrict__ AUX,
                                         float* __restrict__ OUT) {
    constexpr int LDW = CIN + 8;       // padded weight row (halfs): 16-byte rows shifted by 4 banks -> conflict-free b128
    constexpr int MTW = COUT / 32 > 4 ? 4 : COUT / 32;  // co tiles per wave (<= 64 accumulator registers)
    constexpr int CSPLIT = COUT / 32 / MTW;             // waves sharing one 32-pixel tile (each re-reads its x)
    constexpr int PTB = 32 * S * (4 / CSPLIT);          // pixels per workgroup tile (S adjacent pixels per lane)
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
    int* s_next = reinterpret_cast<int*>(cvb + (CAF ? CIN : 0));  // [2] next tile from the counter (launcher: + 16 bytes)

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
    const size_t CS = (size_t)a.cs;  // channel stride (>= P; the fused separator pads it to whole 128-byte lines)
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (a.tile_ctr && tid == 0) s_next[it & 1] = (int)atomicAdd(a.tile_ctr, 1u) + (int)gridDim.x;
        const int b = tile / tiles_per_sample;
        // lane r owns the S adjacent pixels p0 .. p0+S-1 (slot s = column r of MFMA tile s); S == 2: one unaligned 8-byte
        // access per channel row instead of two dword accesses (a dword stream tops out at 4.9 TB/s, dwordx2 at 7)
        const int p0 = (tile - b * tiles_per_sample) * PTB + (wave / CSPLIT) * 32 * S + S * r;
        const int m0 = (wave % CSPLIT) * MTW;
        bool live[S];
#pragma unroll
        for (int sl = 0; sl < S; ++sl) live[sl] = p0 + sl < P;
        // S == 2 load position: (p0, p0+1) normally; the sample's last pixel and dead lanes read (P-2, P-1)
        const int pl = S == 2 ? min(p0, P - 2) : (live[0] ? p0 : P - 1);
        const bool tail = S == 2 && p0 == P - 1;  // lane holds pixel P-1 in .y
        const size_t xb = (size_t)b * CIN * CS + pl;
        size_t cafb[S];
        if (CAF) {  // nearest up-sampling of the video-side terms: tv = floor(t * Tv / T)
#pragma unroll
            for (int sl = 0; sl < S; ++sl) {
                const int t = min(p0 + sl, P - 1) / a.caf_F;
                cafb[sl] = (size_t)b * CIN * a.caf_Tv + nearest_src(t, a.caf_Tv, a.caf_T);
            }
        }
        auto load = [&](const float* __restrict__ src, size_t off, float (&d)[S]) {
            if (S == 2) {
                typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                const f2u t2 = *reinterpret_cast<const f2u*>(src + off);
                d[0] = tail ? t2.y : t2.x;
                d[S - 1] = t2.y;
            } else {
                d[0] = src[off];
            }
        };
        auto store = [&](float* __restrict__ dst, size_t off, const float (&d)[S]) {  // off = position of pixel p0
            if (S == 2) {
                typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                if (live[S - 1]) *reinterpret_cast<f2u*>(dst + off) = f2u{d[0], d[S - 1]};
                else if (live[0]) dst[off] = d[0];
            } else {
                if (live[0]) dst[off] = d[0];
            }
        };
        const size_t ob_in = (size_t)b * CIN * CS + p0;  // store position (pixel p0) in a CIN-channel tensor
        f32x16 acc[MTW][S];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int sl = 0; sl < S; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[m][sl][q] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < CIN; ks += 16) {
            float v[8][S];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t off = xb + (size_t)(ks + 8 * h + j) * CS;
                load(X, off, v[j]);  // dead lanes read valid (clamped) pixels; nothing of theirs is stored
                if (CAF) {
                    const int ci = ks + 8 * h + j;
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) {
                        const size_t co_ = cafb[sl] + (size_t)ci * a.caf_Tv;
                        v[j][sl] = fmaf(fmaxf(fmaf(v[j][sl], cks[ci], ckb[ci]), 0.f), a.caf_r[co_], a.caf_att[co_] * fmaf(v[j][sl], cvs[ci], cvb[ci]));
                    }
                }
                if (PRO == PRO_GATEWAY && HAS_X2) {
                    float x2v[S];
                    load(X2, off, x2v);
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) v[j][sl] += x2v[sl];
                }
            }
            if (PRO == PRO_GATEWAY) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(gsc + ks + 8 * h), s1 = *reinterpret_cast<const f32x4*>(gsc + ks + 8 * h + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(gsh + ks + 8 * h), t1 = *reinterpret_cast<const f32x4*>(gsh + ks + 8 * h + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int sl = 0; sl < S; ++sl)
                        v[j][sl] = preluf_(fmaf(v[j][sl], j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? t0[j & 3] : t1[j & 3]), slope);
                    store(RES, ob_in + (size_t)(ks + 8 * h + j) * CS, v[j]);
                }
            }
            half8 bh[S], bl[S];
#pragma unroll
            for (int sl = 0; sl < S; ++sl)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const _Float16 hi = (_Float16)v[j][sl];
                    bh[sl][j] = hi;
                    bl[sl][j] = (_Float16)(v[j][sl] - (float)hi);
                }
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const half8 ah = *reinterpret_cast<const half8*>(Wh + ((m0 + m) * 32 + r) * LDW + ks + 8 * h);
                const half8 al = *reinterpret_cast<const half8*>(Wl + ((m0 + m) * 32 + r) * LDW + ks + 8 * h);
#pragma unroll
                for (int sl = 0; sl < S; ++sl) {
                    acc[m][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[sl], acc[m][sl], 0, 0, 0);
                    acc[m][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[sl], acc[m][sl], 0, 0, 0);
                    acc[m][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[sl], acc[m][sl], 0, 0, 0);
                }
            }
        }
        if (live[0]) {
            const size_t ob = (size_t)b * COUT * CS + p0, lb = (size_t)b * COUT * CS + pl;
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                float res[16][S];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = (m0 + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) res[q][sl] = 0.f;
                    if (EPI == EPI_BIAS_RES) load(AUX, lb + (size_t)co * CS, res[q]);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = (m0 + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    float o_[S];
#pragma unroll
                    for (int sl = 0; sl < S; ++sl) o_[sl] = fmaf(acc[m][sl][q], WINV, a.bias[co]) + res[q][sl];
                    store(OUT, ob + (size_t)co * CS, o_);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the 16-load / 16-store groups apart (register pressure)
            }
        }
        if (a.tile_ctr) {  // uniform
            __syncthreads();
            tile = s_next[it & 1];
        } else {
            tile += gridDim.x;
        }
    }
}

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF, int S>
__global__ __launch_bounds__(256, 2) void pws_kernel(PwArgs a, int ntiles, int tiles_per_sample) {
    pws_body<CIN, COUT, PRO, EPI, HAS_X2, CAF, S>(a, ntiles, tiles_per_sample, a.x, a.x2, a.res_out, a.aux, a.out);
}

template <int CIN, int COUT, int PRO, int EPI, bool HAS_X2, bool CAF = false>
static int launch_pws_t(const PwArgs& a_, int B, hipStream_t st) {
    constexpr int S = COUT <= 64 ? 2 : 1;  // two pixels per lane where the accumulators allow it (COUT 64: 64 registers)
    if (a_.P < 2 || (a_.cs && a_.cs < a_.P)) return RTFS_ERR_SHAPE;
    PwArgs a = a_;
    if (!a.cs) a.cs = a.P;
    const size_t lds = (size_t)2 * COUT * (CIN + 8) * 2 + (size_t)(CAF ? 6 : 2) * CIN * 4 + 16;
    if (rtfs_set_max_lds((const void*)pws_kernel<CIN, COUT, PRO, EPI, HAS_X2, CAF, S>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    constexpr int PTB = 32 * S * (4 / (COUT / 32 > 4 ? COUT / 32 / 4 : 1));
    const int tps = cdiv(a.P, PTB), ntiles = tps * B;
    const int grid = ntiles < 512 ? ntiles : 512;  // 2 resident workgroups per CU
    hipLaunchKernelGGL((pws_kernel<CIN, COUT, PRO, EPI, HAS_X2, CAF, S>), dim3(grid), dim3(256), lds, st, a, ntiles, tps);
    return rtfs_launch_status();
}

int launch_pws_gateway_proj(const PwArgs& a, int B, hipStream_t st) {
    if (a.caf_r) return a.x2 ? launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, true, true>(a, B, st) : RTFS_ERR_ARG;
    return a.x2 ? launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, true>(a, B, st) : launch_pws_t<256, 64, PRO_GATEWAY, EPI_BIAS, false>(a, B, st);
}


// ---------------------------------------------------------------- block tail: residual_conv (64 -> 256) + bias + residual
// (separators/tdanet.py:129) with TWO adjacent pixels per lane: every access of the 256-channel residual / output rows is
// one unaligned 8-byte load / store (the row-walk pattern runs at 4.4 TB/s with 8-byte accesses against 3.1 with dwords,
// tools/bench_stream.hip).  Same dataflow as GEMM 1 of the block-boundary kernel below: a wave keeps the 64 input channels
// of its 64 pixels as B fragments and produces one 32-channel output tile at a time (16 + 16 accumulator registers).
typedef float f32x2u_ __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void pws_res2_body(const PwArgs& a, int ntiles, int tiles_per_sample, const float* __restrict__ X,
                                              const float* __restrict__ AUX, float* __restrict__ OUT) {
    constexpr int L1 = 64 + 8;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* W1h = reinterpret_cast<_Float16*>(smem);  // [256][L1]
    _Float16* W1l = W1h + 256 * L1;
    float* b1 = reinterpret_cast<float*>(W1l + 256 * L1);
    int* s_next = reinterpret_cast<int*>(b1 + 256);  // [2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    {
        const half8* s1 = reinterpret_cast<const half8*>(a.w16);  // [2 chunks][hi|lo][256][32]
        for (int i = tid; i < 2 * 2 * 256 * 4; i += 256) {
            const int pc = i & 3, co = (i >> 2) & 255, part = (i >> 10) & 1, chunk = i >> 11;
            *reinterpret_cast<half8*>((part ? W1l : W1h) + co * L1 + chunk * 32 + pc * 8) = s1[i];
        }
        b1[tid] = a.bias[tid];
    }
    __syncthreads();
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs;  // channel stride (the launcher checks 256 * cs < 2^31)
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (a.tile_ctr && tid == 0) s_next[it & 1] = (int)atomicAdd(a.tile_ctr, 1u) + (int)gridDim.x;
        const int b = tile / tiles_per_sample;
        const int p0 = (tile - b * tiles_per_sample) * 256 + wave * 64 + 2 * r;  // this lane's pixels p0, p0 + 1
        const bool live0 = p0 < P, live1 = p0 + 1 < P;
        const int pl = min(p0, P - 2);     // load position: the sample's last pixel and dead lanes read (P-2, P-1)
        const bool tail = p0 == P - 1;     // ... and take pixel P-1 from .y
        const float* __restrict__ xs = X + (size_t)b * 64 * CS;
        const float* __restrict__ rs = AUX + (size_t)b * 256 * CS;
        float* __restrict__ os = OUT + (size_t)b * 256 * CS;
        half8 xh[4][2], xl[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x2u_ v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x2u_*>(xs + ((unsigned)(ks * 16 + 8 * h + j) * CS + pl));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v0 = tail ? v[j].y : v[j].x, v1 = v[j].y;
                const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                xh[ks][0][j] = h0;
                xl[ks][0][j] = (_Float16)(v0 - (float)h0);
                xh[ks][1][j] = h1;
                xl[ks][1][j] = (_Float16)(v1 - (float)h1);
            }
        }
#pragma unroll 1
        for (int m = 0; m < 8; ++m) {
            const int cob = m * 32 + 4 * h;  // channel of accumulator register q: cob + (q&3) + 8*(q>>2)
            f32x2u_ res[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) res[q] = *reinterpret_cast<const f32x2u_*>(rs + ((unsigned)(cob + (q & 3) + 8 * (q >> 2)) * CS + pl));
            f32x16 acc[2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[sl][q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(W1h + (m * 32 + r) * L1 + ks * 16 + 8 * h);
                const half8 al = *reinterpret_cast<const half8*>(W1l + (m * 32 + r) * L1 + ks * 16 + 8 * h);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks][sl], acc[sl], 0, 0, 0);
                    acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks][sl], acc[sl], 0, 0, 0);
                    acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks][sl], acc[sl], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = cob + (q & 3) + 8 * (q >> 2);
                const float bq = b1[co];
                const float y0 = fmaf(acc[0][q], WINV, bq) + (tail ? res[q].y : res[q].x);
                const float y1 = fmaf(acc[1][q], WINV, bq) + res[q].y;
                if (live1) *reinterpret_cast<f32x2u_*>(os + ((unsigned)co * CS + p0)) = f32x2u_{y0, y1};
                else if (live0) os[(unsigned)co * CS + p0] = y0;
            }
        }
        if (a.tile_ctr) {  // uniform
            __syncthreads();
            tile = s_next[it & 1];
        } else {
            tile += gridDim.x;
        }
    }
}
__global__ __launch_bounds__(256, 2) void pws_res2_kernel(PwArgs a, int ntiles, int tiles_per_sample) {
    pws_res2_body(a, ntiles, tiles_per_sample, a.x, a.aux, a.out);
}
static int launch_pws_res2(const PwArgs& a_, int B, hipStream_t st) {
    if (a_.P < 2 || (a_.cs && a_.cs < a_.P)) return RTFS_ERR_SHAPE;
    PwArgs a = a_;
    if (!a.cs) a.cs = a.P;
    if ((size_t)256 * a.cs >= ((size_t)1 << 31)) return RTFS_ERR_SHAPE;  // 32-bit element offsets inside a sample
    const size_t lds = (size_t)2 * 256 * 72 * 2 + 256 * 4 + 16;
    if (rtfs_set_max_lds((const void*)pws_res2_kernel, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    const int tps = cdiv(a.P, 256), ntiles = tps * B;
    const int grid = ntiles < 512 ? ntiles : 512;  // 2 resident workgroups per CU
    hipLaunchKernelGGL(pws_res2_kernel, dim3(grid), dim3(256), lds, st, a, ntiles, tps);
    return rtfs_launch_status();
}

// ---------------------------------------------------------------- back-to-back block boundary
// residual_conv of block i (64 -> 256, + residual_i) fused with the gateway + projection of block i+1
// (input = out_i [CAF'ed after the first block] + a1; dw 1x1 + PReLU -> residual_{i+1}; 1x1 256 -> 64 -> x_enc_{i+1}):
// out_i never goes to HBM.  Per wave (32 pixels): GEMM 1 leaves the 256 output channels of its pixels in 128
// accumulator registers; the epilogue rewrites them in place as residual_{i+1}; converted to f16 hi/lo they are the B
// operand of GEMM 2 *as they stand* -- registers 8s..8s+7 of co-tile m hold channels 32m+16s + (j&3) + 8(j>>2) + 4h,
// so the projection weight image is stored with its K axis permuted to that order (packing.proj_perm).
// Both weight matrices (141 KB of f16 hi/lo) stay resident in LDS; 8 waves per workgroup, no barriers in the loop.
#define B2B_NT 512  // threads per workgroup (8 waves, 256 registers each; 4 waves with 512 registers measured slower)
template <bool CAF>
__device__ __forceinline__ void pws_b2b_body(const B2bArgs& a, int ntiles, int tiles_per_sample, const float* __restrict__ X,
                                             float* __restrict__ RES, const float* __restrict__ A1, float* __restrict__ XENC) {
    constexpr int L1 = 64 + 8, L2 = 256 + 8;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* W1h = reinterpret_cast<_Float16*>(smem);  // [256][L1]
    _Float16* W1l = W1h + 256 * L1;
    _Float16* W2h = W1l + 256 * L1;                      // [64][L2], K permuted
    _Float16* W2l = W2h + 64 * L2;
    float* b1 = reinterpret_cast<float*>(W2l + 64 * L2);  // residual_conv bias (256)
    float* gsc = b1 + 256;
    float* gsh = gsc + 256;
    float* cks = gsh + 256;
    float* ckb = cks + (CAF ? 256 : 0);
    float* cvs = ckb + (CAF ? 256 : 0);
    float* cvb = cvs + (CAF ? 256 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    {
        const half8* s1 = reinterpret_cast<const half8*>(a.w1_16);  // [2 chunks][hi|lo][256][32]
        for (int i = tid; i < 2 * 2 * 256 * 4; i += B2B_NT) {
            const int pc = i & 3, co = (i >> 2) & 255, part = (i >> 10) & 1, chunk = i >> 11;
            *reinterpret_cast<half8*>((part ? W1l : W1h) + co * L1 + chunk * 32 + pc * 8) = s1[i];
        }
        const half8* s2 = reinterpret_cast<const half8*>(a.w2_16);  // [8 chunks][hi|lo][64][32]
        for (int i = tid; i < 8 * 2 * 64 * 4; i += B2B_NT) {
            const int pc = i & 3, co = (i >> 2) & 63, part = (i >> 8) & 1, chunk = i >> 9;
            *reinterpret_cast<half8*>((part ? W2l : W2h) + co * L2 + chunk * 32 + pc * 8) = s2[i];
        }
        for (int c = tid; c < 256; c += B2B_NT) {
            b1[c] = a.b1[c];
            gsc[c] = a.gw[c];
            gsh[c] = a.gb[c];
            if (CAF) {
                const float sk = a.caf_bn_key[c] / sqrtf(a.caf_bn_key[768 + c] + RTFS_EPS);
                const float sv = a.caf_bn_val[c] / sqrtf(a.caf_bn_val[768 + c] + RTFS_EPS);
                cks[c] = a.caf_w_key[c] * sk;
                ckb[c] = a.caf_bn_key[256 + c] - a.caf_bn_key[512 + c] * sk;
                cvs[c] = a.caf_w_val[c] * sv;
                cvb[c] = a.caf_bn_val[256 + c] - a.caf_bn_val[512 + c] * sv;
            }
        }
    }
    const float slope = a.slope[0];
    __syncthreads();
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs;  // channel stride (the launcher checks 256 * cs < 2^31)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_sample;
        // two adjacent pixels per lane (slots 0 / 1 = column r of two MFMA tiles): every row access is one unaligned
        // 8-byte load / store (row walks run at 4.4 TB/s with 8-byte accesses against 3.1 with dwords)
        const int p0 = (tile - b * tiles_per_sample) * (B2B_NT / 64 * 64) + wave * 64 + 2 * r;
        const bool live0 = p0 < P, live1 = p0 + 1 < P;
        const int pl = min(p0, P - 2);  // load position: the sample's last pixel and dead lanes read (P-2, P-1)
        const bool tail = p0 == P - 1;  // ... and take pixel P-1 from .y
        unsigned cafb[2] = {0, 0};
        if (CAF) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                cafb[sl] = (unsigned)(b * 256 * a.caf_Tv + nearest_src(min(p0 + sl, P - 1) / a.caf_F, a.caf_Tv, a.caf_T));
        }
        // ---- B fragments of GEMM 1 (expanded_i, 64 channels of this lane's pixels): loaded once per tile
        half8 xh[4][2], xl[4][2];
        {
            const float* __restrict__ xs = X + (size_t)b * 64 * CS;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x2u_ v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x2u_*>(xs + ((unsigned)(ks * 16 + 8 * h + j) * CS + pl));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v0 = tail ? v[j].y : v[j].x, v1 = v[j].y;
                    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
                    xh[ks][0][j] = h0;
                    xl[ks][0][j] = (_Float16)(v0 - (float)h0);
                    xh[ks][1][j] = h1;
                    xl[ks][1][j] = (_Float16)(v1 - (float)h1);
                }
            }
        }
        f32x16 acc2[2][2];  // [projection tile][slot]
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc2[m2][sl][q] = 0.f;
        float* __restrict__ ress = RES + (size_t)b * 256 * CS;
        const float* __restrict__ a1s = A1 + (size_t)b * 256 * CS;
        // ---- one output-channel tile of the residual conv at a time: GEMM 1 tile (both slots) -> epilogue 1 in two halves
        //      of 8 accumulator registers (-> residual_{i+1}) -> each half is one K step of GEMM 2
#pragma unroll 1
        for (int m = 0; m < 8; ++m) {
            const int cob = m * 32 + 4 * h;  // channel of accumulator register q: cob + (q&3) + 8*(q>>2)
            f32x16 acc1[2];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc1[sl][q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(W1h + (m * 32 + r) * L1 + ks * 16 + 8 * h);
                const half8 al = *reinterpret_cast<const half8*>(W1l + (m * 32 + r) * L1 + ks * 16 + 8 * h);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks][sl], acc1[sl], 0, 0, 0);
                    acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks][sl], acc1[sl], 0, 0, 0);
                    acc1[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks][sl], acc1[sl], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // accumulator registers 8s .. 8s+7 = K step 2m+s of GEMM 2
                half8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {  // quarter tiles: 4 channels x 2 pixels of residual and a1 in flight at a time
                    f32x2u_ res[4], a1v[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int q = 8 * s + 4 * u + jj;
                        const unsigned o = (unsigned)(cob + (q & 3) + 8 * (q >> 2)) * CS + pl;
                        res[jj] = *reinterpret_cast<const f32x2u_*>(ress + o);
                        a1v[jj] = *reinterpret_cast<const f32x2u_*>(a1s + o);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = 4 * u + jj, q = 8 * s + j;
                        const int co = cob + (q & 3) + 8 * (q >> 2);
                        float y[2];
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            float t = fmaf(acc1[sl][q], WINV, b1[co]) + (sl == 0 ? (tail ? res[jj].y : res[jj].x) : res[jj].y);  // out_i
                            if (CAF) {
                                const unsigned ci = cafb[sl] + (unsigned)(co * a.caf_Tv);
                                t = fmaf(fmaxf(fmaf(t, cks[co], ckb[co]), 0.f), a.caf_r[ci], a.caf_att[ci] * fmaf(t, cvs[co], cvb[co]));
                            }
                            t += sl == 0 ? (tail ? a1v[jj].y : a1v[jj].x) : a1v[jj].y;
                            t = preluf_(fmaf(t, gsc[co], gsh[co]), slope);
                            y[sl] = t;
                            const _Float16 hi = (_Float16)t;
                            bh[sl][j] = hi;
                            bl[sl][j] = (_Float16)(t - (float)hi);
                        }
                        if (live1) *reinterpret_cast<f32x2u_*>(ress + ((unsigned)co * CS + p0)) = f32x2u_{y[0], y[1]};
                        else if (live0) ress[(unsigned)co * CS + p0] = y[0];
                    }
                }
                const int kk = (2 * m + s) * 16 + 8 * h;
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const half8 ah = *reinterpret_cast<const half8*>(W2h + (m2 * 32 + r) * L2 + kk);
                    const half8 al = *reinterpret_cast<const half8*>(W2l + (m2 * 32 + r) * L2 + kk);
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) {
                        acc2[m2][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[sl], acc2[m2][sl], 0, 0, 0);
                        acc2[m2][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[sl], acc2[m2][sl], 0, 0, 0);
                        acc2[m2][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[sl], acc2[m2][sl], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (live0) {
            float* __restrict__ xes = XENC + (size_t)b * 64 * CS;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = m2 * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    const float y0 = fmaf(acc2[m2][0][q], WINV, a.bp[co]), y1 = fmaf(acc2[m2][1][q], WINV, a.bp[co]);
                    if (live1) *reinterpret_cast<f32x2u_*>(xes + ((unsigned)co * CS + p0)) = f32x2u_{y0, y1};
                    else xes[(unsigned)co * CS + p0] = y0;
                }
        }
    }
}

template <bool CAF>
__global__ __launch_bounds__(B2B_NT) void pws_b2b_kernel(B2bArgs a, int ntiles, int tiles_per_sample) {
    pws_b2b_body<CAF>(a, ntiles, tiles_per_sample, a.x, a.res, a.a1, a.xenc);
}

template <bool CAF>
static int launch_b2b_t(const B2bArgs& a_, int B, hipStream_t st) {
    if (a_.cs && a_.cs < a_.P) return RTFS_ERR_SHAPE;
    B2bArgs a = a_;
    if (!a.cs) a.cs = a.P;
    if ((size_t)256 * a.cs >= ((size_t)1 << 31)) return RTFS_ERR_SHAPE;  // 32-bit element offsets inside a sample
    const size_t lds = (size_t)2 * 256 * 72 * 2 + (size_t)2 * 64 * 264 * 2 + (size_t)(CAF ? 7 : 3) * 256 * 4;
    if (rtfs_set_max_lds((const void*)pws_b2b_kernel<CAF>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    if (a.P < 2) return RTFS_ERR_SHAPE;
    const int tps = cdiv(a.P, B2B_NT / 64 * 64), ntiles = tps * B;
    const int grid = ntiles < 256 ? ntiles : 256;  // one resident 8-wave workgroup per CU
    hipLaunchKernelGGL((pws_b2b_kernel<CAF>), dim3(grid), dim3(B2B_NT), lds, st, a, ntiles, tps);
    return rtfs_launch_status();
}

int launch_pws_b2b(const B2bArgs& a, int B, hipStream_t st) { return a.caf_r ? launch_b2b_t<true>(a, B, st) : launch_b2b_t<false>(a, B, st); }

int launch_pws_residual(const PwArgs& a, int B, hipStream_t st) { return launch_pws_res2(a, B, st); }
