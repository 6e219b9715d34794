// Head of the fused separator in ONE kernel (padded channel rows only; api.hip separator_part routes here):
//   a1       = conv1x1_256->256(ReLU(gLN(a0))) + bias            TDAVNet/av_model.py audio bottleneck (gLN -> ReLU -> Conv)   (kept: every block adds it)
//   residual = PReLU(dw1x1(a1))                                   separators/tdanet.py:106  (gateway of the first block application)
//   x_enc    = conv1x1_256->64(residual) + bias                   separators/tdanet.py:107  (projection)
// Until round 3 this was two launches (pwr_kernel<BN> 0.55 ms + pws_head4_kernel 0.51 ms) with a1 written and read back in between.  Here
// the bottleneck GEMM is K-streaming like k_s3f.hip: its whole 256-channel x 64-pixel output tile lives in 256 accumulator registers
// (AGPRs; 4 waves x 512 registers, one wave per SIMD), the K axis advances one 32-channel chunk of a0 (= one weight chunk, staged through
// LDS, double buffered, one barrier per 96 MFMAs) at a time; the epilogue walks the eight 32-channel output tiles: a1 written through, the
// gateway's PReLU output written through and - split into f16 hi / lo - fed back to the matrix cores as the B operand of the projection
// (weights resident in LDS with their K axis in accumulator-register order, as in k_b2b.hip).  Two pixels per lane everywhere: 8-byte
// accesses of whole 128-byte lines.  Per pixel the kernel reads 256 floats and writes 576, where the two kernels read 512 and wrote 576.
// The small accumulators (projection: 64 registers) are VGPR-form matrix instructions written by hand: see k_s3f.hip.
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

namespace {

constexpr int N_NT = 256;                      // 4 waves, one per SIMD
constexpr int N_ROWB = 80;                     // bottleneck chunk row: 32 k halfs (64 B) + 16 B pad: 20-bank stride, ds_read_b128 conflict-free
constexpr int N_PART = 256 * N_ROWB;
constexpr int N_BUF = 2 * N_PART;
constexpr int N_L2 = 256 + 8;                  // projection weight row (halfs)
constexpr size_t N_LDS = (size_t)2 * N_BUF + (size_t)2 * 64 * N_L2 * 2 + (size_t)(5 * 256 + 64) * 4 + 16;
constexpr int N_RING = 3;                      // chunks of a0 rows in flight

__device__ __forceinline__ void mfma_v0(f32x16& c, half8 a, half8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, half8 a, half8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v_fence4(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
}
__device__ __forceinline__ float acc_rd(float v) {  // one accumulator register -> VGPR (k_s3f.hip)
    float o;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(o) : "a"(v));
    return o;
}

__global__ __launch_bounds__(N_NT) void bn_head_kernel(BnHeadArgs a, int ntiles, int tps) {
    constexpr int L2 = N_L2;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS map: the resident tables first (every constant offset from a lane base fits the 16-bit DS offset field), the chunk buffers last
    _Float16* W2h = reinterpret_cast<_Float16*>(smem);  // [64][L2], K in accumulator order
    _Float16* W2l = W2h + 64 * L2;
    float* sc = reinterpret_cast<float*>(W2l + 64 * L2);  // gLN fold of the current mixture
    float* sh = sc + 256;
    float* bb = sh + 256;   // bottleneck bias
    float* gsc = bb + 256;  // gateway scale / bias
    float* gsh = gsc + 256;
    float* bp = gsh + 256;  // projection bias (64)
    int* s_next = reinterpret_cast<int*>(bp + 64);
    constexpr unsigned OFF_C = 2 * 64 * L2 * 2;                  // sc
    constexpr unsigned OFF_W = OFF_C + (5 * 256 + 64) * 4 + 16;  // [2 buffers][hi|lo][256 co][N_ROWB]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const half8* s2 = reinterpret_cast<const half8*>(a.w2_16);  // [8 chunks][hi|lo][64][32]
        for (int i = tid; i < 8 * 2 * 64 * 4; i += N_NT) {
            const int pc = i & 3, co = (i >> 2) & 63, part = (i >> 8) & 1, chunk = i >> 9;
            *reinterpret_cast<half8*>((part ? W2l : W2h) + co * L2 + chunk * 32 + pc * 8) = s2[i];
        }
        bb[tid] = a.bias[tid];
        gsc[tid] = a.gw[tid];
        gsh[tid] = a.gb[tid];
        if (tid < 64) bp[tid] = a.bp[tid];
    }
    const float slope = a.slope[0];
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs, CS4_ = CS * 4u;
    const int lastw = (cdiv(P, 64) - 1) * 64;
    const __amdgpu_buffer_rsrc_t ws = rsrc_of(reinterpret_cast<const float*>(a.w16));  // bottleneck: [chunk 8][hi|lo][256 co][4 pieces of 8 k]
    const unsigned voffW = (unsigned)tid * 16u;
    const unsigned voffB = ((unsigned)(8 * h) * CS + 2u * r) * 4u;
    const unsigned voffC = ((unsigned)(4 * h) * CS + 2u * r) * 4u;

    // lane byte offsets into LDS, opaque to the compiler: every access is one of these + a constant in the instruction's offset field (left
    // alone, it materialises one address register per distinct constant - 150 of them - and spills them before the tile loop)
    unsigned lw2 = (unsigned)(r * L2 + 8 * h) * 2u;                                  // projection fragments
    unsigned lc8 = OFF_C + 32u * h, lc4 = OFF_C + 16u * h;                           // per-channel constants, B-fragment / accumulator order
    unsigned lwr0 = OFF_W + (unsigned)(r * N_ROWB + 16 * h), lwr1 = lwr0 + N_BUF;    // bottleneck fragments, buffer 0 / 1
    unsigned lww0 = OFF_W + (unsigned)((tid >> 2) * N_ROWB + (tid & 3) * 16), lww1 = lww0 + N_BUF;  // staging writes
    asm volatile("" : "+v"(lw2), "+v"(lc8), "+v"(lc4), "+v"(lwr0), "+v"(lwr1), "+v"(lww0), "+v"(lww1));
    half8 pre[8];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = ld_h8(ws, voffW, (unsigned)(c * 2048 + 256 * j) * 16u);
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // piece tid + 256 j: row (tid >> 2) + 64 (j & 3) of part j >> 2
            *reinterpret_cast<half8*>(smem + (buf ? lww1 : lww0) + (j >> 2) * N_PART + 64 * (j & 3) * N_ROWB) = pre[j];
        }
    };
    stage_load(0);
    stage_write(0);

    int cur_b = -1;
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (tid == 0) s_next[it & 1] = (a.tile_ctr ? (int)atomicAdd(a.tile_ctr, 1u) : tile) + (int)gridDim.x;
        unsigned CS4 = CS4_;
        asm volatile("" : "+s"(CS4));  // row offsets are formed where they are used (one s_mul each), not hoisted out of the tile loop
        const int b = tile / tps;
        // a wave segment past the sample's end repeats the last real one (same values to the same addresses): every wave takes part in every barrier
        const int wp0 = min((tile - b * tps) * (N_NT / 64 * 64) + wave * 64, lastw);
        const __amdgpu_buffer_rsrc_t as = rsrc_of(a.x + (size_t)b * 256 * CS + wp0);
        // ---- a0 rows of chunk c: channels 32 c + 16 s + 8 h + j (B-fragment order), N_RING chunks ahead
        f32x2 X[N_RING][16];
        auto load_x = [&](int c, f32x2 (&d)[16]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = ld2(as, voffB, (unsigned)(c * 32 + (i >> 3) * 16 + (i & 7)) * CS4);
        };
#pragma unroll
        for (int c = 0; c < N_RING; ++c) load_x(c, X[c]);
        if (b != cur_b) {     // block-uniform: the gLN fold of this mixture
            __syncthreads();  // the previous tile's fragments are built
            gln_fold(a.stats + 2 * b, a.inv_count, a.gamma[tid], a.beta[tid], sc[tid], sh[tid]);
            cur_b = b;
        }
        __syncthreads();  // sc / sh (and, first tile, the resident tables) visible
        f32x16 acc[8][2];  // bottleneck conv: [output tile][pixel slot]
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[m][sl][q] = 0.f;
        auto chunk = [&](int kc, f32x2 (&Xb)[16], int buf) {
            half8 bh[2][2], bl[2][2];  // [K step][pixel slot]
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float y0[8], y1[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const unsigned co = (unsigned)(kc * 32 + 16 * s + 4 * g) * 4u;  // channel kc*32 + 16 s + 8 h + 4 g
                    const f32x4 ks = *reinterpret_cast<const f32x4*>(smem + lc8 + co), kt = *reinterpret_cast<const f32x4*>(smem + lc8 + 1024 + co);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * g + i;
                        y0[j] = fmaxf(fmaf(Xb[8 * s + j].x, ks[i], kt[i]), 0.f);
                        y1[j] = fmaxf(fmaf(Xb[8 * s + j].y, ks[i], kt[i]), 0.f);
                    }
                }
                unsigned h0[4], l0[4], h1[4], l1[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    split2(y0[2 * jp], y0[2 * jp + 1], h0[jp], l0[jp]);
                    split2(y1[2 * jp], y1[2 * jp + 1], h1[jp], l1[jp]);
                }
                bh[s][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h0));
                bl[s][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l0));
                bh[s][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h1));
                bl[s][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l1));
            }
            if (kc + N_RING < 8) load_x(kc + N_RING, Xb);  // (uniform) the ring slot is free again
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();  // chunk kc staged (buffer `buf`) and visible; everyone is done reading chunk kc - 1 (the other buffer)
            stage_load((kc + 1) & 7);  // chunk 0 again behind chunk 7: the next tile's first
            auto afrag = [&](int m, int s, int part) {
                return *reinterpret_cast<const half8*>(smem + (buf ? lwr1 : lwr0) + part * N_PART + m * 32 * N_ROWB + s * 32);
            };
            half8 ah[2][2], al[2][2];  // [buffer][K step]
            ah[0][0] = afrag(0, 0, 0); ah[0][1] = afrag(0, 1, 0);
            al[0][0] = afrag(0, 0, 1); al[0][1] = afrag(0, 1, 1);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m + 1 < 8) {
                    ah[(m + 1) & 1][0] = afrag(m + 1, 0, 0); ah[(m + 1) & 1][1] = afrag(m + 1, 1, 0);
                    al[(m + 1) & 1][0] = afrag(m + 1, 0, 1); al[(m + 1) & 1][1] = afrag(m + 1, 1, 1);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1][s], bh[s][0], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1][s], bh[s][1], acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1][s], bl[s][0], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1][s], bl[s][1], acc[m][1], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m & 1][s], bh[s][0], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m & 1][s], bh[s][1], acc[m][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_write(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) chunk(kc, X[kc % N_RING], kc & 1);
        // ---- epilogue: a1 and the gateway output written through, projection from the accumulator registers
        const __amdgpu_buffer_rsrc_t a1s = rsrc_of(a.a1 + (size_t)b * 256 * CS + wp0);
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.res + (size_t)b * 256 * CS + wp0);
        const __amdgpu_buffer_rsrc_t xes = rsrc_of(a.xenc + (size_t)b * 64 * CS + wp0);
        f32x16 acc2[2][2];  // [projection tile][pixel slot]
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // accumulator registers 8s .. 8s+7 = K step 2m + s of the projection
                float y0[8], y1[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const unsigned co = (unsigned)(m * 32 + 8 * (2 * s + g)) * 4u;  // channel m*32 + 4 h + 8 (2 s + g)
                    const f32x4 kb = *reinterpret_cast<const f32x4*>(smem + lc4 + 2048 + co);
                    const f32x4 kg = *reinterpret_cast<const f32x4*>(smem + lc4 + 3072 + co), kh = *reinterpret_cast<const f32x4*>(smem + lc4 + 4096 + co);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * g + i, q = 8 * s + j;
                        const unsigned ro = (unsigned)(m * 32 + (q & 3) + 8 * (q >> 2)) * CS4;
                        const float v0 = fmaf(acc_rd(acc[m][0][q]), WINV, kb[i]), v1 = fmaf(acc_rd(acc[m][1][q]), WINV, kb[i]);
                        st2(a1s, voffC, ro, f32x2{v0, v1});
                        y0[j] = preluf_(fmaf(v0, kg[i], kh[i]), slope);
                        y1[j] = preluf_(fmaf(v1, kg[i], kh[i]), slope);
                        st2(rs, voffC, ro, f32x2{y0[j], y1[j]});
                    }
                }
                unsigned h0[4], l0[4], h1[4], l1[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    split2(y0[2 * jp], y0[2 * jp + 1], h0[jp], l0[jp]);
                    split2(y1[2 * jp], y1[2 * jp + 1], h1[jp], l1[jp]);
                }
                const half8 bh0 = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h0)), bl0 = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l0));
                const half8 bh1 = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h1)), bl1 = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l1));
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const unsigned wo = (unsigned)(m2 * 32 * L2 + (2 * m + s) * 16) * 2u;
                    const half8 ph = *reinterpret_cast<const half8*>(smem + lw2 + wo);
                    const half8 pl = *reinterpret_cast<const half8*>(smem + lw2 + 64 * L2 * 2 + wo);
                    if (m == 0 && s == 0) {
                        mfma_v0(acc2[m2][0], ph, bh0);
                        mfma_v0(acc2[m2][1], ph, bh1);
                    } else {
                        mfma_v(acc2[m2][0], ph, bh0);
                        mfma_v(acc2[m2][1], ph, bh1);
                    }
                    mfma_v(acc2[m2][0], ph, bl0);
                    mfma_v(acc2[m2][1], ph, bl1);
                    mfma_v(acc2[m2][0], pl, bh0);
                    mfma_v(acc2[m2][1], pl, bh1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mfma_v_fence4(acc2[0][0], acc2[0][1], acc2[1][0], acc2[1][1]);
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 kb = *reinterpret_cast<const f32x4*>(smem + lc4 + 5120 + (unsigned)(m2 * 32 + 8 * g) * 4u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = 4 * g + i;
                    st2(xes, voffC, (unsigned)(m2 * 32 + i + 8 * g) * CS4, f32x2{fmaf(acc2[m2][0][q], WINV, kb[i]), fmaf(acc2[m2][1][q], WINV, kb[i])});
                }
            }
        }
        __syncthreads();  // everyone is done with this tile (s_next slot; sc / sh)
        tile = s_next[it & 1];
    }
}

}  // namespace

// RTFS_ERR_ARG = the call does not qualify (contiguous rows, tiny input): the caller runs the two separate kernels
int launch_bn_head(const BnHeadArgs& a, int B, hipStream_t st) {
    if (a.cs <= 0 || (a.cs & 63) || a.cs < (a.P + 63) / 64 * 64 || a.P < 64 || !a.w16 || !a.w2_16 || !a.stats) return RTFS_ERR_ARG;
    if ((size_t)256 * a.cs * 4 >= ((size_t)1 << 31)) return RTFS_ERR_ARG;
    if (rtfs_set_max_lds((const void*)bn_head_kernel, N_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
    const int tps = cdiv(a.P, N_NT / 64 * 64), ntiles = tps * B;
    const int grid = ntiles < 256 ? ntiles : 256;  // one resident workgroup per CU
    hipLaunchKernelGGL(bn_head_kernel, dim3(grid), dim3(N_NT), N_LDS, st, a, ntiles, tps);
    return rtfs_launch_status();
}
