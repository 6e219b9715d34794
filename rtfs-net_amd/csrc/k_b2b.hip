// Block boundary of the fused separator, second generation (padded channel rows only; launch_pws_b2b routes here).
//   out_i          = residual_conv(expanded_i) + residual_i                 separators/tdanet.py:129
//   x              = out_i + a1                                             TDAVNet/refinement_module.py:58-60 (shared block, "+ residual")
//   residual_{i+1} = PReLU(dw1x1(x)),  x_enc_{i+1} = proj(residual_{i+1})   separators/tdanet.py:106-107
// Same arithmetic and GEMM chaining as pws_b2b_kernel (k_pws.hip): GEMM 1 (64 -> 256) one 32-channel tile at a time, its accumulator
// registers rewritten in place as residual_{i+1} and fed back as the B operand of GEMM 2 (256 -> 64, K-permuted weight image).
//
// What changed, and why (round 3, tools/bench_stream{2..5}.hip + the ISA of the old kernel):
//  * The old kernel issued 8 loads, waited for (nearly) all of them, did a quarter tile of arithmetic, stored, and only then issued the next 8:
//    the memory pipe drained four times per 32-channel tile, 8 waves x <= 4 KB in flight per CU.  Its conditional stores (ragged last tile
//    of a sample) put every store in its own basic block, which made the compiler's in-order vmcnt waits conservative on top.
//  * Here a wave is a software pipeline over 32-channel tiles: the 32 loads (residual + a1 rows, 16 KB per wave) of tile m + 1 are issued
//    before tile m's arithmetic, into a second register buffer.  That needs 128 registers of load buffers, so a workgroup is 4 waves with
//    the 512-register budget (one wave per SIMD) instead of 8 with 256.
//  * Channel rows are padded to a multiple of 64 floats (api.hip pitch()): every wave's 64-pixel segment of a row is two whole 128-byte lines
//    INSIDE the row's allocation, so loads and stores are unconditional - no live / tail predicates anywhere in the loop.  (Partial lines
//    shared by workgroups on different XCDs were what held row-walk stores at 3.35 TB/s; whole lines: 5.5.)
//  * Tiles are handed out by an atomic counter instead of a static stride: CUs do not stream at equal rates, and a static partition ends with
//    the slowest (4.1 -> 4.9 TB/s on the skeleton of this kernel, one workgroup per CU).
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

namespace {

constexpr int B4_NT = 256;              // threads per workgroup: 4 waves, one per SIMD, 512 registers each
constexpr int B4_L1 = 64 + 8, B4_L2 = 256 + 8;
constexpr size_t B4_LDS = (size_t)2 * 256 * B4_L1 * 2 + (size_t)2 * 64 * B4_L2 * 2 + (size_t)(7 * 256 + 64) * 4 + 16;

// CAF: the block input is CAF(out_0, video) + a1 (TDAVNet/fusion.py:204-212 after the first block): out <- ReLU(key(out)) * r[tv] + att[tv] * value(out)
// with key / value = dw 1x1 . eval BatchNorm (layers/fusion.py:205-226, folded to one FMA each) and r / att the video-side terms of
// caf_video_kernel, read from their (B, Tv, 256) transposed copies: four consecutive channels of one video frame are one 16-byte load.
// AM: bit 0 = the residual read does not contain a1 yet: read a1 and add it (the block input is out + a1); bit 1 = write the new residual WITH
// a1 added (the next boundary then runs with bit 0 clear and does not read a1 at all: one boundary in two saves 256 of its 576 row reads);
// bit 2 (first boundary only, CAF): residual_0 is not read either - it is the gateway of a1, PReLU(gw a1 + gb) with the block's own gateway
// weights (tdanet.py:106 applied to the bottleneck output), three instructions per value from the a1 rows that are loaded anyway; the head
// kernel then does not write it.
template <bool CAF, int AM>
__global__ __launch_bounds__(B4_NT) void pws_b2b4_kernel(B2bArgs a, int ntiles, int tiles_per_sample, unsigned* __restrict__ ctr) {
    constexpr int L1 = B4_L1, L2 = B4_L2;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* W1h = reinterpret_cast<_Float16*>(smem);  // [256][L1]
    _Float16* W1l = W1h + 256 * L1;
    _Float16* W2h = W1l + 256 * L1;                      // [64][L2], K permuted
    _Float16* W2l = W2h + 64 * L2;
    float* cA = reinterpret_cast<float*>(W2l + 64 * L2);  // per output channel of GEMM 1: gateway scale / 256
    float* cB = cA + 256;                                 //   gateway scale
    float* cC = cB + 256;                                 //   residual_conv bias * gateway scale + gateway bias
    float* bp = cC + 256;                                 // projection bias (64)
    float* cks = bp + 64;                                 // CAF: folded key / value embeddings (4 x 256)
    float* ckb = cks + 256;
    float* cvs = ckb + 256;
    float* cvb = cvs + 256;
    int* s_next = reinterpret_cast<int*>(cvb + 256);      // [2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    {
        const half8* s1 = reinterpret_cast<const half8*>(a.w1_16);  // [2 chunks][hi|lo][256][32]
        for (int i = tid; i < 2 * 2 * 256 * 4; i += B4_NT) {
            const int pc = i & 3, co = (i >> 2) & 255, part = (i >> 10) & 1, chunk = i >> 11;
            *reinterpret_cast<half8*>((part ? W1l : W1h) + co * L1 + chunk * 32 + pc * 8) = s1[i];
        }
        const half8* s2 = reinterpret_cast<const half8*>(a.w2_16);  // [8 chunks][hi|lo][64][32]
        for (int i = tid; i < 8 * 2 * 64 * 4; i += B4_NT) {
            const int pc = i & 3, co = (i >> 2) & 63, part = (i >> 8) & 1, chunk = i >> 9;
            *reinterpret_cast<half8*>((part ? W2l : W2h) + co * L2 + chunk * 32 + pc * 8) = s2[i];
        }
        {
            const float g = a.gw[tid];
            if (CAF) {  // the CAF sits between the residual conv and the gateway: the constants stay separate (cA = 1/256, cB = gateway scale, cC = gateway bias)
                cA[tid] = a.b1[tid];
                cB[tid] = g;
                cC[tid] = a.gb[tid];
                const float sk = a.caf_bn_key[tid] / sqrtf(a.caf_bn_key[768 + tid] + RTFS_EPS);
                const float sv = a.caf_bn_val[tid] / sqrtf(a.caf_bn_val[768 + tid] + RTFS_EPS);
                cks[tid] = a.caf_w_key[tid] * sk;
                ckb[tid] = a.caf_bn_key[256 + tid] - a.caf_bn_key[512 + tid] * sk;
                cvs[tid] = a.caf_w_val[tid] * sv;
                cvb[tid] = a.caf_bn_val[256 + tid] - a.caf_bn_val[512 + tid] * sv;
            } else {
                cA[tid] = g * WINV;
                cB[tid] = g;
                cC[tid] = fmaf(a.b1[tid], g, a.gb[tid]);
            }
            if (tid < 64) bp[tid] = a.bp[tid];
        }
    }
    const float slope = a.slope[0];
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs;
    // lane (r, h): pixels 2r, 2r + 1 of its wave's 64; B-fragment rows are channels 8h + j of a 16-channel K step, accumulator rows are
    // channels 4h + (q & 3) + 8 (q >> 2) of a 32-channel tile: two lane byte offsets serve every access of the kernel
    const unsigned voffB = ((unsigned)(8 * h) * CS + 2u * r) * 4u;
    const unsigned voffC = ((unsigned)(4 * h) * CS + 2u * r) * 4u;
    int it = 0;
    int tile = blockIdx.x;
    while (tile < ntiles) {
        if (tid == 0) s_next[it & 1] = (ctr ? (int)atomicAdd(ctr, 1u) : tile) + (int)gridDim.x;
        if (it == 0) __syncthreads();  // resident weights visible
        const int b = tile / tiles_per_sample;
        const int wp0 = (tile - b * tiles_per_sample) * (B4_NT / 64 * 64) + wave * 64;  // first pixel of this wave (wave-uniform)
        if (wp0 < P) {
            const __amdgpu_buffer_rsrc_t xs = rsrc_of(a.x + (size_t)b * 64 * CS + wp0);
            const __amdgpu_buffer_rsrc_t ress = rsrc_of(a.res + (size_t)b * 256 * CS + wp0);
            const __amdgpu_buffer_rsrc_t a1s = rsrc_of(a.a1 + (size_t)b * 256 * CS + wp0);
            const __amdgpu_buffer_rsrc_t xes = rsrc_of(a.xenc + (size_t)b * 64 * CS + wp0);
            const unsigned CS4 = CS * 4u;  // row pitch in bytes
            // CAF: this lane's two video frames (legacy nearest: tv = floor(t Tv / T), t = pixel / F) in the (B, Tv, 256) tables
            const float *crt0 = nullptr, *crt1 = nullptr, *cat0 = nullptr, *cat1 = nullptr;
            if (CAF) {
                const int px = wp0 + 2 * r;
                const int tv0 = nearest_src(min(px, P - 1) / a.caf_F, a.caf_Tv, a.caf_T), tv1 = nearest_src(min(px + 1, P - 1) / a.caf_F, a.caf_Tv, a.caf_T);
                crt0 = a.caf_rt + ((size_t)b * a.caf_Tv + tv0) * 256;
                crt1 = a.caf_rt + ((size_t)b * a.caf_Tv + tv1) * 256;
                cat0 = a.caf_attt + ((size_t)b * a.caf_Tv + tv0) * 256;
                cat1 = a.caf_attt + ((size_t)b * a.caf_Tv + tv1) * 256;
            }
            // ---- loads of 32-channel tile m (residual_i and a1 rows): 32 x 8 bytes per lane
            f32x2 R[2][16], A[2][16];
            auto load_tile = [&](int m, f32x2 (&Rb)[16], f32x2 (&Ab)[16]) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned ro = (unsigned)(m * 32 + (q & 3) + 8 * (q >> 2)) * CS4;  // uniform
                    if (!(AM & 4)) Rb[q] = ld2(ress, voffC, ro);
                    if (AM & 1) Ab[q] = ld2(a1s, voffC, ro);
                }
            };
            // ---- B fragments of GEMM 1 (expanded_i: 64 channels of this lane's two pixels)
            half8 xh[4][2], xl[4][2];
            {
                f32x2 v[4][8];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[ks][j] = ld2(xs, voffB, (unsigned)(ks * 16 + j) * CS4);
                load_tile(0, R[0], A[0]);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    unsigned hi0[4], lo0[4], hi1[4], lo1[4];
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        split2(v[ks][2 * jp].x, v[ks][2 * jp + 1].x, hi0[jp], lo0[jp]);
                        split2(v[ks][2 * jp].y, v[ks][2 * jp + 1].y, hi1[jp], lo1[jp]);
                    }
                    xh[ks][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi0));
                    xl[ks][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo0));
                    xh[ks][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi1));
                    xl[ks][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo1));
                }
            }
            f32x16 acc2[2][2];  // [projection tile][pixel slot]
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc2[m2][sl][q] = 0.f;
            // one 32-channel tile: GEMM 1 -> epilogue (residual_{i+1} written through) -> two K steps of GEMM 2
            auto tile_body = [&](int m, const f32x2 (&Rb)[16], const f32x2 (&Ab)[16]) {
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
                const int cob = m * 32 + 4 * h;  // channel of accumulator register q: cob + (q & 3) + 8 (q >> 2)
                f32x4 vr0[4], vr1[4], va0[4], va1[4];  // CAF: video terms of the tile's 16 channels, both pixels (L2-resident tables)
                if (CAF) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        vr0[g4] = *reinterpret_cast<const f32x4*>(crt0 + cob + 8 * g4);
                        vr1[g4] = *reinterpret_cast<const f32x4*>(crt1 + cob + 8 * g4);
                        va0[g4] = *reinterpret_cast<const f32x4*>(cat0 + cob + 8 * g4);
                        va1[g4] = *reinterpret_cast<const f32x4*>(cat1 + cob + 8 * g4);
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {  // accumulator registers 8s .. 8s+7 = K step 2m + s of GEMM 2
                    float y0[8], y1[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {  // four consecutive channels: one 16-byte LDS read per constant
                        const int c4 = cob + 8 * (2 * s + g);
                        const f32x4 kA = *reinterpret_cast<const f32x4*>(cA + c4);
                        const f32x4 kB = *reinterpret_cast<const f32x4*>(cB + c4);
                        const f32x4 kC = *reinterpret_cast<const f32x4*>(cC + c4);
                        f32x4 kks, kkb, kvs, kvb;
                        if (CAF) {
                            kks = *reinterpret_cast<const f32x4*>(cks + c4);
                            kkb = *reinterpret_cast<const f32x4*>(ckb + c4);
                            kvs = *reinterpret_cast<const f32x4*>(cvs + c4);
                            kvb = *reinterpret_cast<const f32x4*>(cvb + c4);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int j = 4 * g + i, q = 8 * s + j;
                            float t0, t1;
                            if (CAF) {
                                // out_0 = residual_conv(expanded_0) + residual_0 (CAF constants: kB = gateway scale, kC = gateway bias)
                                const float r0_ = (AM & 4) ? preluf_(fmaf(Ab[q].x, kB[i], kC[i]), slope) : Rb[q].x;
                                const float r1_ = (AM & 4) ? preluf_(fmaf(Ab[q].y, kB[i], kC[i]), slope) : Rb[q].y;
                                float o0 = fmaf(acc1[0][q], WINV, kA[i]) + r0_, o1 = fmaf(acc1[1][q], WINV, kA[i]) + r1_;
                                o0 = fmaf(fmaxf(fmaf(o0, kks[i], kkb[i]), 0.f), vr0[2 * s + g][i], va0[2 * s + g][i] * fmaf(o0, kvs[i], kvb[i]));
                                o1 = fmaf(fmaxf(fmaf(o1, kks[i], kkb[i]), 0.f), vr1[2 * s + g][i], va1[2 * s + g][i] * fmaf(o1, kvs[i], kvb[i]));
                                t0 = fmaf(o0 + Ab[q].x, kB[i], kC[i]);
                                t1 = fmaf(o1 + Ab[q].y, kB[i], kC[i]);
                            } else {
                                const f32x2 ra = (AM & 1) ? Rb[q] + Ab[q] : Rb[q];
                                t0 = fmaf(acc1[0][q], kA[i], fmaf(ra.x, kB[i], kC[i]));
                                t1 = fmaf(acc1[1][q], kA[i], fmaf(ra.y, kB[i], kC[i]));
                            }
                            y0[j] = preluf_(t0, slope);
                            y1[j] = preluf_(t1, slope);
                            st2(ress, voffC, (unsigned)(m * 32 + (q & 3) + 8 * (q >> 2)) * CS4, (AM & 2) ? f32x2{y0[j], y1[j]} + Ab[q] : f32x2{y0[j], y1[j]});
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
                    const int kk = (2 * m + s) * 16 + 8 * h;
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        const half8 ah = *reinterpret_cast<const half8*>(W2h + (m2 * 32 + r) * L2 + kk);
                        const half8 al = *reinterpret_cast<const half8*>(W2l + (m2 * 32 + r) * L2 + kk);
                        acc2[m2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc2[m2][0], 0, 0, 0);
                        acc2[m2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc2[m2][0], 0, 0, 0);
                        acc2[m2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc2[m2][0], 0, 0, 0);
                        acc2[m2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, acc2[m2][1], 0, 0, 0);
                        acc2[m2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, acc2[m2][1], 0, 0, 0);
                        acc2[m2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, acc2[m2][1], 0, 0, 0);
                    }
                }
            };
#pragma unroll 1
            for (int m = 0; m < 8; m += 2) {
                load_tile(m + 1, R[1], A[1]);
                __builtin_amdgcn_sched_barrier(0);
                tile_body(m, R[0], A[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (m + 2 < 8) load_tile(m + 2, R[0], A[0]);  // uniform
                __builtin_amdgcn_sched_barrier(0);
                tile_body(m + 1, R[1], A[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + m2 * 32 + 4 * h), b1_ = *reinterpret_cast<const f32x4*>(bp + m2 * 32 + 4 * h + 8);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + m2 * 32 + 4 * h + 16), b3 = *reinterpret_cast<const f32x4*>(bp + m2 * 32 + 4 * h + 24);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float bq = (q >> 2) == 0 ? b0[q & 3] : (q >> 2) == 1 ? b1_[q & 3] : (q >> 2) == 2 ? b2[q & 3] : b3[q & 3];
                    st2(xes, voffC, (unsigned)(m2 * 32 + (q & 3) + 8 * (q >> 2)) * CS4, f32x2{fmaf(acc2[m2][0][q], WINV, bq), fmaf(acc2[m2][1][q], WINV, bq)});
                }
            }
        }
        __syncthreads();
        tile = s_next[it & 1];
        ++it;
    }
}


// ---------------------------------------------------------------- block head (first application): gateway + projection, padded rows
//   residual = PReLU(dw1x1(x)),  x_enc = proj(residual)                                  separators/tdanet.py:106-107
// Same arithmetic as pws_kernel<256, 64, PRO_GATEWAY> (k_pws.hip); the K loop is a ring: the eight rows of K step ks + 3 are requested
// before step ks is consumed, stores are unconditional (padded rows), tiles come from the counter.  Two workgroups per CU (68 KB of LDS each).
constexpr int H4_L = 256 + 8;
constexpr size_t H4_LDS = (size_t)2 * 64 * H4_L * 2 + (size_t)(2 * 256 + 64) * 4 + 16;

__global__ __launch_bounds__(256, 2) void pws_head4_kernel(PwArgs a, int ntiles, int tiles_per_sample) {
    constexpr int L = H4_L;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Wh = reinterpret_cast<_Float16*>(smem);  // [64][L]
    _Float16* Wl = Wh + 64 * L;
    float* gsc = reinterpret_cast<float*>(Wl + 64 * L);
    float* gsh = gsc + 256;
    float* bp = gsh + 256;
    int* s_next = reinterpret_cast<int*>(bp + 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    {
        const half8* src = reinterpret_cast<const half8*>(a.w16);  // [8 chunks][hi|lo][64][32]
        for (int i = tid; i < 8 * 2 * 64 * 4; i += 256) {
            const int pc = i & 3, co = (i >> 2) & 63, part = (i >> 8) & 1, chunk = i >> 9;
            *reinterpret_cast<half8*>((part ? Wl : Wh) + co * L + chunk * 32 + pc * 8) = src[i];
        }
        gsc[tid] = a.gw[tid];
        gsh[tid] = a.gb[tid];
        if (tid < 64) bp[tid] = a.bias[tid];
    }
    const float slope = a.slope[0];
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs, CS4 = CS * 4u;
    const unsigned voffB = ((unsigned)(8 * h) * CS + 2u * r) * 4u;
    const unsigned voffC = ((unsigned)(4 * h) * CS + 2u * r) * 4u;
    int it = 0;
    int tile = blockIdx.x;
    while (tile < ntiles) {
        if (tid == 0) s_next[it & 1] = (a.tile_ctr ? (int)atomicAdd(a.tile_ctr, 1u) : tile) + (int)gridDim.x;
        if (it == 0) __syncthreads();
        const int b = tile / tiles_per_sample;
        const int wp0 = (tile - b * tiles_per_sample) * 256 + wave * 64;
        if (wp0 < P) {
            const __amdgpu_buffer_rsrc_t xs = rsrc_of(a.x + (size_t)b * 256 * CS + wp0);
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.res_out + (size_t)b * 256 * CS + wp0);
            const __amdgpu_buffer_rsrc_t os = rsrc_of(a.out + (size_t)b * 64 * CS + wp0);
            f32x2 X[4][8];
            auto load_x = [&](int ks, f32x2 (&d)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = ld2(xs, voffB, (unsigned)(ks * 16 + j) * CS4);
            };
            load_x(0, X[0]);
            load_x(1, X[1]);
            load_x(2, X[2]);
            f32x16 acc[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[m][sl][q] = 0.f;
            auto k_step = [&](int ks, const f32x2 (&v)[8]) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(gsc + ks * 16 + 8 * h), s1 = *reinterpret_cast<const f32x4*>(gsc + ks * 16 + 8 * h + 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(gsh + ks * 16 + 8 * h), t1 = *reinterpret_cast<const f32x4*>(gsh + ks * 16 + 8 * h + 4);
                float y0[8], y1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sc_ = j < 4 ? s0[j & 3] : s1[j & 3], sh_ = j < 4 ? t0[j & 3] : t1[j & 3];
                    y0[j] = preluf_(fmaf(v[j].x, sc_, sh_), slope);
                    y1[j] = preluf_(fmaf(v[j].y, sc_, sh_), slope);
                    st2(rs, voffB, (unsigned)(ks * 16 + j) * CS4, f32x2{y0[j], y1[j]});
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
                for (int m = 0; m < 2; ++m) {
                    const half8 ah = *reinterpret_cast<const half8*>(Wh + (m * 32 + r) * L + ks * 16 + 8 * h);
                    const half8 al = *reinterpret_cast<const half8*>(Wl + (m * 32 + r) * L + ks * 16 + 8 * h);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, acc[m][1], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, acc[m][1], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, acc[m][1], 0, 0, 0);
                }
            };
#pragma unroll 1
            for (int k4 = 0; k4 < 16; k4 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ks = k4 + u;
                    if (ks + 3 < 16) load_x(ks + 3, X[(u + 3) & 3]);  // uniform
                    __builtin_amdgcn_sched_barrier(0);
                    k_step(ks, X[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bp + m * 32 + 4 * h + 8 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = 4 * g + i;
                        st2(os, voffC, (unsigned)(m * 32 + i + 8 * g) * CS4, f32x2{fmaf(acc[m][0][q], WINV, bq[i]), fmaf(acc[m][1][q], WINV, bq[i])});
                    }
                }
            }
        }
        __syncthreads();
        tile = s_next[it & 1];
        ++it;
    }
}

}  // namespace

// cs must be a multiple of 64 floats covering every wave segment (api.hip pitch()); ctr: one zeroed counter word for this launch, or null
// (static stride).  Returns RTFS_ERR_ARG when the call does not qualify (the caller then uses the first-generation kernel).
bool launch_pws_b2b4_qualifies(const B2bArgs& a) {
    if ((a.caf_r && !(a.caf_rt && a.caf_attt)) || a.cs <= 0 || (a.cs & 63) || a.cs < (a.P + 63) / 64 * 64 || a.P < 2) return false;
    return (size_t)256 * a.cs * 4 < ((size_t)1 << 31);  // buffer offsets inside one sample: 31 bits
}
int launch_pws_b2b4(const B2bArgs& a, int B, unsigned* ctr, hipStream_t st) {
    if (!launch_pws_b2b4_qualifies(a)) return RTFS_ERR_ARG;
    const int tps = cdiv(a.P, B4_NT / 64 * 64), ntiles = tps * B;
    const int grid = ntiles < 256 ? ntiles : 256;  // one resident workgroup per CU
    const int am = a.a1_mode;
    if ((am != 0 && am != 1 && am != 3 && am != 5 && am != 7) || (a.caf_r && !(am & 1)) || ((am & 4) && !a.caf_r)) return RTFS_ERR_ARG;
#define B4_LAUNCH(CAF_, AM_)                                                                                                  \
    do {                                                                                                                      \
        if (rtfs_set_max_lds((const void*)pws_b2b4_kernel<CAF_, AM_>, B4_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;             \
        hipLaunchKernelGGL((pws_b2b4_kernel<CAF_, AM_>), dim3(grid), dim3(B4_NT), B4_LDS, st, a, ntiles, tps, ctr);           \
    } while (0)
    if (a.caf_r) {
        if (am == 7) B4_LAUNCH(true, 7);
        else if (am == 5) B4_LAUNCH(true, 5);
        else if (am == 3) B4_LAUNCH(true, 3);
        else B4_LAUNCH(true, 1);
    } else if (am == 3) {
        B4_LAUNCH(false, 3);
    } else if (am == 1) {
        B4_LAUNCH(false, 1);
    } else {
        B4_LAUNCH(false, 0);
    }
#undef B4_LAUNCH
    return rtfs_launch_status();
}

// block head on padded rows, no second addend, no CAF; RTFS_ERR_ARG = use pws_kernel
int launch_pws_head4(const PwArgs& a, int B, hipStream_t st) {
    if (a.x2 || a.caf_r || a.cs <= 0 || (a.cs & 63) || a.cs < (a.P + 63) / 64 * 64 || a.P < 2) return RTFS_ERR_ARG;
    if ((size_t)256 * a.cs * 4 >= ((size_t)1 << 31)) return RTFS_ERR_ARG;
    if (rtfs_set_max_lds((const void*)pws_head4_kernel, H4_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
    const int tps = cdiv(a.P, 256), ntiles = tps * B;
    const int grid = ntiles < 512 ? ntiles : 512;  // two resident workgroups per CU
    hipLaunchKernelGGL(pws_head4_kernel, dim3(grid), dim3(256), H4_LDS, st, a, ntiles, tps);
    return rtfs_launch_status();
}
