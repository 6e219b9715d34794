// Tail of the fused separator in ONE kernel (padded channel rows only; api.hip separator_part routes here):
//   refined = residual_conv(expanded) + residual                       separators/tdanet.py:129   (last block application)
//   mask    = ReLU(conv1x1_256->256(PReLU(refined)) + bias)            TDAVNet/mask_generator.py:67-88
//   sep     = mask (x) a0  (complex product with the encoder output)   TDAVNet/mask_generator.py:89-99   (a0 rebuilt from the spectrogram, not read)
//   z       = the decoder's 18 per-tap 1x1 maps of sep                 TDAVNet/decoder.py:110-117 (ConvTranspose2d as taps + shift-sum)
// Until round 3 this was two launches (pws_res2_kernel 0.44 ms + pwr_kernel<S3T> 0.78 ms) with the 1 GB `refined` tensor written and read back
// in between, and the second one bound by the texture addresser (one pixel per lane: 512 dword accesses per 32 pixels).  Here `refined` never
// exists in memory: like the block-boundary kernel (k_b2b.hip), the residual conv produces one 32-channel tile at a time in accumulator
// registers, and the tile - bias, residual, PReLU, f16 hi / lo split - is the B operand of the mask GEMM as it stands.  The mask GEMM is
// K-streaming: its whole 256 x 64-pixel output tile lives in 256 accumulator registers (AGPRs; 4 waves x 512 registers, one wave per SIMD),
// the K axis advances one residual-conv tile (32 channels = one weight chunk) at a time.  Two pixels per lane everywhere: 8-byte accesses of
// whole 128-byte lines.  Per pixel the kernel reads 64 + 256 floats (+ 18 spectrogram taps) and writes 18, where the two kernels read 832 and wrote 274.
//   weights: residual conv resident in LDS (72 KB); mask conv streamed through LDS one 32-channel chunk (32 KB) at a time, double buffered,
//            one barrier per chunk (= per 120 MFMAs); the chunk's A fragments are read in ACCUMULATOR-REGISTER K order (two 8-byte reads
//            from two adjacent 16-byte pieces), so no permuted copy of the weight image is needed;
//   taps:    as in pwr_kernel<S3T> - the separated spectrum goes back into the matrix cores as the B operand of the taps GEMM (K-permuted
//            taps image from L2), the encoder rows are range-normalised by the power of two nearest 1 / rms(a0).
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

namespace {

constexpr int F_NT = 256;                      // 4 waves, one per SIMD
constexpr int F_L1 = 64 + 8;                   // residual-conv weight row (halfs)
constexpr int F_ROWB = 72;                     // mask-conv chunk row: 32 k halfs (64 B) + 8 B pad: 18-bank stride, the 8-byte fragment reads are conflict-free
constexpr int F_PART = 256 * F_ROWB;
constexpr int F_BUF = 2 * F_PART;
constexpr size_t F_LDS = (size_t)2 * 256 * F_L1 * 2 + (size_t)2 * F_BUF + 2 * 256 * 4 + 16;

typedef unsigned long long u64_;

// (the residual conv's and the taps GEMM's small accumulators are VGPR-form matrix instructions written by hand: pipe_helpers.h)

// Diagnostic build only (tools/bench_s3f.hip defines S3F_STAMP and a global s3f_stamps buffer): s_memtime at the phase boundaries of the
// first tiles of workgroup 0.
#ifdef S3F_STAMP
__device__ unsigned s3f_stamps[16 * 32];
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(F_NT) void tail_s3t_kernel(TailS3Args a, int ntiles, int tps) {
    constexpr int L1 = F_L1;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* W1h = reinterpret_cast<_Float16*>(smem);  // [256][L1]
    _Float16* W1l = W1h + 256 * L1;
    unsigned char* Wb = reinterpret_cast<unsigned char*>(W1l + 256 * L1);  // [2 buffers][hi|lo][256 co][F_ROWB]
    float* b1 = reinterpret_cast<float*>(Wb + 2 * F_BUF);                  // residual-conv bias
    float* bs = b1 + 256;                                                  // mask-conv bias
    int* s_next = reinterpret_cast<int*>(bs + 256);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const half8* s1 = reinterpret_cast<const half8*>(a.w1_16);  // [2 chunks][hi|lo][256][32]
        for (int i = tid; i < 2 * 2 * 256 * 4; i += F_NT) {
            const int pc = i & 3, co = (i >> 2) & 255, part = (i >> 10) & 1, chunk = i >> 11;
            *reinterpret_cast<half8*>((part ? W1l : W1h) + co * L1 + chunk * 32 + pc * 8) = s1[i];
        }
        b1[tid] = a.b1[tid];
        bs[tid] = a.bias[tid];
    }
    const float slope = a.slope[0];
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs, CS4_ = CS * 4u;
    const int lastw = (cdiv(P, 64) - 1) * 64;
    // weight images through buffer descriptors too: lane offset in one VGPR, everything else constant (flat addressing keeps a 64-bit
    // address per piece alive across the tile loop: 40 register pairs, spilled)
    const __amdgpu_buffer_rsrc_t ws = rsrc_of(reinterpret_cast<const float*>(a.w16));   // mask conv: [chunk 8][hi|lo][256 co][4 pieces of 8 k]
    const __amdgpu_buffer_rsrc_t ts = rsrc_of(reinterpret_cast<const float*>(a.w16b));  // taps: [m 4][part 2][s 2][hi|lo][32 taps][16 k]
    const __amdgpu_buffer_rsrc_t is = rsrc_of(reinterpret_cast<const float*>(a.enc_img));  // encoder fragments (enc_stats_kernel)
    const unsigned voffT = (unsigned)(r * 2 + h) * 16u;
    const unsigned voffB = ((unsigned)(8 * h) * CS + 2u * r) * 4u;
    const unsigned voffC = ((unsigned)(4 * h) * CS + 2u * r) * 4u;

    // mask-conv weight chunk c -> LDS buffer by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass).  A DMA
    // instruction writes 1 KB of LDS linearly (lane x 16 bytes), so the image in memory already has the LDS image's 72-byte rows
    // (enc_stats_kernel's extra block row pads the pack's 64-byte rows, kernels.h EncPadJobs): a chunk is 36 KB = 36 pieces, 9 per wave.
    // The pieces count on vmcnt like loads; `__syncthreads()` waits for them (the compiler knows an LDS write is pending).
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned voffL = (unsigned)lane * 16u;
    int wave1k = wave * 1024;  // (laundered per tile below: the 72 piece offsets of a tile are loop invariants, and hoisted they spill 120 SGPRs)
    auto stage_dma_piece = [&](int c, int buf, int j) {
        const int po = j * 4096 + wave1k;  // uniform: piece j * 4 + wave
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_ptr)(Wb + buf * F_BUF + po), 16, voffL, c * F_BUF + po, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) stage_dma_piece(0, 0, j);


    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (tid == 0) s_next[it & 1] = (a.tile_ctr ? (int)atomicAdd(a.tile_ctr, 1u) : tile) + (int)gridDim.x;
#ifdef S3F_STAMP
        unsigned stamp[32];
#endif
        STAMP(0);
        unsigned CS4 = CS4_;
        int nlive = a.cout_live;
        asm volatile("" : "+s"(CS4), "+s"(nlive), "+s"(wave1k));  // row offsets are formed where they are used (one s_mul each): hoisted out of the tile loop they cost 300 SGPRs
        const int b = tile / tps;
        // a wave segment past the sample's end repeats the last real one (same values to the same addresses): every wave takes part in every barrier
        const int wp0 = min((tile - b * tps) * (F_NT / 64 * 64) + wave * 64, lastw);
        const __amdgpu_buffer_rsrc_t xs = rsrc_of(a.x + (size_t)b * 64 * CS + wp0);
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.res + (size_t)b * 256 * CS + wp0);
        // ---- residual rows of 32-channel tile m, one tile ahead
        f32x2 R[16];
        auto load_res = [&](int m, f32x2 (&Rb)[16]) {
#pragma unroll
            for (int q = 0; q < 16; ++q) Rb[q] = ld2(rs, voffC, (unsigned)(m * 32 + (q & 3) + 8 * (q >> 2)) * CS4);
        };
        // ---- B fragments of the residual conv (expanded: 64 channels of this lane's two pixels)
        half8 xh[4][2], xl[4][2];
        {
            f32x2 v[4][8];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[ks][j] = ld2(xs, voffB, (unsigned)(ks * 16 + j) * CS4);
            load_res(0, R);
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
        // The encoder output a0 (the complex product's other factor) is not read: its tiles are rebuilt on the matrix cores from this lane's
        // nine spectrogram taps (pipe_helpers.h), requested under the last chunk's MFMAs together with the first tiles' encoder fragments
        f32x2 V[9];
        half8 ea[8];  // encoder A fragments of tiles m (real part) and m + 4 (imaginary part): [tile][K step][hi|lo]
        auto load_ea = [&](int m) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ea[i] = ld_h8(is, voffT, (unsigned)(m * 4 + i) * 1024u);
                ea[4 + i] = ld_h8(is, voffT, (unsigned)((m + 4) * 4 + i) * 1024u);
            }
        };
        STAMP(1);
        f32x16 acc[8][2];  // mask conv: [output tile][pixel slot]
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[m][sl][q] = 0.f;
        // one K chunk: residual-conv tile kc -> refined tile -> PReLU -> B fragments -> 96 MFMAs of the mask conv
        auto chunk = [&](int kc, f32x2 (&Rb)[16], int buf) {
            f32x16 acc1[2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(W1h + (kc * 32 + r) * L1 + ks * 16 + 8 * h);
                const half8 al = *reinterpret_cast<const half8*>(W1l + (kc * 32 + r) * L1 + ks * 16 + 8 * h);
                if (ks == 0) {
                    mfma_v0(acc1[0], ah, xh[0][0]);
                    mfma_v0(acc1[1], ah, xh[0][1]);
                } else {
                    mfma_v(acc1[0], ah, xh[ks][0]);
                    mfma_v(acc1[1], ah, xh[ks][1]);
                }
                mfma_v(acc1[0], ah, xl[ks][0]);
                mfma_v(acc1[1], ah, xl[ks][1]);
                mfma_v(acc1[0], al, xh[ks][0]);
                mfma_v(acc1[1], al, xh[ks][1]);
            }
            mfma_v_fence(acc1[0], acc1[1]);
            if (kc == 3) STAMP(20);
            // refined = conv + bias + residual; the mask head's PReLU; split: registers 8s .. 8s+7 of the tile = K step s of this chunk
            half8 bh[2][2], bl[2][2];  // [K step][pixel slot]
            const int cob = kc * 32 + 4 * h;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float y0[8], y1[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 kb = *reinterpret_cast<const f32x4*>(b1 + cob + 8 * (2 * s + g));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * g + i, q = 8 * s + j;
                        y0[j] = preluf_(fmaf(acc1[0][q], WINV, kb[i]) + Rb[q].x, slope);
                        y1[j] = preluf_(fmaf(acc1[1][q], WINV, kb[i]) + Rb[q].y, slope);
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
            if (kc == 3) STAMP(21);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (kc == 3) STAMP(22);  // chunk kc staged (buffer `buf`) and visible; everyone is done reading chunk kc - 1 (the other buffer)
            // the next chunk's residual rows: requested BEHIND the barrier (in front of it the barrier's wait for the weight DMA would also wait
            // for them); the next weight chunk - chunk 0 again behind chunk 7: the next tile's first - follows piece by piece in the matrix phase
            if (kc + 1 < 8) load_res(kc + 1, Rb);
            if (kc == 7) {
                patch_load(spec_rsrc(a.spec + (size_t)b * 2 * P, P), wp0 + 2 * r, h, P, a.F, V);
                load_ea(0);
            }
            // A fragments in accumulator-register K order: K slot (h, j) of step s is channel 16 s + 4 h + (j & 3) + 8 (j >> 2) of the chunk, i.e.
            // elements 4h .. 4h+3 of piece 2s and of piece 2s + 1
            const unsigned char* wb = Wb + buf * F_BUF + r * F_ROWB + 8 * h;
            auto afrag = [&](int m, int s, int part) {
                const u64_* p = reinterpret_cast<const u64_*>(wb + part * F_PART + m * 32 * F_ROWB + s * 32);
                u64_ v[2] = {p[0], p[2]};  // +16 bytes: the next piece
                return __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(v));
            };
            half8 ah[2][2], al[2][2];  // [buffer][K step]
            ah[0][0] = afrag(0, 0, 0); ah[0][1] = afrag(0, 1, 0);
            al[0][0] = afrag(0, 0, 1); al[0][1] = afrag(0, 1, 1);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m + 1 < 8) {  // tile m + 1's fragments are requested before tile m's twelve MFMAs; nothing moves across a tile
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
                    if (2 * m + s < 9) stage_dma_piece((kc + 1) & 7, buf ^ 1, 2 * m + s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int kc = 0; kc < 8; kc += 2) {
            chunk(kc, R, 0);
            STAMP(2 + kc);
            chunk(kc + 1, R, 1);
            STAMP(3 + kc);
        }
        // ---- mask, complex product with the encoder output, taps GEMM (k_pwr.hip PWR_S3T), eight encoder rows (one K step of the taps GEMM) at a time
        float esc = 1.0f, eisc = WINV;
        if (a.stats) rms_pow2(a.stats + 2 * b, a.inv_count, esc, eisc);  // power of two nearest 1 / rms(a0) of this mixture (wave-uniform)
        PatchFrag pf;
        patch_build(V, wp0 + 2 * r, a.T, a.F, P, esc, pf);
        f32x16 acc2[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc2[sl][q] = 0.f;
        f32x16 eR[2], eI[2];  // a0 x esc x 256 of tiles m / m + 4, per pixel slot
        auto group = [&](int grp) {
            const int m = grp >> 1, s2 = grp & 1;
            // taps image: [m 4][part 2][s 2][hi|lo][32 taps][16 k] halfs -> 16-byte piece index ((idx*2 + hl)*32 + r)*2 + h
            const int idx_r = (m * 2 + 0) * 2 + s2, idx_i = (m * 2 + 1) * 2 + s2;
            const half8 trh = ld_h8(ts, voffT, (unsigned)(idx_r * 2 + 0) * 1024u), trl = ld_h8(ts, voffT, (unsigned)(idx_r * 2 + 1) * 1024u);
            const half8 tih = ld_h8(ts, voffT, (unsigned)(idx_i * 2 + 0) * 1024u), til = ld_h8(ts, voffT, (unsigned)(idx_i * 2 + 1) * 1024u);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {  // one pixel slot at a time: half the temporaries (the encoder-row ring needs the registers)
                float o_r[8], o_i[8];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c4 = m * 32 + 4 * h + 8 * (2 * s2 + g);
                    const f32x4 br = *reinterpret_cast<const f32x4*>(bs + c4), bi = *reinterpret_cast<const f32x4*>(bs + c4 + 128);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 4 * g + i, q = 8 * s2 + j;
                        const float mr = fmaxf(fmaf(acc_rd(acc[m][sl][q]), WINV, br[i]), 0.f), mi = fmaxf(fmaf(acc_rd(acc[m + 4][sl][q]), WINV, bi[i]), 0.f);
                        const float er = eR[sl][q] * WINV, ei = eI[sl][q] * WINV;  // (already normalised: the patches carry esc)
                        o_r[j] = er * mr - ei * mi;
                        o_i[j] = er * mi + ei * mr;
                    }
                }
                unsigned rh[4], rl[4], ih[4], il[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    split2(o_r[2 * jp], o_r[2 * jp + 1], rh[jp], rl[jp]);
                    split2(o_i[2 * jp], o_i[2 * jp + 1], ih[jp], il[jp]);
                }
#define S3F_H8(x) __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(x))
                mfma_v(acc2[sl], trh, S3F_H8(rh));
                mfma_v(acc2[sl], trh, S3F_H8(rl));
                mfma_v(acc2[sl], trl, S3F_H8(rh));
                mfma_v(acc2[sl], tih, S3F_H8(ih));
                mfma_v(acc2[sl], tih, S3F_H8(il));
                mfma_v(acc2[sl], til, S3F_H8(ih));
#undef S3F_H8
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // encoder conv of channels 32 m .. (real) and 128 + 32 m .. (imaginary) at this wave's 64 pixels: K = 2 steps, f16x3
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                mfma_v0(eR[sl], ea[0], pf.h[0][sl]);
                mfma_v0(eI[sl], ea[4], pf.h[0][sl]);
                mfma_v(eR[sl], ea[0], pf.l[0][sl]);
                mfma_v(eI[sl], ea[4], pf.l[0][sl]);
                mfma_v(eR[sl], ea[1], pf.h[0][sl]);
                mfma_v(eI[sl], ea[5], pf.h[0][sl]);
                mfma_v(eR[sl], ea[2], pf.h[1][sl]);
                mfma_v(eI[sl], ea[6], pf.h[1][sl]);
                mfma_v(eR[sl], ea[2], pf.l[1][sl]);
                mfma_v(eI[sl], ea[6], pf.l[1][sl]);
                mfma_v(eR[sl], ea[3], pf.h[1][sl]);
                mfma_v(eI[sl], ea[7], pf.h[1][sl]);
            }
            mfma_v_fence4(eR[0], eR[1], eI[0], eI[1]);
            if (m + 1 < 4) load_ea(m + 1);  // (the fragments are free again; they land under the two groups below)
            __builtin_amdgcn_sched_barrier(0);
            group(2 * m);
            STAMP(10 + 2 * m);
            group(2 * m + 1);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(11 + 2 * m);
        }
        mfma_v_fence(acc2[0], acc2[1]);
        const __amdgpu_buffer_rsrc_t zs = rsrc_of(a.z + (size_t)b * a.cout_live * CS + wp0);  // z (B, 18, cs)
#pragma unroll
        for (int q = 0; q < 12; ++q) {  // tap rows (q & 3) + 8 (q >> 2) + 4 h; q >= 12 would be taps 24 ..: never live
            const int tap = (q & 3) + 8 * (q >> 2) + 4 * h;
            if (tap < nlive) st2(zs, voffC, (unsigned)((q & 3) + 8 * (q >> 2)) * CS4, f32x2{acc2[0][q] * eisc, acc2[1][q] * eisc});
        }
        STAMP(18);
        __syncthreads();  // everyone is done with this tile (s_next slot; the weight buffers are ordered by the chunk barriers)
        STAMP(19);
#ifdef S3F_STAMP
        if (blockIdx.x == 0 && tid == 0 && it < 16)
            for (int i = 0; i < 24; ++i) s3f_stamps[it * 32 + i] = stamp[i];
#endif
        tile = s_next[it & 1];
    }
}

}  // namespace

// RTFS_ERR_ARG = the call does not qualify (contiguous rows, tiny input): the caller runs the two separate kernels
int launch_tail_s3t(const TailS3Args& a, int B, hipStream_t st) {
    if (a.cs <= 0 || (a.cs & 63) || a.cs < (a.P + 63) / 64 * 64 || a.P < 64 || !a.w16b || a.cout_live > 24 || !a.spec || !a.enc_img || a.T * a.F != a.P) return RTFS_ERR_ARG;
    if ((size_t)256 * a.cs * 4 >= ((size_t)1 << 31)) return RTFS_ERR_ARG;
    if (rtfs_set_max_lds((const void*)tail_s3t_kernel, F_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
    const int tps = cdiv(a.P, F_NT / 64 * 64), ntiles = tps * B;
    const int grid = ntiles < 256 ? ntiles : 256;  // one resident workgroup per CU
    hipLaunchKernelGGL(tail_s3t_kernel, dim3(grid), dim3(F_NT), F_LDS, st, a, ntiles, tps);
    return rtfs_launch_status();
}
