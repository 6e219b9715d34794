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
    // The pieces count on vmcnt like loads; the barrier that publishes a chunk is preceded by an explicit `s_waitcnt vmcnt(0)` (the compiler's
    // own wait for a pending LDS write is missing at a loop header, see below).
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
        // (exact size: the pipeline below requests the rows of "chunks 8 and 9" behind the last one - out of range, they return 0 without a memory access)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res + (size_t)b * 256 * CS + wp0), 0, (int)((256u * CS - (unsigned)wp0) * 4u), 0x00020000);
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
        // ---- software pipeline over the eight K chunks (the recipe of k_bnh.hip): while the 96 matrix instructions of chunk kc issue, chunk
        // kc + 1 is prepared in their shadow.  Its residual-conv tile was computed in FRONT of the previous barrier (24 matrix instructions the
        // pipe works on while the four waves meet); bias + residual + PReLU + f16 split run in 48 slices of 3-4 VALU instructions, one per gap
        // between two matrix instructions, writing a second set of B fragments; every residual row is re-requested (two chunks ahead) by the
        // slice that consumed it; the weight chunk's nine DMA pieces sit in gaps too.  A chunk cost 6.4 k cycles with the phases in sequence
        // (1.0 k residual GEMM, 1.5 k VALU, 3.7 k matrix phase).
        half8 bh[2][2][2], bl[2][2][2];  // [chunk parity][K step][pixel slot]
        f32x16 acc1[2];
        float y[4];
        f32x4 kk[2];  // residual-conv bias of four channels, read one group ahead
        unsigned hh[2][4], ll[2][4];
        const float pc1 = 0.5f * (1.0f + slope), pc2 = 0.5f * (1.0f - slope);  // PReLU(x) = pc1 x + pc2 |x| (two instructions, any slope)
        auto gemm1 = [&](int c) {  // residual conv of channels 32 c .. (weights resident in LDS): K = 64, f16x3
            const int row = ((c & 7) * 32 + r) * L1 + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(W1h + row + ks * 16);
                const half8 al = *reinterpret_cast<const half8*>(W1l + row + ks * 16);
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
        };
        auto side_consts = [&](int c, int gi) {  // channels 32 c + 4 h + 8 gi ..
            kk[gi & 1] = *reinterpret_cast<const f32x4*>(b1 + (c & 7) * 32 + 4 * h + 8 * gi);
        };
        // slice `step` (0 .. 47) of chunk c.  Per group of four channels gi = 2 s + g (12 slices; the group's bias is live in kk[gi & 1] for all of
        // them, the next group's is requested by the first): pixel slot 0's four values, its two splits, slot 1's four values (each re-requests
        // its residual row for chunk `creload` once both slots have read it), its two splits
        auto side = [&](int c, int step, int creload) {
            const int gi = step / 12, s = gi >> 1, g = gi & 1, u = step % 12, sl = u / 6, v = u % 6;
            if (v < 4) {
                const int q = 8 * s + 4 * g + v;
                if (u == 0 && gi < 3) side_consts(c, gi + 1);
                const float x = fmaf(acc1[sl][q], WINV, kk[gi & 1][v]) + (sl ? R[q].y : R[q].x);
                y[v] = fmaf(pc2, __builtin_fabsf(x), pc1 * x);
                if (sl == 1) R[q] = ld2(rs, voffC, (unsigned)(creload * 32 + (q & 3) + 8 * (q >> 2)) * CS4);
            } else {
                const int jl = v - 4, jp = 2 * g + jl;  // pair (4 g + 2 jl, 4 g + 2 jl + 1) of K step s
                split2(y[2 * jl], y[2 * jl + 1], hh[sl][jp], ll[sl][jp]);
                if (g == 1 && jl == 1) {
                    bh[c & 1][s][sl] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hh[sl]));
                    bl[c & 1][s][sl] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(ll[sl]));
                }
            }
        };
        gemm1(0);
        side_consts(0, 0);
        mfma_v_fence(acc1[0], acc1[1]);
#pragma unroll
        for (int step = 0; step < 48; ++step) side(0, step, 1);
        gemm1(1);
        side_consts(1, 0);
#pragma unroll 2
        for (int kc = 0; kc < 8; ++kc) {
            const int buf = kc & 1;
            __builtin_amdgcn_sched_barrier(0);
            // chunk kc's DMA pieces must have landed before the barrier publishes them.  The compiler inserts this wait only where it sees the
            // DMA on a straight path: at the loop header - pieces issued on the back edge - it emitted none, and one run in three a mixture
            // came out different from its batch-1 run.  (It also drains the residual rows requested three output tiles ago: they are due anyway.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // chunk kc staged (buffer `buf`) and visible; everyone is done reading chunk kc - 1 (the other buffer)
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
                if (m + 1 < 8) {  // tile m + 1's fragments are requested before tile m's twelve MFMAs
                    ah[(m + 1) & 1][0] = afrag(m + 1, 0, 0); ah[(m + 1) & 1][1] = afrag(m + 1, 1, 0);
                    al[(m + 1) & 1][0] = afrag(m + 1, 0, 1); al[(m + 1) & 1][1] = afrag(m + 1, 1, 1);
                }
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    const int s = t / 6, sl = t & 1, v = (t % 6) >> 1;  // products hi*hi, hi*lo, lo*hi of K step s, the two pixel slots alternating
                    acc[m][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v == 2 ? al[m & 1][s] : ah[m & 1][s], v == 1 ? bl[buf][s][sl] : bh[buf][s][sl], acc[m][sl], 0, 0, 0);
                    if (m >= 1 && m <= 4) side(kc + 1, (m - 1) * 12 + t, kc + 2);  // (chunk 8's slices work on chunk 0's constants and are dropped)
                    if ((m == 0 || m == 5) && t % 3 == 0) stage_dma_piece((kc + 1) & 7, buf ^ 1, (m == 0 ? 0 : 4) + t / 3);
                    if (m == 6 && t == 0) stage_dma_piece((kc + 1) & 7, buf ^ 1, 8);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // chunk kc + 2's residual-conv tile, issued BEFORE the barrier; its first reader is a slice 12+ matrix instructions behind the barrier
            gemm1(kc + 2);
            side_consts(kc + 2, 0);
            STAMP(2 + kc);
        }
        mfma_v_fence(acc1[0], acc1[1]);  // the dropped tile of "chunk 9" is still being written: nothing may move into its registers yet (tools/isa_check.py)
        // the epilogue's spectrogram taps and first encoder fragments (the residual-conv operand is dead now: their registers)
        patch_load(spec_rsrc(a.spec + (size_t)b * 2 * P, P), wp0 + 2 * r, h, P, a.F, V);
        load_ea(0);
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
