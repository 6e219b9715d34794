// Head of the fused separator in ONE kernel (padded channel rows only; api.hip separator_part routes here):
//   a0       = Conv2d(2->256, 3x3)(spectrogram)                   TDAVNet/encoder.py:146-157   (rebuilt tile by tile on the matrix cores, never stored)
//   a1       = conv1x1_256->256(ReLU(gLN(a0))) + bias            TDAVNet/av_model.py audio bottleneck (gLN -> ReLU -> Conv)   (kept: every block adds it)
//   residual = PReLU(dw1x1(a1))                                   separators/tdanet.py:106  (gateway of the first block application)
//   x_enc    = conv1x1_256->64(residual) + bias                   separators/tdanet.py:107  (projection)
// Until round 3 this was two launches (pwr_kernel<BN> 0.55 ms + pws_head4_kernel 0.51 ms) with a1 written and read back in between.  Here
// the bottleneck GEMM is K-streaming like k_s3f.hip: its whole 256-channel x 64-pixel output tile lives in 256 accumulator registers
// (AGPRs; 4 waves x 512 registers, one wave per SIMD), the K axis advances one 32-channel chunk of a0 (= one weight chunk, staged through
// LDS, double buffered, one barrier per 96 MFMAs) at a time; the epilogue walks the eight 32-channel output tiles: a1 written through, the
// gateway's PReLU output written through and - split into f16 hi / lo - fed back to the matrix cores as the B operand of the projection
// (weights resident in LDS with their K axis in accumulator-register order, as in k_b2b.hip).  Two pixels per lane everywhere: 8-byte
// accesses of whole 128-byte lines.  Per pixel the kernel reads 18 spectrogram taps and writes 576 floats, where the three kernels it
// replaces (encoder conv, bottleneck, block head) read 530 and wrote 832: it is a pure write stream.  The encoder conv is a K = 18 (padded
// to 32) GEMM per 32-channel tile - 12 matrix instructions - whose accumulator tile, normalised (the gLN statistics of a0 come from
// enc_stats_kernel, which does not form a0 either), ReLU'd and split, is the B operand of the bottleneck GEMM as it stands: the bottleneck
// weights' A fragments are read in accumulator-register K order (two 8-byte reads from adjacent 16-byte pieces, as in k_s3f.hip).
// The small accumulators (projection: 64 registers) are VGPR-form matrix instructions written by hand: see k_s3f.hip.
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

namespace {

constexpr int N_NT = 256;                      // 4 waves, one per SIMD
constexpr int N_ROWB = 72;                     // bottleneck chunk row: 32 k halfs (64 B) + 8 B pad: 18-bank stride, the 8-byte fragment reads are conflict-free
constexpr int N_PART = 256 * N_ROWB;
constexpr int N_BUF = 2 * N_PART;
constexpr int N_L2 = 256 + 8;                  // projection weight row (halfs)
constexpr size_t N_LDS = (size_t)2 * N_BUF + (size_t)2 * 64 * N_L2 * 2 + (size_t)(5 * 256 + 64) * 4 + 16;
typedef unsigned long long u64_;


// Diagnostic build only (tools/bench_bnh.hip defines BNH_STAMP): s_memtime at the phase boundaries of workgroup 0's first tiles.
#ifdef BNH_STAMP
__device__ unsigned bnh_stamps[16 * 32];
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <bool WRES>  // write the gateway output (false: the first block boundary forms it from a1, k_b2b.hip AM bit 2)
__global__ __launch_bounds__(N_NT) void bn_head_kernel(BnHeadArgs a, int ntiles, int tps) {
    constexpr int L2 = N_L2;
    constexpr float WINV = 1.0f / 256.0f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS map: the resident tables first (every constant offset from a lane base fits the 16-bit DS offset field), the chunk buffers last
    _Float16* W2h = reinterpret_cast<_Float16*>(smem);  // [64][L2], K in accumulator order
    _Float16* W2l = W2h + 64 * L2;
    float* sc = reinterpret_cast<float*>(W2l + 64 * L2);  // gLN fold of the current mixture (scale x the patch normalisation's inverse / 256)
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
    const float pc1 = 0.5f * (1.0f + slope), pc2 = 0.5f * (1.0f - slope);
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs, CS4_ = CS * 4u;
    const int lastw = (cdiv(P, 64) - 1) * 64;
    const __amdgpu_buffer_rsrc_t ws = rsrc_of(reinterpret_cast<const float*>(a.w16));  // bottleneck: [chunk 8][hi|lo][256 co][4 pieces of 8 k]
    const __amdgpu_buffer_rsrc_t is = rsrc_of(reinterpret_cast<const float*>(a.enc_img));  // encoder fragments (enc_stats_kernel)
    const unsigned voffT = (unsigned)(r * 2 + h) * 16u;
    const int T = a.T, F = a.F;
    const unsigned voffC = ((unsigned)(4 * h) * CS + 2u * r) * 4u;

    // lane byte offsets into LDS, opaque to the compiler: every access is one of these + a constant in the instruction's offset field (left
    // alone, it materialises one address register per distinct constant - 150 of them - and spills them before the tile loop)
    unsigned lw2 = (unsigned)(r * L2 + 8 * h) * 2u;                                  // projection fragments
    unsigned lc4 = OFF_C + 16u * h;                                                  // per-channel constants, accumulator order
    unsigned lwr0 = OFF_W + (unsigned)(r * N_ROWB + 8 * h), lwr1 = lwr0 + N_BUF;     // bottleneck fragments, buffer 0 / 1
    asm volatile("" : "+v"(lw2), "+v"(lc4), "+v"(lwr0), "+v"(lwr1));
    // bottleneck weight chunk c -> LDS buffer by LDS-DMA (`buffer_load_dwordx4 ... lds`; k_s3f.hip has the details): the image in memory already
    // has the 72-byte rows (enc_stats_kernel's extra block row writes it); a chunk is 36 pieces of 1 KB, 9 per wave; the barrier that publishes
    // a chunk is preceded by an explicit wait (the compiler's own is missing at a loop header)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned voffL = (unsigned)lane * 16u;
    int wave1k = wave * 1024;  // (laundered per tile: the piece offsets are loop invariants and spill as SGPRs when hoisted)
    auto stage_dma_piece = [&](int c, int buf, int j) {
        const int po = j * 4096 + wave1k;  // uniform: piece j * 4 + wave
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_ptr)(smem + OFF_W + buf * N_BUF + po), 16, voffL, c * N_BUF + po, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) stage_dma_piece(0, 0, j);
    // Phase stagger: every workgroup does the same work per tile, so without it all 256 CUs write their tiles' 576 rows at the same time and
    // then leave the memory system idle through the next K loop (the first output tile of an epilogue took 8-13 k cycles, the others 3 k).
    // Workgroups start a quarter of a tile apart in four groups; the tile counter keeps them apart.
    if (a.stagger)
        for (int i = 0; i < (int)((blockIdx.x >> 3) & 3) * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);

    // this lane's nine spectrogram taps (two pixels each) of the tile: carried across tiles, the next tile's are requested BEFORE this tile's
    // 576 row stores (behind them the first loads of a tile waited 13k cycles of a 97k-cycle tile)
    f32x2 V[9];
    int cur_b = -1;
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (tid == 0) s_next[it & 1] = (a.tile_ctr ? (int)atomicAdd(a.tile_ctr, 1u) : tile) + (int)gridDim.x;
#ifdef BNH_STAMP
        unsigned stamp[32];
#endif
        STAMP(0);
        unsigned CS4 = CS4_;
        asm volatile("" : "+s"(CS4), "+s"(wave1k));  // row offsets are formed where they are used (one s_mul each), not hoisted out of the tile loop
        const int b = tile / tps;
        // a wave segment past the sample's end repeats the last real one (same values to the same addresses): every wave takes part in every barrier
        const int wp0 = min((tile - b * tps) * (N_NT / 64 * 64) + wave * 64, lastw);
        const int p0 = wp0 + 2 * r;
        float esc, eisc;
        rms_pow2(a.stats + 2 * b, a.inv_count, esc, eisc);
        if (it == 0) patch_load(spec_rsrc(a.spec + (size_t)b * 2 * P, P), p0, h, P, F, V);  // (later tiles: requested before the previous tile's epilogue)
        if (b != cur_b) {     // block-uniform: the gLN fold of this mixture; the scale also undoes the patch normalisation and the weights' x 256
            __syncthreads();  // the previous tile's fragments are built
            float fs, ft;
            gln_fold(a.stats + 2 * b, a.inv_count, a.gamma[tid], a.beta[tid], fs, ft);
            sc[tid] = fs * eisc;
            sh[tid] = ft;
            cur_b = b;
        }
        __syncthreads();  // sc / sh (and, first tile, the resident tables) visible
        PatchFrag pf;
        patch_build(V, p0, T, F, P, esc, pf);
        // encoder A fragments of 32-channel tile c: [K step][hi|lo] (L2-resident 32 KB image), requested one chunk before their use
        half8 ea[4];
        auto load_ea_piece = [&](int c, int i) { ea[i] = ld_h8(is, voffT, (unsigned)(c * 4 + i) * 1024u); };
        auto load_ea = [&](int c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_ea_piece(c, i);
        };
        load_ea(0);
        STAMP(1);
        f32x16 acc[8][2];  // bottleneck conv: [output tile][pixel slot]
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[m][sl][q] = 0.f;
        // ---- software pipeline over the eight K chunks: while the 96 matrix instructions of chunk kc issue, chunk kc + 1 is prepared in their
        // shadow - its encoder tile (12 matrix instructions, at the head of the phase), then gLN + ReLU + f16 split in slices of 4-6 VALU
        // instructions, one slice per gap between two matrix instructions (one wave per SIMD: nothing else would fill the gaps), then the
        // staging writes of the next weight chunk.  A chunk cost 5.5 k cycles with the phases in sequence (0.5 k encoder GEMM, 0.9 k VALU,
        // 3.5 k matrix instructions, 0.5 k staging writes).
        half8 bh[2][2][2], bl[2][2][2];  // [chunk parity][K step][pixel slot]
        f32x16 acc1[2];
        float y0[8], y1[8];
        f32x4 kk[2][2];  // gLN scale / shift of four channels, [group parity][scale | shift]: read one group ahead (a wait for them also waits
                         // for the fragment reads queued in front: 4 x ~250 exposed cycles per chunk when they were read at their use)
        unsigned h0[4], l0[4], h1[4], l1[4];
        auto gemm1 = [&]() {  // encoder conv of the tile whose fragments are in ea: K = 2 steps, f16x3
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                mfma_v0(acc1[sl], ea[0], pf.h[0][sl]);
                mfma_v(acc1[sl], ea[0], pf.l[0][sl]);
                mfma_v(acc1[sl], ea[1], pf.h[0][sl]);
                mfma_v(acc1[sl], ea[2], pf.h[1][sl]);
                mfma_v(acc1[sl], ea[2], pf.l[1][sl]);
                mfma_v(acc1[sl], ea[3], pf.h[1][sl]);
            }
        };
        auto side_consts = [&](int c, int gi) {  // channels c*32 + 4 h + 8 gi ..
            const unsigned co = (unsigned)(c * 32 + 8 * gi) * 4u;
            kk[gi & 1][0] = *reinterpret_cast<const f32x4*>(smem + lc4 + co);
            kk[gi & 1][1] = *reinterpret_cast<const f32x4*>(smem + lc4 + 1024 + co);
        };
        // slice `step` (0 .. 31) of chunk c's gLN + ReLU + split: registers 8s .. 8s+7 of the encoder tile = K step s of the chunk (accumulator
        // order).  A slice is 3-4 VALU instructions: what fits beside one matrix instruction without delaying the next
        auto side = [&](int c, int step) {
            const int s = step >> 4, u = step & 15;
            if (u < 8) {
                const int g = u >> 2, i = u & 3, j = 4 * g + i, q = 8 * s + j, gi = 2 * s + g;
                if (i == 0 && gi < 3) side_consts(c, gi + 1);
                y0[j] = fmaxf(fmaf(acc1[0][q], kk[gi & 1][0][i], kk[gi & 1][1][i]), 0.f);
                y1[j] = fmaxf(fmaf(acc1[1][q], kk[gi & 1][0][i], kk[gi & 1][1][i]), 0.f);
            } else {
                const int jp = (u - 8) >> 1;
                if (u & 1) split2(y1[2 * jp], y1[2 * jp + 1], h1[jp], l1[jp]);
                else split2(y0[2 * jp], y0[2 * jp + 1], h0[jp], l0[jp]);
                if (u == 15) {
                    bh[c & 1][s][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h0));
                    bl[c & 1][s][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l0));
                    bh[c & 1][s][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h1));
                    bl[c & 1][s][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l1));
                }
            }
        };
        gemm1();
        side_consts(0, 0);
        mfma_v_fence(acc1[0], acc1[1]);
        load_ea(1);
#pragma unroll
        for (int step = 0; step < 32; ++step) side(0, step);
        gemm1();  // chunk 1's encoder tile (its slices run in the shadow of chunk 0's matrix instructions)
        side_consts(1, 0);
        // (two chunks per trip - the buffer parity is static - and four trips: fully unrolled the kernel was 80 KB of code, more than the
        // instruction cache two CUs share; the work for "chunk 8" behind the last one is done on wrapped indices and dropped: no branches)
#pragma unroll 2
        for (int kc = 0; kc < 8; ++kc) {
            const int buf = kc & 1;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces of chunk kc have landed
            __syncthreads();  // chunk kc staged (buffer `buf`) and visible; everyone is done reading chunk kc - 1 (the other buffer)
            // (the 12 loads of a chunk - next weight chunk, next encoder fragments - are spread over the matrix-instruction gaps below: issued
            // together behind the barrier by all four waves they queued on the CU's one address unit for ~800 cycles, a fifth of a chunk)
#ifdef BNH_STAMP
            if (kc == 2) STAMP(20);
#endif
            // A fragments in accumulator-register K order: K slot (h, j) of step s is channel 16 s + 4 h + (j & 3) + 8 (j >> 2) of the chunk, i.e.
            // elements 4h .. 4h+3 of piece 2s and of piece 2s + 1
            auto afrag = [&](int m, int s, int part) {
                const u64_* pp = reinterpret_cast<const u64_*>(smem + (buf ? lwr1 : lwr0) + part * N_PART + m * 32 * N_ROWB + s * 32);
                u64_ v[2] = {pp[0], pp[2]};  // +16 bytes: the next piece
                return __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(v));
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
                for (int t = 0; t < 12; ++t) {
                    const int s = t / 6, sl = t & 1, v = (t % 6) >> 1;  // products hi*hi, hi*lo, lo*hi of K step s, the two pixel slots alternating
                    acc[m][sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v == 2 ? al[m & 1][s] : ah[m & 1][s], v == 1 ? bl[buf][s][sl] : bh[buf][s][sl], acc[m][sl], 0, 0, 0);
                    if (m >= 1 && (m - 1) * 12 + t < 32) side((kc + 1) & 7, (m - 1) * 12 + t);
                    if ((m == 0 || m == 4) && t % 3 == 0) stage_dma_piece((kc + 1) & 7, buf ^ 1, (m >> 2) * 4 + t / 3);  // chunk 0 again behind chunk 7: the next tile's first
                    if (m == 5 && t == 0) stage_dma_piece((kc + 1) & 7, buf ^ 1, 8);
                    if (m == 3 && t >= 8) load_ea_piece((kc + 2) & 7, t - 8);
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef BNH_STAMP
                if (kc == 2 && m < 7) STAMP(21 + m);
#endif
            }
            // chunk kc + 2's encoder tile, issued BEFORE the barrier: the matrix pipe works on it while the four waves meet.  Its first reader
            // is a slice 12+ matrix instructions behind the barrier (no wait states needed); chunk kc + 1's slices are through with acc1.
            gemm1();
            side_consts((kc + 2) & 7, 0);
            STAMP(2 + kc);
        }
        mfma_v_fence(acc1[0], acc1[1]);  // the dropped encoder tile of "chunk 9" is still being written: nothing may move into its registers yet (tools/isa_check.py)
        // ---- epilogue: a1 and the gateway output written through, projection from the accumulator registers
        const __amdgpu_buffer_rsrc_t a1s = rsrc_of(a.a1 + (size_t)b * 256 * CS + wp0);
        const __amdgpu_buffer_rsrc_t rs = rsrc_of((WRES ? a.res : a.a1) + (size_t)b * 256 * CS + wp0);
        const __amdgpu_buffer_rsrc_t xes = rsrc_of(a.xenc + (size_t)b * 64 * CS + wp0);
        const int ntile = s_next[it & 1];  // (written before this tile's first barrier)
        if (ntile < ntiles) {              // block-uniform
            const int nb = ntile / tps;
            const int nwp0 = min((ntile - nb * tps) * (N_NT / 64 * 64) + wave * 64, lastw);
            patch_load(spec_rsrc(a.spec + (size_t)nb * 2 * P, P), nwp0 + 2 * r, h, P, F, V);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc2[2][2];  // [projection tile][pixel slot]
        // The epilogue is a software pipeline too: block k = (output tile m, K step s of the projection) = 16 accumulator values per pixel
        // slot.  While the 12 projection matrix instructions of block k issue, block k + 1 is built in their gaps: eight value slices (a1 =
        // acc / 256 + bias written through, gateway, PReLU) and four split slices, one per gap; the per-channel constants of a group of four
        // channels are read one group ahead.  In sequence the 192 matrix instructions of a tile added 6 k cycles to 20 k of VALU and stores.
        half8 pbh[2][2], pbl[2][2];  // [block parity][pixel slot]
        float ey0[8], ey1[8];
        f32x4 ek[2][3];  // [group parity][bottleneck bias | gateway scale | gateway bias]
        unsigned eh0[4], el0[4], eh1[4], el1[4];
        auto econsts = [&](int gg) {  // group gg = 2 k + g: channels 32 (gg >> 2) + 4 h + 8 (gg & 3) ..
            const unsigned co = (unsigned)((gg >> 2) * 32 + 8 * (gg & 3)) * 4u;
            ek[gg & 1][0] = *reinterpret_cast<const f32x4*>(smem + lc4 + 2048 + co);
            ek[gg & 1][1] = *reinterpret_cast<const f32x4*>(smem + lc4 + 3072 + co);
            ek[gg & 1][2] = *reinterpret_cast<const f32x4*>(smem + lc4 + 4096 + co);
        };
        auto ebuild = [&](int k, int step) {
            const int m = k >> 1, sq = k & 1;
            if (step < 8) {
                const int g = step >> 2, i = step & 3, j = 4 * g + i, q = 8 * sq + j, gg = 2 * k + g;
                if (i == 1 && gg + 1 < 32) econsts(gg + 1);
                const unsigned ro = (unsigned)(m * 32 + (q & 3) + 8 * (q >> 2)) * CS4;
                const float v0 = fmaf(acc_rd(acc[m][0][q]), WINV, ek[gg & 1][0][i]), v1 = fmaf(acc_rd(acc[m][1][q]), WINV, ek[gg & 1][0][i]);
                st2(a1s, voffC, ro, f32x2{v0, v1});
                const float t0 = fmaf(v0, ek[gg & 1][1][i], ek[gg & 1][2][i]), t1 = fmaf(v1, ek[gg & 1][1][i], ek[gg & 1][2][i]);
                ey0[j] = fmaf(pc2, __builtin_fabsf(t0), pc1 * t0);  // PReLU(t) = pc1 t + pc2 |t| (one instruction fewer than compare + select)
                ey1[j] = fmaf(pc2, __builtin_fabsf(t1), pc1 * t1);
                if (WRES) st2(rs, voffC, ro, f32x2{ey0[j], ey1[j]});
            } else {
                const int jp = step - 8;
                split2(ey0[2 * jp], ey0[2 * jp + 1], eh0[jp], el0[jp]);
                split2(ey1[2 * jp], ey1[2 * jp + 1], eh1[jp], el1[jp]);
                if (jp == 3) {
                    pbh[k & 1][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(eh0));
                    pbl[k & 1][0] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(el0));
                    pbh[k & 1][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(eh1));
                    pbl[k & 1][1] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(el1));
                }
            }
        };
        econsts(0);
#pragma unroll
        for (int step = 0; step < 12; ++step) ebuild(0, step);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            half8 ph, pl;
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const int m2 = t / 6, v = (t % 6) >> 1, sl = t & 1;  // products hi*hi, hi*lo, lo*hi of projection tile m2, the two pixel slots alternating
                if (t % 6 == 0) {
                    const unsigned wo = (unsigned)(m2 * 32 * L2 + k * 16) * 2u;  // K step 2 m + s = k
                    ph = *reinterpret_cast<const half8*>(smem + lw2 + wo);
                    pl = *reinterpret_cast<const half8*>(smem + lw2 + 64 * L2 * 2 + wo);
                }
                const half8 af = v == 2 ? pl : ph, bf = v == 1 ? pbl[k & 1][sl] : pbh[k & 1][sl];
                if (k == 0 && v == 0) mfma_v0(acc2[m2][sl], af, bf);
                else mfma_v(acc2[m2][sl], af, bf);
                if (k + 1 < 16) ebuild(k + 1, t);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k & 1) STAMP(10 + (k >> 1));
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
        STAMP(18);
        __syncthreads();  // everyone is done with this tile (s_next slot; sc / sh)
        STAMP(19);
#ifdef BNH_STAMP
        if (blockIdx.x == 0 && tid == 0 && it < 16)
            for (int i = 0; i < 28; ++i) bnh_stamps[it * 32 + i] = stamp[i];
#endif
        tile = ntile;
    }
}

}  // namespace

bool launch_bn_head_qualifies(const BnHeadArgs& a) {
    if (a.cs <= 0 || (a.cs & 63) || a.cs < (a.P + 63) / 64 * 64 || a.P < 64 || !a.w16 || !a.w2_16 || !a.stats || !a.spec || !a.enc_img || a.T * a.F != a.P) return false;
    return (size_t)256 * a.cs * 4 < ((size_t)1 << 31);
}

// RTFS_ERR_ARG = the call does not qualify (contiguous rows, tiny input): the caller runs the separate kernels
int launch_bn_head(const BnHeadArgs& a, int B, hipStream_t st) {
    if (!launch_bn_head_qualifies(a)) return RTFS_ERR_ARG;
    const int tps = cdiv(a.P, N_NT / 64 * 64), ntiles = tps * B;
    const int grid = ntiles < 256 ? ntiles : 256;  // one resident workgroup per CU
    if (a.res) {
        if (rtfs_set_max_lds((const void*)bn_head_kernel<true>, N_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
        hipLaunchKernelGGL(bn_head_kernel<true>, dim3(grid), dim3(N_NT), N_LDS, st, a, ntiles, tps);
    } else {
        if (rtfs_set_max_lds((const void*)bn_head_kernel<false>, N_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
        hipLaunchKernelGGL(bn_head_kernel<false>, dim3(grid), dim3(N_NT), N_LDS, st, a, ntiles, tps);
    }
    return rtfs_launch_status();
}
