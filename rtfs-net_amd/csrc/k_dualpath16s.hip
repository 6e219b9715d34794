// Fused dual-path SRU sweep, generation 3: the generation-2 dataflow (k_dualpath16.hip) cut for TWO co-resident workgroups per CU.
// DualPathRNN.forward, reference src/models/layers/rnn_layers.py:136-162, with the third-party sru.SRU cell (call site
// rnn_layers.py:99-105,150): LayerNorm over channels, Unfold(8) as an addressing mode, four bidirectional SRU layers (f16x3
// split-precision MFMA GEMM + in-register recurrence), ConvTranspose1d, bias, residual - one launch per sweep.
//
// Why a third generation: the generation-2 workgroup (512 threads, 155 KB of LDS, 219 VGPRs) is alone on its CU, so nothing runs
// under its serial phases - the recurrences (24 % / 35 % of the F / T sweep), the load + LayerNorm phase, the epilogue, the
// barrier waits of the weight stream (profiles/r01_sweep_stamps.txt).  Here a workgroup is 256 threads (one wave per SIMD) that
// owns HALF as many sequences and needs < 80 KB of LDS, so two independent workgroups share every CU and every SIMD: while one
// wave walks its recurrence (38 cycles a step alone, issue-bound: tools/chain_rate.hip) its neighbour from the other workgroup has
// the matrix pipe.  What had to change for the 80 KB:
//   * the weight stream is double-buffered in K = 16 steps of 16 KB (was K = 32, 2 x 40 KB of padded rows).  The pack carries a
//     second copy of the f16 hi / lo images in FRAGMENT ORDER ([K step][direction][gate tile][hi|lo][lane] x 16 bytes,
//     packing.frag_image_gate / frag_image_ct): staging a step is a straight 16 KB copy (four coalesced 16-byte loads per thread,
//     two steps ahead in registers, then four ds_write_b128) and every fragment read is 64 consecutive 16-byte pieces - conflict
//     free without padding or swizzling.  (Tried and measured on the way, tools/sweep_stamps.py, cycles per K step of a workgroup
//     alone on its CU against 768 of MFMA issue: the same steps written by LDS-DMA 1420 - a DMA piece costs the issuing wave
//     60+ cycles and the barrier drains it; B fragments straight from L2 to registers with no LDS at all 1063 alone but 1750
//     with the second workgroup present - the L1 delivers ~35 B/clk/CU of 16-byte fragment loads, four waves x 8 KB per step
//     saturate it.)
//   * F sweep (Ls <= 64): 2 sequences per workgroup as one PAIR - accumulator register q of lane half h is time step q of
//     sequence h, wave = (time part of 32 steps, direction);
//   * T sweep (Ls <= 128): 1 sequence per workgroup, wave = (time part of 64 steps, direction): the A-operand rows of a tile are
//     ordered so that lane half h holds 16 CONSECUTIVE steps (16 h + q); the halves take turns 16 steps at a time (one
//     v_permlane32_swap per hand-off), the resting half masked by EXEC;
//   * Ls <= 256 (the 4 s time sweep): the same single-sequence program with FOUR time parts - 512 threads, 102 KB, one workgroup per CU.
// Both variants keep the generation-2 tricks: only the cell-state chain c_t = u0 + (c_{t-1} - u0) sigmoid(u1 + v_f c_{t-1}) is
// serial, the reset gate / highway output are evaluated after the hand-off; sigmoid = rcp(1 + exp2(z)) with -log2(e) folded into
// the gate weights; the highway input of layers 1-3 comes out of the same MFMAs through an identity block in the weight image (whose
// exact zeros are not multiplied: kstep's N3).  The write-back after the hand-off is branch-free with immediate row offsets (see there).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

#define HLD 136         // activation row (halfs): [hi 64 | lo 64 | pad 8] = 272 bytes = 68 dwords (4 mod 32): conflict-free b128 fragment reads,
#define HLO 64          // and the lo half of an element sits a constant 128 bytes behind its hi half (one address, two immediates)
#define PADR 8          // rows in front of a sequence's plane that the backward write-back may overshoot into
#define WBUF 16384      // one staged K step of the weight stream (bytes)
#define WINV (1.0f / 256.0f)
#ifndef DP16S_SGB
#define DP16S_SGB 1
#endif
#ifndef DP16S_WBG
#define DP16S_WBG 4     // write-back steps evaluated together (stage by stage); 8 costs the same per step but skips less behind the sequence end
#endif
#ifndef DP16S_SGB_CT
#define DP16S_SGB_CT 1  // the conv-transpose taps pinned the same way
#endif

namespace {

__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 hh = (_Float16)v[i];
        hi[i] = hh;
        unsigned lb;  // (f16)(v - hh) in one instruction (the compiler emits convert, subtract, convert)
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v[i]), "v"(hh));
        lo[i] = __builtin_bit_cast(_Float16, (unsigned short)lb);
    }
}

// element at (wave-uniform base) + (32-bit lane BYTE offset): global_load / global_store in their saddr + voffset form, no 64-bit vector
// address arithmetic and no 64-bit address registers (the launcher checks that the whole tensor spans < 4 GB); the empty asm keeps the
// zero-extension next to the access
__device__ __forceinline__ float ldo(const float* __restrict__ base, unsigned boff) {
    asm volatile("" : "+v"(boff));
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ void sto(float* __restrict__ base, unsigned boff, float v) {
    asm volatile("" : "+v"(boff));
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v;
}

// sigmoid with the -log2(e) factor already folded into z
__device__ __forceinline__ float sig2(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }

// value held by lane half `ph` (0 = lanes 0-31, 1 = lanes 32-63), in both halves
__device__ __forceinline__ float take_half(float v, int ph) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(ph == 0 ? r[0] : r[1]);
}
__device__ __forceinline__ float sum_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace

// NT = row tiles per wave.  NT = 2: 4 waves (2 time parts x 2 directions), 128 accumulators per wave, <= 256 registers: one wave of the
// workgroup per SIMD.  NT = 1: 8 waves (4 time parts x 2 directions), 64 accumulators, <= 128 registers: two waves of the workgroup per SIMD
// and, with the CU's second workgroup, four per SIMD.
// NP = time parts per workgroup (default 4 / NT: 128 covered steps unpaired, 64 paired).  NT = 2 with NP = 4 is the LONG variant for the 4 s
// shapes (Ls <= 256: one sequence, 4 time parts x 2 directions = 512 threads, 102 KB of LDS, ONE workgroup per CU with two of its own waves per
// SIMD): the same program, the chain handed through four parts.
template <int NSEQ, bool PAIRED, int NT, bool STAMP = false, int NP = 4 / NT>
__global__ __launch_bounds__(128 * NP, NT == 2 ? 2 : 4) void dp16s_kernel(Dp16Args a) {
    static_assert((NSEQ == 2 && PAIRED) || (NSEQ == 1 && !PAIRED), "F sweep: one sequence pair; T sweep: one sequence");
    static_assert(NT == 1 || NT == 2, "row tiles per wave");
    static_assert(NP == 2 || NP == 4, "time parts");
    constexpr int STEPS = (PAIRED ? 16 : 32) * NT;  // time steps covered by one wave
    constexpr int NPART = NP;                       // time parts per workgroup
    constexpr int NTHR = 128 * NPART;               // 2 directions x NPART waves
    constexpr int NPIECE = 1024 / NTHR;             // 16-byte pieces of a staged K step per thread
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int Ls = a.Ls, L = Ls - 7, rowsH = Ls + 1 + PADR;  // per sequence: PADR scratch rows (-PADR .. -1), rows 0 .. Ls - 1, row Ls all zero (conv-transpose borders)
    half8* Wst = reinterpret_cast<half8*>(smem);                          // [2 buffers][1024 pieces of 16 B], fragment order
    _Float16* Hh = reinterpret_cast<_Float16*>(smem + 2 * WBUF) + PADR * HLD;  // row 0 of sequence 0; [NSEQ][rowsH][HLD]: hi halves of a row, then its lo halves
    _Float16* Hl = Hh + HLO;
    float* chand = reinterpret_cast<float*>(Hh + (NSEQ * rowsH - PADR) * HLD);  // [NSEQ][2 dirs][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int part = wave >> 1, dir = wave & 1;           // time part MAJOR (the two waves of a part run their chains together)
    const int seq = PAIRED ? h : 0;                       // the sequence this LANE's accumulator registers belong to
    constexpr int CPART = PAIRED ? NPART / 2 : NPART;     // conv-transpose roles: (sequence, co tile, part of 32 NT positions); STEPS NPART positions in all
    const int cseq = wave / (2 * CPART), ccot = (wave / CPART) & 1, cpart = wave % CPART;
    const int n0 = blockIdx.x * NSEQ;

    int nstamp = 0;
    auto stamp = [&]() {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (tid == 0 && nstamp < 16) a.stamps[(size_t)blockIdx.x * 16 + nstamp] = t;
            ++nstamp;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stamp_at = [&](int slot) {  // diagnostic sub-stamps of layer 1's scan (slots 12-15, thread 0 = time part 0, forward)
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (tid == 0) a.stamps[(size_t)blockIdx.x * 16 + slot] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stamp();  // 0: start
    auto seq_base = [&](int s) {
        int n = n0 + s;
        n = n < a.nseq ? n : a.nseq - 1;
        return (size_t)(n / a.R) * a.bstride + (size_t)(n % a.R) * a.rstride;
    };

    // ---------------- weight stream.  Global images in fragment order, 1024 pieces (16 KB) per K step:
    //   gate step:  piece ((dir * 4 + m) * 2 + part) * 64 + lane      conv-transpose tap: piece ((co tile * 4 + ks) * 2 + part) * 64 + lane
    // Staging copies the step linearly: thread tid moves pieces tid + 256 j.  `pre` holds the step after the one in LDS.
    // Two register sets: `pre[s]` holds stream step g + 1 + s while step g is multiplied.  Under full load an L2 round trip of this stream
    // takes longer than one K step (the issue-time stamps showed ~760 ticks of a 1400-tick step waiting for `pre` with one set), two steps
    // cover it.  All phases have an even number of steps, so set = step parity and the sets alternate without moves.
    half8 pre[2][NPIECE];
    auto stage_load = [&](auto set_c, const half8* __restrict__ step) {
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) pre[S][j] = step[tid + NTHR * j];
    };
    auto stage_write = [&](auto set_c, int buf) {
        constexpr int S = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) Wst[buf * 1024 + tid + NTHR * j] = pre[S][j];
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    stage_load(S0, a.wf_l0);  // step 0 of layer 0: in flight under the load + LayerNorm phase

    // ---------------- phase 0: load rows, LayerNorm over channels, split to f16 planes (normalizations.py:33-37).
    // lane = (position, channel half): 32 consecutive positions x 2 halves per wave, the halves meet in one permlane swap
    if (wave * 32 < NSEQ * Ls) {  // (uniform; 32 rows a wave: Ls <= 64 paired, <= 128 / 256 unpaired - the launcher routes by Ls)
        const int task = wave * 32 + r, ntask = NSEQ * Ls;
        const bool live = task < ntask;
        const int tk = live ? task : ntask - 1;
        const int s = (NSEQ == 2 && tk >= Ls) ? 1 : 0, pos = tk - s * Ls;
        const unsigned xoff = (unsigned)(seq_base(s) + pos + (size_t)(32 * h) * a.cstride) * 4u;
        float v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = ldo(a.x + (size_t)c * a.cstride, xoff);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) sum += v[c];
        const float mean = sum_halves(sum) * (1.0f / 64);
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float d = v[c] - mean;
            var = fmaf(d, d, var);
        }
        const float rstd = 1.0f / sqrtf(sum_halves(var) * (1.0f / 64) + RTFS_EPS);
        _Float16* dh = Hh + (s * rowsH + pos) * HLD + 32 * h;
        _Float16* dl = Hl + (s * rowsH + pos) * HLD + 32 * h;
        if (live) {
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = fmaf((v[c8 * 8 + i] - mean) * rstd, a.ln_gamma[32 * h + c8 * 8 + i], a.ln_beta[32 * h + c8 * 8 + i]);
                half8 hi, lo;
                split8(y, hi, lo);
                *reinterpret_cast<half8*>(dh + c8 * 8) = hi;
                *reinterpret_cast<half8*>(dl + c8 * 8) = lo;
            }
        }
        if (tid < NSEQ * 16) {  // the zero rows (hi + lo: 128 halfs = 16 x 16 B each)
            half8 z;
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.f;
            const int zs = tid / 16, pc = tid % 16;
            *reinterpret_cast<half8*>(Hh + (zs * rowsH + Ls) * HLD + pc * 8) = z;
        }
    }

    // activation rows of this wave's two row tiles (virtual time tau; the backward direction reads position L-1-tau).
    // A-operand row r of tile t: PAIRED   -> sequence (r >> 2) & 1, step 16 t + (r & 3) + 4 (r >> 3)
    //                            unpaired -> step 32 t + 16 ((r >> 2) & 1) + (r & 3) + 4 (r >> 3)
    // either way accumulator register q of lane half h (row (q & 3) + 8 (q >> 2) + 4 h) is step q of that half's 16-step run
    int rowbase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int rs = PAIRED ? (r >> 2) & 1 : 0;
        int tau = STEPS * part + (r & 3) + 4 * (r >> 3) + (PAIRED ? 16 * t : 32 * t + 16 * ((r >> 2) & 1));
        tau = tau < L ? tau : L - 1;
        rowbase[t] = (rs * rowsH + (dir ? L - 1 - tau : tau)) * HLD + 8 * h;
    }
    stamp();  // 1: after load + LN issue (before first barrier)
    stage_write(S0, 0);
    stage_load(S0, a.wf_l0 + 1024);      // step 1 -> set 0 (written to LDS during step 0)
    stage_load(S1, a.wf_l0 + 2 * 1024);  // step 2 -> set 1 (written during step 1)
    __syncthreads();  // activation planes filled, step 0 staged

    int g = 0;  // steps consumed so far: step g sits in buffer g & 1, steps g + 1 and g + 2 in the register sets (g + 1) in set g & 1
    // ---------------- four SRU layers
    // (layer 0 - K = 512, 32 steps in a loop - and layers 1-3 - K = 64, four unrolled steps with the identity tile's zeros skipped - are two
    // instantiations of the layer body: as two branches inside ONE loop body the register allocator spilled around both)
    auto do_layer = [&](const int layer, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int nchunk = FIRST ? 32 : 4;
        const float vf = a.wc16[layer * 128 + dir * 32 + r], vr = a.wc16[layer * 128 + 64 + dir * 32 + r];
        const float bf = a.bias16[layer * 128 + dir * 32 + r] * 256.f, br = a.bias16[layer * 128 + 64 + dir * 32 + r] * 256.f;
        f32x16 acc[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[t][0][q] = 0.f;
                acc[t][1][q] = bf;
                acc[t][2][q] = br;
                acc[t][3][q] = 0.f;
            }
        const half8* const gsrc = FIRST ? a.wf_l0 : a.wf_l + (size_t)(layer - 1) * 4 * 1024;
        // what follows this layer in the stream: the next layer's steps, then the conv-transpose's taps
        const half8* const gnext = layer < 3 ? a.wf_l + (size_t)layer * 4 * 1024 : a.wf_ct;
        // (MFMA order: tools/mfma_rate.hip measures 32.5 ticks per v_mfma_f32_32x32x16_f16 whether 1, 2, 4 or 8 accumulators rotate - a
        // dependent chain costs nothing, so the order of the three split-precision terms is free; term-major is kept, it is harmless.)
        // `tiles` consecutive gate tiles from m0, term-major: NT = 2 issues pairs (4 accumulators), NT = 1 all four gate tiles at once
        auto gate_tiles = [&](int m0, auto ntile_c, const half8 (&ah)[NT], const half8 (&al)[NT], const half8* __restrict__ wb) {
            constexpr int NM = decltype(ntile_c)::value;
            half8 bh[NM], bl[NM];
#pragma unroll
            for (int mm = 0; mm < NM; ++mm) {
                bh[mm] = wb[(m0 + mm) * 128];
                bl[mm] = wb[(m0 + mm) * 128 + 64];
            }
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int mm = 0; mm < NM; ++mm) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t][m0 + mm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[t] : ah[t], term == 1 ? bl[mm] : bh[mm], acc[t][m0 + mm], 0, 0, 0);
                }
            }
        };
        // One K step: fragments of step g from buffer g & 1, 12 NT MFMAs; in the middle step g + 1 leaves its register set for the other buffer
        // (every wave left it at the barrier that ended step g - 1) and step g + 3 is requested into the freed set; one barrier.
        // (Cycles per step of a 4-wave workgroup alone on its CU, 768 of them MFMA issue: ~1350; variants measured on the way, none better
        // alone: LDS-DMA staging 1420; all fragment reads pinned to the top of the step 1350; a mid-step barrier with the next step's first
        // fragments requested behind it 1360-1490 (+36 registers); gate tile 3 deferred across the barrier 1650.)
        // N3 = MFMAs of gate tile 3 in this step.  Layer 0: 6 like every tile.  Layers 1-3: tile 3 is the highway input, an IDENTITY block
        // of the weight image (packing._dualpath_parts: 256 at k = 32 dir + j, exactly representable, lo half zero): only the two K steps
        // that hold this direction's 32 input channels contribute - stream steps 0, 1 for both directions, the backward one's steps are
        // rotated - and only through the two terms with the hi weights: 4 MFMAs there, none in steps 2, 3 (the skipped products are
        // exact zeros): 80 instead of 96 MFMAs per layer and wave.
        auto kstep = [&](int q, auto set_c, auto n3_c) {
            constexpr int N3 = (NT == 2 && DP16S_SGB) ? decltype(n3_c)::value : 6;
            // (layers 1-3: the backward direction's K steps come rotated by two, packing.frag_image_gate_rot)
            const int aoff = FIRST ? (q >> 2) * HLD + (q & 3) * 16 : ((q + 2 * dir) & 3) * 16;
            half8 ah[NT], al[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ah[t] = *reinterpret_cast<const half8*>(Hh + rowbase[t] + aoff);
                al[t] = *reinterpret_cast<const half8*>(Hl + rowbase[t] + aoff);
            }
            const half8* wb = Wst + (g & 1) * 1024 + dir * 512 + lane;
            if (NT == 2 && DP16S_SGB) {
                // Hand-pinned issue order (sched_group_barrier): the wave's 12 fragment reads, 4 staging writes and 4 prefetch loads are
                // spread over the gaps between its 24 MFMAs instead of clustered in front of them (an MFMA holds the issue port for 8 of
                // its 32 cycles): only the first six reads and the barrier stay exposed.
                half8 bh[4], bl[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m < 3 || N3 > 0) bh[m] = wb[m * 128];
                    if (m < 3 || N3 == 6) bl[m] = wb[m * 128 + 64];
                }
                auto tile6 = [&](int m) {
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[t] : ah[t], term == 1 ? bl[m] : bh[m], acc[t][m], 0, 0, 0);
                };
                tile6(0);
                tile6(1);
                stage_write(set_c, (g + 1) & 1);
                tile6(2);
                stage_load(set_c, q + 3 < nchunk ? gsrc + (size_t)(q + 3) * 1024 : gnext + (size_t)(q + 3 - nchunk) * 1024);
                if (N3 == 6) tile6(3);
                if (N3 == 4) {
#pragma unroll
                    for (int term = 0; term < 3; term += 2)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[t][3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[t] : ah[t], bh[3], acc[t][3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // A (4) + B tile 0 (2)
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // tile 0: the reads of tile 1 in its first gaps
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // tile 1: reads of tile 2, then the staging writes
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                constexpr int R3 = N3 == 6 ? 2 : (N3 == 4 ? 1 : 0);  // fragment reads of tile 3
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // tile 2: reads of tile 3, then the prefetch loads
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < R3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                if (N3 > 0) __builtin_amdgcn_sched_group_barrier(0x008, N3, 0);
            } else if (NT == 2) {
                gate_tiles(0, std::integral_constant<int, 2>(), ah, al, wb);
                stage_write(set_c, (g + 1) & 1);
                stage_load(set_c, q + 3 < nchunk ? gsrc + (size_t)(q + 3) * 1024 : gnext + (size_t)(q + 3 - nchunk) * 1024);
                gate_tiles(2, std::integral_constant<int, 2>(), ah, al, wb);
            } else {
                stage_write(set_c, (g + 1) & 1);
                stage_load(set_c, q + 3 < nchunk ? gsrc + (size_t)(q + 3) * 1024 : gnext + (size_t)(q + 3 - nchunk) * 1024);
                gate_tiles(0, std::integral_constant<int, 4>(), ah, al, wb);
            }
            __syncthreads();  // step g consumed by every wave, step g + 1 visible
            ++g;
        };
        const std::integral_constant<int, 6> N6;
        const std::integral_constant<int, 4> N4;
        const std::integral_constant<int, 0> N0;
        if (FIRST) {
            for (int q = 0; q < nchunk; q += 2) {
                kstep(q, S0, N6);
                kstep(q + 1, S1, N6);
            }
        } else {
            kstep(0, S0, N4);
            kstep(1, S1, N4);
            kstep(2, S0, N0);
            kstep(3, S1, N0);
        }
        // (that barrier also means: every wave has finished reading the activation planes - the scan may overwrite them in place)
        stamp();  // 2,4,6,8: GEMM of layer done
        // undo the 2^8 weight prescale.  Time part 0 is on the critical path: only what its chain reads (gates 0, 1) now, the rest in
        // front of its write-back; the later parts scale everything while they wait for the hand-off.  (The empty asm pins the products
        // here: the compiler sinks the multiplies into the serialised sections otherwise.)
        auto unscale = [&](int m0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = m0; m < m0 + 2; ++m) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[t][m][q] *= WINV;
                    asm volatile("" : "+v"(acc[t][m]));
                }
        };
        unscale(0);
        if (part != 0) unscale(2);
        __builtin_amdgcn_sched_barrier(0);
        if (layer == 1) stamp_at(12);  // prescale undone
        float cin[NT];  // c_{t-1} of register 0 of tile t (this lane's run)
#pragma unroll
        for (int t = 0; t < NT; ++t) cin[t] = 0.f;
        for (int hp = 0; hp < NPART; ++hp) {
            if (part == hp) {
                float c = hp > 0 ? chand[(seq * 2 + dir) * 32 + r] : 0.f;
                if (PAIRED) {
                    // register q of tile t = time step 16 t + q of this lane's own sequence.  Only the cell-state chain is serial; it
                    // overwrites u0 in place.  The reset gate and the hidden output depend on c_{t-1}, c_t but nothing depends on
                    // them: they are evaluated after the hand-off below, concurrently with the next time part's chain.
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        cin[t] = c;
#pragma unroll
                        for (int q4 = 0; q4 < 16; q4 += 4) {
                            if (STEPS * hp + 16 * t + q4 < L) {  // (uniform: groups of four steps behind the sequence end are not walked)
#pragma unroll
                                for (int q = q4; q < q4 + 4; ++q) {
                                    const float u0 = acc[t][0][q];
                                    const float f = sig2(fmaf(vf, c, acc[t][1][q]));
                                    c = fmaf(c - u0, f, u0);
                                    acc[t][0][q] = c;
                                }
                            }
                        }
                    }
                    chand[(seq * 2 + dir) * 32 + r] = c;
                } else {
                    // the two lane halves hold steps 16 h + q of a tile: they take turns, 16 steps each
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            float cr = c;
                            if (h == ph) {  // (an exec mask, not a select per step: the other half's registers stay as they are)
                                cin[t] = c;
#pragma unroll
                                for (int q4 = 0; q4 < 16; q4 += 4) {
                                    if (STEPS * hp + 32 * t + 16 * ph + q4 < L) {  // (uniform)
#pragma unroll
                                        for (int q = q4; q < q4 + 4; ++q) {
                                            const float u0 = acc[t][0][q];
                                            const float f = sig2(fmaf(vf, cr, acc[t][1][q]));
                                            cr = fmaf(cr - u0, f, u0);
                                            acc[t][0][q] = cr;
                                        }
                                    }
                                }
                            }
                            c = take_half(cr, ph);
                        }
                    }
                    if (h == 0) chand[dir * 32 + r] = c;
                }
            }
            if (layer == 1 && hp == 0) stamp_at(13);  // part 0's chain done
            if (hp + 1 < NPART) __syncthreads();  // cell state published: the next time part starts while this one writes back
            if (layer == 1 && hp == 0) stamp_at(14);
            if (part == hp) {
                if (hp == 0) unscale(2);
                // deferred reset gate + highway: h = x' + (c_t - x') r(c_{t-1}); hidden outputs (this wave's direction half of the
                // channels) go back into the planes in place: 9 vector instructions + 2 two-byte stores a step, no branch per step (that
                // made every step its own basic block - exp, rcp, converts and stores at their full latencies, ~125 cycles a step).
                // Addresses in BYTES: one lane base + an IMMEDIATE per step (the direction is a compile-time constant inside `wb`: per-step scalar
                // row arithmetic - add, min, multiply, all dependent - cost 20 of a step's 85 cycles, tools/chain_rate.hip).  WBG steps at a
                // time, stage by stage (a step is a chain of dependent instructions with two transcendentals in it); a group is skipped when
                // its first step lies beyond the sequence end, the up to WBG - 1 steps behind the end inside a group land in rows L .. L + WBG - 2
                // (forward; free since layer 0's GEMM) or in the PADR rows in front of the plane (backward).
                const int tau0 = STEPS * part;
                char* const Hb = reinterpret_cast<char*>(Hh);
                constexpr int ROWB = HLD * 2, TSTEP = PAIRED ? 16 : 32, KMAX = (NT - 1) * TSTEP + 15, WBG = DP16S_WBG;
                static_assert(WBG <= PADR && WBG <= 8 && 16 % WBG == 0, "overshoot of a group: at most WBG - 1 rows, 7 free rows behind the sequence end");
                const int lane_t0 = PAIRED ? tau0 : tau0 + 16 * h;  // first step of this lane's run in tile 0
                const int col = (seq * rowsH * HLD + dir * 32 + r) * 2;
                auto wb = [&](auto bw_c) {
                    constexpr bool BW = decltype(bw_c)::value;
                    int base = col + (BW ? L - 1 - lane_t0 - KMAX : lane_t0) * ROWB;
                    asm volatile("" : "+v"(base));  // opaque per layer
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int q4 = 0; q4 < 16; q4 += WBG) {
                            if (lane_t0 + TSTEP * t + q4 < L) {
                                float z[WBG], d[WBG], hv[WBG];
                                _Float16 hh[WBG];
                                unsigned lo[WBG];
#pragma unroll
                                for (int i = 0; i < WBG; ++i) {
                                    const int q = q4 + i;
                                    z[i] = fmaf(vr, q == 0 ? cin[t] : acc[t][0][q - 1], acc[t][2][q]);
                                    d[i] = acc[t][0][q] - acc[t][3][q];
                                }
#pragma unroll
                                for (int i = 0; i < WBG; ++i) z[i] = __builtin_amdgcn_exp2f(z[i]);
#pragma unroll
                                for (int i = 0; i < WBG; ++i) z[i] = 1.0f + z[i];
#pragma unroll
                                for (int i = 0; i < WBG; ++i) z[i] = __builtin_amdgcn_rcpf(z[i]);
#pragma unroll
                                for (int i = 0; i < WBG; ++i) {
                                    hv[i] = fmaf(d[i], z[i], acc[t][3][q4 + i]);
                                    hh[i] = (_Float16)hv[i];
                                }
#pragma unroll
                                for (int i = 0; i < WBG; ++i)  // (f16)(hv - hh) in one instruction
                                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo[i]) : "v"(hv[i]), "v"(hh[i]));
#pragma unroll
                                for (int i = 0; i < WBG; ++i) {
                                    const int k = TSTEP * t + q4 + i;
                                    char* const o = Hb + base + (BW ? KMAX - k : k) * ROWB;
                                    *reinterpret_cast<_Float16*>(o) = hh[i];
                                    *reinterpret_cast<unsigned short*>(o + HLO * 2) = (unsigned short)lo[i];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                if (dir)
                    wb(std::true_type());
                else
                    wb(std::false_type());
            }
        }
        if (layer == 1) stamp_at(15);  // part 0's write-back done and the loop's last barrier passed: waiting for part 1's write-back
        __syncthreads();  // all hidden outputs of this layer are in the planes
        stamp();  // 3,5,7,9: scan of layer done
    };
    do_layer(0, std::true_type());
    for (int layer = 1; layer < 4; ++layer) do_layer(layer, std::false_type());

    // ---------------- ConvTranspose1d(64->64, k=8) + bias + residual (rnn_layers.py:153-156), transposed:
    //   y[co][t] = bt[co] + sum_{kk,ci} Wt[co][kk*64+ci] * H[t-kk][ci];  wave = (sequence, co tile, position part)
    {
        // one accumulator per position tile (a dependent MFMA chain issues at the full rate, tools/mfma_rate.hip: separate partial sums
        // per k-step parity bought nothing and cost 32 registers)
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
        // the residual rows of the epilogue are requested now and arrive under the GEMM (clamped addresses: dead
        // sequences / positions read a valid element that is never stored)
        // channel co = 32 ccot + (q & 3) + 8 (q >> 2) + 4 h: the q part of the address is wave-uniform (scalar base), the rest one lane offset
        float res[NT][16];
        const size_t rbase = seq_base(cseq) + (size_t)(ccot * 32) * a.cstride;  // uniform
        unsigned eoff[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) eoff[t] = (unsigned)(rbase + (size_t)(4 * h) * a.cstride + min(32 * NT * cpart + 32 * t + r, Ls - 1)) * 4u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cq = (q & 3) + 8 * (q >> 2);
            const float btq = a.bt[ccot * 32 + cq + 4 * h];
#pragma unroll
            for (int t = 0; t < NT; ++t) res[t][q] = ldo(a.x + (size_t)cq * a.cstride, eoff[t]) + btq;  // residual + conv-transpose bias, fetched under the GEMM
        }
        __builtin_amdgcn_sched_barrier(0);  // (the residual requests stay out of the taps' pinned schedule)
        // tap 0 is staged in buffer g & 1, taps 1 and 2 in the register sets; 8 taps of 64 k' each
        auto tap = [&](int q, auto set_c) {
            int hrow[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p = 32 * NT * cpart + 32 * t + r - q;
                hrow[t] = (cseq * rowsH + ((p >= 0 && p < L) ? p : Ls)) * HLD + 8 * h;
            }
            const half8* wb = Wst + (g & 1) * 1024 + ccot * 512 + lane;
            // all fragment reads of the tap first in program order, in the order they are needed (the staging write below may alias the
            // weight reads as far as the compiler knows, so it can only be scheduled behind the last of them); two k steps x two position
            // tiles = four accumulators.  Term order (wh xh), (wl xh), (wh xl): xh and wl are dead after eight of a block's twelve MFMAs,
            // so the next block's fragments arrive into their registers (with xl last the pinned order needed 18 live fragments and spilled)
            half8 xh[4][NT], xl[4][NT], wh[4], wl[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; k2 += 2) {
#pragma unroll
                for (int kk = k2; kk < k2 + 2; ++kk) wh[kk] = wb[kk * 128];
#pragma unroll
                for (int kk = k2; kk < k2 + 2; ++kk)
#pragma unroll
                    for (int t = 0; t < NT; ++t) xh[kk][t] = *reinterpret_cast<const half8*>(Hh + hrow[t] + kk * 16);
#pragma unroll
                for (int kk = k2; kk < k2 + 2; ++kk) wl[kk] = wb[kk * 128 + 64];
#pragma unroll
                for (int kk = k2; kk < k2 + 2; ++kk)
#pragma unroll
                    for (int t = 0; t < NT; ++t) xl[kk][t] = *reinterpret_cast<const half8*>(Hl + hrow[t] + kk * 16);
            }
            stage_write(set_c, (g + 1) & 1);  // (the last tap rewrites tap 7's neighbour with a clamped copy: never read)
#pragma unroll
            for (int k2 = 0; k2 < 4; k2 += 2) {
#pragma unroll
                for (int term = 0; term < 3; ++term) {
#pragma unroll
                    for (int kk = k2; kk < k2 + 2; ++kk)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 1 ? wl[kk] : wh[kk], term == 2 ? xl[kk][t] : xh[kk][t], acc[t], 0, 0, 0);
                }
            }
            stage_load(set_c, a.wf_ct + (size_t)(q + 3 < 8 ? q + 3 : 7) * 1024);  // unconditional, clamped
            if (NT == 2 && DP16S_SGB_CT) {
                // issue order of a tap pinned like a gate step's: 24 fragment reads (a tap reads twice a gate step's - each fragment feeds
                // six MFMAs, not twelve), 4 staging writes and 4 prefetch loads in the gaps between the 24 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // k2 = 0: hi fragments
#pragma unroll
                for (int i = 0; i < 18; ++i) {  // then one read per gap: k2 = 0 lo activations, lo weights; k2 = 2 the same way
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i >= 14) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // the staging writes behind the last weight read
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // prefetch loads of the tap three ahead
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __syncthreads();
            ++g;
        };
        for (int q = 0; q < 8; q += 2) {
            tap(q, S0);
            tap(q + 1, S1);
        }
        stamp();  // 10: conv-transpose GEMM done
        if (n0 + cseq < a.nseq) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p = 32 * NT * cpart + 32 * t + r;
                if (p < Ls) {  // (then the clamped position of eoff is p itself)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        sto(a.out + (size_t)((q & 3) + 8 * (q >> 2)) * a.cstride, eoff[t], fmaf(acc[t][q], WINV, res[t][q]));
                }
            }
        }
        stamp();  // 11: end
    }
}

size_t dp16s_lds_bytes(int Ls, int nseq_per_wg) {
    return (size_t)2 * WBUF + (size_t)nseq_per_wg * (Ls + 1 + PADR) * HLD * 2 + (size_t)nseq_per_wg * 2 * 32 * 4;
}

template <int NSEQ, bool PAIRED, int NT, int NP = 4 / NT>
static int launch_dp16s_t(const Dp16Args& a, hipStream_t st) {
    const size_t lds = dp16s_lds_bytes(a.Ls, NSEQ);
    constexpr bool LONG = NT == 2 && NP == 4;             // one 512-thread workgroup per CU
    if (lds > (LONG ? 160 : 80) * 1024) return RTFS_ERR_SHAPE;  // (else: two workgroups per CU)
    constexpr int NTHR = 128 * NP;
    if (a.stamps) {
        if (rtfs_set_max_lds((const void*)dp16s_kernel<NSEQ, PAIRED, NT, true, NP>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
        hipLaunchKernelGGL((dp16s_kernel<NSEQ, PAIRED, NT, true, NP>), dim3(cdiv(a.nseq, NSEQ)), dim3(NTHR), lds, st, a);
        return rtfs_launch_status();
    }
    if (rtfs_set_max_lds((const void*)dp16s_kernel<NSEQ, PAIRED, NT, false, NP>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    void* slot = dualpath_timing_begin(a.Ls, a.nseq, st);
    hipLaunchKernelGGL((dp16s_kernel<NSEQ, PAIRED, NT, false, NP>), dim3(cdiv(a.nseq, NSEQ)), dim3(NTHR), lds, st, a);
    dualpath_timing_end(slot, st);
    return rtfs_launch_status();
}

// Ls <= 64 (the F sweep: 64): one sequence pair per workgroup; Ls <= 128 (the 2 s T sweep: 125): one sequence per workgroup; Ls <= 256 (the
// 4 s T sweep: 250): one sequence per 512-thread workgroup.  The limits are
// on Ls = L + 7, the row count of the load phase and of the conv-transpose output (routing by L left Ls = 65 .. 71 and 129 .. 135 with their
// last positions unwritten - found by tests/test_hip_parity.py::test_dualpath_sweep_lengths)
int launch_dualpath16s(const Dp16Args& a0, hipStream_t st) {
    const int L = a0.Ls - 7;
    if (L < 1 || a0.Ls > 256) return RTFS_ERR_SHAPE;
    // 32-bit byte offsets from the tensor base inside the kernel
    if ((((size_t)(a0.nseq - 1) / a0.R) * a0.bstride + (size_t)(a0.R - 1) * a0.rstride + 63 * a0.cstride + a0.Ls) * 4 >= ((size_t)1 << 32)) return RTFS_ERR_SHAPE;
    const Dp16Args& a = a0;
    const bool pair = a0.Ls <= 64;
    if (a0.Ls > 128) return launch_dp16s_t<1, false, 2, 4>(a, st);  // the 4 s shapes: four time parts, one workgroup per CU
    return pair ? launch_dp16s_t<2, true, 2>(a, st) : launch_dp16s_t<1, false, 2>(a, st);
}
