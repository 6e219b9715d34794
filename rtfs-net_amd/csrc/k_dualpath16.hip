// Fused dual-path SRU sweep, generation 2: f16x3 split-precision MFMA + in-register scan.
// Same arithmetic as k_dualpath.hip (DualPathRNN.forward, reference src/models/layers/rnn_layers.py:136-162,
// with the third-party sru.SRU cell, call site rnn_layers.py:99-105,150) but restructured for gfx950:
//
//   * one 512-thread workgroup (8 waves, 2 per SIMD) owns NSEQ sequences end to end; HBM sees only the input
//     row, the output row and (through L2) the weights;
//   * activations live in LDS as two f16 planes (hi, lo) [position][64 ch], 144-byte rows -> every MFMA A/B
//     fragment (8 consecutive channels of one position) is one conflict-free ds_read_b128.  The Unfold(k=8)
//     is an addressing mode: K index k' = kk*64 + c reads row (pos + kk);
//   * all GEMMs use v_mfma_f32_32x32x16_f16 on split operands: x = xh + xl, 256*w = wh + wl,
//     x.w ~ (xh.wh + xh.wl + xl.wh)/256 (error 2^-22; measured end to end 6e-7);
//   * weights are streamed from L2 in 32-deep K chunks through a double-buffered LDS image shared by the 8 waves
//     (register prefetch of chunk q+1 during the MFMAs of chunk q, one barrier per chunk);
//   * wave = (sequence, direction, time part): its 2 row tiles x 4 gate tiles of U stay in 128 accumulator
//     registers; the recurrence runs on those registers (both lane halves take turns 4 steps at a time, the
//     cell state crosses halves with v_permlane32_swap); the highway input x' of layers 1-3 is produced by the
//     same MFMAs through an identity block in the weight image; sigmoid = rcp(1 + exp2(z)) with -log2(e)
//     folded into the gate weights; hidden outputs are written back to the LDS planes in place.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>


#define HLD 72   // activation row stride (halfs)
#define WLD 40   // staged gate-weight row stride (halfs): 32 k + 8 pad
#define CLD 72   // staged conv-transpose weight row stride (halfs): 64 k + 8 pad
#define WINV (1.0f / 256.0f)

namespace {

__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 hh = (_Float16)v[i];
        hi[i] = hh;
        lo[i] = (_Float16)(v[i] - (float)hh);
    }
}

// sigmoid with the -log2(e) factor already folded into z
__device__ __forceinline__ float sig2(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }

// broadcast the values held by lanes of half `ph` (0 = lanes 0-31, 1 = lanes 32-63) to the partner lanes
__device__ __forceinline__ float take_half(float v, int ph) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(ph == 0 ? r[0] : r[1]);
}

}  // namespace

// STAMP = diagnostic build: wave 0 of every workgroup records s_memtime at phase boundaries into a.stamps
// ([workgroup][16] u64); never used by the product path.
// PAIRED: the two lane halves of an MFMA tile (rows with bit 2 clear / set) carry two different sequences, 16 time
// steps each, so accumulator register q of lane (h, j) is time step q of sequence (pair, h): the scan is a plain
// loop over registers with every lane busy and no cross-lane traffic.  A wave then covers 32 time steps of a
// sequence pair; NHALF = number of 32-step parts.  !PAIRED (4 s inputs, one sequence per workgroup): a wave covers
// 64 steps of one sequence and the halves take turns (v_permlane32_swap hand-off every 4 steps).
template <int NSEQ, int NHALF, bool PAIRED, bool STAMP = false>
__global__ __launch_bounds__(512) void dp16_kernel(Dp16Args a) {
    static_assert((PAIRED ? NSEQ / 2 : NSEQ) * NHALF == 4, "8 waves = sequences (pairs) x 2 directions x NHALF time parts");
    constexpr int STEPS = PAIRED ? 32 : 64;  // time steps covered by one wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Ls = a.Ls, L = Ls - 7, rowsH = Ls + 1;  // one extra all-zero row for the conv-transpose borders
    _Float16* Hh = reinterpret_cast<_Float16*>(smem);
    _Float16* Hl = Hh + NSEQ * rowsH * HLD;
    _Float16* Wst = Hl + NSEQ * rowsH * HLD;  // [buf 2][part 2][256][WLD]
    float* chand = reinterpret_cast<float*>(Wst + 2 * 2 * 256 * WLD);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // time part MAJOR: the waves that run one part of the serialised recurrence together (8 / NHALF of them) have
    // consecutive ids and therefore sit on different SIMDs -- the recurrence is VALU-issue bound (4 transcendentals
    // per step), two scanning waves on one SIMD with the neighbouring SIMD idle would double its time
    constexpr int WPP = 8 / NHALF;
    const int part = wave / WPP, grp = (wave % WPP) >> 1, dir = wave & 1;  // grp = sequence or pair
    const int seq = PAIRED ? grp * 2 + h : grp;  // the sequence this LANE's accumulator rows / A-operand rows belong to
    // conv-transpose roles (always one sequence per wave): sequence, co tile, 64-position part
    constexpr int CPART = 4 / NSEQ;
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
    stamp();  // 0: start
    auto seq_base = [&](int s) {
        int n = n0 + s;
        n = n < a.nseq ? n : a.nseq - 1;
        return (size_t)(n / a.R) * a.bstride + (size_t)(n % a.R) * a.rstride;
    };

    // ---------------- phase 0: load rows, LayerNorm over channels, split to f16 planes (normalizations.py:33-37)
    for (int task = tid; task < NSEQ * Ls; task += 512) {
        const int s = task / Ls, pos = task - s * Ls;
        const float* xp = a.x + seq_base(s) + pos;
        float v[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) v[c] = xp[(size_t)c * a.cstride];
        float mean = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) mean += v[c];
        mean *= (1.0f / 64);
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const float d = v[c] - mean;
            var = fmaf(d, d, var);
        }
        const float rstd = 1.0f / sqrtf(var * (1.0f / 64) + RTFS_EPS);
        _Float16* dh = Hh + (s * rowsH + pos) * HLD;
        _Float16* dl = Hl + (s * rowsH + pos) * HLD;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = fmaf((v[c8 * 8 + i] - mean) * rstd, a.ln_gamma[c8 * 8 + i], a.ln_beta[c8 * 8 + i]);
            half8 hi, lo;
            split8(y, hi, lo);
            *reinterpret_cast<half8*>(dh + c8 * 8) = hi;
            *reinterpret_cast<half8*>(dl + c8 * 8) = lo;
        }
    }
    if (tid < NSEQ * 9) {  // the zero rows (72 halfs = 9 x 16 B each)
        half8 z;
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.f;
        const int s = tid / 9, pc = tid % 9;
        *reinterpret_cast<half8*>(Hh + (s * rowsH + Ls) * HLD + pc * 8) = z;
        *reinterpret_cast<half8*>(Hl + (s * rowsH + Ls) * HLD + pc * 8) = z;
    }

    // weight chunk staging: image [part 2][rows][32 halfs] in global -> padded rows in LDS buffer `buf`
    half8 pre[4];
    // rows_c = 256: gate chunk [hi|lo][256 cols][32 k] -> rows of WLD halfs;  64: conv-transpose chunk [hi|lo][64 co][64 k] -> rows of CLD
    auto stage_load = [&](const half8* __restrict__ img, auto rows_c) {
        constexpr int rows = decltype(rows_c)::value;
        constexpr int npiece = rows == 256 ? 2 * 256 * 4 : 2 * 64 * 8;
#pragma unroll
        for (int j = 0; j < npiece / 512; ++j) pre[j] = img[tid + 512 * j];
    };
    auto stage_write = [&](int buf, auto rows_c) {
        constexpr int rows = decltype(rows_c)::value;
        constexpr int npiece = rows == 256 ? 2 * 256 * 4 : 2 * 64 * 8;
        constexpr int ppr = rows == 256 ? 4 : 8;  // 16-byte pieces per row
#pragma unroll
        for (int j = 0; j < npiece / 512; ++j) {
            const int i = tid + 512 * j;
            const int pt = i / (rows * ppr), rem = i - pt * rows * ppr;
            if (rows == 256)
                *reinterpret_cast<half8*>(Wst + ((buf * 2 + pt) * 256 + rem / ppr) * WLD + (rem % ppr) * 8) = pre[j];
            else
                *reinterpret_cast<half8*>(Wst + buf * (2 * 64 * CLD) + (pt * 64 + rem / ppr) * CLD + (rem % ppr) * 8) = pre[j];
        }
    };
    const std::integral_constant<int, 256> R256;
    const std::integral_constant<int, 64> R64;

    // activation rows of this wave's two row tiles (virtual time tau; the backward direction reads position L-1-tau)
    int rowbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // A-operand row r of tile t: PAIRED -> sequence (pair, bit 2 of r), step (r&3) + 4*(r>>3); else step r
        const int rs = PAIRED ? grp * 2 + ((r >> 2) & 1) : grp;
        int tau = STEPS * part + (PAIRED ? 16 * t + (r & 3) + 4 * (r >> 3) : 32 * t + r);
        tau = tau < L ? tau : L - 1;
        rowbase[t] = (rs * rowsH + (dir ? L - 1 - tau : tau)) * HLD;
    }

    stamp();  // 1: after load + LN issue (before first barrier)
    // the weight chunks of all phases form one stream: chunk c+1 (or the next phase's chunk 0) is loaded into
    // registers while chunk c is multiplied and written to the other LDS buffer before the barrier that ends chunk c.
    stage_load(a.w16_l0, R256);
    __syncthreads();  // phase 0 has filled the activation planes
    stage_write(0, R256);
    __syncthreads();
    // ---------------- four SRU layers
    for (int layer = 0; layer < 4; ++layer) {
        const int nchunk = layer == 0 ? 16 : 2;
        const half8* wimg = layer == 0 ? a.w16_l0 : a.w16_l + (size_t)(layer - 1) * 2 * (2 * 256 * 4);
        const float vf = a.wc16[layer * 128 + dir * 32 + r], vr = a.wc16[layer * 128 + 64 + dir * 32 + r];
        const float bf = a.bias16[layer * 128 + dir * 32 + r] * 256.f, br = a.bias16[layer * 128 + 64 + dir * 32 + r] * 256.f;
        f32x16 acc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc[t][0][q] = 0.f;
                acc[t][1][q] = bf;
                acc[t][2][q] = br;
                acc[t][3][q] = 0.f;
            }
        for (int q = 0; q < nchunk; ++q) {
            const bool last = q + 1 == nchunk;  // nchunk is even, so the next phase's chunk 0 lands in buffer 0
            // ONE unconditional load from a selected pointer: a load behind a branch is waited for at the join, which
            // would serialise the prefetch in front of the MFMAs.  (The conv-transpose chunk only needs 2 of the 4 pieces.)
            const half8* nextp = !last ? wimg + (size_t)(q + 1) * (2 * 256 * 4)
                                       : (layer < 3 ? a.w16_l + (size_t)layer * 2 * (2 * 256 * 4) : a.w16_ct);
            stage_load(nextp, R256);
            const int aoff = layer == 0 ? (q >> 1) * HLD + (q & 1) * 32 : q * 32;
            const _Float16* wb = Wst + ((q & 1) * 2) * 256 * WLD + (dir * 128 + r) * WLD + 8 * h;
            // software-pipelined fragment reads: the LDS reads of gate tile (ks, m+1) are issued before the six MFMAs
            // of (ks, m), so a wave never sits on an LDS round trip with the matrix pipe idle (sched_group_barrier pins
            // the order: 2 DS reads, then 6 MFMAs)
            {
                half8 ah[2][2], al[2][2];  // [ks][tile]
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        ah[k2][t] = *reinterpret_cast<const half8*>(Hh + rowbase[t] + aoff + 16 * k2 + 8 * h);
                        al[k2][t] = *reinterpret_cast<const half8*>(Hl + rowbase[t] + aoff + 16 * k2 + 8 * h);
                    }
                half8 bh = *reinterpret_cast<const half8*>(wb);
                half8 bl = *reinterpret_cast<const half8*>(wb + 256 * WLD);
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // the 8 activation + 2 weight fragments above
#pragma unroll
                for (int step = 0; step < 8; ++step) {
                    const int k2 = step >> 2, m = step & 3;
                    half8 nh = bh, nl = bl;
                    if (step < 7) {
                        const int n2 = (step + 1) >> 2, nm = (step + 1) & 3;
                        nh = *reinterpret_cast<const half8*>(wb + nm * 32 * WLD + 16 * n2);
                        nl = *reinterpret_cast<const half8*>(wb + 256 * WLD + nm * 32 * WLD + 16 * n2);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k2][t], bh, acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k2][t], bl, acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[k2][t], bh, acc[t][m], 0, 0, 0);
                    }
                    if (step < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read x2 (next fragments) ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                // ... in front of this step's 6 MFMAs
                    bh = nh;
                    bl = nl;
                    if (step == 3) {
                        // the next chunk (loaded into registers at the top of this iteration) goes to the other LDS buffer
                        // under the second half of this chunk's MFMAs instead of after them
                        if (!last || layer < 3) stage_write((q + 1) & 1, R256);
                        else stage_write(0, R64);
                        __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);  // DS write x4
                    }
                }
            }
            __syncthreads();
        }
        stamp();  // 2,4,6,8: GEMM of layer done
        // every wave has finished reading the activation planes: the scan may overwrite them in place.
        // ---- recurrence on the accumulator registers (row of register q: (q&3) + 8*(q>>2) + 4*h)
        if (PAIRED) {
            // undo the 2^8 weight prescale on all 128 accumulators now, on every wave at once (independent work that
            // packs into v_pk_mul_f32), so the serialised per-part recurrence below is the bare dependency chain
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[t][m][q] *= WINV;
            __builtin_amdgcn_sched_barrier(0);
        }
        float cin = 0.f;
        for (int hp = 0; hp < NHALF; ++hp) {
            if (part == hp) {
                float c = 0.f;
                if (NHALF > 1 && hp > 0) c = chand[(seq * 2 + dir) * 32 + r];
                if (PAIRED) {
                    // register q of tile t = time step 16t + q of this lane's own sequence.  Only the cell-state chain
                    // c_t = u0 + (c_{t-1} - u0) f(c_{t-1}) is serial; it overwrites u0 in place.  The reset gate and the
                    // hidden output depend on c_{t-1}, c_t but nothing depends on them: they are evaluated after the
                    // hand-off below, concurrently with the next time part's chain.
                    cin = c;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float u0 = acc[t][0][q];
                            const float f = sig2(fmaf(vf, c, acc[t][1][q]));
                            c = fmaf(c - u0, f, u0);
                            acc[t][0][q] = c;
                        }
                    if (NHALF > 1) chand[(seq * 2 + dir) * 32 + r] = c;
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
                            for (int ph = 0; ph < 2; ++ph) {
                                float cr = c;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int q = rg * 4 + i;
                                    const float u0 = acc[t][0][q] * WINV;
                                    const float f = sig2(fmaf(acc[t][1][q], WINV, vf * cr));
                                    const float g = sig2(fmaf(acc[t][2][q], WINV, vr * cr));
                                    const float xp = acc[t][3][q] * WINV;
                                    cr = fmaf(cr - u0, f, u0);
                                    const float hv = fmaf(cr - xp, g, xp);
                                    if (h == ph) acc[t][0][q] = hv;  // this lane's own rows: keep the hidden output
                                }
                                c = take_half(cr, ph);
                            }
                        }
                    }
                    if (NHALF > 1 && h == 0) chand[(seq * 2 + dir) * 32 + r] = c;
                }
            }
            if (NHALF > 1) __syncthreads();  // cell state published: the next time part starts while this one writes back
            if (part == hp) {
                // write the hidden outputs (this wave's direction half of the channels) back in place
                {
                    const int tau0 = STEPS * part;
                    int o0 = (seq * rowsH + (dir ? L - 1 - tau0 : tau0)) * HLD + dir * 32 + r;
                    asm volatile("" : "+v"(o0));  // opaque per layer: keeps 32 derived addresses from being hoisted + spilled
                    const int ostep = dir ? -HLD : HLD;
                    const int nvalid = L - tau0;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int idx = PAIRED ? 16 * t + q : 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
                            float hv = acc[t][0][q];
                            if (PAIRED) {  // deferred reset gate + highway: h = x' + (c_t - x') r(c_{t-1})
                                const float g = sig2(fmaf(vr, cin, acc[t][2][q])), xp = acc[t][3][q];
                                cin = hv;
                                hv = fmaf(hv - xp, g, xp);
                            }
                            if (idx < nvalid) {
                                const _Float16 hh = (_Float16)hv;
                                const int o = o0 + idx * ostep;
                                Hh[o] = hh;
                                Hl[o] = (_Float16)(hv - (float)hh);
                            }
                        }
                }
            }
        }
        __syncthreads();  // all hidden outputs of this layer are in the planes
        stamp();  // 3,5,7,9: scan of layer done
    }

    // ---------------- ConvTranspose1d(64->64, k=8) + bias + residual (rnn_layers.py:153-156), transposed:
    //   y[co][t] = bt[co] + sum_{kk,ci} Wt[co][kk*64+ci] * H[t-kk][ci];  wave = (sequence, co tile, position part)
    {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
        // the residual rows of the epilogue are requested now and arrive under the GEMM (clamped addresses: dead
        // sequences / positions read a valid element that is never stored)
        float res[2][16];
        const size_t rbase = seq_base(cseq);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = min(64 * cpart + 32 * t + r, Ls - 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) {  // residual + conv-transpose bias, both fetched under the GEMM (the bias loads in the epilogue cost 5 %)
                const int co = ccot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                res[t][q] = a.x[rbase + (size_t)co * a.cstride + p] + a.bt[co];
            }
        }
        // chunk 0 is already staged; the barrier that ended the layer-3 scan ordered the hidden outputs.
        // 8 chunks of 64 k' (one tap kk each): staged image [hi|lo][64 co][64 + 8 pad]
        for (int q = 0; q < 8; ++q) {
            stage_load(a.w16_ct + (size_t)(q + 1 < 8 ? q + 1 : 7) * (2 * 64 * 8), R64);  // unconditional (clamped)
            const _Float16* wb = Wst + (q & 1) * (2 * 64 * CLD) + (ccot * 32 + r) * CLD + 8 * h;
            int hrow[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int p = 64 * cpart + 32 * t + r - q;
                hrow[t] = (cseq * rowsH + ((p >= 0 && p < L) ? p : Ls)) * HLD + 8 * h;
            }
#pragma unroll
            for (int ks = 0; ks < 64; ks += 16) {
                const half8 wh = *reinterpret_cast<const half8*>(wb + ks);
                const half8 wl = *reinterpret_cast<const half8*>(wb + 64 * CLD + ks);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const half8 xh = *reinterpret_cast<const half8*>(Hh + hrow[t] + ks);
                    const half8 xl = *reinterpret_cast<const half8*>(Hl + hrow[t] + ks);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[t], 0, 0, 0);
                }
            }
            if (q + 1 < 8) stage_write((q + 1) & 1, R64);
            __syncthreads();
        }
        stamp();  // 10: conv-transpose GEMM done
        if (n0 + cseq < a.nseq) {
            const size_t base = seq_base(cseq);
            const int cot = ccot;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int p = 64 * cpart + 32 * t + r;
                if (p < Ls) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int co = cot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        a.out[base + (size_t)co * a.cstride + p] = fmaf(acc[t][q], WINV, res[t][q]);
                    }
                }
            }
        }
        stamp();  // 11: end
    }
}

size_t dp16_lds_bytes(int Ls, int nseq_per_wg) {
    return (size_t)2 * nseq_per_wg * (Ls + 1) * HLD * 2 + (size_t)2 * 2 * 256 * WLD * 2 + (size_t)nseq_per_wg * 2 * 32 * 4;
}

template <int NSEQ, int NHALF, bool PAIRED>
static int launch_dp16_stamp_t(const Dp16Args& a, hipStream_t st) {
    const size_t lds = dp16_lds_bytes(a.Ls, NSEQ);
    if (hipFuncSetAttribute((const void*)dp16_kernel<NSEQ, NHALF, PAIRED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return RTFS_ERR_LAUNCH;
    hipLaunchKernelGGL((dp16_kernel<NSEQ, NHALF, PAIRED, true>), dim3(cdiv(a.nseq, NSEQ)), dim3(512), lds, st, a);
    return rtfs_launch_status();
}

template <int NSEQ, int NHALF, bool PAIRED>
static int launch_dp16_t(const Dp16Args& a, hipStream_t st) {
    const size_t lds = dp16_lds_bytes(a.Ls, NSEQ);
    if (rtfs_set_max_lds((const void*)dp16_kernel<NSEQ, NHALF, PAIRED>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    void* slot = dualpath_timing_begin(a.Ls, a.nseq, st);
    hipLaunchKernelGGL((dp16_kernel<NSEQ, NHALF, PAIRED>), dim3(cdiv(a.nseq, NSEQ)), dim3(512), lds, st, a);
    dualpath_timing_end(slot, st);
    return rtfs_launch_status();
}

int launch_dualpath16(const Dp16Args& a, hipStream_t st) {
    const int L = a.Ls - 7;
    if (L < 1 || a.Ls > 256) return RTFS_ERR_SHAPE;
    // (all routing is by Ls = L + 7: the load phase and the conv-transpose output cover Ls rows - 64 / 128 / 256 per configuration)
    // generation 3 (k_dualpath16s.hip: two workgroups per CU up to Ls = 128, its 512-thread variant above); RTFS_SWEEP_GEN2=1 keeps this file's
    // kernels for A/B
    static const bool gen2 = getenv("RTFS_SWEEP_GEN2") != nullptr;
    if (!gen2) {
        const int rc = launch_dualpath16s(a, st);
        if (rc != RTFS_ERR_SHAPE) return rc;  // (a tensor spanning >= 4 GB: this file's kernels address with 64 bits)
    }
    if (a.stamps) {
        if (a.Ls <= 64) return launch_dp16_stamp_t<4, 2, true>(a, st);
        if (a.Ls <= 128) return launch_dp16_stamp_t<2, 4, true>(a, st);
        return launch_dp16_stamp_t<1, 4, false>(a, st);
    }
    if (a.Ls <= 64) return launch_dp16_t<4, 2, true>(a, st);    // 4 sequences: 2 pairs x 2 dirs x 2 parts of 32 steps
    if (a.Ls <= 128) return launch_dp16_t<2, 4, true>(a, st);   // 2 sequences: 1 pair  x 2 dirs x 4 parts of 32 steps
    return launch_dp16_t<1, 4, false>(a, st);                // 4 s inputs: 1 sequence x 2 dirs x 4 parts of 64 steps
}
