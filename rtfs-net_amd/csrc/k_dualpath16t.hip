// Fused dual-path SRU sweep, generation 4 (experiment): generation 3 (k_dualpath16s.hip) cut once more so that THREE workgroups share a CU.
// DualPathRNN.forward, reference src/models/layers/rnn_layers.py:136-162 with the third-party sru.SRU cell (rnn_layers.py:99-105,150).
//
// Generation 3 keeps the matrix pipe ~46 % busy: two 4-wave workgroups per CU, each alternating GEMM phases with a latency-bound recurrence.
// A third resident workgroup needs <= 168 registers and <= 53 KB of LDS.  Both come from splitting a layer's GEMM in TWO PASSES over K:
//   pass A  gate tiles 0-1 (candidate u0, forget gate u1: 64 accumulators) -> prescale -> the cell-state chain, c_t left in 32 registers;
//   pass B  gate tiles 2-3 (reset gate u2, highway input u3: 64 accumulators) -> reset gate + highway output -> hidden states to the planes.
// A staged K step then holds only the two gate tiles of the pass: 8 KB, double-buffered 16 KB (generation 3: 32 KB), and the planes share
// one zero row: 54,048 B per workgroup.  The price: twice the K steps (12 MFMAs per wave and barrier instead of 24) and the A fragments read
// twice.  Weight images: packing.frag_image_gate2 / frag_image_ct2 ([pass][K step][direction][gate tile of the pass][hi|lo][lane] x 16 B).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

#define HLD 72          // activation row stride (halfs)
#define WPIECES 512     // 16-byte pieces of one staged step (8 KB)
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
__device__ __forceinline__ float sig2(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }
__device__ __forceinline__ float take_half(float v, int ph) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(ph == 0 ? r[0] : r[1]);
}
__device__ __forceinline__ float sum_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
}  // namespace

template <int NSEQ, bool PAIRED, bool STAMP = false>
__global__ __launch_bounds__(256, 3) void dp16t_kernel(Dp16Args a) {
    static_assert((NSEQ == 2 && PAIRED) || (NSEQ == 1 && !PAIRED), "F sweep: one sequence pair; T sweep: one sequence");
    constexpr int STEPS = PAIRED ? 32 : 64;  // time steps covered by one wave; 2 parts per workgroup
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int Ls = a.Ls, L = Ls - 7;
    const int zrow = NSEQ * Ls;  // ONE all-zero row behind the sequences' rows (conv-transpose borders)
    half8* Wst = reinterpret_cast<half8*>(smem);                                   // [2 buffers][512 pieces]
    _Float16* Hh = reinterpret_cast<_Float16*>(smem + 2 * WPIECES * 16);           // [NSEQ * Ls + 1][HLD]
    _Float16* Hl = Hh + (NSEQ * Ls + 1) * HLD;
    float* chand = reinterpret_cast<float*>(Hl + (NSEQ * Ls + 1) * HLD);           // [NSEQ][2 dirs][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int part = wave >> 1, dir = wave & 1;
    const int seq = PAIRED ? h : 0;
    constexpr int CPART = 2 / NSEQ;
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
    stamp();  // 0
    auto seq_base = [&](int s) {
        int n = n0 + s;
        n = n < a.nseq ? n : a.nseq - 1;
        return (size_t)(n / a.R) * a.bstride + (size_t)(n % a.R) * a.rstride;
    };

    // weight stream: 512 pieces per step, thread tid moves pieces tid and tid + 256; two register sets, two steps ahead
    half8 pre[2][2];
    auto stage_load = [&](auto set_c, const half8* __restrict__ step) {
        constexpr int S = decltype(set_c)::value;
        pre[S][0] = step[tid];
        pre[S][1] = step[tid + 256];
    };
    auto stage_write = [&](auto set_c, int buf) {
        constexpr int S = decltype(set_c)::value;
        Wst[buf * WPIECES + tid] = pre[S][0];
        Wst[buf * WPIECES + tid + 256] = pre[S][1];
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    stage_load(S0, a.wg_l0);

    // ---------------- phase 0: load rows, LayerNorm over channels, split to f16 planes
    {
        const int task = wave * 32 + r, ntask = NSEQ * Ls;  // <= 128
        const bool live = task < ntask;
        const int tk = live ? task : ntask - 1;
        const int s = (NSEQ == 2 && tk >= Ls) ? 1 : 0, pos = tk - s * Ls;
        const float* xp = a.x + seq_base(s) + pos + (size_t)(32 * h) * a.cstride;
        float v[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = xp[(size_t)c * a.cstride];
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
        _Float16* dh = Hh + tk * HLD + 32 * h;  // row of (s, pos) = s * Ls + pos = tk
        _Float16* dl = Hl + tk * HLD + 32 * h;
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
        if (tid < 9) {
            half8 z;
#pragma unroll
            for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.f;
            *reinterpret_cast<half8*>(Hh + zrow * HLD + tid * 8) = z;
            *reinterpret_cast<half8*>(Hl + zrow * HLD + tid * 8) = z;
        }
    }
    int rowbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rs = PAIRED ? (r >> 2) & 1 : 0;
        int tau = STEPS * part + (r & 3) + 4 * (r >> 3) + (PAIRED ? 16 * t : 32 * t + 16 * ((r >> 2) & 1));
        tau = tau < L ? tau : L - 1;
        rowbase[t] = (rs * Ls + (dir ? L - 1 - tau : tau)) * HLD + 8 * h;
    }
    stamp();  // 1
    stage_write(S0, 0);
    stage_load(S0, a.wg_l0 + WPIECES);
    stage_load(S1, a.wg_l0 + 2 * WPIECES);
    __syncthreads();

    int g = 0;
    // one pass of one layer: 12 MFMAs per step into acc2 (2 row tiles x 2 gate tiles), term-major; `cur` = this pass's image, `nxt` = what
    // follows it in the stream (>= 4 steps long)
    auto gemm_pass = [&](f32x16 (&acc2)[2][2], int nstep, bool layer0, const half8* __restrict__ cur, const half8* __restrict__ nxt) {
        auto kstep = [&](int q, auto set_c) {
            const int aoff = layer0 ? (q >> 2) * HLD + (q & 3) * 16 : q * 16;
            half8 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const half8*>(Hh + rowbase[t] + aoff);
                al[t] = *reinterpret_cast<const half8*>(Hl + rowbase[t] + aoff);
            }
            const half8* wb = Wst + (g & 1) * WPIECES + dir * 256 + lane;  // piece ((dir * 2 + m') * 2 + part) * 64 + lane
            const half8 b0h = wb[0], b0l = wb[64], b1h = wb[128], b1l = wb[192];
            stage_write(set_c, (g + 1) & 1);
            stage_load(set_c, q + 3 < nstep ? cur + (size_t)(q + 3) * WPIECES : nxt + (size_t)(q + 3 - nstep) * WPIECES);
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc2[t][mm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? al[t] : ah[t], term == 1 ? (mm ? b1l : b0l) : (mm ? b1h : b0h),
                                                                             acc2[t][mm], 0, 0, 0);
                }
            }
            __syncthreads();
            ++g;
        };
        for (int q = 0; q < nstep; q += 2) {
            kstep(q, S0);
            kstep(q + 1, S1);
        }
    };

    for (int layer = 0; layer < 4; ++layer) {
        const int nstep = layer == 0 ? 32 : 4;
        const float vf = a.wc16[layer * 128 + dir * 32 + r], vr = a.wc16[layer * 128 + 64 + dir * 32 + r];
        const float bf = a.bias16[layer * 128 + dir * 32 + r] * 256.f, br = a.bias16[layer * 128 + 64 + dir * 32 + r] * 256.f;
        const half8* const imgA = layer == 0 ? a.wg_l0 : a.wg_l + (size_t)((layer - 1) * 2) * 4 * WPIECES;
        const half8* const imgB = imgA + (size_t)nstep * WPIECES;
        const half8* const after = layer < 3 ? a.wg_l + (size_t)(layer * 2) * 4 * WPIECES : a.wg_ct;
        // ---- pass A: u0, u1 -> the cell-state chain
        f32x16 accA[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                accA[t][0][q] = 0.f;
                accA[t][1][q] = bf;
            }
        gemm_pass(accA, nstep, layer == 0, imgA, imgB);
        stamp();  // 2, 5, 8, 11 -> (the stamp slots wrap at 16: diagnostic build reports the first phases only)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) accA[t][m][q] *= WINV;
        __builtin_amdgcn_sched_barrier(0);
        float cin[2] = {0.f, 0.f};
        for (int hp = 0; hp < 2; ++hp) {
            if (part == hp) {
                float c = hp > 0 ? chand[(seq * 2 + dir) * 32 + r] : 0.f;
                if (PAIRED) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        cin[t] = c;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float u0 = accA[t][0][q];
                            const float f = sig2(fmaf(vf, c, accA[t][1][q]));
                            c = fmaf(c - u0, f, u0);
                            accA[t][0][q] = c;
                        }
                    }
                    chand[(seq * 2 + dir) * 32 + r] = c;
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            if (h == ph) cin[t] = c;
                            float cr = c;
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                const float u0 = accA[t][0][q];
                                const float f = sig2(fmaf(vf, cr, accA[t][1][q]));
                                cr = fmaf(cr - u0, f, u0);
                                accA[t][0][q] = h == ph ? cr : u0;
                            }
                            c = take_half(cr, ph);
                        }
                    }
                    if (h == 0) chand[dir * 32 + r] = c;
                }
            }
            if (hp == 0) __syncthreads();  // cell state of part 0 published (nothing waits for part 1's)
        }
        stamp();  // chain done
        // ---- pass B: u2, u3 -> reset gate + highway output
        f32x16 accB[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                accB[t][0][q] = br;
                accB[t][1][q] = 0.f;
            }
        gemm_pass(accB, nstep, layer == 0, imgB, after);
        // every wave's fragment reads of this layer are complete (last barrier of the pass): the planes may be overwritten in place
        {
            const int tau0 = STEPS * part;
            int o0 = (seq * Ls + (dir ? L - 1 - tau0 : tau0)) * HLD + dir * 32 + r;
            asm volatile("" : "+v"(o0));
            const int ostep = dir ? -HLD : HLD;
            const int nvalid = L - tau0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float cprev = cin[t];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int idx = PAIRED ? 16 * t + q : 32 * t + 16 * h + q;
                    const float ct = accA[t][0][q];
                    const float gte = sig2(fmaf(vr, cprev, accB[t][0][q] * WINV)), xp = accB[t][1][q] * WINV;
                    cprev = ct;
                    const float hv = fmaf(ct - xp, gte, xp);
                    if (idx < nvalid) {
                        const _Float16 hh = (_Float16)hv;
                        const int o = o0 + idx * ostep;
                        Hh[o] = hh;
                        Hl[o] = (_Float16)(hv - (float)hh);
                    }
                }
            }
        }
        __syncthreads();  // all hidden outputs of this layer are in the planes
        stamp();
    }

    // ---------------- ConvTranspose1d + bias + residual; 16 half-taps of 32 k' each ([tap][k half][co tile][ks'][hi|lo][lane])
    {
        f32x16 acc[2], accB[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = accB[t][q] = 0.f;
        float res[2][16];
        const size_t rbase = seq_base(cseq);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int p = min(64 * cpart + 32 * t + r, Ls - 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = ccot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                res[t][q] = a.x[rbase + (size_t)co * a.cstride + p] + a.bt[co];
            }
        }
        auto halftap = [&](int q, auto set_c) {  // q = tap * 2 + k half
            const int tapi = q >> 1, kh = q & 1;
            int hrow[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int p = 64 * cpart + 32 * t + r - tapi;
                hrow[t] = ((p >= 0 && p < L) ? cseq * Ls + p : zrow) * HLD + 8 * h + kh * 32;
            }
            const half8* wb = Wst + (g & 1) * WPIECES + ccot * 256 + lane;  // piece ((co tile * 2 + ks') * 2 + part) * 64 + lane
            half8 xh[2][2], xl[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xh[kk][t] = *reinterpret_cast<const half8*>(Hh + hrow[t] + kk * 16);
                    xl[kk][t] = *reinterpret_cast<const half8*>(Hl + hrow[t] + kk * 16);
                }
            const half8 w0h = wb[0], w0l = wb[64], w1h = wb[128], w1l = wb[192];
            stage_write(set_c, (g + 1) & 1);
            stage_load(set_c, a.wg_ct + (size_t)(q + 3 < 16 ? q + 3 : 15) * WPIECES);
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? w0l : w0h, term == 1 ? xl[0][t] : xh[0][t], acc[t], 0, 0, 0);
                    accB[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? w1l : w1h, term == 1 ? xl[1][t] : xh[1][t], accB[t], 0, 0, 0);
                }
            }
            __syncthreads();
            ++g;
        };
        for (int q = 0; q < 16; q += 2) {
            halftap(q, S0);
            halftap(q + 1, S1);
        }
        stamp();
        if (n0 + cseq < a.nseq) {
            const size_t base = seq_base(cseq);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int p = 64 * cpart + 32 * t + r;
                if (p < Ls) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int co = ccot * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        a.out[base + (size_t)co * a.cstride + p] = fmaf(acc[t][q] + accB[t][q], WINV, res[t][q]);
                    }
                }
            }
        }
        stamp();
    }
}

size_t dp16t_lds_bytes(int Ls, int nseq_per_wg) {
    return (size_t)2 * WPIECES * 16 + (size_t)2 * (nseq_per_wg * Ls + 1) * HLD * 2 + (size_t)nseq_per_wg * 2 * 32 * 4;
}

template <int NSEQ, bool PAIRED>
static int launch_dp16t_t(const Dp16Args& a, hipStream_t st) {
    const size_t lds = dp16t_lds_bytes(a.Ls, NSEQ);
    if (lds > 53 * 1024) return RTFS_ERR_SHAPE;  // three workgroups per CU
    if (a.stamps) {
        if (rtfs_set_max_lds((const void*)dp16t_kernel<NSEQ, PAIRED, true>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
        hipLaunchKernelGGL((dp16t_kernel<NSEQ, PAIRED, true>), dim3(cdiv(a.nseq, NSEQ)), dim3(256), lds, st, a);
        return rtfs_launch_status();
    }
    if (rtfs_set_max_lds((const void*)dp16t_kernel<NSEQ, PAIRED>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    void* slot = dualpath_timing_begin(a.Ls, a.nseq, st);
    hipLaunchKernelGGL((dp16t_kernel<NSEQ, PAIRED>), dim3(cdiv(a.nseq, NSEQ)), dim3(256), lds, st, a);
    dualpath_timing_end(slot, st);
    return rtfs_launch_status();
}

int launch_dualpath16t(const Dp16Args& a, hipStream_t st) {
    const int L = a.Ls - 7;
    if (L < 1 || a.Ls > 128 || !a.wg_l0) return RTFS_ERR_SHAPE;
    return a.Ls <= 64 ? launch_dp16t_t<2, true>(a, st) : launch_dp16t_t<1, false>(a, st);
}
