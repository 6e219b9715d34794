// TF-domain self-attention (reference MultiHeadSelfAttention2D, src/models/layers/attention.py:76-189).
//   row_can_kernel<96>   per (b,t) row: all heads' Q/K/V ConvActNorm = 1x1 conv -> PReLU -> LN over (E,F)
//                        (conv_layers.py:196-205, normalizations.py:20-41), written head-major
//                        Q,K (B,H,T,E*F)  V (B,H,T,Cv*F)     (attention.py:156-168)
//   attn_core_kernel     softmax_keys(Q K^T / sqrt(E*F)) V on the f32 matrix cores  (attention.py:171-175)
//   row_can_kernel<64>   attn_concat_proj ConvActNorm + residual (attention.py:182-184)
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

#define AF 64  // n_freqs: LayerNormalization4D gamma/beta are (1,C,1,64), so F' is tied to 64

// X row (64 ch x 64 f) -> NOUT channels: D[o][f] = sum_c W[o][c] X[c][f] on the f16 matrix cores with the 3-term hi/lo
// split (x = xh + xl, 256 w = wh + wl; same scheme and error as k_pw16.hip).  Wave w owns f-tile (w & 1) and output
// tiles {w >> 1, (w >> 1) + 2}.  A workgroup sweeps RCAN_ROWS consecutive frames with the weight fragments and the
// LayerNorm affine of its rows held in registers; the X fragments come straight from HBM (lane = f -> coalesced 128 B
// segments) and the next row's X / residual loads are issued before the current row's GEMM + LayerNorm, so the
// kernel (few waves per CU: register-heavy) never waits on a cold load.
#define RCAN_ROWS 8
#define RCAN_LD 72  // Ys row stride: D-layout writes (rows +4 on the upper half-wave) land in the other 32 banks
template <int NOUT>
__global__ __launch_bounds__(256, 2) void row_can_kernel(RowCanArgs a) {
    constexpr int MT = NOUT / 32, MTW = (MT + 1) / 2;
    constexpr int NR = NOUT / 4;  // LayerNorm rows per wave: 96 -> Q_w (4) + K_w (4) + V_w (16); 64 -> 16 of the one group
    constexpr float WSC = 256.f, WINV = 1.f / 256.f;
    __shared__ float Ys[NOUT][RCAN_LD];
    __shared__ float bias_s[NOUT], slope_s[NOUT];
    __shared__ float red[8];
    const int T = a.T, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 1, mp = wave >> 1;
    for (int o = tid; o < NOUT; o += 256) {
        bias_s[o] = a.bias[o] * WSC;
        slope_s[o] = a.slope[NOUT == 96 ? a.group_of[o] : 0];
    }
    half8 wh[MTW][4], wl[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int mt = min(mp + 2 * m, MT - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = a.wt[(ks * 16 + 8 * h + j) * NOUT + mt * 32 + r] * WSC;
                const _Float16 hi = (_Float16)w;
                wh[m][ks][j] = hi;
                wl[m][ks][j] = (_Float16)(w - (float)hi);
            }
    }
    // LayerNorm rows of this wave (lane = f) and their affine
    auto ln_row = [&](int i) -> int {
        if (NOUT == 64) return wave * 16 + i;
        return i < 4 ? 4 * wave + i : i < 8 ? 16 + 4 * wave + (i - 4) : 32 + 16 * wave + (i - 8);
    };
    // the LayerNorm affine of this wave's rows lives in LDS (48 registers when it was held per lane: with them the kernel needed 319 registers,
    // one workgroup per CU, and every latency of its serial per-frame phases - GEMM, two reductions, stores - was exposed)
    __shared__ float gam_s[NOUT][AF], bet_s[NOUT][AF];
    for (int i = tid; i < NOUT * AF; i += 256) {
        gam_s[0][i] = a.gamma[i];
        bet_s[0][i] = a.beta[i];
    }
    const int t0 = blockIdx.x * a.rpw;
    const int nrows = min(a.rpw, T - t0);
    const float* __restrict__ X = a.x + (size_t)b * 64 * T * AF + nt * 32 + r;
    const float* __restrict__ RES = NOUT == 64 ? a.res + ((size_t)b * 64 + wave * 16) * T * AF + lane : nullptr;
    float v[4][8];
    float rs[NOUT == 64 ? 16 : 1];
    auto load_x = [&](int t) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[ks][j] = X[((size_t)(ks * 16 + 8 * h + j) * T + t) * AF];
    };
    load_x(t0);
    __syncthreads();
    for (int rr = 0; rr < nrows; ++rr) {
        const int t = t0 + rr;
        half8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) split2(v[ks][2 * jp], v[ks][2 * jp + 1], hi[jp], lo[jp]);  // 3 instructions per pair (pipe_helpers.h)
            bh[ks] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi));
            bl[ks] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo));
        }
        if (NOUT == 64) {
#pragma unroll
            for (int i = 0; i < 16; ++i) rs[i] = RES[((size_t)i * T + t) * AF];
        }
        load_x(min(t + 1, T - 1));  // prefetch; consumed next iteration
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const int mt = mp + 2 * m;
            if (mt < MT) {  // wave-uniform
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = bias_s[mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m][ks], bh[ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[m][ks], bl[ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[m][ks], bh[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int o = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    Ys[o][nt * 32 + r] = preluf_(acc[q] * WINV, slope_s[o]);
                }
            }
        }
        __syncthreads();
        float y[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) y[i] = Ys[ln_row(i)][lane];
        if (NOUT == 96) {
            // LayerNorm over (channels of the group, F) for Q_w (rows 0-3), K_w (4-7), V_w (8-23); outputs head-major
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int i0 = g == 0 ? 0 : g == 1 ? 4 : 8, gs = g == 2 ? 16 : 4;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < gs; ++i) s += y[i0 + i];
                const float mean = wave_sum(s) / (float)(gs * AF);
                float vv = 0.f;
#pragma unroll
                for (int i = 0; i < gs; ++i) {
                    y[i0 + i] -= mean;
                    vv = fmaf(y[i0 + i], y[i0 + i], vv);
                }
                const float rstd = 1.0f / sqrtf(wave_sum(vv) / (float)(gs * AF) + RTFS_EPS);
                float* __restrict__ dst = (g == 0 ? a.q : g == 1 ? a.k : a.v) + (((size_t)b * 4 + wave) * T + t) * (gs * AF) + lane;
#pragma unroll
                for (int i = 0; i < gs; ++i) dst[i * AF] = fmaf(y[i0 + i] * rstd, gam_s[ln_row(i0 + i)][lane], bet_s[ln_row(i0 + i)][lane]);
            }
        } else {
            // one LayerNorm group over all (64, F): 16 channels per wave, partial sums exchanged through LDS
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += y[i];
            s = wave_sum(s);
            if (lane == 0) red[wave] = s;
            __syncthreads();
            const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)(64 * AF);
            float vv = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                y[i] -= mean;
                vv = fmaf(y[i], y[i], vv);
            }
            vv = wave_sum(vv);
            if (lane == 0) red[4 + wave] = vv;
            __syncthreads();
            const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)(64 * AF) + RTFS_EPS);
            float* __restrict__ O = a.out + ((size_t)b * 64 + wave * 16) * T * AF + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) O[((size_t)i * T + t) * AF] = fmaf(y[i] * rstd, gam_s[ln_row(i)][lane], bet_s[ln_row(i)][lane]) + rs[i];
        }
        __syncthreads();  // Ys / red are rewritten by the next row
    }
}

// One workgroup per (query tile of 32, head, b).  E = 256 (Q/K row), D = 1024 (V row).  Both products run on the f16
// matrix cores with the 3-term hi/lo split.  Q, K and V fragments are read straight from HBM/L2 in fragment order (Q/K
// rows are k-contiguous: two 16 B loads per fragment; V columns are lane-contiguous), only the 32 x T score tile lives
// in LDS (softmax rows, then the A operand of P V).
#define ATT_PSC 4096.f  // P in [0,1] is scaled so its low half stays above the f16 flush threshold
__device__ __forceinline__ void split8(const float (&v)[8], float sc, half8& hi, half8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) split2(v[2 * jp] * sc, v[2 * jp + 1] * sc, h[jp], l[jp]);  // 3 instructions per pair (pipe_helpers.h)
    hi = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(h));
    lo = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(l));
}
__global__ __launch_bounds__(256) void attn_core_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float S[];  // [32][ldp]
    const int T = a.T;
    const int NKT = (T + 31) >> 5;  // key tiles
    const int NK = NKT * 32;
    const int ldp = NK + 4;         // 16 B aligned rows, ds_read_b128 conflict-free
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block order: workgroup L runs on XCD L % 8, so the query tiles that share one (b, head)'s K and V are
    // given the same residue and hit the same L2 instead of pulling V across the fabric once per tile.
    const int nqt = (T + 31) >> 5;
    const int slot = blockIdx.x >> 3;
    const int pair = (slot / nqt) * 8 + (blockIdx.x & 7);
    if (pair >= a.npairs) return;  // uniform; before any barrier
    const int q0 = (slot % nqt) * 32, hd = pair & 3, b = pair >> 2;
    const float* __restrict__ Q = a.q + ((size_t)b * 4 + hd) * T * 256;
    const float* __restrict__ K = a.k + ((size_t)b * 4 + hd) * T * 256;
    const float* __restrict__ V = a.v + ((size_t)b * 4 + hd) * T * 1024;

    // ---- scores S = Q K^T / 16: wave w owns key tiles w and w + 4
    {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        const float* __restrict__ qp = Q + (size_t)min(q0 + r, T - 1) * 256 + 8 * h;
        const float* __restrict__ kp0 = K + (size_t)min(wave * 32 + r, T - 1) * 256 + 8 * h;
        const float* __restrict__ kp1 = K + (size_t)min((wave + 4) * 32 + r, T - 1) * 256 + 8 * h;
        const bool two = wave + 4 < NKT;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            float qv[8], k0[8], k1[8];
            *reinterpret_cast<f32x4*>(qv) = *reinterpret_cast<const f32x4*>(qp + ks * 16);
            *reinterpret_cast<f32x4*>(qv + 4) = *reinterpret_cast<const f32x4*>(qp + ks * 16 + 4);
            *reinterpret_cast<f32x4*>(k0) = *reinterpret_cast<const f32x4*>(kp0 + ks * 16);
            *reinterpret_cast<f32x4*>(k0 + 4) = *reinterpret_cast<const f32x4*>(kp0 + ks * 16 + 4);
            *reinterpret_cast<f32x4*>(k1) = *reinterpret_cast<const f32x4*>(kp1 + ks * 16);
            *reinterpret_cast<f32x4*>(k1 + 4) = *reinterpret_cast<const f32x4*>(kp1 + ks * 16 + 4);
            half8 ah, al, bh, bl;
            split8(qv, 1.f, ah, al);
            split8(k0, 1.f, bh, bl);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
            if (two) {  // wave-uniform
                split8(k1, 1.f, bh, bl);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kt = wave + 4 * j;
            if (kt < NKT) {
#pragma unroll
                for (int q = 0; q < 16; ++q) S[((q & 3) + 8 * (q >> 2) + 4 * h) * ldp + kt * 32 + r] = acc[j][q] * a.scale;
            }
        }
    }
    __syncthreads();
    // ---- softmax over keys, 8 rows per wave; keys >= T get probability 0
    for (int rr = 0; rr < 8; ++rr) {
        float* row = S + (wave * 8 + rr) * ldp;
        float m = -3.0e38f;
        for (int j = lane; j < T; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < NK; j += 64) {
            const float e = j < T ? expf(row[j] - m) : 0.f;
            row[j] = e;
            s += e;
        }
        const float inv = 1.0f / wave_sum(s);
        for (int j = lane; j < NK; j += 64) row[j] *= inv;
    }
    __syncthreads();
    // ---- O = P V : 32 column tiles of 32, wave w owns tiles [8w, 8w+8) in two groups of 4
    for (int grp = 0; grp < 2; ++grp) {
        f32x16 o[4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int q = 0; q < 16; ++q) o[n][q] = 0.f;
        // the group's four column tiles interleaved: tile n holds columns 128 (2 wave + grp) + 4 r + n, so that a lane's four B-operand values
        // of a key are ONE 16-byte load (and its four outputs of a query one 16-byte store).  With tile-contiguous columns these were four
        // dword accesses each: 512 load instructions per wave, ~30 % of the kernel on the CU's address unit.
        const int col0 = (wave * 8 + grp * 4) * 32 + 4 * r;
        const float* __restrict__ vp = V + col0;
        f32x4 vv[2][8];
        auto load_v = [&](int ks, f32x4 (&dst)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const f32x4*>(vp + (size_t)min(ks * 16 + 8 * h + j, T - 1) * 1024);  // P is 0 on the padded keys
        };
        auto step = [&](int ks, const f32x4 (&src)[8]) {
            float pv[8];
            *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(S + r * ldp + ks * 16 + 8 * h);
            *reinterpret_cast<f32x4*>(pv + 4) = *reinterpret_cast<const f32x4*>(S + r * ldp + ks * 16 + 8 * h + 4);
            half8 ph, pl;
            split8(pv, ATT_PSC, ph, pl);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = src[j][n];
                half8 vh, vl;
                split8(t8, 1.f, vh, vl);
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, o[n], 0, 0, 0);
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, o[n], 0, 0, 0);
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, o[n], 0, 0, 0);
            }
        };
        const int nks = NK / 16;  // even
        load_v(0, vv[0]);
        for (int ks = 0; ks < nks; ks += 2) {  // two key steps per trip, the next one's V loads always in flight
            load_v(ks + 1, vv[1]);
            step(ks, vv[0]);
            load_v(min(ks + 2, nks - 1), vv[0]);
            step(ks + 1, vv[1]);
        }
        {
            const int c = col0 >> 6, f = col0 & 63;  // four adjacent columns: one channel, f .. f + 3
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int qq = q0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (qq < T)
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)b * 64 + hd * 16 + c) * T + qq) * AF + f) =
                        f32x4{o[0][q], o[1][q], o[2][q], o[3][q]} * (1.f / ATT_PSC);
            }
        }
    }
}

size_t attn_core_lds_bytes(int T) {
    const int NK = ((T + 31) >> 5) * 32;
    return (size_t)32 * (NK + 4) * sizeof(float);
}

// frames per workgroup: RCAN_ROWS when the launch fills the chip anyway, fewer for small batches (batch 1: 125 frames were 16 workgroups)
static int rcan_rows(int T, int B) {
    if ((long)T * B >= 3500) return RCAN_ROWS;
    const int r = (int)((long)T * B / 512);
    return r < 1 ? 1 : (r > RCAN_ROWS ? RCAN_ROWS : r);
}
int launch_row_can_qkv(const RowCanArgs& a_, int B, hipStream_t st) {
    RowCanArgs a = a_;
    a.rpw = rcan_rows(a.T, B);
    if (a.ngroups != 12 || a.group_start[8] != 32 || a.group_start[12] != 96) return RTFS_ERR_SHAPE;  // Q_h x4, K_h x4 (4 ch), V_h x4 (16 ch)
    hipLaunchKernelGGL(row_can_kernel<96>, dim3(cdiv(a.T, a.rpw), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_row_can_proj(const RowCanArgs& a_, int B, hipStream_t st) {
    RowCanArgs a = a_;
    a.rpw = rcan_rows(a.T, B);
    if (a.ngroups != 1) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(row_can_kernel<64>, dim3(cdiv(a.T, a.rpw), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_attn_core(const AttnArgs& a, int B, hipStream_t st) {
    if (a.T < 1 || a.T > 256) return RTFS_ERR_SHAPE;
    const size_t lds = attn_core_lds_bytes(a.T);
    if (rtfs_set_max_lds((const void*)attn_core_kernel, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    AttnArgs a2 = a;
    a2.npairs = B * 4;
    hipLaunchKernelGGL(attn_core_kernel, dim3(cdiv(a2.npairs, 8) * 8 * cdiv(a.T, 32)), dim3(256), lds, st, a2);
    return rtfs_launch_status();
}
