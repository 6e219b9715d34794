// TF-domain self-attention (reference MultiHeadSelfAttention2D, src/models/layers/attention.py:76-189).
//   row_can_kernel<96>   per (b,t) row: all heads' Q/K/V ConvActNorm = 1x1 conv -> PReLU -> LN over (E,F)
//                        (conv_layers.py:196-205, normalizations.py:20-41), written head-major
//                        Q,K (B,H,T,E*F)  V (B,H,T,Cv*F)     (attention.py:156-168)
//   attn_core_kernel     softmax_keys(Q K^T / sqrt(E*F)) V on the f32 matrix cores  (attention.py:171-175)
//   row_can_kernel<64>   attn_concat_proj ConvActNorm + residual (attention.py:182-184)
#include "common.h"
#include "kernels.h"

#define AF 64  // n_freqs: LayerNormalization4D gamma/beta are (1,C,1,64), so F' is tied to 64

// X row (64 ch x 64 f) -> NOUT channels; wave w owns NOUT/4 consecutive output channels, lane = f.
template <int NOUT>
__global__ __launch_bounds__(256) void row_can_kernel(RowCanArgs a) {
    constexpr int PER = NOUT / 4;
    __shared__ float Xs[64][AF];
    __shared__ float Ys[NOUT][AF];
    const int t = blockIdx.x, b = blockIdx.y, T = a.T;
    const int tid = threadIdx.x, f = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < 64 * AF; idx += 256) {
        const int c = idx >> 6, ff = idx & 63;
        Xs[c][ff] = a.x[(((size_t)b * 64 + c) * T + t) * AF + ff];
    }
    __syncthreads();
    float acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = a.bias[wave * PER + i];
    for (int c = 0; c < 64; ++c) {
        const float xv = Xs[c][f];
        const float* w = a.wt + c * NOUT + wave * PER;  // transposed weight (64, NOUT): wave-uniform -> scalar loads
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] = fmaf(w[i], xv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int o = wave * PER + i;
        Ys[o][f] = preluf_(acc[i], a.slope[a.group_of[o]]);
    }
    __syncthreads();
    // LayerNorm per group over (channels of the group, F)
    for (int g = wave; g < a.ngroups; g += 4) {
        const int o0 = a.group_start[g], gs = a.group_start[g + 1] - o0;
        float s = 0.f;
        for (int i = 0; i < gs; ++i) s += Ys[o0 + i][f];
        const float mean = wave_sum(s) / (float)(gs * AF);
        float v = 0.f;
        for (int i = 0; i < gs; ++i) {
            const float d = Ys[o0 + i][f] - mean;
            v = fmaf(d, d, v);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)(gs * AF) + RTFS_EPS);
        for (int i = 0; i < gs; ++i) {
            const int o = o0 + i;
            const float y = fmaf((Ys[o][f] - mean) * rstd, a.gamma[o * AF + f], a.beta[o * AF + f]);
            if (NOUT == 96) {
                // groups 0-3 Q_h, 4-7 K_h, 8-11 V_h
                const int kind = g >> 2, h = g & 3;
                if (kind < 2) {
                    float* dst = kind == 0 ? a.q : a.k;
                    dst[(((size_t)b * 4 + h) * T + t) * (4 * AF) + i * AF + f] = y;
                } else {
                    a.v[(((size_t)b * 4 + h) * T + t) * (16 * AF) + i * AF + f] = y;
                }
            } else {
                const size_t off = (((size_t)b * 64 + o) * T + t) * AF + f;
                a.out[off] = y + a.res[off];
            }
        }
    }
}

// One workgroup per (query tile of 32, head, b).  E = 256 (Q/K row), D = 1024 (V row).
__global__ __launch_bounds__(256) void attn_core_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = a.T;
    const int NKT = (T + 31) >> 5;      // key tiles
    const int NK = NKT * 32;
    const int ldp = NK + 1;
    float* Qs = lds;                    // [32][65]
    float* Ks = Qs + 32 * 65;           // [NK][65]
    float* S = Ks + (size_t)NK * 65;    // [32][NK+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const float* Q = a.q + ((size_t)b * 4 + h) * T * 256;
    const float* K = a.k + ((size_t)b * 4 + h) * T * 256;
    const float* V = a.v + ((size_t)b * 4 + h) * T * 1024;

    // ---- scores S = Q K^T / 16
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kc = 0; kc < 256; kc += 64) {
        __syncthreads();
        for (int idx = tid; idx < 32 * 64; idx += 256) {
            const int i = idx >> 6, k = idx & 63;
            const int q = min(q0 + i, T - 1);
            Qs[i * 65 + k] = Q[(size_t)q * 256 + kc + k];
        }
        for (int idx = tid; idx < NK * 64; idx += 256) {
            const int j = idx >> 6, k = idx & 63;
            const float kv = K[(size_t)min(j, T - 1) * 256 + kc + k];
            Ks[j * 65 + k] = j < T ? kv : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k0 = 0; k0 < 64; k0 += 2) {
            const int k = k0 + (lane >> 5);
            const float av = Qs[(lane & 31) * 65 + k];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kt = wave + 4 * j;
                if (kt < NKT) {
                    const float bv = Ks[(kt * 32 + (lane & 31)) * 65 + k];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kt = wave + 4 * j;
        if (kt < NKT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                S[row * ldp + kt * 32 + (lane & 31)] = acc[j][r] * a.scale;
            }
        }
    }
    __syncthreads();
    // ---- softmax over keys, 8 rows per wave
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
            for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
        const int nt0 = wave * 8 + grp * 4;
        const int KP = (T + 1) & ~1;
#pragma unroll 2
        for (int k0 = 0; k0 < KP; k0 += 2) {
            const int k = k0 + (lane >> 5);
            const float av = S[(lane & 31) * ldp + k];  // 0 for k >= T
            const float* vrow = V + (size_t)min(k, T - 1) * 1024 + nt0 * 32 + (lane & 31);
#pragma unroll
            for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, vrow[n * 32], o[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = (nt0 + n) * 32 + (lane & 31);
            const int c = col >> 6, f = col & 63;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (q < T) a.out[(((size_t)b * 64 + h * 16 + c) * T + q) * AF + f] = o[n][r];
            }
        }
    }
}

size_t attn_core_lds_bytes(int T) {
    const int NK = ((T + 31) >> 5) * 32;
    return ((size_t)32 * 65 + (size_t)NK * 65 + (size_t)32 * (NK + 1)) * sizeof(float);
}

int launch_row_can_qkv(const RowCanArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(row_can_kernel<96>, dim3(a.T, B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_row_can_proj(const RowCanArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(row_can_kernel<64>, dim3(a.T, B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_attn_core(const AttnArgs& a, int B, hipStream_t st) {
    if (a.T < 1 || a.T > 256) return RTFS_ERR_SHAPE;
    const size_t lds = attn_core_lds_bytes(a.T);
    static size_t configured = 0;
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)attn_core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RTFS_ERR_LAUNCH;
        configured = lds;
    }
    hipLaunchKernelGGL(attn_core_kernel, dim3(cdiv(a.T, 32), 4, B), dim3(256), lds, st, a);
    return rtfs_launch_status();
}
