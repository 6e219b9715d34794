// Training-side kernels of the recurrent part: SRU / LSTM / GRU scans (forward with saved state, backward) and the dual-path
// layout kernels around them.  GEMMs: k_train_gemm.hip; overview: the header of that file and DESIGN.md (Widening, rank 1).
#include "train_common.h"

// ------------------------------------------------------------------------------------------------ SRU scans
// Row addressing: step t of sequence n lives at row t*ts + n*ns of every (rows, width) array.  The operator alone uses the
// upstream (L, N, .) order (ts = N, ns = 1); the dual-path layout is sequence-major with pitch Ls = L + 7 (ts = 1, ns = Ls),
// where `pad` asks the wave to zero the 7 rows of its sequence that are not steps (see the layout notes further down).
// forward with saved state: h and c of every step go to HBM (the backward reads c; h feeds the next layer)
// Both scans are software-pipelined by hand: the loads of the next SRU_LOOK steps are issued into a second register set before the
// current SRU_LOOK steps are computed and stored.  (Left to the compiler, every step's loads stay behind the previous step's stores -
// it cannot prove they do not alias - and each step pays a full global-memory latency: 68 -> 30 us for the backward scan.)
#define SRU_LOOK 8
__global__ __launch_bounds__(256) void sru_scan_fwd_kernel(SruScanArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, dir = lane >> 5;
    const int n = blockIdx.x * 4 + wave;
    if (n >= a.N) return;
    const float vf = a.wc[lane], vr = a.wc[64 + lane], bf = a.bias[lane], br = a.bias[64 + lane];
    const int KC = a.KC, L = a.L;
    const size_t ts = a.ts, nb = (size_t)n * a.ns;
    auto row_of = [&](int s) { const int sc = min(s, L - 1); return (size_t)(dir ? L - 1 - sc : sc) * ts + nb; };
    auto load = [&](int s, float (&v)[4]) {
        const size_t row = row_of(s);
        const float* u = a.U + row * KC + lane;
        v[0] = u[0]; v[1] = u[64]; v[2] = u[128];
        v[3] = *(a.xin ? a.xin + row * 64 + lane : u + 192);
    };
    float cur[SRU_LOOK][4], nx[SRU_LOOK][4];
#pragma unroll
    for (int i = 0; i < SRU_LOOK; ++i) load(i, cur[i]);
    float c = 0.f;
    for (int s0 = 0; s0 < L; s0 += SRU_LOOK) {
#pragma unroll
        for (int i = 0; i < SRU_LOOK; ++i) load(s0 + SRU_LOOK + i, nx[i]);
#pragma unroll
        for (int i = 0; i < SRU_LOOK; ++i) {
            const int s = s0 + i;
            if (s < L) {
                const float u0 = cur[i][0], xp = cur[i][3];
                const float f = sigmoidf_(cur[i][1] + vf * c + bf), rg = sigmoidf_(cur[i][2] + vr * c + br);
                c = u0 + (c - u0) * f;
                const size_t row = row_of(s);
                a.c[row * 64 + lane] = c;
                a.h[row * 64 + lane] = xp + (c - xp) * rg;
            }
        }
#pragma unroll
        for (int i = 0; i < SRU_LOOK; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[i][q] = nx[i][q];
    }
    if (a.pad)  // h is stored 7 rows into its sequence slot: rows -7..-1 are the zero steps the conv-transpose windows read
        for (int i = 1; i <= 7; ++i) a.h[((long)nb - i) * 64 + lane] = 0.f;
}

// backward: walks each direction's steps in reverse; dc is the only carried quantity
__global__ __launch_bounds__(256) void sru_scan_bwd_kernel(SruScanArgs a) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, dir = lane >> 5;
    const int n = blockIdx.x * 4 + wave;
    const float vf = a.wc[lane], vr = a.wc[64 + lane], bf = a.bias[lane], br = a.bias[64 + lane];
    const int KC = a.KC, L = a.L;
    const size_t ts = a.ts, nb = (size_t)n * a.ns;
    float s_vf = 0.f, s_bf = 0.f, s_vr = 0.f, s_br = 0.f;
    if (n < a.N) {
        // step index s counts forward-scan order; the walk is s = L-1 .. 0.  k = L-1-s is the walk position.
        auto row_at = [&](int s) { const int sc = min(max(s, 0), L - 1); return (size_t)(dir ? L - 1 - sc : sc) * ts + nb; };
        auto load = [&](int k, float (&v)[6]) {  // walk position k -> step s = L-1-k (clamped: positions past the end are never used)
            const int s = L - 1 - k;
            const size_t row = row_at(s);
            const float* u = a.U + row * KC + lane;
            v[0] = u[0]; v[1] = u[64]; v[2] = u[128];
            v[3] = *(a.xin ? a.xin + row * 64 + lane : u + 192);
            v[4] = a.g[row * 64 + lane];
            v[5] = a.c[row_at(s - 1) * 64 + lane];  // c of the step before (ignored at s = 0)
        };
        float cur[SRU_LOOK][6], nx[SRU_LOOK][6];
#pragma unroll
        for (int i = 0; i < SRU_LOOK; ++i) load(i, cur[i]);
        float dc = 0.f;
        float ct = a.c[row_at(L - 1) * 64 + lane];
        for (int k0 = 0; k0 < L; k0 += SRU_LOOK) {
#pragma unroll
            for (int i = 0; i < SRU_LOOK; ++i) load(k0 + SRU_LOOK + i, nx[i]);
#pragma unroll
            for (int i = 0; i < SRU_LOOK; ++i) {
                const int s = L - 1 - (k0 + i);
                if (s >= 0) {
                    const float u0 = cur[i][0], u1 = cur[i][1], u2 = cur[i][2], xp = cur[i][3], gh = cur[i][4];
                    const float cp = s > 0 ? cur[i][5] : 0.f;
                    const float f = sigmoidf_(u1 + vf * cp + bf), rg = sigmoidf_(u2 + vr * cp + br);
                    const float dr = gh * (ct - xp), dct = dc + gh * rg, dxp = gh * (1.f - rg);
                    const float du0 = dct * (1.f - f), df = dct * (cp - u0);
                    const float dzf = df * f * (1.f - f), dzr = dr * rg * (1.f - rg);
                    dc = dct * f + dzf * vf + dzr * vr;
                    s_vf = fmaf(dzf, cp, s_vf);
                    s_bf += dzf;
                    s_vr = fmaf(dzr, cp, s_vr);
                    s_br += dzr;
                    const size_t row = row_at(s);
                    float* d = a.dU + row * KC + lane;
                    d[0] = du0;
                    d[64] = dzf;
                    d[128] = dzr;
                    *(a.xin ? a.dxp + row * 64 + lane : d + 192) = dxp;
                    ct = cp;
                }
            }
#pragma unroll
            for (int i = 0; i < SRU_LOOK; ++i)
#pragma unroll
                for (int q = 0; q < 6; ++q) cur[i][q] = nx[i][q];
        }
        if (a.pad)  // rows L..L+6 of the slot are not steps: the weight-gradient GEMMs sum over every row, so they must be zero
            for (int i = 0; i < 7; ++i) {
                const size_t row = nb + L + i;
                for (int m = 0; m < KC; m += 64) a.dU[row * KC + m + lane] = 0.f;
                if (a.xin) a.dxp[row * 64 + lane] = 0.f;
            }
    }
    red[wave][0][lane] = s_vf;
    red[wave][1][lane] = s_vr;
    red[wave][2][lane] = s_bf;
    red[wave][3][lane] = s_br;
    __syncthreads();
    // thread (wave = quantity, lane = unit) sums the four sequences
    const float tot = red[0][wave][lane] + red[1][wave][lane] + red[2][wave][lane] + red[3][wave][lane];
    float* dst = (wave < 2 ? a.dwc : a.dbias) + (wave & 1) * 64 + lane;
    unsafeAtomicAdd(dst, tot);
}

int launch_sru_scan_fwd(const SruScanArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(sru_scan_fwd_kernel, dim3(cdiv(a.N, 4)), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_sru_scan_bwd(const SruScanArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(sru_scan_bwd_kernel, dim3(cdiv(a.N, 4)), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ dual-path layout kernels
// Training layout of DualPathRNN (rnn_layers.py:136-162): sequence-major, channel-last.  Sequence n owns a slot of Ls = L + 7
// rows of 64 floats; an Unfold(8) window of step l is then the 512 contiguous floats starting at row n*Ls + l (feature order
// k*64 + c, the weights are permuted to match), so the unfolded matrix is an addressing mode of the GEMM's A operand and the
// ConvTranspose1d and both of their adjoints are the same GEMMs over windows.  Rows l >= L of a slot are not steps: GEMM
// outputs there are ignored, and everything the weight-gradient GEMMs sum over is kept zero there.
// Source tensor: (B, 64, R, Ls) with the sweep axis contiguous (the T-sweep goes through launch_transpose first).
namespace {
__device__ __forceinline__ size_t seq_base(int n, int R, int Ls) { return ((size_t)(n / R) * 64 * R + (n % R)) * Ls; }
}

// LayerNorm over channels per position + layout change: x -> xn[n*Ls + s][c].  blockIdx.y = chunk of 256 positions (any Ls).
__global__ __launch_bounds__(256) void dp_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ xn, int R, int Ls) {
    extern __shared__ float tile[];  // [64][len + 1]
    __shared__ float mu[256], rs[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int s0 = blockIdx.y * 256, len = min(256, Ls - s0), P = len + 1;
    const size_t base = seq_base(n, R, Ls) + s0, cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * len; idx += 256) {
        const int c = idx / len, s = idx - c * len;
        tile[c * P + s] = x[base + c * cs + s];
    }
    __syncthreads();
    if (tid < len) {
        const int s = tid;
        float m = 0.f;
        for (int c = 0; c < 64; ++c) m += tile[c * P + s];
        m *= (1.f / 64);
        float v = 0.f;
        for (int c = 0; c < 64; ++c) {
            const float d = tile[c * P + s] - m;
            v = fmaf(d, d, v);
        }
        mu[s] = m;
        rs[s] = 1.0f / sqrtf(v * (1.f / 64) + RTFS_EPS);
    }
    __syncthreads();
    const int c = tid & 63;
    const float g = gamma[c], b = beta[c];
    for (int s = tid >> 6; s < len; s += 4) xn[((size_t)n * Ls + s0 + s) * 64 + c] = fmaf((tile[c * P + s] - mu[s]) * rs[s], g, b);
}

// out = y[n*Ls + s][c] + bias[c] + x  (back to the (B, 64, R, Ls) layout).  blockIdx.y = chunk of 256 positions.
__global__ __launch_bounds__(256) void dp_out_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                     const float* __restrict__ x, float* __restrict__ out, int R, int Ls) {
    extern __shared__ float tile[];  // [len][65]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int s0 = blockIdx.y * 256, len = min(256, Ls - s0);
    const size_t base = seq_base(n, R, Ls) + s0, cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * len; idx += 256) tile[(idx >> 6) * 65 + (idx & 63)] = y[((size_t)n * Ls + s0) * 64 + idx] + bias[idx & 63];
    __syncthreads();
    for (int idx = tid; idx < 64 * len; idx += 256) {
        const int c = idx / len, s = idx - c * len;
        out[base + c * cs + s] = tile[s * 65 + c] + x[base + c * cs + s];
    }
}

// dy[n*Ls + s][c] = dout; dbias[c] += sum_s dout
__global__ __launch_bounds__(256) void dp_dy_kernel(const float* __restrict__ dout, float* __restrict__ dy, float* __restrict__ dbias,
                                                    int R, int Ls) {
    extern __shared__ float tile[];  // [Ls][65]
    __shared__ float part[4][64];
    const int n = blockIdx.x, tid = threadIdx.x;
    const size_t base = seq_base(n, R, Ls), cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * Ls; idx += 256) {
        const int c = idx / Ls, s = idx - c * Ls;
        tile[s * 65 + c] = dout[base + c * cs + s];
    }
    __syncthreads();
    const int c = tid & 63;
    float acc = 0.f;
    for (int s = tid >> 6; s < Ls; s += 4) {
        const float v = tile[s * 65 + c];
        dy[((size_t)n * Ls + s) * 64 + c] = v;
        acc += v;
    }
    part[tid >> 6][c] = acc;
    __syncthreads();
    if (tid < 64) unsafeAtomicAdd(dbias + tid, part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]);
}

// LayerNorm backward + residual: dx = rstd * (gamma*dxn - mean_c(gamma*dxn) - xhat * mean_c(gamma*dxn*xhat)) + dout,
// dgamma[c] += sum dxn * xhat, dbeta[c] += sum dxn  (normalizations.py:33-37 differentiated; statistics recomputed from x)
__global__ __launch_bounds__(256) void dp_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dxn,
                                                        const float* __restrict__ dout, const float* __restrict__ gamma,
                                                        float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        int R, int Ls) {
    extern __shared__ float lds[];  // X [64][Ls + 1] (becomes xhat), D [64][Ls + 1] (dxn)
    __shared__ float rs[256], ma[256], mb[256], part[2][4][64];
    const int n = blockIdx.x, tid = threadIdx.x, P = Ls + 1;
    float* X = lds;
    float* D = lds + 64 * P;
    const size_t base = seq_base(n, R, Ls), cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * Ls; idx += 256) {
        const int c = idx / Ls, s = idx - c * Ls;
        X[c * P + s] = x[base + c * cs + s];
    }
    for (int idx = tid; idx < 64 * Ls; idx += 256) D[(idx & 63) * P + (idx >> 6)] = dxn[(size_t)n * Ls * 64 + idx];
    __syncthreads();
    for (int s = tid; s < Ls; s += 256) {
        float m = 0.f;
        for (int c = 0; c < 64; ++c) m += X[c * P + s];
        m *= (1.f / 64);
        float v = 0.f;
        for (int c = 0; c < 64; ++c) {
            const float d = X[c * P + s] - m;
            v = fmaf(d, d, v);
        }
        const float r = 1.0f / sqrtf(v * (1.f / 64) + RTFS_EPS);
        float a = 0.f, b = 0.f;
        for (int c = 0; c < 64; ++c) {
            const float xh = (X[c * P + s] - m) * r;
            X[c * P + s] = xh;
            const float gd = gamma[c] * D[c * P + s];
            a += gd;
            b = fmaf(gd, xh, b);
        }
        rs[s] = r;
        ma[s] = a * (1.f / 64);
        mb[s] = b * (1.f / 64);
    }
    __syncthreads();
    {   // parameter gradients: thread (c = tid & 63) sums its quarter of the positions
        const int c = tid & 63;
        float sg = 0.f, sb = 0.f;
        for (int s = tid >> 6; s < Ls; s += 4) {
            const float d = D[c * P + s];
            sg = fmaf(d, X[c * P + s], sg);
            sb += d;
        }
        part[0][tid >> 6][c] = sg;
        part[1][tid >> 6][c] = sb;
    }
    for (int idx = tid; idx < 64 * Ls; idx += 256) {
        const int c = idx / Ls, s = idx - c * Ls;
        const float v = rs[s] * (gamma[c] * D[c * P + s] - ma[s] - X[c * P + s] * mb[s]);
        dx[base + c * cs + s] = v + dout[base + c * cs + s];
    }
    __syncthreads();
    if (tid < 128) {
        const int w = tid >> 6, c = tid & 63;
        unsafeAtomicAdd((w ? dbeta : dgamma) + c, part[w][0][c] + part[w][1][c] + part[w][2][c] + part[w][3][c]);
    }
}


int launch_dp_ln_fwd(const float* x, const float* gamma, const float* beta, float* xn, int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)64 * ((Ls < 256 ? Ls : 256) + 1) * sizeof(float);
    int rc = set_lds(dp_ln_fwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_ln_fwd_kernel, dim3(nseq, cdiv(Ls, 256)), dim3(256), lds, st, x, gamma, beta, xn, R, Ls);
    return rtfs_launch_status();
}
int launch_dp_out(const float* y, const float* bias, const float* x, float* out, int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)(Ls < 256 ? Ls : 256) * 65 * sizeof(float);
    int rc = set_lds(dp_out_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_out_kernel, dim3(nseq, cdiv(Ls, 256)), dim3(256), lds, st, y, bias, x, out, R, Ls);
    return rtfs_launch_status();
}
int launch_dp_dy(const float* dout, float* dy, float* dbias, int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8 || Ls > 256) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)Ls * 65 * sizeof(float);
    int rc = set_lds(dp_dy_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_dy_kernel, dim3(nseq), dim3(256), lds, st, dout, dy, dbias, R, Ls);
    return rtfs_launch_status();
}
int launch_dp_ln_bwd(const float* x, const float* dxn, const float* dout, const float* gamma, float* dx, float* dgamma, float* dbeta,
                     int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8 || Ls > 256) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)2 * 64 * (Ls + 1) * sizeof(float);
    int rc = set_lds(dp_ln_bwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_ln_bwd_kernel, dim3(nseq), dim3(256), lds, st, x, dxn, dout, gamma, dx, dgamma, dbeta, R, Ls);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ rows-layout helpers
// out = y + bias[c] + x over rows of C floats (the dual path's output in rows layout)
__global__ __launch_bounds__(256) void rows_bias_res_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ x,
                                                            float* __restrict__ out, size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = y[i] + bias[i & (C - 1)] + x[i];
}
// (B, H, W, C) -> (B, W, H, C): the T-sweep's sequences become contiguous runs of rows
__global__ __launch_bounds__(256) void rows_permute_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const size_t total = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H), b = (int)(p / H);
        y[(((size_t)b * W + w) * H + h) * C + c] = x[i];
    }
}
int launch_rows_bias_res(const float* y, const float* bias, const float* x, float* out, size_t n, int C, hipStream_t st) {
    if (C < 1 || (C & (C - 1))) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(rows_bias_res_kernel, dim3(grid_for(n)), dim3(256), 0, st, y, bias, x, out, n, C);
    return rtfs_launch_status();
}
int launch_rows_permute(const float* x, float* y, int B, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(rows_permute_kernel, dim3(grid_for((size_t)B * H * W * C)), dim3(256), 0, st, x, y, B, H, W, C);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ LSTM cell, training side
// nn.LSTM(512, 32, 4 layers, bidirectional) as DualPathRNN's other cell (rnn_layers.py:116-122); gates i, f, g, o.
// U = x . W_ih^T + (b_ih + b_hh) for both directions comes from the GEMM as rows x 256 (column dir*128 + gate*32 + j).  One wave per
// (sequence, direction): lane l < 32 owns gate rows i_j and g_j (j = l), lane l >= 32 rows f_j and o_j (j = l - 32), each with its
// 2 x 32 recurrent weights in registers; h_{t-1} is broadcast through LDS.  Saved for the backward: the four activated gates G
// (rows x 256), c, h (in the zero-padded slot layout the windows read) and h_{t-1} (rows x 64, for the W_hh gradient GEMM).
__global__ __launch_bounds__(256) void lstm_scan_fwd_kernel(LstmScanArgs a) {
    __shared__ float hs[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
    const long id = (long)blockIdx.x * 4 + wave;
    const bool live = id < 2L * a.N;
    const int n = live ? (int)(id >> 1) : 0, dir = (int)(id & 1);
    const int r0 = hi ? 32 + j : j, r1 = hi ? 96 + j : 64 + j;  // i|f and g|o rows
    float w0[32], w1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        w0[k] = a.whh[((size_t)dir * 128 + r0) * 32 + k];
        w1[k] = a.whh[((size_t)dir * 128 + r1) * 32 + k];
    }
    const size_t nb = (size_t)n * a.ns;
    float c = 0.f, hprev = 0.f;
    if (lane < 32) hs[wave][j] = 0.f;
    __syncthreads();
    for (int s = 0; s < a.L; ++s) {
        const int t = dir ? a.L - 1 - s : s;
        const size_t row = (size_t)t * a.ts + nb;
        float z0 = 0.f, z1 = 0.f;
        if (live) {
            z0 = a.U[row * 256 + dir * 128 + r0];
            z1 = a.U[row * 256 + dir * 128 + r1];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float hk = hs[wave][k];
            z0 = fmaf(w0[k], hk, z0);
            z1 = fmaf(w1[k], hk, z1);
        }
        const float a0 = sigmoidf_(z0), a1 = hi ? sigmoidf_(z1) : tanhf_(z1);
        const float fg = __shfl(a0, j + 32, 64), og = __shfl(a1, j + 32, 64);
        __syncthreads();  // every lane has read h_{t-1}
        if (live) {
            a.G[row * 256 + dir * 128 + r0] = a0;
            a.G[row * 256 + dir * 128 + r1] = a1;
        }
        if (lane < 32) {
            c = fg * c + a0 * a1;
            const float h = og * tanhf_(c);
            if (live) {
                a.c[row * 64 + dir * 32 + j] = c;
                a.h[row * 64 + dir * 32 + j] = h;
                a.hprev[row * 64 + dir * 32 + j] = hprev;
            }
            hprev = h;
            hs[wave][j] = h;
        }
        __syncthreads();
    }
    if (a.pad && live && dir == 0)
        for (int i = 1; i <= 7; ++i) a.h[((long)nb - i) * 64 + lane] = 0.f;
}

// backward: reverse walk; carried: dc and the recurrent part of dh.  Writes dU (= gradient w.r.t. the gate pre-activations).
__global__ __launch_bounds__(256) void lstm_scan_bwd_kernel(LstmScanArgs a) {
    __shared__ float dzs[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
    const long id = (long)blockIdx.x * 4 + wave;
    const bool live = id < 2L * a.N;
    const int n = live ? (int)(id >> 1) : 0, dir = (int)(id & 1);
    // W_hh^T: lane (k = j, half hi) holds W_hh[hi*64 + r][k] for r = 0..63
    float wt[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) wt[r] = a.whh[((size_t)dir * 128 + hi * 64 + r) * 32 + j];
    const size_t nb = (size_t)n * a.ns;
    float dc = 0.f, dh_rec = 0.f;
    for (int s = a.L - 1; s >= 0; --s) {
        const int t = dir ? a.L - 1 - s : s;
        const size_t row = (size_t)t * a.ts + nb;
        if (lane < 32) {
            float dz[4] = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                const float* g = a.G + row * 256 + dir * 128;
                const float ig = g[j], fg = g[32 + j], gg = g[64 + j], og = g[96 + j];
                const float ct = a.c[row * 64 + dir * 32 + j];
                const int tp = dir ? t + 1 : t - 1;
                const float cp = s > 0 ? a.c[((size_t)tp * a.ts + nb) * 64 + dir * 32 + j] : 0.f;
                const float dh = a.g[row * 64 + dir * 32 + j] + dh_rec;
                const float tc = tanhf_(ct);
                const float d_o = dh * tc;
                const float dct = fmaf(dh * og, 1.f - tc * tc, dc);
                dz[0] = dct * gg * ig * (1.f - ig);
                dz[1] = dct * cp * fg * (1.f - fg);
                dz[2] = dct * ig * (1.f - gg * gg);
                dz[3] = d_o * og * (1.f - og);
                dc = dct * fg;
                float* d = a.dU + row * 256 + dir * 128;
                d[j] = dz[0]; d[32 + j] = dz[1]; d[64 + j] = dz[2]; d[96 + j] = dz[3];
            }
            dzs[wave][j] = dz[0]; dzs[wave][32 + j] = dz[1]; dzs[wave][64 + j] = dz[2]; dzs[wave][96 + j] = dz[3];
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 64; ++r) acc = fmaf(wt[r], dzs[wave][hi * 64 + r], acc);
        acc += __shfl_xor(acc, 32, 64);
        dh_rec = acc;
        __syncthreads();
    }
    if (a.pad && live && dir == 0)
        for (int i = 0; i < 7; ++i)
            for (int m = 0; m < 256; m += 64) a.dU[(nb + a.L + i) * 256 + m + lane] = 0.f;
}

int launch_lstm_scan(const LstmScanArgs& a, bool bwd, hipStream_t st) {
    const long waves = 2L * a.N;
    if (bwd) hipLaunchKernelGGL(lstm_scan_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(lstm_scan_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ GRU cell (forward with saved state + backward)
// nn.GRU(512, 32, 4 layers, bidirectional): DualPathRNN's third cell (rnn_layers.py:116-122, rnn_type "GRU"); gates r, z, n:
//   r = s(U_r + hr_r), z = s(U_z + hr_z), n = tanh(U_n + r * hr_n), h' = (1 - z) n + z h,   hr = W_hh h + b_hh,  U = W_ih x + b_ih.
// U comes from the GEMM as rows x 192 (column dir*96 + gate*32 + j).  One wave per (sequence, direction): lane j < 32 owns rows r_j and
// n_j, lane 32 + j row z_j.  Saved: r, z, n, hr_n as S (rows x 256, column dir*128 + q*32 + j), h (slot layout), h_{t-1}.
__global__ __launch_bounds__(256) void gru_scan_fwd_kernel(GruScanArgs a) {
    __shared__ float hs[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
    const long id = (long)blockIdx.x * 4 + wave;
    const bool live = id < 2L * a.N;
    const int n = live ? (int)(id >> 1) : 0, dir = (int)(id & 1);
    const int r0 = hi ? 32 + j : j, r1 = 64 + j;  // r|z row, n row (lanes < 32 only)
    float w0[32], w1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        w0[k] = a.whh[((size_t)dir * 96 + r0) * 32 + k];
        w1[k] = a.whh[((size_t)dir * 96 + r1) * 32 + k];
    }
    const float b0 = a.bhh[dir * 96 + r0], b1 = a.bhh[dir * 96 + r1];
    const size_t nb = (size_t)n * a.ns;
    float h = 0.f;
    if (lane < 32) hs[wave][j] = 0.f;
    __syncthreads();
    for (int s = 0; s < a.L; ++s) {
        const int t = dir ? a.L - 1 - s : s;
        const size_t row = (size_t)t * a.ts + nb;
        float z0 = b0, z1 = b1;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float hk = hs[wave][k];
            z0 = fmaf(w0[k], hk, z0);
            z1 = fmaf(w1[k], hk, z1);
        }
        const float u0 = live ? a.U[row * 192 + dir * 96 + r0] : 0.f;
        const float un = (live && !hi) ? a.U[row * 192 + dir * 96 + r1] : 0.f;
        const float g0 = sigmoidf_(u0 + z0);             // r (lanes < 32) or z (lanes >= 32)
        const float zg = __shfl(g0, j + 32, 64);
        __syncthreads();  // every lane has read h_{t-1}
        if (lane < 32) {
            const float ng = tanhf_(fmaf(g0, z1, un));
            const float hn = fmaf(1.f - zg, ng, zg * h);
            if (live) {
                float* sv = a.S + row * 256 + dir * 128;
                sv[j] = g0; sv[32 + j] = zg; sv[64 + j] = ng; sv[96 + j] = z1;
                a.hprev[row * 64 + dir * 32 + j] = h;
                a.h[row * 64 + dir * 32 + j] = hn;
            }
            h = hn;
            hs[wave][j] = hn;
        }
        __syncthreads();
    }
    if (a.pad && live && dir == 0)
        for (int i = 1; i <= 7; ++i) a.h[((long)nb - i) * 64 + lane] = 0.f;
}

// backward: writes dU (gradient w.r.t. W_ih x + b_ih) and dHR (w.r.t. W_hh h + b_hh), both rows x 192
__global__ __launch_bounds__(256) void gru_scan_bwd_kernel(GruScanArgs a) {
    __shared__ float dhr[4][96];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
    const long id = (long)blockIdx.x * 4 + wave;
    const bool live = id < 2L * a.N;
    const int n = live ? (int)(id >> 1) : 0, dir = (int)(id & 1);
    // W_hh^T: lane (k = j, half hi) holds W_hh[hi*48 + r][k] for r = 0..47
    float wt[48];
#pragma unroll
    for (int r = 0; r < 48; ++r) wt[r] = a.whh[((size_t)dir * 96 + hi * 48 + r) * 32 + j];
    const size_t nb = (size_t)n * a.ns;
    float dh_rec = 0.f;
    for (int s = a.L - 1; s >= 0; --s) {
        const int t = dir ? a.L - 1 - s : s;
        const size_t row = (size_t)t * a.ts + nb;
        if (lane < 32) {
            float d_r = 0.f, d_z = 0.f, d_n = 0.f, d_hn = 0.f;
            if (live) {
                const float* sv = a.S + row * 256 + dir * 128;
                const float rg = sv[j], zg = sv[32 + j], ng = sv[64 + j], hrn = sv[96 + j];
                const float hp = a.hprev[row * 64 + dir * 32 + j];
                const float dh = a.g[row * 64 + dir * 32 + j] + dh_rec;
                const float dn = dh * (1.f - zg);
                d_z = dh * (hp - ng) * zg * (1.f - zg);
                d_n = dn * (1.f - ng * ng);
                d_r = d_n * hrn * rg * (1.f - rg);
                d_hn = d_n * rg;
                dh_rec = dh * zg;  // the direct path; the recurrent-matrix part is added below
                float* du = a.dU + row * 192 + dir * 96;
                du[j] = d_r; du[32 + j] = d_z; du[64 + j] = d_n;
                float* dq = a.dHR + row * 192 + dir * 96;
                dq[j] = d_r; dq[32 + j] = d_z; dq[64 + j] = d_hn;
            } else {
                dh_rec = 0.f;
            }
            dhr[wave][j] = d_r; dhr[wave][32 + j] = d_z; dhr[wave][64 + j] = d_hn;
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 48; ++r) acc = fmaf(wt[r], dhr[wave][hi * 48 + r], acc);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 32) dh_rec += acc;
        __syncthreads();
    }
    if (a.pad && live && dir == 0)
        for (int i = 0; i < 7; ++i)
            for (int m = 0; m < 192; m += 64) {
                a.dU[(nb + a.L + i) * 192 + m + lane] = 0.f;
                a.dHR[(nb + a.L + i) * 192 + m + lane] = 0.f;
            }
}

int launch_gru_scan(const GruScanArgs& a, bool bwd, hipStream_t st) {
    const long waves = 2L * a.N;
    if (bwd) hipLaunchKernelGGL(gru_scan_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gru_scan_fwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

