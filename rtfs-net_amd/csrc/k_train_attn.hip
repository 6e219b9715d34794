// Training-side kernels of the two attention modules: MultiHeadSelfAttention2D (TF attention of the RTFS block) and the video-side
// MultiHeadSelfAttention (LayerNorm rows, nn.MultiheadAttention core).
#include "train_common.h"

// ------------------------------------------------------------------------------------------------ TF attention, training side
// MultiHeadSelfAttention2D (attention.py:149-189) on channel-last rows (b, t, f) x CZ.  "LNG" = the tail of a ConvActNorm
// (conv_layers.py:201-205): PReLU, then LayerNormalization4D((C_out, F)) = statistics over (channels of the module, F) per (b, t)
// with a (C_out, F) affine (normalizations.py:26,33-37).  The twelve Q/K/V modules are evaluated side by side: their channels are
// stacked (CZ = 128, 96 used) and each module is one "group".
__global__ __launch_bounds__(256) void att_lng_fwd_kernel(LngArgs a) {
    extern __shared__ float tile[];  // [64 f][CZ + 1]
    __shared__ float gm[16], gr[16];
    const int bt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, CZ = a.CZ, P = CZ + 1;
    const float* z = a.Z + (size_t)bt * 64 * CZ;
    for (int idx = tid; idx < 64 * CZ; idx += 256) {
        const int f = idx / CZ, c = idx - f * CZ;
        const float v = z[idx];
        tile[f * P + c] = v >= 0.f ? v : a.slope[c] * v;
    }
    __syncthreads();
    for (int g = wave; g < a.ngroups; g += 4) {
        const int c0 = a.gstart[g], gs = a.gstart[g + 1] - c0, n = 64 * gs;
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += tile[(i / gs) * P + c0 + i % gs];
        const float mean = wave_sum(s) / n;
        float v = 0.f;
        for (int i = lane; i < n; i += 64) {
            const float d = tile[(i / gs) * P + c0 + i % gs] - mean;
            v = fmaf(d, d, v);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / n + RTFS_EPS);
        if (lane == 0) {
            gm[g] = mean;
            gr[g] = rstd;
            a.stats[((size_t)bt * 16 + g) * 2] = mean;
            a.stats[((size_t)bt * 16 + g) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    float* y = a.Y + (size_t)bt * 64 * CZ;
    for (int idx = tid; idx < 64 * CZ; idx += 256) {
        const int f = idx / CZ, c = idx - f * CZ, g = a.gof[c];
        float v = 0.f;
        if (g < 16) v = fmaf((tile[f * P + c] - gm[g]) * gr[g], a.gamma[c * 64 + f], a.beta[c * 64 + f]);
        if (a.res) v += a.res[(size_t)bt * 64 * CZ + idx];
        y[idx] = v;
    }
}

// One workgroup per (b, t) slice.  256 % CZ == 0, so a thread keeps its channel c (hence its group, slope and statistics) over its
// 64 * CZ / 256 elements and only f moves.  The (C, F) affine's gradient contributions leave as one coalesced row of per-workgroup
// partials in slice order [f][c] (att_lng_reduce_kernel folds and transposes them: atomics from every workgroup onto the 2 x 8192
// addresses serialise, 346 -> 60 us); the PReLU slope gradient is summed in a register and merged once per thread (an LDS atomic per
// negative element - ~4000 onto 16 addresses per slice - was a quarter of this kernel).
__global__ __launch_bounds__(256) void att_lng_bwd_kernel(LngArgs a) {
    extern __shared__ float lds[];  // A [64][CZ+1] (xhat), D [64][CZ+1] (gamma * dY)
    __shared__ float g1[16], g2[16], gsl[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, CZ = a.CZ, P = CZ + 1, bt = blockIdx.x;
    float* A = lds;
    float* D = lds + 64 * P;
    const int c = tid & (CZ - 1), f0 = tid / CZ, fstep = 256 / CZ, per = 64 * CZ / 256;
    const int g = a.gof[c];
    const bool on = g < 16;
    const float slope = a.slope[c];
    const float mean = on ? a.stats[((size_t)bt * 16 + g) * 2] : 0.f, rstd = on ? a.stats[((size_t)bt * 16 + g) * 2 + 1] : 0.f;
    if (tid < 16) gsl[tid] = 0.f;
    const float* __restrict__ z = a.Z + (size_t)bt * 64 * CZ;
    const float* __restrict__ dy = a.dY + (size_t)bt * 64 * CZ;
    float* __restrict__ sc = a.scratch + (size_t)bt * 2 * CZ * 64;
    const int n = 64 * CZ;
#pragma unroll 8
    for (int k = 0; k < per; ++k) {
        const int idx = tid + 256 * k, f = f0 + fstep * k;
        const float v = z[idx], d = dy[idx];
        const float act = v >= 0.f ? v : slope * v;
        const float xh = on ? (act - mean) * rstd : 0.f;
        A[f * P + c] = xh;
        D[f * P + c] = on ? a.gamma[c * 64 + f] * d : 0.f;
        sc[idx] = on ? d * xh : 0.f;
        sc[n + idx] = on ? d : 0.f;
    }
    __syncthreads();
    for (int q = wave; q < a.ngroups; q += 4) {
        const int c0 = a.gstart[q], gs = a.gstart[q + 1] - c0, m = 64 * gs;
        float s1 = 0.f, s2 = 0.f;
        for (int i = lane; i < m; i += 64) {
            const int o = (i / gs) * P + c0 + i % gs;
            s1 += D[o];
            s2 = fmaf(D[o], A[o], s2);
        }
        s1 = wave_sum(s1) / m;
        s2 = wave_sum(s2) / m;
        if (lane == 0) {
            g1[q] = s1;
            g2[q] = s2;
        }
    }
    __syncthreads();
    float* __restrict__ dz = a.dZ + (size_t)bt * 64 * CZ;
    const float m1 = on ? g1[g] : 0.f, m2 = on ? g2[g] : 0.f;
    float dsl = 0.f;
#pragma unroll 8
    for (int k = 0; k < per; ++k) {
        const int idx = tid + 256 * k, f = f0 + fstep * k;
        const float dA = rstd * (D[f * P + c] - m1 - A[f * P + c] * m2);  // rstd = 0 for padding channels
        const float v = z[idx];
        dz[idx] = v >= 0.f ? dA : dA * slope;
        dsl += v >= 0.f ? 0.f : dA * v;
    }
    if (on && dsl != 0.f) atomicAdd(&gsl[g], dsl);
    __syncthreads();
    if (tid < a.ngroups && gsl[tid] != 0.f) unsafeAtomicAdd(a.dslope + tid, gsl[tid]);
}

// nwg rows of [d*xhat | d] in slice order (f, c) -> dgamma, dbeta (c, f) (+=)
__global__ __launch_bounds__(256) void att_lng_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int nwg, int n, int CZ) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // f * CZ + c
    if (i >= n) return;
    float g = 0.f, b = 0.f;
#pragma unroll 4
    for (int w = blockIdx.y; w < nwg; w += gridDim.y) {
        g += scratch[(size_t)w * 2 * n + i];
        b += scratch[(size_t)w * 2 * n + n + i];
    }
    const int o = (i % CZ) * 64 + i / CZ;
    unsafeAtomicAdd(dgamma + o, g);
    unsafeAtomicAdd(dbeta + o, b);
}

// Y rows (b,t,f) x 128 <-> Qp, Kp (4B, Tp, 256 = f*4 + e), Vp (4B, Tp, 1024 = f*16 + c); head-major batch index h*B + b
// (attention.py:160-168).  dir 0: rows -> packed, 1: packed -> rows (channels 96..127 of the rows get zero).
__global__ __launch_bounds__(256) void att_pack_qkv_kernel(float* __restrict__ rows, float* __restrict__ Qp, float* __restrict__ Kp,
                                                           float* __restrict__ Vp, int B, int T, int Tp, int dir) {
    const size_t total = (size_t)B * T * 64 * 128;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 127);
        const size_t r = i >> 7;
        const int f = (int)(r & 63);
        const size_t bt = r >> 6;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float* p;
        if (c < 16) p = Qp + (((size_t)(c >> 2) * B + b) * Tp + t) * 256 + f * 4 + (c & 3);
        else if (c < 32) p = Kp + (((size_t)((c - 16) >> 2) * B + b) * Tp + t) * 256 + f * 4 + (c & 3);
        else if (c < 96) p = Vp + (((size_t)((c - 32) >> 4) * B + b) * Tp + t) * 1024 + f * 16 + (c & 15);
        else p = nullptr;
        if (dir == 0) {
            if (p) *p = rows[i];
        } else {
            rows[i] = p ? *p : 0.f;
        }
    }
}
// O (4B, Tp, 1024 = f*16 + c) <-> rows (b,t,f) x 64 with channel h*16 + c (attention.py:178-181)
__global__ __launch_bounds__(256) void att_pack_o_kernel(float* __restrict__ rows, float* __restrict__ Op, int B, int T, int Tp, int dir) {
    const size_t total = (size_t)B * T * 64 * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 63);
        const size_t r = i >> 6;
        const int f = (int)(r & 63);
        const size_t bt = r >> 6;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float* p = Op + (((size_t)(c >> 4) * B + b) * Tp + t) * 1024 + f * 16 + (c & 15);
        if (dir == 0) rows[i] = *p;
        else *p = rows[i];
    }
}

// one wave per score row: P = softmax(scale * S[:T]) (zeros in the padding columns);  backward in place on dP:
// dS = scale * P * (dP - sum(P * dP))
__global__ __launch_bounds__(256) void att_softmax_kernel(float* __restrict__ S, const float* __restrict__ Pm, size_t nrows_total, int T,
                                                          int Tp, float scale, int bwd) {
    const int lane = threadIdx.x & 63;
    const size_t rid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rid >= nrows_total) return;
    // rows are stored (batch, Tp rows, Tp columns) but only the first T rows of a batch are scores
    const size_t batch = rid / T, row = rid % T;
    float* s = S + (batch * Tp + row) * Tp;
    if (Tp > 256) {  // long rows (inputs past 4 s): three passes over the row instead of four register-resident values per lane
        if (!bwd) {
            float m = -INFINITY;
            for (int k = lane; k < T; k += 64) m = fmaxf(m, scale * s[k]);
            m = wave_max(m);
            float sum = 0.f;
            for (int k = lane; k < T; k += 64) sum += __expf(scale * s[k] - m);
            const float inv = 1.0f / wave_sum(sum);
            for (int k = lane; k < Tp; k += 64) s[k] = k < T ? __expf(scale * s[k] - m) * inv : 0.f;
        } else {
            const float* pm = Pm + (batch * Tp + row) * Tp;
            float dot = 0.f;
            for (int k = lane; k < T; k += 64) dot = fmaf(pm[k], s[k], dot);
            dot = wave_sum(dot);
            for (int k = lane; k < Tp; k += 64) s[k] = k < T ? scale * pm[k] * (s[k] - dot) : 0.f;
        }
        return;
    }
    if (!bwd) {
        float v[4], m = -INFINITY;
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            v[i] = k < T ? scale * s[k] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int i = 0; i < 4; ++i) {
            v[i] = (lane + 64 * i) < T ? __expf(v[i] - m) : 0.f;
            sum += v[i];
        }
        const float inv = 1.0f / wave_sum(sum);
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < Tp) s[lane + 64 * i] = v[i] * inv;
    } else {
        const float* pm = Pm + (batch * Tp + row) * Tp;
        float pv[4], dv[4], dot = 0.f;
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            pv[i] = k < T ? pm[k] : 0.f;
            dv[i] = k < T ? s[k] : 0.f;
            dot = fmaf(pv[i], dv[i], dot);
        }
        dot = wave_sum(dot);
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < Tp) s[lane + 64 * i] = scale * pv[i] * (dv[i] - dot);
    }
}

size_t att_lng_scratch_floats(int nbt) { return (size_t)nbt * 2 * 128 * 64; }
int launch_att_lng(const LngArgs& a, int nbt, bool bwd, hipStream_t st) {
    if (a.CZ != 64 && a.CZ != 128) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)(bwd ? 2 : 1) * 64 * (a.CZ + 1) * sizeof(float);
    int rc = bwd ? set_lds(att_lng_bwd_kernel, lds) : set_lds(att_lng_fwd_kernel, lds);
    if (rc) return rc;
    if (bwd) {
        LngArgs b = a;
        b.nbt = nbt;
        if (!b.scratch) return RTFS_ERR_WORKSPACE;
        const int nwg = nbt;
        hipLaunchKernelGGL(att_lng_bwd_kernel, dim3(nwg), dim3(256), lds, st, b);
        hipLaunchKernelGGL(att_lng_reduce_kernel, dim3(cdiv(a.CZ * 64, 256), nwg >= 16 ? 16 : nwg), dim3(256), 0, st, b.scratch, a.dgamma, a.dbeta, nwg,
                           a.CZ * 64, a.CZ);
    }
    else hipLaunchKernelGGL(att_lng_fwd_kernel, dim3(nbt), dim3(256), lds, st, a);
    return rtfs_launch_status();
}
int launch_att_pack_qkv(float* rows, float* Qp, float* Kp, float* Vp, int B, int T, int Tp, int dir, hipStream_t st) {
    hipLaunchKernelGGL(att_pack_qkv_kernel, dim3(grid_for((size_t)B * T * 64 * 128)), dim3(256), 0, st, rows, Qp, Kp, Vp, B, T, Tp, dir);
    return rtfs_launch_status();
}
int launch_att_pack_o(float* rows, float* Op, int B, int T, int Tp, int dir, hipStream_t st) {
    hipLaunchKernelGGL(att_pack_o_kernel, dim3(grid_for((size_t)B * T * 64 * 64)), dim3(256), 0, st, rows, Op, B, T, Tp, dir);
    return rtfs_launch_status();
}
int launch_att_softmax(float* S, const float* P, int nbatch, int T, int Tp, float scale, bool bwd, hipStream_t st) {
    if (T < 1) return RTFS_ERR_SHAPE;
    const size_t rows = (size_t)nbatch * T;
    hipLaunchKernelGGL(att_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, rows, T, Tp, scale, bwd ? 1 : 0);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ video-side attention (1-D) kernels
// nn.LayerNorm(C) over the last axis of rows (N, C), C in {64, 128, ..., 1024 with C % 64 == 0}; one wave per row.
// bwd: dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); dgamma += dy*xhat, dbeta += dy (per-workgroup LDS sums, then atomics)
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, size_t N, int C, int bwd,
                                                      const float* __restrict__ res) {
    __shared__ float pg[4][1024], pb[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, per = C >> 6;
    // a lane owns channels lane + 64k in every row: the affine's gradients accumulate in registers
    float ag[16], ab[16], gm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        ag[k] = ab[k] = 0.f;
        gm[k] = k < per ? gamma[lane + 64 * k] : 0.f;
    }
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < N; row += (size_t)gridDim.x * 4) {
        const float* xr = x + row * C;
        float v[16], s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = k < per ? xr[lane + 64 * k] : 0.f;
            s += v[k];
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = k < per ? v[k] - mean : 0.f;
            q = fmaf(v[k], v[k], q);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / C + RTFS_EPS);
        if (!bwd) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < per) y[row * C + lane + 64 * k] = fmaf(v[k] * rstd, gm[k], beta[lane + 64 * k]);
        } else {
            float gd[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float d = k < per ? dy[row * C + lane + 64 * k] : 0.f;
                v[k] *= rstd;  // xhat
                gd[k] = gm[k] * d;
                s1 += gd[k];
                s2 = fmaf(gd[k], v[k], s2);
                ag[k] = fmaf(d, v[k], ag[k]);
                ab[k] += d;
            }
            s1 = wave_sum(s1) / C;
            s2 = wave_sum(s2) / C;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < per) dx[row * C + lane + 64 * k] = rstd * (gd[k] - s1 - v[k] * s2) + (res ? res[row * C + lane + 64 * k] : 0.f);
        }
    }
    if (bwd) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < per) {
                pg[wave][lane + 64 * k] = ag[k];
                pb[wave][lane + 64 * k] = ab[k];
            }
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += 256) {
            unsafeAtomicAdd(dgamma + i, pg[0][i] + pg[1][i] + pg[2][i] + pg[3][i]);
            unsafeAtomicAdd(dbeta + i, pb[0][i] + pb[1][i] + pb[2][i] + pb[3][i]);
        }
    }
}

// nn.MultiheadAttention's core for self-attention on packed projections: qkv rows (B*T, 3E) = [q | k | v], E = nh * hd, hd <= 16,
// T <= 256.  One workgroup per (b, head); thread t owns query row t (forward, dq) and key/value row t (dk, dv).
// pmask (optional, (B*nh, T, T)): dropout keep-mask on the attention probabilities already scaled by 1/(1-p) (train mode).
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ qkv, const float* __restrict__ pmask, float* __restrict__ o,
                                                       const float* __restrict__ dout, float* __restrict__ dqkv, int T, int nh, int hd, int bwd) {
    extern __shared__ float sm[];  // q, k, v [T][hd]; bwd: do [T][hd], m [T], l [T], D [T]
    const int bh = blockIdx.x, b = bh / nh, h = bh % nh, t = threadIdx.x, E = nh * hd;
    float* q = sm;
    float* k = q + T * hd;
    float* v = k + T * hd;
    float* dO = v + T * hd;
    float* rm = dO + T * hd;
    float* rl = rm + T;
    float* rD = rl + T;
    const float scale = rsqrtf((float)hd);
    for (int i = t; i < T * hd; i += 256) {
        const int tt = i / hd, d = i - tt * hd;
        const size_t base = ((size_t)b * T + tt) * 3 * E + h * hd + d;
        q[i] = qkv[base];
        k[i] = qkv[base + E];
        v[i] = qkv[base + 2 * E];
        if (bwd) dO[i] = dout[((size_t)b * T + tt) * E + h * hd + d];
    }
    __syncthreads();
    const float* pm = pmask ? pmask + (size_t)bh * T * T : nullptr;
    float qt[16], acc[16];
    if (t < T) {
        for (int d = 0; d < hd; ++d) qt[d] = q[t * hd + d];
        float m = -INFINITY;
        for (int j = 0; j < T; ++j) {
            float sc = 0.f;
            for (int d = 0; d < hd; ++d) sc = fmaf(qt[d], k[j * hd + d], sc);
            m = fmaxf(m, sc * scale);
        }
        float l = 0.f;
        for (int d = 0; d < hd; ++d) acc[d] = 0.f;
        float D = 0.f;
        for (int j = 0; j < T; ++j) {
            float sc = 0.f;
            for (int d = 0; d < hd; ++d) sc = fmaf(qt[d], k[j * hd + d], sc);
            const float e = __expf(sc * scale - m);
            l += e;
            const float w = pm ? e * pm[(size_t)t * T + j] : e;
            for (int d = 0; d < hd; ++d) acc[d] = fmaf(w, v[j * hd + d], acc[d]);
            if (bwd) {
                float dp = 0.f;
                for (int d = 0; d < hd; ++d) dp = fmaf(dO[t * hd + d], v[j * hd + d], dp);
                D = fmaf(w, dp, D);  // sum_j p_tj * mask_tj * dP_tj  (unnormalised by l here)
            }
        }
        const float inv = 1.0f / l;
        if (!bwd) {
            for (int d = 0; d < hd; ++d) o[((size_t)b * T + t) * E + h * hd + d] = acc[d] * inv;
        } else {
            rm[t] = m;
            rl[t] = inv;
            rD[t] = D * inv;
            // dq_t = scale * sum_j dS_tj k_j,  dS_tj = p_tj * (mask_tj * dP_tj - D_t)
            float dq[16];
            for (int d = 0; d < hd; ++d) dq[d] = 0.f;
            for (int j = 0; j < T; ++j) {
                float sc = 0.f, dp = 0.f;
                for (int d = 0; d < hd; ++d) {
                    sc = fmaf(qt[d], k[j * hd + d], sc);
                    dp = fmaf(dO[t * hd + d], v[j * hd + d], dp);
                }
                const float p = __expf(sc * scale - m) * inv;
                const float ds = p * ((pm ? pm[(size_t)t * T + j] : 1.f) * dp - D * inv);
                for (int d = 0; d < hd; ++d) dq[d] = fmaf(ds, k[j * hd + d], dq[d]);
            }
            for (int d = 0; d < hd; ++d) dqkv[((size_t)b * T + t) * 3 * E + h * hd + d] = dq[d] * scale;
        }
    }
    if (!bwd) return;
    __syncthreads();
    if (t < T) {  // thread t now owns key / value row j = t
        const int j = t;
        float kj[16], vj[16], dk[16], dv[16];
        for (int d = 0; d < hd; ++d) {
            kj[d] = k[j * hd + d];
            vj[d] = v[j * hd + d];
            dk[d] = dv[d] = 0.f;
        }
        for (int tt = 0; tt < T; ++tt) {
            float sc = 0.f, dp = 0.f;
            for (int d = 0; d < hd; ++d) {
                sc = fmaf(q[tt * hd + d], kj[d], sc);
                dp = fmaf(dO[tt * hd + d], vj[d], dp);
            }
            const float p = __expf(sc * scale - rm[tt]) * rl[tt];
            const float mk = pm ? pm[(size_t)tt * T + j] : 1.f;
            const float ds = p * (mk * dp - rD[tt]);
            for (int d = 0; d < hd; ++d) {
                dk[d] = fmaf(ds, q[tt * hd + d], dk[d]);
                dv[d] = fmaf(p * mk, dO[tt * hd + d], dv[d]);
            }
        }
        for (int d = 0; d < hd; ++d) {
            dqkv[((size_t)b * T + j) * 3 * E + E + h * hd + d] = dk[d] * scale;
            dqkv[((size_t)b * T + j) * 3 * E + 2 * E + h * hd + d] = dv[d];
        }
    }
}

int launch_ln_rows(const float* x, const float* gamma, const float* beta, float* y, const float* dy, float* dx, float* dgamma, float* dbeta,
                   size_t N, int C, bool bwd, hipStream_t st, const float* res) {
    if (C < 64 || C > 1024 || (C & 63)) return RTFS_ERR_SHAPE;
    size_t g = (N + 3) / 4;
    g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
    hipLaunchKernelGGL(ln_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, x, gamma, beta, y, dy, dx, dgamma, dbeta, N, C, bwd ? 1 : 0, res);
    return rtfs_launch_status();
}
int launch_mha_core(const float* qkv, const float* pmask, float* o, const float* dout, float* dqkv, int B, int T, int nh, int hd, bool bwd,
                    hipStream_t st) {
    if (T < 1 || T > 256 || hd < 1 || hd > 16 || nh < 1) return RTFS_ERR_SHAPE;
    const size_t lds = ((size_t)4 * T * hd + 3 * T) * sizeof(float);
    hipLaunchKernelGGL(mha_core_kernel, dim3(B * nh), dim3(256), lds, st, qkv, pmask, o, dout, dqkv, T, nh, hd, bwd ? 1 : 0);
    return rtfs_launch_status();
}

