// Training-side kernels of the convolutional modules on channel-last rows: ConvNormAct stages (gLN / BatchNorm / activations),
// depthwise convolutions, and the glue with adjoints (pooling, TFAR / CAF combine, encoder / decoder / S^3).
#include "train_common.h"
#define CL_STAGE_MAX_WG 256                      // workgroups per sample of the backward reduction pass
#define CL_STAGE_PITCH(C) (2 * (C) + 6)          // floats per workgroup row: dgamma C | dbeta C | dslope, pad | S1, S2 as doubles

// ------------------------------------------------------------------------------------------------ channel-last training kernels
// ConvNormAct (conv_layers.py:65-129) in training: activations as rows (b, h, w) x C channels, C fastest.  A stage
// "norm + act" is y = act((x - mean_b) * rstd_b * gamma_c + beta_c) with gLN statistics per sample (normalizations.py:8-17).
// act: 0 none, 1 ReLU, 2 PReLU (one slope), 3 Sigmoid.
namespace {
__device__ __forceinline__ void stats_of(const double* st, int b, double inv_n, float& mean, float& rstd) {
    const double m = st[2 * b] * inv_n;
    double var = st[2 * b + 1] * inv_n - m * m;
    var = var < 0 ? 0 : var;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)RTFS_EPS));
}
__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
    if (act == 1) return fmaxf(z, 0.f);
    if (act == 2) return z >= 0.f ? z : slope * z;
    if (act == 3) return 1.0f / (1.0f + __expf(-z));
    return z;
}
// d act / dz times dy; for PReLU also the slope's gradient contribution
__device__ __forceinline__ float act_bwd(float z, float dy, int act, float slope, float& dslope) {
    if (act == 1) return z > 0.f ? dy : 0.f;
    if (act == 2) {
        if (z >= 0.f) return dy;
        dslope += dy * z;
        return dy * slope;
    }
    if (act == 3) {
        const float y = 1.0f / (1.0f + __expf(-z));
        return dy * y * (1.f - y);
    }
    return dy;
}
}  // namespace

// grid (chunks, B): each workgroup a strided share of one sample's n = rows*C elements.
// norm: 0 none, 1 gLN (per-sample statistics), 2 BatchNorm with frozen running statistics (per-channel mean / variance; the
// eval-mode arithmetic of conv_layers.py's BatchNorm stages, used when a model is fine-tuned with its BN layers in eval mode),
// 3 BatchNorm in train mode (per-channel statistics of the batch, cl_chan_stats_kernel; dx = gamma*rstd*(da - dbeta/n - xhat*dgamma/n),
// where dgamma, dbeta are exactly the per-channel sums the reduction pass produces anyway).
namespace {
__device__ __forceinline__ void norm_of(const ClStageArgs& a, int b, int c, float& mean, float& rstd) {
    if (a.norm == 2) {
        mean = a.rmean[c];
        rstd = 1.0f / sqrtf(a.rvar[c] + RTFS_EPS);
    } else if (a.norm == 3) {  // BatchNorm in train mode: statistics of this batch, per channel (biased variance)
        const double m = a.cstats[2 * c] * a.inv_rows;
        double var = a.cstats[2 * c + 1] * a.inv_rows - m * m;
        var = var < 0 ? 0 : var;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)RTFS_EPS));
    }
}
}  // namespace
// The three stage kernels walk a sample in units of V adjacent channels per thread (V = 4: 16-byte accesses - a dword stream runs at
// 60 % of their rate - whenever 4 | C; V = 1 for narrower tensors).  The grid stride is a multiple of C / V units, so a thread keeps
// its channels and loads their gamma / beta / statistics once.
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
namespace {
template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, size_t i, float (&o)[V]) {
    if constexpr (V == 4) {
        const f32x4u_t v = reinterpret_cast<const f32x4u_t*>(p)[i];
        o[0] = v[0], o[1] = v[1], o[2] = v[2], o[3] = v[3];
    } else {
        o[0] = p[i];
    }
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, size_t i, const float (&o)[V]) {
    if constexpr (V == 4) {
        const f32x4u_t v = {o[0], o[1], o[2], o[3]};
        reinterpret_cast<f32x4u_t*>(p)[i] = v;
    } else {
        p[i] = o[0];
    }
}
template <int V>
struct ChanNorm {
    float g[V], be[V], mean[V], rstd[V];
    __device__ __forceinline__ ChanNorm(const ClStageArgs& a, int b, int c0) {
        float m0 = 0.f, r0 = 1.f;
        if (a.norm == 1) stats_of(a.stats, b, 1.0 / (double)a.n, m0, r0);
#pragma unroll
        for (int k = 0; k < V; k++) {
            mean[k] = m0, rstd[k] = r0;
            norm_of(a, b, c0 + k, mean[k], rstd[k]);
            g[k] = a.norm ? a.gamma[c0 + k] : 1.f;
            be[k] = a.norm ? a.beta[c0 + k] : 0.f;
        }
    }
};
}  // namespace
template <int V>
__global__ __launch_bounds__(256) void cl_norm_act_fwd_kernel(ClStageArgs a) {
    const int b = blockIdx.y;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, nv = a.n / V;
    const ChanNorm<V> p(a, b, (int)((i0 * V) & (size_t)(a.C - 1)));
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const float* __restrict__ x = a.x + (size_t)b * a.n;
    float* __restrict__ y = a.y + (size_t)b * a.n;
#pragma unroll 4
    for (size_t i = i0; i < nv; i += (size_t)gridDim.x * 256) {
        float v[V];
        ldv<V>(x, i, v);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const float z = a.norm ? fmaf((v[k] - p.mean[k]) * p.rstd[k], p.g[k], p.be[k]) : v[k];
            v[k] = act_fwd(z, a.act, slope);
        }
        stv<V>(y, i, v);
    }
}

// reductions of the stage's backward: per sample S1 = sum da*gamma, S2 = sum da*gamma*xhat (f64 atomics into S[2b..], gLN only);
// per channel dgamma += sum da*xhat, dbeta += sum da; dslope.
template <int V>
__global__ __launch_bounds__(256) void cl_norm_act_bwd_reduce_kernel(ClStageArgs a) {
    __shared__ double red[16];
    __shared__ float part[2 * V + 1][256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * 256 + tid, nv = a.n / V;
    const int c0 = (int)((i0 * V) & (size_t)(a.C - 1));
    const ChanNorm<V> p(a, b, c0);
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const float* __restrict__ x = a.x + (size_t)b * a.n;
    const float* __restrict__ dy = a.dy + (size_t)b * a.n;
    float s1 = 0.f, s2 = 0.f, dsl = 0.f, dg[V], db[V];
#pragma unroll
    for (int k = 0; k < V; k++) dg[k] = db[k] = 0.f;
#pragma unroll 4
    for (size_t i = i0; i < nv; i += (size_t)gridDim.x * 256) {
        float xv[V], dv[V];
        ldv<V>(x, i, xv);
        ldv<V>(dy, i, dv);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const float xh = a.norm ? (xv[k] - p.mean[k]) * p.rstd[k] : xv[k];
            const float z = a.norm ? fmaf(xh, p.g[k], p.be[k]) : xv[k];
            const float da = act_bwd(z, dv[k], a.act, slope, dsl);
            dg[k] = fmaf(da, xh, dg[k]);
            db[k] += da;
            s1 = fmaf(da, p.g[k], s1);
            s2 = fmaf(da * p.g[k], xh, s2);
        }
    }
    // No atomics here: a thousand workgroups landing on the same 2C + 3 addresses serialise (~0.1 us each: 100 us for a 33 MB tensor
    // whose streaming takes 15).  Every workgroup stores its sums as one row of `partial`; cl_stage_reduce2_kernel folds the rows.
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    float* __restrict__ row = a.partial + (size_t)wg * CL_STAGE_PITCH(a.C);
    {
        const double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
        if ((tid & 63) == 0) red[2 * (tid >> 6)] = d1, red[2 * (tid >> 6) + 1] = d2;
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
        part[2 * k][tid] = dg[k];
        part[2 * k + 1][tid] = db[k];
    }
    part[2 * V][tid] = dsl;
    __syncthreads();
    const int CV = a.C / V;  // threads with distinct channels (<= 256: V = 1 serves C < 4 only)
    if (tid < CV) {
#pragma unroll
        for (int k = 0; k < V; k++) {
            float sg = 0.f, sb = 0.f;
            for (int j = tid; j < 256; j += CV) {
                sg += part[2 * k][j];
                sb += part[2 * k + 1][j];
            }
            row[c0 + k] = sg;
            row[a.C + c0 + k] = sb;
        }
    }
    if (tid < 64) {
        float v = part[2 * V][tid] + part[2 * V][tid + 64] + part[2 * V][tid + 128] + part[2 * V][tid + 192];
        v = wave_sum(v);
        if (tid == 0) {
            row[2 * a.C] = v;
            double* sd = reinterpret_cast<double*>(row + 2 * a.C + 2);  // 8-byte aligned: the pitch and 2C + 2 are even
            sd[0] = (red[0] + red[2]) + (red[4] + red[6]);
            sd[1] = (red[1] + red[3]) + (red[5] + red[7]);
        }
    }
}

// Second stage: rows of [dgamma C | dbeta C | dslope, -, S1 (f64), S2 (f64)] -> dgamma / dbeta / dslope (+=) and S (B, 2) (=).
// grid (ceil(2C / 64) + 1, Y): 64 columns x 4 row lanes per workgroup, Y row slices meet in one atomic per column; the last column block
// (y = 0 only) folds the PReLU slope and the per-sample pair.
__global__ __launch_bounds__(256) void cl_stage_reduce2_kernel(const float* __restrict__ partial, int gx, int B, int C, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dslope, double* __restrict__ S,
                                                               int norm, int act) {
    __shared__ float red[4][64];
    __shared__ double dred[2][256];
    const int tid = threadIdx.x, pitch = CL_STAGE_PITCH(C), nwg = gx * B;
    if (blockIdx.x + 1 < gridDim.x) {
        if (!norm) return;
        const int cl = tid & 63, lane = tid >> 6, col = blockIdx.x * 64 + cl;
        float v = 0.f;
        if (col < 2 * C) {
#pragma unroll 8
            for (int w = blockIdx.y * 4 + lane; w < nwg; w += gridDim.y * 4) v += partial[(size_t)w * pitch + col];
        }
        red[lane][cl] = v;
        __syncthreads();
        if (lane == 0 && col < 2 * C) {
            v = (v + red[1][cl]) + (red[2][cl] + red[3][cl]);
            unsafeAtomicAdd(col < C ? dgamma + col : dbeta + (col - C), v);
        }
        return;
    }
    if (blockIdx.y) return;
    if (act == 2) {
        float v = 0.f;
        for (int w = tid; w < nwg; w += 256) v += partial[(size_t)w * pitch + 2 * C];
        v = wave_sum(v);
        if ((tid & 63) == 0) red[0][tid >> 6] = v;
        __syncthreads();
        if (tid == 0) unsafeAtomicAdd(dslope, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    }
    if (norm == 1) {
        for (int b = 0; b < B; ++b) {
            double s1 = 0, s2 = 0;
            for (int w = tid; w < gx; w += 256) {
                const double* sd = reinterpret_cast<const double*>(partial + (size_t)(b * gx + w) * pitch + 2 * C + 2);
                s1 += sd[0];
                s2 += sd[1];
            }
            s1 = wave_sum_d(s1), s2 = wave_sum_d(s2);
            __syncthreads();
            if ((tid & 63) == 0) dred[0][tid >> 6] = s1, dred[1][tid >> 6] = s2;
            __syncthreads();
            if (tid == 0) {
                S[2 * b] = (dred[0][0] + dred[0][1]) + (dred[0][2] + dred[0][3]);
                S[2 * b + 1] = (dred[1][0] + dred[1][1]) + (dred[1][2] + dred[1][3]);
            }
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void cl_norm_act_bwd_apply_kernel(ClStageArgs a) {
    const int b = blockIdx.y;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, nv = a.n / V;
    const int c0 = (int)((i0 * V) & (size_t)(a.C - 1));
    const ChanNorm<V> p(a, b, c0);
    float m1 = 0.f, m2 = 0.f, cb[V], cg[V];
    if (a.norm == 1) {
        m1 = (float)(a.S[2 * b] / (double)a.n);
        m2 = (float)(a.S[2 * b + 1] / (double)a.n);
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
        cb[k] = a.norm == 3 ? (float)a.inv_rows * a.dbeta[c0 + k] : 0.f;
        cg[k] = a.norm == 3 ? (float)a.inv_rows * a.dgamma[c0 + k] : 0.f;
    }
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const float* __restrict__ x = a.x + (size_t)b * a.n;
    const float* __restrict__ dy = a.dy + (size_t)b * a.n;
    float* __restrict__ dx = a.dx + (size_t)b * a.n;
    float dummy = 0.f;
#pragma unroll 4
    for (size_t i = i0; i < nv; i += (size_t)gridDim.x * 256) {
        float xv[V], dv[V];
        ldv<V>(x, i, xv);
        ldv<V>(dy, i, dv);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const float xh = a.norm ? (xv[k] - p.mean[k]) * p.rstd[k] : xv[k];
            const float z = a.norm ? fmaf(xh, p.g[k], p.be[k]) : xv[k];
            const float da = act_bwd(z, dv[k], a.act, slope, dummy);
            float out = da;
            if (a.norm == 1) out = p.rstd[k] * (da * p.g[k] - m1 - xh * m2);
            else if (a.norm == 2) out = da * p.g[k] * p.rstd[k];
            else if (a.norm == 3) out = p.g[k] * p.rstd[k] * (da - (cb[k] + xh * cg[k]));
            dv[k] = out;
        }
        stv<V>(dx, i, dv);
    }
}

// ------------------------------------------------------------------------------------------------ RTFS block gateway, one pass each way
// tdanet.py:30-38,106-108: gateway = ConvNormAct(C, C, 1, groups = C, act PReLU) applied to x (+ x_res) - on rows that is
// y = PReLU(w_c * (x + x_res) + b_c).  As separate modules it cost an add, a depthwise pass and an activation pass forward and five
// passes backward over the block's largest tensor (133 MB at batch 4); here each direction is one pass: the backward recomputes z from
// x (+ x_res), writes dx (the gradient of both addends) and leaves per-workgroup rows [dw C | db C | dslope] for cl_stage_reduce2_kernel.
template <bool BWD>
__global__ __launch_bounds__(256) void gateway_kernel(GatewayArgs a) {
    __shared__ float part[9][256];
    const int tid = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * 256 + tid;
    const int c0 = (int)((i0 * 4) & (size_t)(a.C - 1));
    const f32x4u_t w4 = *reinterpret_cast<const f32x4u_t*>(a.w + c0), b4 = *reinterpret_cast<const f32x4u_t*>(a.b + c0);
    const float slope = a.slope[0];
    f32x4u_t dw = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    float dsl = 0.f;
#pragma unroll 4
    for (size_t i = i0; i < a.n4; i += (size_t)gridDim.x * 256) {
        f32x4u_t v = reinterpret_cast<const f32x4u_t*>(a.x)[i];
        if (a.xr) v += reinterpret_cast<const f32x4u_t*>(a.xr)[i];
        const f32x4u_t z = w4 * v + b4;
        if (!BWD) {
            f32x4u_t y;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = z[k] >= 0.f ? z[k] : slope * z[k];
            reinterpret_cast<f32x4u_t*>(a.y)[i] = y;
        } else {
            const f32x4u_t d = reinterpret_cast<const f32x4u_t*>(a.dy)[i];
            f32x4u_t da;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                da[k] = z[k] >= 0.f ? d[k] : slope * d[k];
                dsl += z[k] >= 0.f ? 0.f : d[k] * z[k];
            }
            dw += da * v;
            db += da;
            reinterpret_cast<f32x4u_t*>(a.dx)[i] = da * w4;
        }
    }
    if (!BWD) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        part[2 * k][tid] = dw[k];
        part[2 * k + 1][tid] = db[k];
    }
    part[8][tid] = dsl;
    __syncthreads();
    float* __restrict__ row = a.partial + (size_t)blockIdx.x * CL_STAGE_PITCH(a.C);
    const int CV = a.C >> 2;
    if (tid < CV) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sw = 0.f, sb = 0.f;
            for (int j = tid; j < 256; j += CV) {
                sw += part[2 * k][j];
                sb += part[2 * k + 1][j];
            }
            row[c0 + k] = sw;
            row[a.C + c0 + k] = sb;
        }
    }
    if (tid < 64) {
        float v = part[8][tid] + part[8][tid + 64] + part[8][tid + 128] + part[8][tid + 192];
        v = wave_sum(v);
        if (tid == 0) row[2 * a.C] = v;
    }
}

// per-channel sum and sum of squares over all rows (BatchNorm batch statistics), f64 atomics; grid stride a multiple of C
__global__ __launch_bounds__(256) void cl_chan_stats_kernel(const float* __restrict__ x, double* __restrict__ st, size_t n, int C) {
    __shared__ double part[2][256];
    const int tid = threadIdx.x;
    double s = 0, ss = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
        const double v = x[i];
        s += v;
        ss += v * v;
    }
    if (C >= 256) {
        const size_t c = ((size_t)blockIdx.x * 256 + tid) & (C - 1);
        atomicAdd(st + 2 * c, s);
        atomicAdd(st + 2 * c + 1, ss);
        return;
    }
    part[0][tid] = s;
    part[1][tid] = ss;
    __syncthreads();
    if (tid < C) {
        double a = 0, b = 0;
        for (int j = tid; j < 256; j += C) {
            a += part[0][j];
            b += part[1][j];
        }
        atomicAdd(st + 2 * tid, a);
        atomicAdd(st + 2 * tid + 1, b);
    }
}
// running_mean / running_var update of nn.BatchNorm (momentum m, unbiased variance for the running estimate)
__global__ void bn_update_kernel(const double* __restrict__ st, float* __restrict__ rmean, float* __restrict__ rvar, int C, double rows,
                                 float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = st[2 * c] / rows;
    double var = st[2 * c + 1] / rows - m * m;
    var = var < 0 ? 0 : var;
    const double unb = rows > 1 ? var * rows / (rows - 1) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
}

// out[c] += sum over rows of d[row][c]   (bias gradients).  A thread owns four adjacent channels (16-byte loads; 4 | C, and the grid
// stride in float4 columns is a multiple of C / 4), the workgroup folds its 256 partial quads onto C / 4 and ends in C atomics:
// few, long workgroups - every one of them lands on the same C addresses.
__global__ __launch_bounds__(256) void cl_colsum_kernel(const float* __restrict__ d, float* __restrict__ out, size_t n, int C,
                                                        float* __restrict__ partial) {
    __shared__ float4 part[256];
    const int tid = threadIdx.x, C4 = C >> 2;
    const size_t n4 = n >> 2;
    f32x4u_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) s += reinterpret_cast<const f32x4u_t*>(d)[i];
    part[tid] = make_float4(s[0], s[1], s[2], s[3]);
    __syncthreads();
    if (tid < C4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = tid; j < 256; j += C4) {
            const float4 q = part[j];
            v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
        }
        if (partial) {  // two-stage: one row per workgroup, folded by cl_dw_wgrad_reduce_kernel (taps = 1)
            *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * C + 4 * tid) = v;
            return;
        }
        unsafeAtomicAdd(out + 4 * tid, v.x);
        unsafeAtomicAdd(out + 4 * tid + 1, v.y);
        unsafeAtomicAdd(out + 4 * tid + 2, v.z);
        unsafeAtomicAdd(out + 4 * tid + 3, v.w);
    }
}

// depthwise k x k convolution, channel-last: x (B, H, W, C), w (C, kh*kw), y (B, Ho, Wo, C); cross-correlation with
// top/left padding (pt, pl) and stride s (conv_layers.py:100-101: "same" k = 4 -> pt = pl = 1, stride 2 -> symmetric 1)
__global__ __launch_bounds__(256) void cl_dw_fwd_kernel(ClDwArgs a) {
    const unsigned total = (unsigned)a.B * a.Ho * a.Wo * a.C;  // < 2^31 (launcher)
    for (unsigned i = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = (int)(i % (unsigned)a.C);
        unsigned r = i / (unsigned)a.C;
        const int wo = (int)(r % (unsigned)a.Wo);
        r /= (unsigned)a.Wo;
        const int ho = (int)(r % (unsigned)a.Ho), b = (int)(r / (unsigned)a.Ho);
        float acc = a.bias ? a.bias[c] : 0.f;
        const float* xb = a.x + ((size_t)b * a.H * a.W) * a.Cp + c;
        // taps unrolled to 4 x 5 with uniform predicates; loads unconditional on clamped addresses, masked afterwards
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {
                const int h = ho * a.s - a.pt + ki;
                const bool hok = h >= 0 && h < a.H;
                const float* rowp = xb + (size_t)min(max(h, 0), a.H - 1) * a.W * a.Cp;
                float v[5];
#pragma unroll
                for (int kj = 0; kj < 5; ++kj) v[kj] = kj < a.kw ? rowp[(size_t)min(max(wo * a.s - a.pl + kj, 0), a.W - 1) * a.Cp] : 0.f;
#pragma unroll
                for (int kj = 0; kj < 5; ++kj) {
                    const int w = wo * a.s - a.pl + kj;
                    if (kj < a.kw) acc = fmaf(a.w[c * a.kh * a.kw + ki * a.kw + kj], (hok && w >= 0 && w < a.W) ? v[kj] : 0.f, acc);
                }
            }
        }
        a.y[(size_t)(i / (unsigned)a.C) * a.Cp + c] = acc;
    }
}

// stride-1 variant: a thread produces four outputs along W for one channel, so the (kw + 3) inputs of a kernel row are loaded once
// for the four of them (28 loads instead of 64 for 4x4 taps) and the index arithmetic is paid once.  FLIP evaluates the input
// gradient: the same correlation with the taps reversed and the padding mirrored (pt' = kh-1-pt, pl' = kw-1-pl), reading dy.
template <bool FLIP>
__global__ __launch_bounds__(256) void cl_dw_s1_w4_kernel(ClDwArgs a) {
    const float* __restrict__ src = FLIP ? a.dy : a.x;
    float* __restrict__ dst = FLIP ? a.dx : a.y;
    const int W4 = (a.W + 3) >> 2;
    const unsigned total = (unsigned)a.B * a.H * W4 * a.C;
    const int pt = FLIP ? a.kh - 1 - a.pt : a.pt, pl = FLIP ? a.kw - 1 - a.pl : a.pl;
    for (unsigned i = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = (int)(i % (unsigned)a.C);
        unsigned r = i / (unsigned)a.C;
        const int w0 = (int)(r % (unsigned)W4) * 4;
        r /= (unsigned)W4;
        const int h = (int)(r % (unsigned)a.H), b = (int)(r / (unsigned)a.H);
        float wt[4][5];
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
            for (int kj = 0; kj < 5; ++kj) {
                const int si = FLIP ? a.kh - 1 - ki : ki, sj = FLIP ? a.kw - 1 - kj : kj;
                wt[ki][kj] = (ki < a.kh && kj < a.kw) ? a.w[c * a.kh * a.kw + si * a.kw + sj] : 0.f;
            }
        const float b0 = (!FLIP && a.bias) ? a.bias[c] : 0.f;
        float acc[4] = {b0, b0, b0, b0};
        const float* sb = src + ((size_t)b * a.H * a.W) * a.Cp + c;
        // every load is unconditional on a clamped address and masked afterwards: a conditional load compiles to a branch with a wait
        // behind it, which serialises the 28 loads of an iteration (145 -> 70 us per full-resolution convolution at batch 16)
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {  // uniform
                const int hh = h - pt + ki;
                const bool hok = hh >= 0 && hh < a.H;
                const float* rowp = sb + (size_t)min(max(hh, 0), a.H - 1) * a.W * a.Cp;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ww = w0 - pl + j;
                    v[j] = rowp[(size_t)min(max(ww, 0), a.W - 1) * a.Cp];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ww = w0 - pl + j;
                    const float x = (hok && ww >= 0 && ww < a.W) ? v[j] : 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kj = j - o;
                        if (kj >= 0 && kj < 5) acc[o] = fmaf(wt[ki][kj], x, acc[o]);  // taps beyond kw carry zero weights
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (w0 + o < a.W) dst[(((size_t)b * a.H + h) * a.W + w0 + o) * a.Cp + c] = acc[o];
    }
}

// The same with four adjacent channels per thread (4 | C, taps up to 4 x 4): 16-byte loads / stores, a quarter of the memory
// instructions per output - the dword version is bound by the texture addresser (one wave instruction per 16 cycles whatever the
// width), not by HBM.  Weights of the thread's channel quad stay in registers across the grid-stride loop (the stride is a multiple of C / 4).
template <bool FLIP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void cl_dw_s1_w4c4_kernel(ClDwArgs a) {
    // the taps live in LDS as [ki][kj][channel] (one 16-byte read per tap and thread): in registers they cost 64 VGPRs and left two waves
    // per SIMD for a kernel that is all load latency
    __shared__ float wl[16 * 256];
    const float* __restrict__ src = FLIP ? a.dy : a.x;
    float* __restrict__ dst = FLIP ? a.dx : a.y;
    const int W4 = (a.W + 3) >> 2, C4 = a.C >> 2;
    const unsigned total = (unsigned)a.B * a.H * W4 * C4;
    const int pt = FLIP ? a.kh - 1 - a.pt : a.pt, pl = FLIP ? a.kw - 1 - a.pl : a.pl;
    const unsigned i0 = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int c = (int)(i0 % (unsigned)C4) * 4;
    for (int e = threadIdx.x; e < 16 * a.C; e += 256) {
        const int t = e / a.C, ch = e - t * a.C, ki = t >> 2, kj = t & 3;
        const int si = FLIP ? a.kh - 1 - ki : ki, sj = FLIP ? a.kw - 1 - kj : kj;
        wl[e] = (ki < a.kh && kj < a.kw) ? a.w[ch * a.kh * a.kw + si * a.kw + sj] : 0.f;
    }
    __syncthreads();
    const float4* __restrict__ wq = reinterpret_cast<const float4*>(wl) + (c >> 2);  // tap t: wq[t * C4]
    f32x4u_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (!FLIP && a.bias) b4 = *reinterpret_cast<const f32x4u_t*>(a.bias + c);
    for (unsigned i = i0; i < total; i += gridDim.x * 256) {
        unsigned r = i / (unsigned)C4;
        const int w0 = (int)(r % (unsigned)W4) * 4;
        r /= (unsigned)W4;
        const int h = (int)(r % (unsigned)a.H), b = (int)(r / (unsigned)a.H);
        f32x4u_t acc[4] = {b4, b4, b4, b4};
        const float* sb = src + ((size_t)b * a.H * a.W) * a.Cp + c;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {  // uniform
                const int hh = h - pt + ki;
                const bool hok = hh >= 0 && hh < a.H;
                const float* rowp = sb + (size_t)min(max(hh, 0), a.H - 1) * a.W * a.Cp;
                f32x4u_t v[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = *reinterpret_cast<const f32x4u_t*>(rowp + (size_t)min(max(w0 - pl + j, 0), a.W - 1) * a.Cp);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int ww = w0 - pl + j;
                    const float m = (hok && ww >= 0 && ww < a.W) ? 1.f : 0.f;
                    const f32x4u_t x = v[j] * m;
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kj = j - o;
                        if (kj >= 0 && kj < 4) {  // taps beyond kh x kw carry zero weights
                            const float4 w4 = wq[(ki * 4 + kj) * C4];
                            acc[o] += f32x4u_t{w4.x, w4.y, w4.z, w4.w} * x;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (w0 + o < a.W) *reinterpret_cast<f32x4u_t*>(dst + (((size_t)b * a.H + h) * a.W + w0 + o) * a.Cp + c) = acc[o];
    }
}

// input gradient: dx[b,h,w,c] = sum over taps with (h + pt - ki) = ho*s, (w + pl - kj) = wo*s of w[c,ki,kj] * dy[b,ho,wo,c]
// (conditional loads on purpose: this kernel serves the stride-2 case, where three taps in four fail the parity test - loading them
// unconditionally costs 5x the traffic: 141 vs 50 us)
__global__ __launch_bounds__(256) void cl_dw_bwd_data_kernel(ClDwArgs a) {
    const unsigned total = (unsigned)a.B * a.H * a.W * a.C;  // < 2^31 (launcher)
    for (unsigned i = xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = (int)(i % (unsigned)a.C);
        unsigned r = i / (unsigned)a.C;
        const int w = (int)(r % (unsigned)a.W);
        r /= (unsigned)a.W;
        const int h = (int)(r % (unsigned)a.H), b = (int)(r / (unsigned)a.H);
        float acc = 0.f;
        for (int ki = 0; ki < a.kh; ++ki) {
            const int hn = h + a.pt - ki;
            if (hn < 0 || hn % a.s) continue;
            const int ho = hn / a.s;
            if (ho >= a.Ho) continue;
            for (int kj = 0; kj < a.kw; ++kj) {
                const int wn = w + a.pl - kj;
                if (wn < 0 || wn % a.s) continue;
                const int wo = wn / a.s;
                if (wo >= a.Wo) continue;
                acc = fmaf(a.w[c * a.kh * a.kw + ki * a.kw + kj], a.dy[(((size_t)b * a.Ho + ho) * a.Wo + wo) * a.Cp + c], acc);
            }
        }
        a.dx[(size_t)(i / (unsigned)a.C) * a.Cp + c] = acc;
    }
}

// weight gradient: dw[c,ki,kj] += sum_{b,ho,wo} dy * x(shifted).  Thread = (channel, one of 256/C row lanes); taps up to 4 x 5,
// fully unrolled with predicates so the accumulators stay in registers and no tap index is ever divided.
__global__ __launch_bounds__(256) void cl_dw_wgrad_kernel(ClDwArgs a) {
    __shared__ float part[256];
    const int tid = threadIdx.x, c = tid % a.C, lanes = 256 / a.C, rl = tid / a.C;
    float acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = 0.f;
    const unsigned rows = (unsigned)a.B * a.Ho * a.Wo, HoWo = (unsigned)a.Ho * a.Wo;
    for (unsigned r = xcd_block(blockIdx.x, gridDim.x) * lanes + rl; r < rows; r += gridDim.x * lanes) {
        const unsigned b = r / HoWo, q = r - b * HoWo;
        const int ho = (int)(q / (unsigned)a.Wo), wo = (int)(q - (unsigned)ho * a.Wo);
        const float d = a.dy[(size_t)r * a.Cp + c];
        const int hb = ho * a.s - a.pt, wb = wo * a.s - a.pl;
        const float* xb = a.x + ((size_t)b * a.H * a.W) * a.Cp + c;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {  // uniform; loads unconditional on clamped addresses, masked afterwards (see cl_dw_s1_w4_kernel)
                const int h = hb + ki;
                const bool hok = h >= 0 && h < a.H;
                const float* rowp = xb + (size_t)min(max(h, 0), a.H - 1) * a.W * a.Cp;
                float v[5];
#pragma unroll
                for (int kj = 0; kj < 5; ++kj) v[kj] = kj < a.kw ? rowp[(size_t)min(max(wb + kj, 0), a.W - 1) * a.Cp] : 0.f;
#pragma unroll
                for (int kj = 0; kj < 5; ++kj) {
                    const int w = wb + kj;
                    if (kj < a.kw) acc[ki][kj] = fmaf(d, (hok && w >= 0 && w < a.W) ? v[kj] : 0.f, acc[ki][kj]);
                }
            }
        }
    }
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int kj = 0; kj < 5; ++kj) {
            if (ki < a.kh && kj < a.kw) {  // uniform
                part[tid] = acc[ki][kj];
                __syncthreads();
                if (tid < a.C) {
                    float v = 0.f;
                    for (int j = tid; j < 256; j += a.C) v += part[j];
                    // per-workgroup partial; cl_dw_wgrad_reduce_kernel sums them (atomics onto kh*kw*C addresses from thousands
                    // of workgroups serialise: measured 515 us vs 110 us of work)
                    a.scratch[((size_t)blockIdx.x * a.kh * a.kw + ki * a.kw + kj) * a.C + tid] = v;
                }
                __syncthreads();
            }
        }
}

// The same with four adjacent channels per thread (4 | C, taps up to 4 x 4; see cl_dw_s1_w4c4_kernel): thread = (channel quad, one of
// 1024 / C row lanes).
__global__ __launch_bounds__(256) void cl_dw_wgrad_c4_kernel(ClDwArgs a) {
    __shared__ float4 part[256];
    const int tid = threadIdx.x, C4 = a.C >> 2, c = (tid % C4) * 4, lanes = 256 / C4, rl = tid / C4;
    f32x4u_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4u_t{0.f, 0.f, 0.f, 0.f};
    const unsigned rows = (unsigned)a.B * a.Ho * a.Wo, HoWo = (unsigned)a.Ho * a.Wo;
    for (unsigned r = xcd_block(blockIdx.x, gridDim.x) * lanes + rl; r < rows; r += gridDim.x * lanes) {
        const unsigned b = r / HoWo, q = r - b * HoWo;
        const int ho = (int)(q / (unsigned)a.Wo), wo = (int)(q - (unsigned)ho * a.Wo);
        const f32x4u_t d = *reinterpret_cast<const f32x4u_t*>(a.dy + (size_t)r * a.Cp + c);
        const int hb = ho * a.s - a.pt, wb = wo * a.s - a.pl;
        const float* xb = a.x + ((size_t)b * a.H * a.W) * a.Cp + c;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {  // uniform
                const int h = hb + ki;
                const bool hok = h >= 0 && h < a.H;
                const float* rowp = xb + (size_t)min(max(h, 0), a.H - 1) * a.W * a.Cp;
                f32x4u_t v[4];
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) v[kj] = *reinterpret_cast<const f32x4u_t*>(rowp + (size_t)min(max(wb + kj, 0), a.W - 1) * a.Cp);
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) {
                    const int w = wb + kj;
                    const float m = (kj < a.kw && hok && w >= 0 && w < a.W) ? 1.f : 0.f;
                    acc[ki][kj] += d * (v[kj] * m);
                }
            }
        }
    }
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (ki < a.kh && kj < a.kw) {  // uniform
                part[tid] = make_float4(acc[ki][kj][0], acc[ki][kj][1], acc[ki][kj][2], acc[ki][kj][3]);
                __syncthreads();
                if (tid < C4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int j = tid; j < 256; j += C4) {
                        const float4 q = part[j];
                        v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
                    }
                    *reinterpret_cast<float4*>(a.scratch + ((size_t)blockIdx.x * a.kh * a.kw + ki * a.kw + kj) * a.C + 4 * tid) = v;
                }
                __syncthreads();
            }
        }
}

// Stride 1, four adjacent channels AND four adjacent output columns per thread: the 7 input columns of a kernel row serve the four
// outputs (7 + 1 16-byte loads per output row of the window instead of 16 + 4: the per-pixel version is bound by its load instructions).
__global__ __launch_bounds__(256) void cl_dw_wgrad_w4c4_kernel(ClDwArgs a) {
    __shared__ float4 part[256];
    const int tid = threadIdx.x, C4 = a.C >> 2, c = (tid % C4) * 4, lanes = 256 / C4, rl = tid / C4;
    f32x4u_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4u_t{0.f, 0.f, 0.f, 0.f};
    const unsigned W4 = (unsigned)(a.Wo + 3) >> 2, HoW4 = (unsigned)a.Ho * W4, rows4 = (unsigned)a.B * HoW4;
    for (unsigned r = xcd_block(blockIdx.x, gridDim.x) * lanes + rl; r < rows4; r += gridDim.x * lanes) {
        const unsigned b = r / HoW4, q = r - b * HoW4;
        const int ho = (int)(q / W4), wo0 = (int)(q - (unsigned)ho * W4) * 4;
        const float* dyb = a.dy + (((size_t)b * a.Ho + ho) * a.Wo) * a.Cp + c;
        f32x4u_t d[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4u_t t = *reinterpret_cast<const f32x4u_t*>(dyb + (size_t)min(wo0 + o, a.Wo - 1) * a.Cp);
            d[o] = t * (wo0 + o < a.Wo ? 1.f : 0.f);
        }
        const float* xb = a.x + ((size_t)b * a.H * a.W) * a.Cp + c;
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            if (ki < a.kh) {  // uniform
                const int h = ho - a.pt + ki;
                const bool hok = h >= 0 && h < a.H;
                const float* rowp = xb + (size_t)min(max(h, 0), a.H - 1) * a.W * a.Cp;
                f32x4u_t v[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = *reinterpret_cast<const f32x4u_t*>(rowp + (size_t)min(max(wo0 - a.pl + j, 0), a.W - 1) * a.Cp);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int w = wo0 - a.pl + j;
                    v[j] = v[j] * ((hok && w >= 0 && w < a.W) ? 1.f : 0.f);
                }
#pragma unroll
                for (int kj = 0; kj < 4; ++kj)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[ki][kj] += d[o] * v[o + kj];  // taps beyond kw are never stored
            }
        }
    }
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (ki < a.kh && kj < a.kw) {  // uniform
                part[tid] = make_float4(acc[ki][kj][0], acc[ki][kj][1], acc[ki][kj][2], acc[ki][kj][3]);
                __syncthreads();
                if (tid < C4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int j = tid; j < 256; j += C4) {
                        const float4 q = part[j];
                        v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
                    }
                    *reinterpret_cast<float4*>(a.scratch + ((size_t)blockIdx.x * a.kh * a.kw + ki * a.kw + kj) * a.C + 4 * tid) = v;
                }
                __syncthreads();
            }
        }
}

// Second stage of the depthwise weight gradient: nwg partial rows of taps*C floats -> dw (C, taps), accumulated.  A workgroup owns 64
// float4 columns x 4 row lanes (independent 16-byte loads, LDS 4 -> 1), gridDim.y row slices meet in one atomic per element.
__global__ __launch_bounds__(256) void cl_dw_wgrad_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int nwg, int taps,
                                                                 int C) {
    __shared__ float4 red[4][64];
    const int n = taps * C, n4 = n >> 2, cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n4) {
#pragma unroll 8
        for (int w = blockIdx.y * 4 + lane; w < nwg; w += gridDim.y * 4) {
            const float4 q = reinterpret_cast<const float4*>(scratch + (size_t)w * n)[col];
            v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
        }
    }
    red[lane][cl] = v;
    __syncthreads();
    if (lane == 0 && col < n4) {
        const float4 a = red[1][cl], b = red[2][cl], c = red[3][cl];
        const float s[4] = {v.x + a.x + b.x + c.x, v.y + a.y + b.y + c.y, v.z + a.z + b.z + c.z, v.w + a.w + b.w + c.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = col * 4 + k;  // (tap, c)
            unsafeAtomicAdd(dw + (i % C) * taps + i / C, s[k]);
        }
    }
}
// taps * C not a multiple of 4 (1- or 2-channel tensors with odd tap counts): one float per thread
__global__ __launch_bounds__(256) void cl_dw_wgrad_reduce1_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int nwg, int taps,
                                                                  int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // (tap, c)
    if (i >= taps * C) return;
    float v = 0.f;
    for (int w = blockIdx.y; w < nwg; w += gridDim.y) v += scratch[(size_t)w * taps * C + i];
    unsafeAtomicAdd(dw + (i % C) * taps + i / C, v);
}


namespace {
inline bool cl_c_ok(int C) { return C >= 1 && C <= 1024 && !(C & (C - 1)); }
inline unsigned grid4(size_t n, unsigned cap) { return (grid_for(n, cap) + 3) / 4 * 4; }  // stride (grid * 256) % C == 0 for C <= 1024
}  // namespace
size_t cl_stage_partial_floats(int B, int C) { return (size_t)B * CL_STAGE_MAX_WG * CL_STAGE_PITCH(C); }
int launch_cl_norm_act_fwd(const ClStageArgs& a, int B, hipStream_t st) {
    if (!cl_c_ok(a.C)) return RTFS_ERR_SHAPE;
    if (a.C % 4 == 0) hipLaunchKernelGGL(cl_norm_act_fwd_kernel<4>, dim3(grid4(a.n / 4, 2048), B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(cl_norm_act_fwd_kernel<1>, dim3(grid4(a.n, 2048), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
// part 0: reduction + apply; 1: reduction only; 2: apply only (SyncBatchNorm all-reduces dgamma / dbeta in between)
int launch_cl_norm_act_bwd(const ClStageArgs& a, int B, hipStream_t st, int part) {
    if (!cl_c_ok(a.C)) return RTFS_ERR_SHAPE;
    if (part != 2 && (a.norm || a.act == 2)) {
        if (!a.partial) return RTFS_ERR_WORKSPACE;
        const unsigned gx = a.C % 4 == 0 ? grid4(a.n / 4 / 4, CL_STAGE_MAX_WG) : grid4(a.n, CL_STAGE_MAX_WG);
        if (a.C % 4 == 0) hipLaunchKernelGGL(cl_norm_act_bwd_reduce_kernel<4>, dim3(gx, B), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(cl_norm_act_bwd_reduce_kernel<1>, dim3(gx, B), dim3(256), 0, st, a);
        const unsigned rows = gx * (unsigned)B;
        hipLaunchKernelGGL(cl_stage_reduce2_kernel, dim3(cdiv(2 * a.C, 64) + 1, rows >= 128 ? 32 : cdiv((int)rows, 4)), dim3(256), 0, st,
                           a.partial, (int)gx, B, a.C, a.dgamma, a.dbeta, a.dslope, a.S, a.norm, a.act);
    }
    if (part != 1) {
        if (a.C % 4 == 0) hipLaunchKernelGGL(cl_norm_act_bwd_apply_kernel<4>, dim3(grid4(a.n / 4, 2048), B), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(cl_norm_act_bwd_apply_kernel<1>, dim3(grid4(a.n, 2048), B), dim3(256), 0, st, a);
    }
    return rtfs_launch_status();
}
// bwd: a.dx, a.dy, a.partial (cl_stage_partial_floats(2, C) floats), dw / db / dslope accumulate into zeroed buffers
int launch_gateway(const GatewayArgs& a, bool bwd, float* dw, float* db, float* dslope, hipStream_t st) {
    if (a.C < 4 || a.C > 1024 || (a.C & (a.C - 1))) return RTFS_ERR_SHAPE;
    if (!bwd) {
        hipLaunchKernelGGL(gateway_kernel<false>, dim3(grid4(a.n4, 2048)), dim3(256), 0, st, a);
        return rtfs_launch_status();
    }
    if (!a.partial) return RTFS_ERR_WORKSPACE;
    const unsigned gx = grid4(a.n4 / 4, 2 * CL_STAGE_MAX_WG);  // four streams of the block's largest tensor: two workgroups per CU
    hipLaunchKernelGGL(gateway_kernel<true>, dim3(gx), dim3(256), 0, st, a);
    hipLaunchKernelGGL(cl_stage_reduce2_kernel, dim3(cdiv(2 * a.C, 64) + 1, gx >= 128 ? 32 : cdiv((int)gx, 4)), dim3(256), 0, st, a.partial, (int)gx, 1,
                       a.C, dw, db, dslope, (double*)nullptr, 2, 2);
    return rtfs_launch_status();
}
int launch_cl_chan_stats(const float* x, double* stats, size_t n, int C, hipStream_t st) {
    if (!cl_c_ok(C)) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(cl_chan_stats_kernel, dim3(grid4(n, 1024)), dim3(256), 0, st, x, stats, n, C);
    return rtfs_launch_status();
}
int launch_bn_update(const double* stats, float* rmean, float* rvar, int C, double rows, float momentum, hipStream_t st) {
    hipLaunchKernelGGL(bn_update_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, stats, rmean, rvar, C, rows, momentum);
    return rtfs_launch_status();
}
// any C: thread = column, workgroup = a chunk of rows
__global__ __launch_bounds__(256) void cl_colsum_any_kernel(const float* __restrict__ d, float* __restrict__ out, size_t rows, int C, int chunk) {
    const size_t r0 = (size_t)blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (size_t r = r0; r < r1; ++r) s += d[r * C + c];
        unsafeAtomicAdd(out + c, s);
    }
}
// `partial` (optional, >= 256 * C floats): per-workgroup rows + a second-stage fold instead of 256-way contended atomics (35 -> ~15 us
// for a 33 MB tensor)
int launch_cl_colsum(const float* d, float* out, size_t n, int C, hipStream_t st, float* partial) {
    if (C >= 1 && (!cl_c_ok(C) || C < 4)) {
        const size_t rows = n / C;
        const int chunk = 64;
        hipLaunchKernelGGL(cl_colsum_any_kernel, dim3((unsigned)((rows + chunk - 1) / chunk)), dim3(256), 0, st, d, out, rows, C, chunk);
        return rtfs_launch_status();
    }
    if (!cl_c_ok(C)) return RTFS_ERR_SHAPE;
    const unsigned g = grid_for(n / 4 / 8, 256);
    if (g < 16) partial = nullptr;  // few workgroups: the atomics are cheaper than a second launch
    hipLaunchKernelGGL(cl_colsum_kernel, dim3(g), dim3(256), 0, st, d, out, n, C, partial);
    if (partial)
        hipLaunchKernelGGL(cl_dw_wgrad_reduce_kernel, dim3(cdiv(C / 4, 64), g >= 128 ? 32 : cdiv((int)g, 4)), dim3(256), 0, st, partial, out, (int)g,
                           1, C);
    return rtfs_launch_status();
}
namespace {
int launch_cl_dw_chunk(const ClDwArgs& a, int what, hipStream_t st) {
    const bool w4 = a.s == 1 && a.W >= 4 && a.Ho == a.H && a.Wo == a.W;
    const size_t n4 = (size_t)a.B * a.H * ((a.W + 3) / 4) * a.C;
    const bool c4 = a.C % 4 == 0 && a.Cp % 4 == 0 && a.kh <= 4 && a.kw <= 4;
    if (what == 0 && w4 && c4) hipLaunchKernelGGL(cl_dw_s1_w4c4_kernel<false>, dim3(grid8(grid_for(n4 / 4))), dim3(256), 0, st, a);
    else if (what == 1 && w4 && c4) hipLaunchKernelGGL(cl_dw_s1_w4c4_kernel<true>, dim3(grid8(grid_for(n4 / 4))), dim3(256), 0, st, a);
    else if (what == 0 && w4) hipLaunchKernelGGL(cl_dw_s1_w4_kernel<false>, dim3(grid8(grid_for(n4))), dim3(256), 0, st, a);
    else if (what == 1 && w4) hipLaunchKernelGGL(cl_dw_s1_w4_kernel<true>, dim3(grid8(grid_for(n4))), dim3(256), 0, st, a);
    else if (what == 0) hipLaunchKernelGGL(cl_dw_fwd_kernel, dim3(grid8(grid_for((size_t)a.B * a.Ho * a.Wo * a.C))), dim3(256), 0, st, a);
    else if (what == 1) hipLaunchKernelGGL(cl_dw_bwd_data_kernel, dim3(grid8(grid_for((size_t)a.B * a.H * a.W * a.C))), dim3(256), 0, st, a);
    else {
        const bool w4c4 = c4 && a.s == 1 && a.Wo >= 4;
        const size_t rows = w4c4 ? (size_t)a.B * a.Ho * ((a.Wo + 3) / 4) : (size_t)a.B * a.Ho * a.Wo,
                     per_wg = (size_t)(256 / (c4 ? a.C / 4 : a.C)) * (w4c4 ? 2 : 8);
        size_t g = (rows + per_wg - 1) / per_wg;
        if (w4c4) {  // 148 VGPRs: three workgroups per CU = 768 resident; one balanced round instead of 1.35 or 2.7 ragged ones
            const size_t lanes = 256 / (a.C / 4), n = (rows + lanes - 1) / lanes, iters = (n + 767) / 768;
            g = (n + iters - 1) / iters;
        }
        g = g > CL_DW_WGRAD_MAX_WG ? CL_DW_WGRAD_MAX_WG : grid8((unsigned)(g < 1 ? 1 : g));  // a multiple of 8 (xcd_block); idle workgroups store zeros
        if (!a.scratch) return RTFS_ERR_WORKSPACE;
        if (w4c4) hipLaunchKernelGGL(cl_dw_wgrad_w4c4_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
        else if (c4) hipLaunchKernelGGL(cl_dw_wgrad_c4_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(cl_dw_wgrad_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
        const int n = a.kh * a.kw * a.C;
        if (n % 4 == 0)
            hipLaunchKernelGGL(cl_dw_wgrad_reduce_kernel, dim3(cdiv(n / 4, 64), g >= 128 ? 32 : (unsigned)cdiv((int)g, 4)), dim3(256), 0, st,
                               a.scratch, a.dw, (int)g, a.kh * a.kw, a.C);
        else
            hipLaunchKernelGGL(cl_dw_wgrad_reduce1_kernel, dim3(cdiv(n, 256), g >= 64 ? 64 : (unsigned)g), dim3(256), 0, st, a.scratch, a.dw,
                               (int)g, a.kh * a.kw, a.C);
    }
    return rtfs_launch_status();
}
}  // namespace
// C up to 256 in one launch; wider tensors (the VP block's 512-channel gateway) go through in slices of 256 channels of the same rows
int launch_cl_dw(const ClDwArgs& a0, int what, hipStream_t st) {
    ClDwArgs a = a0;
    a.Cp = a.C;
    if (a.kh > 4 || a.kw > 5 || a.C < 1 || (size_t)a.B * a.H * a.W * a.C >= 0x7fffffffu) return RTFS_ERR_SHAPE;
    if (a.C <= 256) {
        if (256 % a.C) return RTFS_ERR_SHAPE;
        return launch_cl_dw_chunk(a, what, st);
    }
    if (a.C % 256) return RTFS_ERR_SHAPE;
    const int taps = a.kh * a.kw;
    for (int c0 = 0; c0 < a0.C; c0 += 256) {
        ClDwArgs b = a;
        b.C = 256;
        if (b.x) b.x += c0;
        if (b.y) b.y += c0;
        if (b.dy) b.dy += c0;
        if (b.dx) b.dx += c0;
        if (b.w) b.w += (size_t)c0 * taps;
        if (b.bias) b.bias += c0;
        if (b.dw) b.dw += (size_t)c0 * taps;
        int rc = launch_cl_dw_chunk(b, what, st);
        if (rc) return rc;
    }
    return RTFS_OK;
}

// ------------------------------------------------------------------------------------------------ pooling / TFAR glue (channel-first planes)
// adaptive average pooling (tdanet.py:116, F.adaptive_avg_pool2d): window i = [floor(i*in/out), ceil((i+1)*in/out))
namespace {
__device__ __forceinline__ int win_lo(int i, int n_in, int n_out) { return (int)(((long)i * n_in) / n_out); }
__device__ __forceinline__ int win_hi(int i, int n_in, int n_out) { return (int)((((long)(i + 1)) * n_in + n_out - 1) / n_out); }
}  // namespace
// All four glue kernels take an inner channel count C: C = 1 is the channel-first case (N = B*C planes of (H, W)), C > 1 the rows case
// (N = B maps of (H, W, C) with channels fastest): element (n, h, w, c) lives at ((n*H + h)*W + w)*C + c.
__global__ __launch_bounds__(256) void pool2d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t N, int H, int W, int Ho,
                                                         int Wo, int C) {
    const size_t total = N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho);
        const size_t n = p / ((size_t)Wo * Ho);
        const int h0 = win_lo(ho, H, Ho), h1 = win_hi(ho, H, Ho), w0 = win_lo(wo, W, Wo), w1 = win_hi(wo, W, Wo);
        float s = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) s += x[((n * H + h) * W + w) * C + c];
        y[i] = s / (float)((h1 - h0) * (w1 - w0));
    }
}
// V adjacent channels per thread (V = 4 on rows: the window arithmetic - a dozen integer divisions - is paid once per 16-byte store)
template <int V>
__global__ __launch_bounds__(256) void pool2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t N, int H, int W, int Ho,
                                                         int Wo, int C) {
    const int CV = C / V;
    const size_t total = N * H * W * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        const size_t p = i / CV;
        const int w = (int)(p % W), h = (int)((p / W) % H);
        const size_t n = p / ((size_t)W * H);
        // candidate windows around floor(h * out / in): window starts are non-decreasing and each is at most in/out + 1 long
        const int hc = (int)(((long)h * Ho) / H), wc = (int)(((long)w * Wo) / W);
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] = 0.f;
        for (int ho = max(hc - 1, 0); ho <= min(hc + 1, Ho - 1); ++ho) {
            const int h0 = win_lo(ho, H, Ho), h1 = win_hi(ho, H, Ho);
            if (h < h0 || h >= h1) continue;
            for (int wo = max(wc - 1, 0); wo <= min(wc + 1, Wo - 1); ++wo) {
                const int w0 = win_lo(wo, W, Wo), w1 = win_hi(wo, W, Wo);
                if (w < w0 || w >= w1) continue;
                const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
                float d[V];
                ldv<V>(dy + ((n * Ho + ho) * Wo + wo) * C + c, 0, d);
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] += d[k] * inv;
            }
        }
        stv<V>(dx + p * C + c, 0, s);
    }
}
// InjectionMultiSum's last line (fusion.py:54-69): out = local * up(gate) + up(global), up = F.interpolate(mode="nearest")
template <int V>
__global__ __launch_bounds__(256) void tfar_combine_fwd_kernel(const float* __restrict__ le, const float* __restrict__ gate,
                                                               const float* __restrict__ ge, float* __restrict__ out, size_t N, int H, int W,
                                                               int Hg, int Wg, int C) {
    const int CV = C / V;
    const size_t total = N * H * W * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        const size_t p = i / CV;
        const int w = (int)(p % W), h = (int)((p / W) % H);
        const size_t n = p / ((size_t)W * H);
        const size_t j = ((n * Hg + nearest_src(h, Hg, H)) * Wg + nearest_src(w, Wg, W)) * C + c;
        float l[V], g[V], e[V];
        ldv<V>(le + p * C + c, 0, l);
        ldv<V>(gate + j, 0, g);
        ldv<V>(ge + j, 0, e);
#pragma unroll
        for (int k = 0; k < V; ++k) l[k] = fmaf(l[k], g[k], e[k]);
        stv<V>(out + p * C + c, 0, l);
    }
}
template <int V>
__global__ __launch_bounds__(256) void tfar_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ le,
                                                               const float* __restrict__ gate, float* __restrict__ dle,
                                                               float* __restrict__ dgate, float* __restrict__ dge, size_t N, int H, int W, int Hg,
                                                               int Wg, int C) {
    // one thread per GLOBAL element (V adjacent channels): it owns the local pixels that read it (a contiguous block of rows x columns)
    const int CV = C / V;
    const size_t total = N * Hg * Wg * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        const size_t p = i / CV;
        const int wg = (int)(p % Wg), hg = (int)((p / Wg) % Hg);
        const size_t n = p / ((size_t)Wg * Hg);
        // local h reads global floor(h * Hg / H) == hg  <=>  h in [ceil(hg*H/Hg), ceil((hg+1)*H/Hg))
        const int h0 = (int)(((long)hg * H + Hg - 1) / Hg), h1 = min(H, (int)(((long)(hg + 1) * H + Hg - 1) / Hg));
        const int w0 = (int)(((long)wg * W + Wg - 1) / Wg), w1 = min(W, (int)(((long)(wg + 1) * W + Wg - 1) / Wg));
        float g[V], sg[V], se[V];
        ldv<V>(gate + p * C + c, 0, g);
#pragma unroll
        for (int k = 0; k < V; ++k) sg[k] = se[k] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const size_t q = ((n * H + h) * W + w) * C + c;
                float d[V], l[V], o[V];
                ldv<V>(dout + q, 0, d);
                ldv<V>(le + q, 0, l);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    o[k] = d[k] * g[k];
                    sg[k] = fmaf(d[k], l[k], sg[k]);
                    se[k] += d[k];
                }
                stv<V>(dle + q, 0, o);
            }
        stv<V>(dgate + p * C + c, 0, sg);
        stv<V>(dge + p * C + c, 0, se);
    }
}
int launch_pool2d(const float* x, float* y, size_t N, int H, int W, int Ho, int Wo, bool bwd, hipStream_t st, int C) {
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1 || Ho > H || Wo > W || C < 1) return RTFS_ERR_SHAPE;
    if (bwd && C % 4 == 0) hipLaunchKernelGGL(pool2d_bwd_kernel<4>, dim3(grid_for(N * H * W * C / 4)), dim3(256), 0, st, x, y, N, H, W, Ho, Wo, C);
    else if (bwd) hipLaunchKernelGGL(pool2d_bwd_kernel<1>, dim3(grid_for(N * H * W * C)), dim3(256), 0, st, x, y, N, H, W, Ho, Wo, C);
    else hipLaunchKernelGGL(pool2d_fwd_kernel, dim3(grid_for(N * Ho * Wo * C)), dim3(256), 0, st, x, y, N, H, W, Ho, Wo, C);
    return rtfs_launch_status();
}
int launch_tfar_combine(const float* le, const float* gate, const float* ge, float* out, size_t N, int H, int W, int Hg, int Wg, hipStream_t st,
                        int C) {
    if (Hg < 1 || Wg < 1 || Hg > H || Wg > W || C < 1) return RTFS_ERR_SHAPE;
    if (C % 4 == 0) hipLaunchKernelGGL(tfar_combine_fwd_kernel<4>, dim3(grid_for(N * H * W * C / 4)), dim3(256), 0, st, le, gate, ge, out, N, H, W, Hg, Wg, C);
    else hipLaunchKernelGGL(tfar_combine_fwd_kernel<1>, dim3(grid_for(N * H * W * C)), dim3(256), 0, st, le, gate, ge, out, N, H, W, Hg, Wg, C);
    return rtfs_launch_status();
}
int launch_tfar_combine_bwd(const float* dout, const float* le, const float* gate, float* dle, float* dgate, float* dge, size_t N, int H, int W,
                            int Hg, int Wg, hipStream_t st, int C) {
    if (Hg < 1 || Wg < 1 || Hg > H || Wg > W || C < 1) return RTFS_ERR_SHAPE;
    if (C % 4 == 0)
        hipLaunchKernelGGL(tfar_combine_bwd_kernel<4>, dim3(grid_for(N * Hg * Wg * C / 4)), dim3(256), 0, st, dout, le, gate, dle, dgate, dge, N, H, W, Hg, Wg,
                           C);
    else
        hipLaunchKernelGGL(tfar_combine_bwd_kernel<1>, dim3(grid_for(N * Hg * Wg * C)), dim3(256), 0, st, dout, le, gate, dle, dgate, dge, N, H, W, Hg, Wg, C);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ encoder / decoder / S^3 adjoints
// 3x3 "same" patches of a 2-channel map as rows: rows[(b, t, f)][(c, ki, kj)] = z[b, c, t - 1 + ki, f - 1 + kj] (18 of 64 columns used).
// Both the encoder's Conv2d(2 -> 256) (encoder.py:146-157) and the adjoint of the decoder's ConvTranspose2d(256 -> 2)
// (decoder.py:96-106) have the weight gradient  dW (256, 18) = big_rows^T . patch_rows.
__global__ __launch_bounds__(256) void patch3x3_rows_kernel(const float* __restrict__ z, float* __restrict__ rows, int B, int T, int F) {
    const unsigned total = (unsigned)B * T * F * 64;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int col = (int)(i & 63);
        unsigned r = i >> 6;
        const int f = (int)(r % (unsigned)F);
        r /= (unsigned)F;
        const int t = (int)(r % (unsigned)T), b = (int)(r / (unsigned)T);
        float v = 0.f;
        if (col < 18) {
            const int c = col / 9, ki = (col % 9) / 3, kj = col % 3;
            const int tt = t - 1 + ki, ff = f - 1 + kj;
            if (tt >= 0 && tt < T && ff >= 0 && ff < F) v = z[(((size_t)b * 2 + c) * T + tt) * F + ff];
        }
        rows[i] = v;
    }
}

// adjoint of torch.istft(n_fft 256, hop 128, periodic Hann, center, length L) as used at decoder.py:122-128:
// dwav (B, L) -> dspec (B, 2 = re|im, T, 129).  Frame t, sample m sits at padded position 128 t + m, output n = that - 128.
__global__ __launch_bounds__(256) void istft_adjoint_kernel(const float* __restrict__ dwav, float* __restrict__ dspec, int T, int L) {
    __shared__ float g[256], ct[256], sn[256];
    const int t = blockIdx.x, b = blockIdx.y, m = threadIdx.x;
    {
        float s_, c_;
        sincospif((float)m * (1.0f / 128.0f), &s_, &c_);
        ct[m] = c_;
        sn[m] = s_;
        const float w = 0.5f - 0.5f * c_;
        const int n = 128 * t + m - 128;
        float v = 0.f;
        if (n >= 0 && n < L) {
            // envelope: this frame plus the one overlapping it on this half
            const int mo = m < 128 ? m + 128 : m - 128, to = m < 128 ? t - 1 : t + 1;
            float env = w * w;
            if (to >= 0 && to < T) {
                const float wo = 0.5f - 0.5f * cospif((float)mo * (1.0f / 128.0f));
                env = fmaf(wo, wo, env);
            }
            v = w * dwav[(size_t)b * L + n] / env;
        }
        g[m] = v;
    }
    __syncthreads();
    if (m < 129) {
        float re = 0.f, im = 0.f;
        for (int j = 0; j < 256; ++j) {
            const int k = (m * j) & 255;
            re = fmaf(g[j], ct[k], re);
            im = fmaf(g[j], sn[k], im);
        }
        const float c = (m == 0 || m == 128) ? 1.0f / 256.0f : 2.0f / 256.0f;
        const size_t o = ((size_t)b * 2 * T + t) * 129 + m;
        dspec[o] = c * re;
        dspec[o + (size_t)T * 129] = (m == 0 || m == 128) ? 0.f : -c * im;
    }
}

// complex multiply of (B, 2 x 128, P) maps split as [real 128 | imag 128] (mask_generator.py:71-82): out = a (x) b, or conj(a) (x) b
__global__ __launch_bounds__(256) void cmul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t half,
                                                   size_t total_half, int conj_a) {
    // half = 128 * P elements per (sample, part); total_half = B * half
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_half; i += (size_t)gridDim.x * 256) {
        const size_t bidx = i / half, r = i - bidx * half, o = bidx * 2 * half + r;
        const float ar = a[o], ai = conj_a ? -a[o + half] : a[o + half], br = b[o], bi = b[o + half];
        out[o] = ar * br - ai * bi;
        out[o + half] = ar * bi + ai * br;
    }
}

int launch_patch3x3_rows(const float* z, float* rows, int B, int T, int F, hipStream_t st) {
    if ((size_t)B * T * F * 64 >= 0x7fffffffu) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(patch3x3_rows_kernel, dim3(grid_for((size_t)B * T * F * 64)), dim3(256), 0, st, z, rows, B, T, F);
    return rtfs_launch_status();
}
int launch_istft_adjoint(const float* dwav, float* dspec, int B, int T, int L, hipStream_t st) {
    hipLaunchKernelGGL(istft_adjoint_kernel, dim3(T, B), dim3(256), 0, st, dwav, dspec, T, L);
    return rtfs_launch_status();
}
int launch_cmul(const float* a, const float* b, float* out, int B, size_t half, int conj_a, hipStream_t st) {
    hipLaunchKernelGGL(cmul_kernel, dim3(grid_for((size_t)B * half)), dim3(256), 0, st, a, b, out, half, (size_t)B * half, conj_a);
    return rtfs_launch_status();
}

// ------------------------------------------------------------------------------------------------ CAF glue (layers/fusion.py:252-274)
// attention weights: in (B, 4C, Tv) -> mean over the 4 channels of each group (reshape (B, C, 4, Tv)) -> softmax over Tv.
// One wave per (b, c); Tv <= 256.  bwd: given dout and out, din[c*4 + j, t] = out * (dout - sum(out * dout)) / 4.
__global__ __launch_bounds__(256) void caf_att_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ dout,
                                                      float* __restrict__ din, int nbc, int Tv, int bwd) {
    const int lane = threadIdx.x & 63, bc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bc >= nbc) return;
    if (!bwd) {
        float v[4], m = -INFINITY;
        for (int i = 0; i < 4; ++i) {
            const int t = lane + 64 * i;
            v[i] = -INFINITY;
            if (t < Tv) {
                const float* p = in + (size_t)bc * 4 * Tv + t;
                v[i] = 0.25f * (p[0] + p[Tv] + p[2 * Tv] + p[3 * Tv]);
            }
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float sum = 0.f;
        for (int i = 0; i < 4; ++i) {
            v[i] = (lane + 64 * i) < Tv ? __expf(v[i] - m) : 0.f;
            sum += v[i];
        }
        const float inv = 1.0f / wave_sum(sum);
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < Tv) out[(size_t)bc * Tv + lane + 64 * i] = v[i] * inv;
    } else {
        float pv[4], dv[4], dot = 0.f;
        for (int i = 0; i < 4; ++i) {
            const int t = lane + 64 * i;
            pv[i] = t < Tv ? out[(size_t)bc * Tv + t] : 0.f;
            dv[i] = t < Tv ? dout[(size_t)bc * Tv + t] : 0.f;
            dot = fmaf(pv[i], dv[i], dot);
        }
        dot = wave_sum(dot);
        for (int i = 0; i < 4; ++i) {
            const int t = lane + 64 * i;
            if (t < Tv) {
                const float d = 0.25f * pv[i] * (dv[i] - dot);
                float* p = din + (size_t)bc * 4 * Tv + t;
                p[0] = d; p[Tv] = d; p[2 * Tv] = d; p[3 * Tv] = d;
            }
        }
    }
}
// fused = key * up(r) + up(att) * value; key, value, fused (B*C, T, F); r, att (B*C, Tv); up = nearest over time, broadcast over F
__global__ __launch_bounds__(256) void caf_combine_fwd_kernel(const float* __restrict__ key, const float* __restrict__ value,
                                                              const float* __restrict__ r, const float* __restrict__ att,
                                                              float* __restrict__ out, size_t N, int T, int F, int Tv) {
    const size_t total = N * T * F;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t nt = i / F;
        const int t = (int)(nt % T);
        const size_t j = (nt / T) * Tv + nearest_src(t, Tv, T);
        out[i] = fmaf(key[i], r[j], att[j] * value[i]);
    }
}
// one wave per (n, tv): it owns the frames t that read tv
__global__ __launch_bounds__(256) void caf_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ key,
                                                              const float* __restrict__ value, const float* __restrict__ r,
                                                              const float* __restrict__ att, float* __restrict__ dkey, float* __restrict__ dvalue,
                                                              float* __restrict__ dr, float* __restrict__ datt, size_t N, int T, int F, int Tv) {
    const int lane = threadIdx.x & 63;
    const size_t id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= N * Tv) return;
    const size_t n = id / Tv;
    const int tv = (int)(id % Tv);
    const int t0 = (int)(((long)tv * T + Tv - 1) / Tv), t1 = min(T, (int)(((long)(tv + 1) * T + Tv - 1) / Tv));
    const float rv = r[id], av = att[id];
    float sr = 0.f, sa = 0.f;
    for (int t = t0; t < t1; ++t)
        for (int f = lane; f < F; f += 64) {
            const size_t k = (n * T + t) * F + f;
            const float d = dout[k];
            dkey[k] = d * rv;
            dvalue[k] = d * av;
            sr = fmaf(d, key[k], sr);
            sa = fmaf(d, value[k], sa);
        }
    sr = wave_sum(sr);
    sa = wave_sum(sa);
    if (lane == 0) {
        dr[id] = sr;
        datt[id] = sa;
    }
}
int launch_caf_att(const float* in, float* out, const float* dout, float* din, int nbc, int Tv, bool bwd, hipStream_t st) {
    if (Tv < 1 || Tv > 256) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(caf_att_kernel, dim3(cdiv(nbc, 4)), dim3(256), 0, st, in, out, dout, din, nbc, Tv, bwd ? 1 : 0);
    return rtfs_launch_status();
}
// The same on rows: key, value, out (B, T, F, C) with C fastest; r, att stay (B, C, Tv) (they come from the video side).
// One workgroup per (b, t) slab of F x C floats; C divides 1024, so a thread keeps its channel quad (c4 = tid % (C / 4)) and its
// r / att values (gathered once, stride Tv) while f moves.  BWD: dkey = dout * r, dvalue = dout * att, and the slab's share of
// dr = sum dout * key, datt = sum dout * value, folded over the threads of a channel quad in LDS and added to the (zeroed) outputs -
// a handful of frames t share a video frame tv.
template <bool BWD>
__global__ __launch_bounds__(256) void caf_combine_rows_kernel(const float* __restrict__ dout, const float* __restrict__ key,
                                                               const float* __restrict__ value, const float* __restrict__ r,
                                                               const float* __restrict__ att, float* __restrict__ o1, float* __restrict__ o2,
                                                               float* __restrict__ dr, float* __restrict__ datt, int T, int F, int C, int Tv) {
    __shared__ float4 part[2][256];
    const int b = blockIdx.x / T, t = blockIdx.x - b * T, tid = threadIdx.x;
    const int C4 = C >> 2, c = (tid % C4) * 4, n4 = F * C4;
    const size_t j = ((size_t)b * C + c) * Tv + nearest_src(t, Tv, T);
    f32x4u_t r4, a4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r4[q] = r[j + (size_t)q * Tv];
        a4[q] = att[j + (size_t)q * Tv];
    }
    const size_t base = (size_t)blockIdx.x * n4;
    f32x4u_t sr = {0.f, 0.f, 0.f, 0.f}, sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int i = tid; i < n4; i += 256) {
        const f32x4u_t k4 = reinterpret_cast<const f32x4u_t*>(key)[base + i], v4 = reinterpret_cast<const f32x4u_t*>(value)[base + i];
        if (!BWD) {
            reinterpret_cast<f32x4u_t*>(o1)[base + i] = k4 * r4 + a4 * v4;
        } else {
            const f32x4u_t d = reinterpret_cast<const f32x4u_t*>(dout)[base + i];
            reinterpret_cast<f32x4u_t*>(o1)[base + i] = d * r4;
            reinterpret_cast<f32x4u_t*>(o2)[base + i] = d * a4;
            sr += d * k4;
            sa += d * v4;
        }
    }
    if (!BWD) return;
    part[0][tid] = make_float4(sr[0], sr[1], sr[2], sr[3]);
    part[1][tid] = make_float4(sa[0], sa[1], sa[2], sa[3]);
    __syncthreads();
    if (tid < C4) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
        for (int k = tid; k < 256; k += C4) {
            const float4 p = part[0][k], q = part[1][k];
            x.x += p.x, x.y += p.y, x.z += p.z, x.w += p.w;
            y.x += q.x, y.y += q.y, y.z += q.z, y.w += q.w;
        }
        const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsafeAtomicAdd(dr + j + (size_t)q * Tv, xs[q]);
            unsafeAtomicAdd(datt + j + (size_t)q * Tv, ys[q]);
        }
    }
}
namespace {
inline bool caf_rows_ok(int T, int C, int Tv) { return Tv >= 1 && Tv <= T && C >= 4 && C <= 1024 && !(1024 % C); }
}  // namespace
int launch_caf_combine_rows(const float* key, const float* value, const float* r, const float* att, float* out, int B, int T, int F, int C, int Tv,
                            hipStream_t st) {
    if (!caf_rows_ok(T, C, Tv)) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(caf_combine_rows_kernel<false>, dim3((unsigned)(B * T)), dim3(256), 0, st, (const float*)nullptr, key, value, r, att, out,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, T, F, C, Tv);
    return rtfs_launch_status();
}
int launch_caf_combine_rows_bwd(const float* dout, const float* key, const float* value, const float* r, const float* att, float* dkey,
                                float* dvalue, float* dr, float* datt, int B, int T, int F, int C, int Tv, hipStream_t st) {
    if (!caf_rows_ok(T, C, Tv)) return RTFS_ERR_SHAPE;
    if (hipMemsetAsync(dr, 0, sizeof(float) * B * C * Tv, st) != hipSuccess || hipMemsetAsync(datt, 0, sizeof(float) * B * C * Tv, st) != hipSuccess)
        return RTFS_ERR_LAUNCH;
    hipLaunchKernelGGL(caf_combine_rows_kernel<true>, dim3((unsigned)(B * T)), dim3(256), 0, st, dout, key, value, r, att, dkey, dvalue, dr, datt, T, F,
                       C, Tv);
    return rtfs_launch_status();
}
int launch_caf_combine(const float* key, const float* value, const float* r, const float* att, float* out, size_t N, int T, int F, int Tv,
                       hipStream_t st) {
    if (Tv < 1 || Tv > T) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(caf_combine_fwd_kernel, dim3(grid_for(N * T * F)), dim3(256), 0, st, key, value, r, att, out, N, T, F, Tv);
    return rtfs_launch_status();
}
int launch_caf_combine_bwd(const float* dout, const float* key, const float* value, const float* r, const float* att, float* dkey,
                           float* dvalue, float* dr, float* datt, size_t N, int T, int F, int Tv, hipStream_t st) {
    if (Tv < 1 || Tv > T) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(caf_combine_bwd_kernel, dim3((unsigned)((N * Tv + 3) / 4)), dim3(256), 0, st, dout, key, value, r, att, dkey, dvalue, dr,
                       datt, N, T, F, Tv);
    return rtfs_launch_status();
}

