// Video-side VP block: the 1-D TDANetBlock (upsampling_depth 4, kernel 3, BatchNorm1d, GlobalAttention = MHSA + FFN)
// applied to the (B, 512, Tv) lip embedding (reference separators/tdanet.py:104-131 with yaml video_params;
// layers/attention.py:28-73,192-220; layers/conv_layers.py:218-259; layers/fusion.py:54-69).
// 0.004 GMAC per sample: one workgroup per sample runs the whole block out of LDS, stage by stage.
// Eval-mode BatchNorm (and the conv bias in front of it) is folded into (scale, shift) at pack time.
#include "common.h"
#include "kernels.h"

#define VC 64     // hidden channels
#define VIN 512   // embedding channels
#define VDEPTH 4
#define VP_NT 512  // threads per workgroup (one workgroup per sample: at small batches this kernel is the forward's critical path, and
                    // every stage below is a short loop over at most 64 x Tv items between two barriers - 256 threads took 0.5 ms)

namespace {
struct Ims {  // InjectionMultiSum parameters: dw conv k3 (no bias) + folded BN for local / global_emb / global_gate
    const float *lw, *ls, *lb, *ew, *es, *eb, *gw, *gs, *gb;
};
__device__ __forceinline__ const float* take(const float*& p, int n) {
    const float* r = p;
    p += (n + 63) / 64 * 64;
    return r;
}
__device__ __forceinline__ Ims take_ims(const float*& p) {
    Ims m;
    m.lw = take(p, VC * 3); m.ls = take(p, VC); m.lb = take(p, VC);
    m.ew = take(p, VC * 3); m.es = take(p, VC); m.eb = take(p, VC);
    m.gw = take(p, VC * 3); m.gs = take(p, VC); m.gb = take(p, VC);
    return m;
}
__device__ __forceinline__ float tap3(const float* x, int n, int t) { return (t >= 0 && t < n) ? x[t] : 0.f; }
// depthwise conv k=3 'same' (pad 1) of row x (length n) at position t with weights w[3]
__device__ __forceinline__ float dw3(const float* x, int n, int t, const float* w) {
    return w[0] * tap3(x, n, t - 1) + w[1] * tap3(x, n, t) + w[2] * tap3(x, n, t + 1);
}
}  // namespace

__global__ __launch_bounds__(VP_NT) void vp_block_kernel(const float* __restrict__ video, const float* __restrict__ pack,
                                                       float* __restrict__ out, int Tv) {
    extern __shared__ float lds[];
    __shared__ double red[2 * (VP_NT / 64)];
    __shared__ float stat[2];
    const int tid = threadIdx.x, b = blockIdx.x;
    int len[VDEPTH];
    len[0] = Tv;
    for (int i = 1; i < VDEPTH; ++i) len[i] = (len[i - 1] - 1) / 2 + 1;
    const int Lg = len[VDEPTH - 1];
    // ---- parameter cursor (order = packing.pack_vp)
    const float* p = pack;
    const float* gw = take(p, VIN); const float* gb = take(p, VIN); const float* gslope = take(p, 1);
    const float* proj_wt = take(p, VIN * VC); const float* proj_b = take(p, VC);
    const float *dsw[VDEPTH], *dss[VDEPTH], *dsb[VDEPTH];
    for (int i = 0; i < VDEPTH; ++i) { dsw[i] = take(p, VC * 3); dss[i] = take(p, VC); dsb[i] = take(p, VC); }
    const float* ln1w = take(p, VC); const float* ln1b = take(p, VC); const float* pe = take(p, 16 * VC);
    const float* inw = take(p, 192 * VC); const float* inb = take(p, 192);
    const float* ow = take(p, VC * VC); const float* ob = take(p, VC);
    const float* ln2w = take(p, VC); const float* ln2b = take(p, VC);
    const float* encw = take(p, 128 * VC); const float* encg = take(p, 128); const float* encb = take(p, 128);
    const float* refw = take(p, 128 * 3); const float* refb = take(p, 128);
    const float* decw = take(p, VC * 128); const float* decg = take(p, VC); const float* decb = take(p, VC);
    Ims fus[VDEPTH], cat[VDEPTH - 1];
    for (int i = 0; i < VDEPTH; ++i) fus[i] = take_ims(p);
    for (int i = 0; i < VDEPTH - 1; ++i) cat[i] = take_ims(p);
    const float* res_wt = take(p, VC * VIN); const float* res_b = take(p, VIN);
    const float slope = gslope[0];

    // ---- LDS carve-up (floats).  R2 is two 64 x Tv buffers for the projection chunk / `expanded` pong / lazily
    // computed x_fused[i]; while the attention + FFN run (nothing else is live there) it holds their work buffers.
    float* D[VDEPTH];  // downsampled pyramids d_i (64 x len_i)
    float* q = lds;
    for (int i = 0; i < VDEPTH; ++i) { D[i] = q; q += VC * len[i]; }
    float* XE = q; q += VC * Tv;           // x_enc, later `expanded` ping
    float* G = q; q += VC * 16;            // global features (64 x Lg)
    float* XF3 = q; q += VC * 16;          // x_fused[depth-1] (64 x Lg)
    float* R2 = q;
    float* EX = R2;                        // `expanded` pong; residual chunk RS during the projection
    float* TMP = R2 + VC * Tv;             // x_fused[i] for the concat step in flight
    float* RS = EX;
    float* Y = R2;                         // (Lg x 64) token-major work buffers
    float* Y2 = Y + 16 * VC;
    float* QKV = Y2 + 16 * VC;
    float* SC = QKV + 16 * 192;
    float* HID = SC + 8 * 16 * 16;
    float* HID2 = HID + 128 * 16;
    const float* vb = video + (size_t)b * VIN * Tv;

    // ---- 1. gateway (dw 1x1 + PReLU) + projection 512 -> 64, K chunked by 64 input channels
    {
        constexpr int NQ = VP_NT / 64, NJ = 128 / NQ;  // time slots tq + NQ j, j < NJ cover Tv <= 120
        const int co = tid & 63, tq = tid >> 6;
        float acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
        for (int c0 = 0; c0 < VIN; c0 += 64) {
            __syncthreads();
            for (int i = tid; i < 64 * Tv; i += VP_NT) {
                const int ci = c0 + i / Tv;
                RS[i] = preluf_(fmaf(vb[(size_t)c0 * Tv + i], gw[ci], gb[ci]), slope);
            }
            __syncthreads();
            for (int k0 = 0; k0 < 64; k0 += 16) {
                float w[16];  // 16 independent weight loads in flight instead of one L2 round trip per k
#pragma unroll
                for (int k = 0; k < 16; ++k) w[k] = proj_wt[(c0 + k0 + k) * VC + co];
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int t = tq + NQ * j;
                        if (t < Tv) acc[j] = fmaf(w[k], RS[(k0 + k) * Tv + t], acc[j]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = tq + NQ * j;
            if (t < Tv) XE[co * Tv + t] = acc[j] + proj_b[co];
        }
    }
    __syncthreads();
    // ---- 2. bottom-up pyramid: d0 = BN(dw3 s1 (x_enc)); d_i = BN(dw3 s2 pad1 (d_{i-1}))
    for (int i = tid; i < VC * Tv; i += VP_NT) {
        const int c = i / Tv, t = i - c * Tv;
        D[0][i] = fmaf(dw3(XE + c * Tv, Tv, t, dsw[0] + c * 3), dss[0][c], dsb[0][c]);
    }
    __syncthreads();
    for (int lv = 1; lv < VDEPTH; ++lv) {
        const int n = len[lv - 1], m = len[lv];
        for (int i = tid; i < VC * m; i += VP_NT) {
            const int c = i / m, t = i - c * m;
            const float* x = D[lv - 1] + c * n;
            const float* w = dsw[lv] + c * 3;
            const float v = w[0] * tap3(x, n, 2 * t - 1) + w[1] * tap3(x, n, 2 * t) + w[2] * tap3(x, n, 2 * t + 1);
            D[lv][i] = fmaf(v, dss[lv][c], dsb[lv][c]);
        }
        __syncthreads();
    }
    // ---- 3. global features: sum of adaptive average pools to length Lg
    for (int i = tid; i < VC * Lg; i += VP_NT) {
        const int c = i / Lg, t = i - c * Lg;
        float s = 0.f;
        for (int lv = 0; lv < VDEPTH; ++lv) {
            const int n = len[lv];
            const int s0 = (t * n) / Lg, s1 = ((t + 1) * n + Lg - 1) / Lg;
            float a = 0.f;
            for (int k = s0; k < s1; ++k) a += D[lv][c * n + k];
            s += a / (float)(s1 - s0);
        }
        G[i] = s;
    }
    __syncthreads();
    // ---- 4. MHSA (8 heads x 8): LayerNorm -> +PE -> self-attention -> +res -> LayerNorm -> + input
    if (tid < Lg) {  // LayerNorm over channels per token, + positional encoding
        const int t = tid;
        float mu = 0.f;
        for (int c = 0; c < VC; ++c) mu += G[c * Lg + t];
        mu /= VC;
        float var = 0.f;
        for (int c = 0; c < VC; ++c) { const float d = G[c * Lg + t] - mu; var += d * d; }
        const float rstd = 1.0f / sqrtf(var / VC + 1e-5f);
        for (int c = 0; c < VC; ++c) Y[t * VC + c] = fmaf((G[c * Lg + t] - mu) * rstd, ln1w[c], ln1b[c]) + pe[t * VC + c];
    }
    __syncthreads();
    for (int i = tid; i < Lg * 192; i += VP_NT) {
        const int t = i / 192, o = i - t * 192;
        float a = inb[o];
#pragma unroll 16
        for (int c = 0; c < VC; ++c) a = fmaf(Y[t * VC + c], inw[o * VC + c], a);
        QKV[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < 8 * Lg * Lg; i += VP_NT) {
        const int hd = i / (Lg * Lg), r = i - hd * Lg * Lg, ti = r / Lg, tj = r - ti * Lg;
        float a = 0.f;
        for (int d = 0; d < 8; ++d) a = fmaf(QKV[ti * 192 + hd * 8 + d], QKV[tj * 192 + 64 + hd * 8 + d], a);
        SC[(hd * 16 + ti) * 16 + tj] = a * 0.35355339059327373f;  // 1/sqrt(8)
    }
    __syncthreads();
    for (int i = tid; i < 8 * Lg; i += VP_NT) {
        float* row = SC + (size_t)(i / Lg * 16 + i % Lg) * 16;
        float mx = -3.0e38f;
        for (int j = 0; j < Lg; ++j) mx = fmaxf(mx, row[j]);
        float s = 0.f;
        for (int j = 0; j < Lg; ++j) { row[j] = expf(row[j] - mx); s += row[j]; }
        for (int j = 0; j < Lg; ++j) row[j] /= s;
    }
    __syncthreads();
    for (int i = tid; i < Lg * VC; i += VP_NT) {  // attention output, token-major (t, head*8+d)
        const int t = i / VC, o = i - t * VC, hd = o >> 3;
        float a = 0.f;
        for (int j = 0; j < Lg; ++j) a = fmaf(SC[(hd * 16 + t) * 16 + j], QKV[j * 192 + 128 + o], a);
        Y2[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < Lg * VC; i += VP_NT) {  // out_proj + residual (the post-PE tokens)
        const int t = i / VC, o = i - t * VC;
        float a = ob[o];
#pragma unroll 16
        for (int c = 0; c < VC; ++c) a = fmaf(Y2[t * VC + c], ow[o * VC + c], a);
        QKV[i] = a + Y[i];
    }
    __syncthreads();
    if (tid < Lg) {  // LayerNorm 2, back to channel-major, + block input
        const int t = tid;
        float mu = 0.f;
        for (int c = 0; c < VC; ++c) mu += QKV[t * VC + c];
        mu /= VC;
        float var = 0.f;
        for (int c = 0; c < VC; ++c) { const float d = QKV[t * VC + c] - mu; var += d * d; }
        const float rstd = 1.0f / sqrtf(var / VC + 1e-5f);
        for (int c = 0; c < VC; ++c) G[c * Lg + t] += fmaf((QKV[t * VC + c] - mu) * rstd, ln2w[c], ln2b[c]);
    }
    __syncthreads();
    // ---- 5. FFN: 1x1 64->128 (gLN) -> dw3 + bias + ReLU -> 1x1 128->64 (gLN) -> + input
    auto gln_stats = [&](const float* x, int n) {  // leaves (mean, rstd) in stat[]
        float s = 0.f, ss = 0.f;
        for (int i = tid; i < n; i += VP_NT) { s += x[i]; ss = fmaf(x[i], x[i], ss); }
        const double ds = wave_sum_d((double)s), dss2 = wave_sum_d((double)ss);
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = ds; red[2 * (tid >> 6) + 1] = dss2; }
        __syncthreads();
        if (tid == 0) {
            double a = 0, c = 0;
            for (int w = 0; w < VP_NT / 64; ++w) { a += red[2 * w]; c += red[2 * w + 1]; }
            const double mean = a / n;
            double var = c / n - mean * mean;
            var = var < 0 ? 0 : var;
            stat[0] = (float)mean;
            stat[1] = (float)(1.0 / sqrt(var + 1e-5));
        }
        __syncthreads();
    };
    for (int i = tid; i < 128 * Lg; i += VP_NT) {
        const int o = i / Lg, t = i - o * Lg;
        float a = 0.f;
#pragma unroll 16
        for (int c = 0; c < VC; ++c) a = fmaf(encw[o * VC + c], G[c * Lg + t], a);
        HID[i] = a;
    }
    __syncthreads();
    gln_stats(HID, 128 * Lg);
    for (int i = tid; i < 128 * Lg; i += VP_NT) HID[i] = fmaf((HID[i] - stat[0]) * stat[1], encg[i / Lg], encb[i / Lg]);
    __syncthreads();
    for (int i = tid; i < 128 * Lg; i += VP_NT) {
        const int o = i / Lg, t = i - o * Lg;
        HID2[i] = fmaxf(dw3(HID + o * Lg, Lg, t, refw + o * 3) + refb[o], 0.f);
    }
    __syncthreads();
    for (int i = tid; i < VC * Lg; i += VP_NT) {
        const int o = i / Lg, t = i - o * Lg;
        float a = 0.f;
#pragma unroll 16
        for (int c = 0; c < 128; ++c) a = fmaf(decw[o * 128 + c], HID2[c * Lg + t], a);
        Y[i] = a;
    }
    __syncthreads();
    gln_stats(Y, VC * Lg);
    for (int i = tid; i < VC * Lg; i += VP_NT) G[i] += fmaf((Y[i] - stat[0]) * stat[1], decg[i / Lg], decb[i / Lg]);
    __syncthreads();
    // ---- 6. InjectionMultiSum: out = BN(dw3(local)) * sigmoid(BN(dw3(glob)))^ + BN(dw3(glob))^  (^ = nearest up-sampling)
    auto ims = [&](const Ims& m, const float* loc, int n, const float* glob, int ng, const float* add, float* dst) {
        for (int i = tid; i < VC * n; i += VP_NT) {
            const int c = i / n, t = i - c * n;
            const int tg = min((t * ng) / n, ng - 1);
            const float l = fmaf(dw3(loc + c * n, n, t, m.lw + c * 3), m.ls[c], m.lb[c]);
            const float e = fmaf(dw3(glob + c * ng, ng, tg, m.ew + c * 3), m.es[c], m.eb[c]);
            const float g = sigmoidf_(fmaf(dw3(glob + c * ng, ng, tg, m.gw + c * 3), m.gs[c], m.gb[c]));
            dst[i] = fmaf(l, g, e) + (add ? add[i] : 0.f);
        }
        __syncthreads();
    };
    // x_fused[i] = fusion_layers[i](d_i, g) is computed right before the concat step that consumes it.
    // expanded = cat[2](xf2, xf3) + d2 ; then cat[1](xf1, expanded) + d1 ; cat[0](xf0, expanded) + d0
    float* cur = XE;
    float* nxt = EX;
    ims(fus[VDEPTH - 1], D[VDEPTH - 1], Lg, G, Lg, nullptr, XF3);
    ims(fus[VDEPTH - 2], D[VDEPTH - 2], len[VDEPTH - 2], G, Lg, nullptr, TMP);
    ims(cat[VDEPTH - 2], TMP, len[VDEPTH - 2], XF3, Lg, D[VDEPTH - 2], cur);
    for (int lv = VDEPTH - 3; lv >= 0; --lv) {
        ims(fus[lv], D[lv], len[lv], G, Lg, nullptr, TMP);
        ims(cat[lv], TMP, len[lv], cur, len[lv + 1], D[lv], nxt);
        float* t = cur; cur = nxt; nxt = t;
    }
    // ---- 7. residual_conv 64 -> 512 + bias + gateway(video) (recomputed)
    {
        float* ob_ = out + (size_t)b * VIN * Tv;
        constexpr int NTH = VP_NT / VIN > 0 ? VP_NT / VIN : 1;  // threads per output channel (they interleave the frames)
        for (int co = tid % VIN, th = tid / VIN; co < VIN && th < NTH; co += VIN) {  // one output channel per thread group: its 64 weights live in registers
            float w[VC];
#pragma unroll
            for (int c = 0; c < VC; ++c) w[c] = res_wt[c * VIN + co];
            const float bias = res_b[co], g0 = gw[co], g1 = gb[co];
            for (int t = th; t < Tv; t += NTH) {
                float a = bias;
#pragma unroll
                for (int c = 0; c < VC; ++c) a = fmaf(w[c], cur[c * Tv + t], a);
                ob_[(size_t)co * Tv + t] = a + preluf_(fmaf(vb[(size_t)co * Tv + t], g0, g1), slope);
            }
        }
    }
}

size_t vp_lds_bytes(int Tv) {
    int len[VDEPTH];
    len[0] = Tv;
    int sum = Tv;
    for (int i = 1; i < VDEPTH; ++i) { len[i] = (len[i - 1] - 1) / 2 + 1; sum += len[i]; }
    const size_t work = 2 * 16 * VC + 16 * 192 + 8 * 16 * 16 + 2 * 128 * 16;  // attention + FFN buffers aliased onto R2
    const size_t r2 = (size_t)2 * VC * Tv > work ? (size_t)2 * VC * Tv : work;
    return ((size_t)VC * (sum + Tv) + 2 * VC * 16 + r2) * sizeof(float);
}

int launch_vp_block(const float* video, const float* pack, float* out, int B, int Tv, hipStream_t st) {
    int Lg = Tv;
    for (int i = 1; i < VDEPTH; ++i) Lg = (Lg - 1) / 2 + 1;
    if (Tv < 1 || Lg > 16 || Tv > 120) return RTFS_ERR_SHAPE;
    const size_t lds = vp_lds_bytes(Tv);
    if (rtfs_set_max_lds((const void*)vp_block_kernel, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    hipLaunchKernelGGL(vp_block_kernel, dim3(B), dim3(VP_NT), lds, st, video, pack, out, Tv);
    return rtfs_launch_status();
}
