// Cross-dimensional Attention Fusion (reference ATTNFusionCell.forward, src/models/layers/fusion.py:252-274).
//   caf_video_kernel  video side, one workgroup per sample:
//       r   = gLN(Conv1d(512->256, k1, groups 256))(video)                          (resize, :255)
//       att = softmax_Tv( mean_k gLN(Conv1d(512->1024, k1, groups 256))(video) )    (attention_embed, :261-264)
//   caf_apply_kernel  audio side, one pass over the (B,256,T,F) features:
//       out = ReLU(BN(dw1x1 audio)) * r[tv(t)] + att[tv(t)] * BN(dw1x1 audio)       (:258-259,268-272)
//     with tv(t) = floor(t*Tv/T) (legacy nearest) and the eval-mode BatchNorm folded into one FMA.
#include "common.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void caf_video_kernel(CafArgs a) {
    __shared__ double red[8];
    __shared__ double tot[4];
    const int b = blockIdx.x, c = threadIdx.x;  // one thread per audio channel (256)
    const int Tv = a.Tv;
    const float* v0 = a.video + ((size_t)b * 512 + 2 * c) * Tv;
    const float* v1 = v0 + Tv;
    const float wr0 = a.w_resize[2 * c], wr1 = a.w_resize[2 * c + 1], br = a.b_resize[c];
    float wa0[4], wa1[4], ba[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wa0[k] = a.w_att[(4 * c + k) * 2];
        wa1[k] = a.w_att[(4 * c + k) * 2 + 1];
        ba[k] = a.b_att[4 * c + k];
    }
    // pass 1: statistics of both pre-norm tensors over (channels, Tv)
    double sr = 0, ssr = 0, sa = 0, ssa = 0;
#pragma unroll 8
    for (int t = 0; t < Tv; ++t) {  // (unrolled: the row loads of eight frames in flight - at batch 1 this kernel is on the critical path)
        const float x0 = v0[t], x1 = v1[t];
        const float r = fmaf(wr1, x1, fmaf(wr0, x0, br));
        sr += r;
        ssr += (double)r * r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float av = fmaf(wa1[k], x1, fmaf(wa0[k], x0, ba[k]));
            sa += av;
            ssa += (double)av * av;
        }
    }
    double vals[4] = {sr, ssr, sa, ssa};
    for (int i = 0; i < 4; ++i) {
        const double w = wave_sum_d(vals[i]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) tot[i] = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
    }
    float rsc, rsh;
    {
        const double st[2] = {tot[0], tot[1]};
        gln_fold(st, 1.0 / (256.0 * Tv), a.g_resize[c], a.be_resize[c], rsc, rsh);
    }
    float asc[4], ash[4];
    {
        const double st[2] = {tot[2], tot[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) gln_fold(st, 1.0 / (1024.0 * Tv), a.g_att[4 * c + k], a.be_att[4 * c + k], asc[k], ash[k]);
    }
    // pass 2: resize output, and the max of the mean-over-k logits
    float* ro = a.r_out + ((size_t)b * 256 + c) * Tv;
    float* ao = a.att_out + ((size_t)b * 256 + c) * Tv;
    float mx = -3.0e38f;
#pragma unroll 8
    for (int t = 0; t < Tv; ++t) {  // (unrolled: the row loads of eight frames in flight - at batch 1 this kernel is on the critical path)
        const float x0 = v0[t], x1 = v1[t];
        ro[t] = fmaf(fmaf(wr1, x1, fmaf(wr0, x0, br)), rsc, rsh);
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) m += fmaf(fmaf(wa1[k], x1, fmaf(wa0[k], x0, ba[k])), asc[k], ash[k]);
        m *= 0.25f;
        ao[t] = m;
        mx = fmaxf(mx, m);
    }
    float s = 0.f;
    for (int t = 0; t < Tv; ++t) {
        const float e = expf(ao[t] - mx);
        ao[t] = e;
        s += e;
    }
    const float inv = 1.0f / s;
    for (int t = 0; t < Tv; ++t) ao[t] *= inv;
    if (a.r_t && a.att_t) {  // (B, Tv, 256) copies: the block-boundary kernel reads four consecutive channels of a frame at once
        for (int t = 0; t < Tv; ++t) {
            a.r_t[((size_t)b * Tv + t) * 256 + c] = ro[t];
            a.att_t[((size_t)b * Tv + t) * 256 + c] = ao[t];
        }
    }
}

__device__ __forceinline__ void caf_apply_body(const CafArgs& a, const float* __restrict__ AUDIO, float* __restrict__ OUT) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int T = a.T, F = a.F, Tv = a.Tv;
    const float ks = a.w_key[c] * a.bn_key[c] / sqrtf(a.bn_key[768 + c] + RTFS_EPS);
    const float kb = a.bn_key[256 + c] - a.bn_key[512 + c] * a.bn_key[c] / sqrtf(a.bn_key[768 + c] + RTFS_EPS);
    const float vs = a.w_val[c] * a.bn_val[c] / sqrtf(a.bn_val[768 + c] + RTFS_EPS);
    const float vb = a.bn_val[256 + c] - a.bn_val[512 + c] * a.bn_val[c] / sqrtf(a.bn_val[768 + c] + RTFS_EPS);
    const size_t plane = ((size_t)b * 256 + c) * T * F;
    const float* r = a.r_out + ((size_t)b * 256 + c) * Tv;
    const float* at = a.att_out + ((size_t)b * 256 + c) * Tv;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < T * F; p += gridDim.x * 256) {
        const int t = p / F;
        const int tv = nearest_src(t, Tv, T);
        const float x = AUDIO[plane + p];
        OUT[plane + p] = fmaf(fmaxf(fmaf(x, ks, kb), 0.f), r[tv], at[tv] * fmaf(x, vs, vb));
    }
}

__global__ __launch_bounds__(256) void caf_apply_kernel(CafArgs a) { caf_apply_body(a, a.audio, a.out); }

int launch_caf_video(const CafArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(caf_video_kernel, dim3(B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
int launch_caf_apply(const CafArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(caf_apply_kernel, dim3(cdiv(a.T * a.F, 256 * 4), 256, B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
