// STFT encoder / iSTFT decoder kernels (gfx950).
//   stft_kernel        reference TDAVNet/encoder.py:164-172  (torch.stft + stack(re,im).transpose)
//   enc_conv_kernel    reference TDAVNet/encoder.py:146-157,173 (Conv2d 2->C 3x3 'same', no bias) + gLN stats of its output
//   dec_shift_sum_istft reference TDAVNet/decoder.py:117-128   (tail of ConvTranspose2d + torch.istft)
#include "common.h"
#include "kernels.h"

#define NFFT 256
#define HOP 128
#define NBIN 129

// One workgroup per (frame t, batch b): 256 threads.  Direct 256-point real DFT with an exact
// table of cos/sin(2*pi*k/256) built in LDS (angle index reduced mod 256 in integers, so the
// bin indexing is exact).  Output spec (B,2,T,F): [b][0][t][f] = Re, [b][1][t][f] = Im.
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wav, float* __restrict__ spec, int L, int T) {
    __shared__ float xw[NFFT];
    __shared__ float ct[NFFT];
    __shared__ float st[NFFT];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    {
        float s, c;
        sincospif((float)tid * (1.0f / 128.0f), &s, &c);
        ct[tid] = c;
        st[tid] = s;
        int n = t * HOP + tid - NFFT / 2;  // index into the un-padded signal
        if (n < 0) n = -n;                 // reflect (no edge repeat)
        if (n >= L) n = 2 * (L - 1) - n;
        n = n < 0 ? 0 : n;
        const float w = 0.5f - 0.5f * cospif((float)tid * (1.0f / 128.0f));  // periodic Hann
        xw[tid] = wav[(size_t)b * L + n] * w;
    }
    __syncthreads();
    if (tid < NBIN) {
        float re = 0.f, im = 0.f;
#pragma unroll 8
        for (int n = 0; n < NFFT; ++n) {
            const int k = (tid * n) & (NFFT - 1);
            const float x = xw[n];
            re = fmaf(x, ct[k], re);
            im = fmaf(-x, st[k], im);
        }
        const size_t plane = (size_t)T * NBIN;
        spec[((size_t)b * 2 + 0) * plane + (size_t)t * NBIN + tid] = re;
        spec[((size_t)b * 2 + 1) * plane + (size_t)t * NBIN + tid] = im;
    }
}

// Conv2d(2 -> C, 3x3, 'same', no bias) over spec (B,2,T,F) -> a0 (B,C,T,F) with channel stride cs.
// One thread per PAIR of adjacent pixels keeps its 2 x 18 taps in registers and walks the C output channels, one
// unaligned 8-byte store per channel (the kernel is a 1 GB write stream); the weights are wave-uniform (scalar loads).
// Also accumulates (sum, sumsq) of a0 per sample.
__global__ __launch_bounds__(256) void enc_conv_kernel(const float* __restrict__ spec, const float* __restrict__ w,
                                                       float* __restrict__ a0, double* __restrict__ stats, int C, int T,
                                                       int F, size_t cs, size_t bs) {
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    __shared__ double red[8];
    const int P = T * F;
    const int b = blockIdx.y;
    const int p0 = 2 * (blockIdx.x * 256 + threadIdx.x);
    float tap[2][18];
    const bool live0 = p0 < P, live1 = p0 + 1 < P;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool live = p0 + k < P;
        const int t = live ? (p0 + k) / F : 0, f = live ? (p0 + k) % F : 0;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int tt = t + dt - 1, ff = f + df - 1;
                    const bool ok = live && tt >= 0 && tt < T && ff >= 0 && ff < F;
                    const int tc = tt < 0 ? 0 : (tt < T ? tt : T - 1), fc = ff < 0 ? 0 : (ff < F ? ff : F - 1);
                    const float v = spec[((size_t)b * 2 + ci) * P + (size_t)tc * F + fc];  // unconditional, clamped
                    tap[k][ci * 9 + dt * 3 + df] = ok ? v : 0.f;
                }
    }
    float s = 0.f, ss = 0.f;
    float* out = a0 + (size_t)b * bs + p0;
    // small batches: the channels are cut into gridDim.z groups (a batch-1 launch was 64 workgroups each walking all 256 channels: 98 us)
    const int cg = C / gridDim.z, c0 = blockIdx.z * cg;
    for (int c = c0; c < c0 + cg; ++c) {
        const float* wc = w + c * 18;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            acc0 = fmaf(tap[0][j], wc[j], acc0);
            acc1 = fmaf(tap[1][j], wc[j], acc1);
        }
        if (live1) {
            *reinterpret_cast<f2u*>(out + (size_t)c * cs) = f2u{acc0, acc1};
            s += acc0 + acc1;
            ss = fmaf(acc0, acc0, fmaf(acc1, acc1, ss));
        } else if (live0) {
            out[(size_t)c * cs] = acc0;
            s += acc0;
            ss = fmaf(acc0, acc0, ss);
        }
    }
    if (stats) block_stats_atomic(s, ss, red, stats + 2 * b);
}

// The fused separator never materialises the encoder output a0: its two consumers (k_bnh.hip, k_s3f.hip) rebuild the tiles they need on the
// matrix cores from the spectrogram patches (18 taps per pixel).  What they cannot rebuild is the gLN statistic of a0 - sum and sum of squares
// over the whole (256, T, F) volume of a mixture - which this kernel computes WITHOUT forming a0:
//     sum_c a0[c][p]   = wbar . patch(p)                    wbar = sum_c W[c]            (18)
//     sum_c a0[c][p]^2 = patch(p)^T G patch(p)              G    = sum_c W[c] W[c]^T     (18 x 18, symmetric)
// 189 f64 FMAs per pixel instead of 4608 f32 ones (the quadratic form cancels: f64).  Every workgroup rebuilds G' (upper triangle, off-diagonal
// entries doubled) in LDS from the 256 x 18 weights (256 iterations per thread); workgroup (0, 0) also writes the encoder's f16 hi / lo
// fragment image for the consumers: [tile 8][K step 2][hi|lo][32 rows][h 2][8 halfs], K slot (h, j) of step 0 = tap j of input channel h
// (re / im), of step 1 = tap 8 of channel h for j = 0 and zero above (weights x 256, as every f16x3 image of this library).
template <int PPT>  // pixels per thread, processed TOGETHER: every G' entry read from LDS (a broadcast read still moves 512 bytes) serves PPT pixels
__global__ __launch_bounds__(256) void enc_stats_kernel(const float* __restrict__ spec, const float* __restrict__ w, double* __restrict__ stats,
                                                        _Float16* __restrict__ img, int T, int F, EncPadJobs pad, int B) {
    if ((int)blockIdx.y == B) {  // extra block row: the head / tail kernels' padded weight images (see EncPadJobs)
        typedef unsigned long long u64;
        for (int job = 0; job < 2; ++job) {
            const u64* src = reinterpret_cast<const u64*>(pad.src[job]);
            u64* dst = reinterpret_cast<u64*>(pad.dst[job]);
            if (!src || !dst) continue;
            for (int i = blockIdx.x * 256 + threadIdx.x; i < 4096 * 9; i += gridDim.x * 256) {
                const int row = i / 9, c = i - row * 9;
                dst[i] = c < 8 ? src[row * 8 + c] : 0ull;
            }
        }
        return;
    }
    __shared__ float W[256 * 18];
    __shared__ double G[171 + 18];  // G' rows i: entries j >= i at i*18 - i(i-1)/2 + (j - i); then wbar
    __shared__ double red[8];
    const int tid = threadIdx.x, b = blockIdx.y, P = T * F;
    for (int i = tid; i < 256 * 18; i += 256) W[i] = w[i];
    __syncthreads();
    if (tid < 171 + 18) {
        double acc = 0;
        if (tid < 171) {
            int i = 0, base = 0;
            while (tid >= base + 18 - i) { base += 18 - i; ++i; }
            const int j = i + tid - base;
#pragma unroll 8
            for (int c = 0; c < 256; ++c) acc = fma((double)W[c * 18 + i], (double)W[c * 18 + j], acc);
            if (j != i) acc *= 2.0;
        } else {
#pragma unroll 8
            for (int c = 0; c < 256; ++c) acc += (double)W[c * 18 + tid - 171];
        }
        G[tid] = acc;
    }
    if (img && blockIdx.x == 0 && b == 0) {
        for (int e = tid; e < 8 * 2 * 32 * 2 * 8; e += 256) {  // (tile, step, row, h, j); hi and lo written together
            const int j = e & 7, h = (e >> 3) & 1, r = (e >> 4) & 31, s = (e >> 9) & 1, kc = e >> 10;
            const int c = kc * 32 + r;
            const float v = s == 0 ? 256.0f * W[c * 18 + h * 9 + j] : (j == 0 ? 256.0f * W[c * 18 + h * 9 + 8] : 0.f);
            const _Float16 hi = (_Float16)v;
            const size_t o = ((size_t)((kc * 2 + s) * 2) * 64 + r * 2 + h) * 8 + j;
            img[o] = hi;
            img[o + 64 * 8] = (_Float16)(v - (float)hi);
        }
    }
    __syncthreads();
    double s1 = 0, s2 = 0;
    {
        double q[PPT][18];
#pragma unroll
        for (int it = 0; it < PPT; ++it) {
            const int p = (blockIdx.x * PPT + it) * 256 + tid;
            const bool live = p < P;
            const int t = live ? p / F : 0, f = live ? p - t * F : 0;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int tt = t + dt - 1, ff = f + df - 1;
                        const bool ok = live && tt >= 0 && tt < T && ff >= 0 && ff < F;
                        const int tc = tt < 0 ? 0 : (tt < T ? tt : T - 1), fc = ff < 0 ? 0 : (ff < F ? ff : F - 1);
                        const float v = spec[((size_t)b * 2 + ci) * P + (size_t)tc * F + fc];  // unconditional, clamped
                        q[it][ci * 9 + dt * 3 + df] = ok ? (double)v : 0.0;
                    }
        }
        int k = 0;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            double u[PPT];
#pragma unroll
            for (int it = 0; it < PPT; ++it) u[it] = 0;
#pragma unroll
            for (int j = i; j < 18; ++j) {
                const double g = G[k++];
#pragma unroll
                for (int it = 0; it < PPT; ++it) u[it] = fma(g, q[it][j], u[it]);
            }
            const double wb = G[171 + i];
#pragma unroll
            for (int it = 0; it < PPT; ++it) {
                s2 = fma(q[it][i], u[it], s2);
                s1 = fma(wb, q[it][i], s1);
            }
        }
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((tid & 63) == 0) {
        red[2 * (tid >> 6)] = s1;
        red[2 * (tid >> 6) + 1] = s2;
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(stats + 2 * b, red[0] + red[2] + red[4] + red[6]);
        atomicAdd(stats + 2 * b + 1, red[1] + red[3] + red[5] + red[7]);
    }
}

// Tail of the decoder.  z (B,18,T,F) holds the per-tap pointwise products
//   z[b][(o*3+dt)*3+df][t][f] = sum_c x[b][c][t][f] * Wdec[c][o][dt][df]
// ConvTranspose2d(pad 1): y[o][t][f] = sum_{dt,df} z[o,dt,df][t+1-dt][f+1-df].
// Then torch.istft: frame-wise irfft(256) * hann, overlap-add, / sum(w^2), drop 128, keep L.
// One workgroup per hop-block s of 128 output samples: the two frames that overlap it are
// t0 = s (second half, m = 128 + n) and t1 = s + 1 (first half, m = n).
__global__ __launch_bounds__(128) void dec_shift_sum_istft_kernel(const float* __restrict__ z, float* __restrict__ wav,
                                                                  int T, int F, int L, size_t zcs, size_t zbs) {
    __shared__ float re[2][NBIN];
    __shared__ float im[2][NBIN];
    __shared__ float ct[NFFT];
    __shared__ float st[NFFT];
    const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < NFFT; k += 128) {
        float sn, cs;
        sincospif((float)k * (1.0f / 128.0f), &sn, &cs);
        ct[k] = cs;
        st[k] = sn;
    }
    for (int idx = tid; idx < 2 * 2 * NBIN; idx += 128) {
        const int fr = idx / (2 * NBIN), o = (idx / NBIN) & 1, f = idx % NBIN;
        const int t = s + fr;
        float acc = 0.f;
        if (t < T) {
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int tt = t + 1 - dt, ff = f + 1 - df;
                    if (tt >= 0 && tt < T && ff >= 0 && ff < F)
                        acc += z[(size_t)b * zbs + (size_t)((o * 3 + dt) * 3 + df) * zcs + (size_t)tt * F + ff];
                }
        }
        (o == 0 ? re : im)[fr][f] = acc;
    }
    __syncthreads();
    const int n = s * HOP + tid;  // output sample
    if (n >= L) return;
    float num = 0.f, env = 0.f;
#pragma unroll
    for (int fr = 0; fr < 2; ++fr) {
        const int t = s + fr;
        if (t >= T) continue;
        const int m = fr == 0 ? tid + HOP : tid;  // sample index inside the frame
        // irfft: bins 0 and 128 contribute their real part only
        float acc = re[fr][0] + ((m & 1) ? -re[fr][128] : re[fr][128]);
        float a2 = 0.f;
        for (int f = 1; f < 128; ++f) {
            const int k = (f * m) & (NFFT - 1);
            a2 = fmaf(re[fr][f], ct[k], a2);
            a2 = fmaf(-im[fr][f], st[k], a2);
        }
        acc = (acc + 2.f * a2) * (1.0f / NFFT);
        const float w = 0.5f - 0.5f * cospif((float)m * (1.0f / 128.0f));
        num = fmaf(acc, w, num);
        env = fmaf(w, w, env);
    }
    wav[(size_t)b * L + n] = num / env;
}

int launch_stft(const float* wav, float* spec, int B, int L, int T, hipStream_t st) {
    hipLaunchKernelGGL(stft_kernel, dim3(T, B), dim3(256), 0, st, wav, spec, L, T);
    return rtfs_launch_status();
}

int launch_enc_conv(const float* spec, const float* w, float* a0, double* stats, int B, int C, int T, int F, size_t cs,
                    size_t bs, hipStream_t st) {
    const int gx = cdiv(cdiv(T * F, 2), 256);
    int gz = 1;
    while (gz < 8 && gx * B * gz < 512 && C % (2 * gz) == 0) gz *= 2;
    hipLaunchKernelGGL(enc_conv_kernel, dim3(gx, B, gz), dim3(256), 0, st, spec, w, a0, stats, C, T, F, cs, bs);
    return rtfs_launch_status();
}

int launch_enc_stats(const float* spec, const float* w, double* stats, void* img, const EncPadJobs& pad, int B, int T, int F, hipStream_t st) {
    const int gy = B + ((pad.src[0] && pad.dst[0]) || (pad.src[1] && pad.dst[1]) ? 1 : 0);
    if (B * cdiv(T * F, 256) >= 2048)
        hipLaunchKernelGGL(enc_stats_kernel<4>, dim3(cdiv(T * F, 256 * 4), gy), dim3(256), 0, st, spec, w, stats, reinterpret_cast<_Float16*>(img), T, F, pad, B);
    else
        hipLaunchKernelGGL(enc_stats_kernel<1>, dim3(cdiv(T * F, 256), gy), dim3(256), 0, st, spec, w, stats, reinterpret_cast<_Float16*>(img), T, F, pad, B);
    return rtfs_launch_status();
}

int launch_dec_istft(const float* z, float* wav, int B, int T, int F, int L, size_t zcs, size_t zbs, hipStream_t st) {
    hipLaunchKernelGGL(dec_shift_sum_istft_kernel, dim3(cdiv(L, HOP), B), dim3(128), 0, st, z, wav, T, F, L, zcs, zbs);
    return rtfs_launch_status();
}
