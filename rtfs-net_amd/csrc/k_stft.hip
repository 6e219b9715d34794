// STFT encoder / iSTFT decoder kernels (gfx950).
//   stft_mfma_kernel   reference TDAVNet/encoder.py:164-172  (torch.stft + stack(re,im).transpose)
//   enc_conv_kernel    reference TDAVNet/encoder.py:146-157,173 (Conv2d 2->C 3x3 'same', no bias) + gLN stats of its output
//   enc_stats_kernel   the gLN statistics of that conv's output without the conv (fused separator path)
//   istft_mfma_kernel  reference TDAVNet/decoder.py:117-128   (tail of ConvTranspose2d + torch.istft)
#include "common.h"
#include "kernels.h"
#include "pipe_helpers.h"

#define NFFT 256
#define HOP 128
#define NBIN 129

// STFT on the matrix cores (round 3; rounds 1-2 had one workgroup per frame, a direct 256-point DFT over an exact cos / sin table in LDS:
// 2 x 256 table reads per output, LDS-bound with bank conflicts on the gather, 105 us).  Output spec (B,2,T,F): [b][0][t][f] = Re, [b][1][t][f] = Im.
// The 256-point real DFT of 32 frames is a (320 x 256) x (256 x 32) GEMM - rows = 129 real
// + 129 imaginary outputs (each padded to 160 = 5 tiles), K = sample index, columns = frames - in the f16 hi / lo split form of the rest of
// the library (3 matrix instructions per product, 2^-22 relative).  One workgroup per (32 frames, mixture), 4 waves:
//   1. the windowed samples of the 32 frames become the B fragments (lane = frame, 8 consecutive samples per K step and lane half), built once
//      and shared through LDS.  Each frame is scaled by the power of two that brings its largest sample to [1, 2) - the split keeps its low
//      half only for values >= 2^-3 or so, and a recording at amplitude 1e-5 would otherwise vanish in f16 - and the column is scaled back
//      at the store: exact;
//   2. a wave owns output tiles {w, w + 4, w + 8}; the twiddle A fragment of (tile, K step) is GATHERED from the exact 256-entry cos / sin
//      table in LDS (angle index f n reduced mod 256 in integers: exact bin indexing) and split on the fly - used once, so no 288 KB image
//      is needed.  105 -> 26 us.
#define STFT_FT 32
__global__ __launch_bounds__(256) void stft_mfma_kernel(const float* __restrict__ wav, float* __restrict__ spec, int L, int T) {
    __shared__ float ct[NFFT], st[NFFT], hann[NFFT];  // 256 cos, -256 sin, window
    __shared__ __attribute__((aligned(16))) half8 Bh[16][64], Bl[16][64];
    __shared__ float pmax[4][STFT_FT];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * STFT_FT, b = blockIdx.y;
    {
        float sn, cs;
        sincospif((float)tid * (1.0f / 128.0f), &sn, &cs);
        ct[tid] = 256.0f * cs;
        st[tid] = -256.0f * sn;
        hann[tid] = 0.5f - 0.5f * cospif((float)tid * (1.0f / 128.0f));  // periodic Hann
    }
    __syncthreads();
    // ---- B fragments: wave w builds K steps 4 w .. 4 w + 3 (samples 64 w .. 64 w + 63) of every frame
    const int t = min(t0 + r, T - 1);  // (frames past the end repeat the last one; their columns are not stored)
    float x[4][8];
    float mx = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = (wave * 4 + k4) * 16 + 8 * h + j;
            int i = t * HOP + n - NFFT / 2;  // index into the un-padded signal
            if (i < 0) i = -i;               // reflect (no edge repeat)
            if (i >= L) i = 2 * (L - 1) - i;
            i = i < 0 ? 0 : i;
            x[k4][j] = wav[(size_t)b * L + i] * hann[n];
            mx = fmaxf(mx, fabsf(x[k4][j]));
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));  // the two lane halves of a frame
    if (h == 0) pmax[wave][r] = mx;
    __syncthreads();
    const float fm = fmaxf(fmaxf(pmax[0][r], pmax[1][r]), fmaxf(pmax[2][r], pmax[3][r]));
    // 2^-floor(log2(fm)) on the exponent bits (fm = 0 or denormal: 1)
    const unsigned eb = (__float_as_uint(fm) >> 23) & 0xFF;
    const float sc = (eb > 1 && eb < 253) ? __uint_as_float((254u - eb) << 23) : 1.0f;
    const float isc = (eb > 1 && eb < 253) ? __uint_as_float(eb << 23) * (1.0f / 256.0f) : (1.0f / 256.0f);  // undoes the scale and the table's x 256
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) split2(x[k4][2 * jp] * sc, x[k4][2 * jp + 1] * sc, hi[jp], lo[jp]);
        Bh[wave * 4 + k4][lane] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi));
        Bl[wave * 4 + k4][lane] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo));
    }
    __syncthreads();
    // ---- GEMM: output tiles mt = wave, wave + 4, wave + 8 (< 10); tile mt: rows f2 = 32 mt + r -> part (mt >= 5), bin f = f2 - 160 part
    f32x16 acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
        const half8 bh = Bh[ks][lane], bl = Bl[ks][lane];
        const int n0 = ks * 16 + 8 * h;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int mt = wave + 4 * m;
            if (mt < 10) {  // wave-uniform
                const int part = mt >= 5, f = mt * 32 + r - 160 * part;
                const float* __restrict__ tab = part ? st : ct;
                const bool live = f < NBIN;
                float tw[8];
                int k = f * n0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    tw[j] = live ? tab[k & (NFFT - 1)] : 0.f;
                    k += f;
                }
                unsigned hi[4], lo[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) split2(tw[2 * jp], tw[2 * jp + 1], hi[jp], lo[jp]);
                const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi)), al = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m], 0, 0, 0);
            }
        }
    }
    // ---- store: accumulator register q of tile mt = row f2 = 32 mt + (q & 3) + 8 (q >> 2) + 4 h, column = frame t0 + r
    if (t0 + r < T) {
        const size_t plane = (size_t)T * NBIN;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int mt = wave + 4 * m;
            if (mt < 10) {
                const int part = mt >= 5;
                float* __restrict__ o = spec + ((size_t)b * 2 + part) * plane + (size_t)(t0 + r) * NBIN;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int f = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h - 160 * part;
                    if (f < NBIN) o[f] = acc[m][q] * isc;
                }
            }
        }
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

// Tail of the decoder on the matrix cores (round 3; rounds 1-2: one workgroup per hop block, direct sums over the table, 119 us -> 43).
// z (B,18,T,F) holds the per-tap pointwise products z[b][(o*3+dt)*3+df][t][f] = sum_c x[b][c][t][f] * Wdec[c][o][dt][df];
// ConvTranspose2d(pad 1): y[o][t][f] = sum_{dt,df} z[o,dt,df][t+1-dt][f+1-df]; then torch.istft: frame-wise irfft(256) * hann, overlap-add,
// / sum(w^2), drop 128, keep L.  The 256 samples of a frame are a (256 x 256) x (256 x frames) GEMM over K = the 129 real and 127 contributing imaginary bins:
//     y[m] = (1/256) (re_0 + (-1)^m re_128 + 2 sum_{f=1..127} re_f cos(2 pi f m / 256) - im_f sin(2 pi f m / 256))
// K order: k = 0 .. 127 -> re_k, k = 128 .. 254 -> im_{k-127}, k = 255 -> re_128.  One workgroup per (31 hop blocks, mixture) = 32 frames
// (block s overlap-adds the second half of frame s and the first half of frame s + 1, so consecutive workgroups share one frame):
//   1. the shift-sum of the decoder's 18 tap maps (ConvTranspose2d pad 1, decoder.py:117-121) for 32 frames x 258 values goes to LDS;
//   2. B fragments (lane = frame) from there, each frame scaled by the power of two that brings its largest value to [1, 2) (the separated
//      spectrum scales with the recording's amplitude; the f16 split wants O(1)), shared through LDS;
//   3. wave w owns sample tiles w and w + 4 - rows m and m + 128, exactly the two halves that meet in the overlap-add; twiddle A
//      fragments gathered from the exact cos / sin table (index f m mod 256 in integers) and split on the fly;
//   4. epilogue: column r + 1's first half comes over by one lane shift, Hann window, envelope, store.
#define ISTFT_FT 32
__global__ __launch_bounds__(256) void istft_mfma_kernel(const float* __restrict__ z, float* __restrict__ wav, int T, int F, int L, size_t zcs,
                                                         size_t zbs) {
    __shared__ float ct[NFFT], st[NFFT], hann[NFFT];          // 256 cos, -256 sin, window
    __shared__ float S[2][NBIN][ISTFT_FT + 1];                 // [re | im][bin][frame]
    __shared__ __attribute__((aligned(16))) half8 Bh[16][64], Bl[16][64];
    __shared__ float pmax[4][ISTFT_FT];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * (ISTFT_FT - 1), b = blockIdx.y;
    {
        float sn, cs;
        sincospif((float)tid * (1.0f / 128.0f), &sn, &cs);
        ct[tid] = 256.0f * cs;
        st[tid] = -256.0f * sn;
        hann[tid] = 0.5f - 0.5f * cospif((float)tid * (1.0f / 128.0f));
    }
    // ---- 1. shift-sum: S[o][f][tt] = sum_{dt,df} z[(o*3+dt)*3+df][t+1-dt][f+1-df], frames past the end = 0
    const float* __restrict__ zb = z + (size_t)b * zbs;
    // (unconditional, clamped loads + selects: a branch per load costs a full wait each; four items = 36 loads in flight per thread)
#pragma unroll 4
    for (int it = 0; it < (2 * ISTFT_FT * NBIN + 255) / 256; ++it) {
        const int idx = min(tid + 256 * it, 2 * ISTFT_FT * NBIN - 1);  // (the last trip's surplus threads redo the last item)
        const int f = idx % NBIN, tt_ = (idx / NBIN) % ISTFT_FT, o = idx / (NBIN * ISTFT_FT);
        const int t = t0 + tt_;
        float acc = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int tt = t + 1 - dt, ff = f + 1 - df;
                const bool ok = t < T && tt >= 0 && tt < T && ff >= 0 && ff < F;
                const int tc = min(max(tt, 0), T - 1), fc = min(max(ff, 0), F - 1);
                const float v = zb[(size_t)((o * 3 + dt) * 3 + df) * zcs + (size_t)tc * F + fc];
                acc += ok ? v : 0.f;
            }
        S[o][f][tt_] = acc;
    }
    __syncthreads();
    // ---- 2. B fragments: wave w builds K steps 4 w .. 4 w + 3; K slot k -> (part, bin)
    auto sval = [&](int k) { return k < 128 ? S[0][k][r] : (k < 255 ? S[1][k - 127][r] : S[0][128][r]); };
    float x[4][8];
    float mx = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[k4][j] = sval((wave * 4 + k4) * 16 + 8 * h + j);
            mx = fmaxf(mx, fabsf(x[k4][j]));
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (h == 0) pmax[wave][r] = mx;
    __syncthreads();
    const float fm = fmaxf(fmaxf(pmax[0][r], pmax[1][r]), fmaxf(pmax[2][r], pmax[3][r]));
    const unsigned eb = (__float_as_uint(fm) >> 23) & 0xFF;
    const bool scaled = eb > 1 && eb < 253;
    const float sc = scaled ? __uint_as_float((254u - eb) << 23) : 1.0f;
    // undoes the frame scale, the table's x 256 and the transform's 1 / 256 (the factor 2 of bins 1 .. 127 is in the fragments)
    const float isc = (scaled ? __uint_as_float(eb << 23) : 1.0f) * (1.0f / 65536.0f);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) split2(x[k4][2 * jp] * sc, x[k4][2 * jp + 1] * sc, hi[jp], lo[jp]);
        Bh[wave * 4 + k4][lane] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi));
        Bl[wave * 4 + k4][lane] = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo));
    }
    __syncthreads();
    // ---- 3. GEMM: sample tiles wave and wave + 4; row m = 32 mt + r; coefficient of K slot k: re_f -> c_f cos(2 pi f m / 256) (c = 1 for f = 0, 128,
    //         else 2), im_f -> -2 sin(2 pi f m / 256)
    f32x16 acc[2];
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m2][q] = 0.f;
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
        const half8 bh = Bh[ks][lane], bl = Bl[ks][lane];
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            const int m = (wave + 4 * m2) * 32 + r;
            float tw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 16 + 8 * h + j;
                const int f = k < 128 ? k : (k < 255 ? k - 127 : 128);
                const float v = (k >= 128 && k < 255 ? st : ct)[(f * m) & (NFFT - 1)];
                tw[j] = (f == 0 || f == 128) ? v : 2.0f * v;
            }
            unsigned hi[4], lo[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) split2(tw[2 * jp], tw[2 * jp + 1], hi[jp], lo[jp]);
            const half8 ah = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(hi)), al = __builtin_bit_cast(half8, *reinterpret_cast<f32x4*>(lo));
            acc[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m2], 0, 0, 0);
            acc[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m2], 0, 0, 0);
            acc[m2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m2], 0, 0, 0);
        }
    }
    // ---- 4. overlap-add: hop block s = t0 + r takes samples m = 128 + j of frame s (tile wave + 4, this lane) and m = j of frame s + 1 (tile wave,
    //         the next lane's column); j = 32 wave + (q & 3) + 8 (q >> 2) + 4 h.  Lane 31's block belongs to the next workgroup.
    const int sblk = t0 + r;
    const bool f0 = sblk < T, f1 = sblk + 1 < T;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        const float y1 = __shfl_down(acc[0][q] * isc, 1);  // frame s + 1, sample j (every lane takes part in the shuffle)
        const float y0 = acc[1][q] * isc;                   // frame s, sample 128 + j
        const float w0 = hann[HOP + j], w1 = hann[j];
        const float num = (f0 ? y0 * w0 : 0.f) + (f1 ? y1 * w1 : 0.f);
        const float env = (f0 ? w0 * w0 : 0.f) + (f1 ? w1 * w1 : 0.f);
        const int n = sblk * HOP + j;
        if (r < ISTFT_FT - 1 && n < L && (f0 || f1)) wav[(size_t)b * L + n] = num / env;
    }
}

int launch_stft(const float* wav, float* spec, int B, int L, int T, hipStream_t st) {
    hipLaunchKernelGGL(stft_mfma_kernel, dim3(cdiv(T, STFT_FT), B), dim3(256), 0, st, wav, spec, L, T);
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
    if (F != NBIN) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(istft_mfma_kernel, dim3(cdiv(cdiv(L, HOP), ISTFT_FT - 1), B), dim3(256), 0, st, z, wav, T, F, L, zcs, zbs);
    return rtfs_launch_status();
}
