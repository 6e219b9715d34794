// 256 -> 256 pointwise convolutions of the separator head and tail with the PIXELS resident in registers:
//   audio_bottleneck   gLN -> ReLU -> 1x1 conv (reference src/models/TDAVNet/av_model.py bottleneck, conv_layers.py:9-74)
//   S3 mask head       PReLU -> 1x1 conv -> ReLU -> complex product with the encoder output
//                      (reference src/models/layers/mask_generator.py, MaskGenerator.forward)
// Same f16x3 split arithmetic as k_pw16.hip (x = xh + xl, 256 w = wh + wl, three MFMAs, f32 accumulate).
//
// A 256 x 256 weight (256 KB as hi/lo f16) does not fit LDS beside anything else, and k_pw16's LDS-staged X tiles cost
// a global -> register -> LDS transposition per K chunk with the load latency exposed eight times per tile.  Here the
// roles are swapped: a wave loads its 32 pixels x 256 input channels ONCE, straight from HBM in MFMA B-fragment order
// (lane = pixel: coalesced 128 B rows), applies the prologue, splits to hi/lo f16 and keeps the 32 fragments in
// registers.  The weights stream through a double-buffered LDS image one 32-row output tile at a time (8 tiles per pixel
// tile; the next tile's rows are in registers while this one multiplies, one barrier per tile), every output tile is
// finished -- bias / mask epilogue, coalesced stores -- as soon as its 48 MFMAs retire.
// 256-thread workgroups, two per CU (69 KB LDS each): while one streams pixels the other multiplies.
// (Measured alternatives, not kept: a weight-stationary variant -- weights in registers, pixels streamed through LDS -- no
// faster for the bottleneck, slower for S3 whose epilogue registers do not fit beside 128 weight registers; a K-streaming
// variant with 128 resident accumulators and two pixels per lane -- register-bound, 860 / 1100 us; 8-byte accesses through
// an adjacent-lane channel exchange on this kernel -- 850 / 1000 us: the kernel is bound by its load-then-multiply phase
// structure, not by the access width.)
#include "common.h"
#include "kernels.h"

namespace {

enum { PWR_BN = 0, PWR_S3 = 1, PWR_S3T = 2 };  // S3T: S3 fused with the decoder's per-tap 1x1 maps (writes z, not sep)

constexpr int PWR_LDW = 264;                       // staged weight row: 256 k + 8 pad halfs (528 B, ds_read_b128 conflict-free)
constexpr int PWR_BUF = 2 * 32 * PWR_LDW;          // halfs per buffer: [hi|lo][32 co][PWR_LDW]
constexpr int PWR_PT = 128;                        // pixels per workgroup tile (4 waves x 32)
constexpr size_t PWR_LDS = (size_t)2 * PWR_BUF * 2 + 3 * 256 * 4 + 16;

template <int MODE>
__device__ __forceinline__ void pwr_body(const PwArgs& a, const float* __restrict__ X, const float* __restrict__ AUX,
                                         float* __restrict__ OUT, const half8* __restrict__ W16, int ntiles, int tps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Wb = reinterpret_cast<_Float16*>(smem);
    float* sc = reinterpret_cast<float*>(smem + (size_t)2 * PWR_BUF * 2);
    float* sh = sc + 256;
    float* bs = sh + 256;
    int* s_next = reinterpret_cast<int*>(bs + 256);  // [2] next tile from the counter
    constexpr float WINV = 1.0f / 256.0f;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.P;
    const unsigned CS = (unsigned)a.cs;  // channel stride of x / aux / out (the launcher checks 256 * cs < 2^31)
    bs[tid] = a.bias[tid];  // visible after the first barrier
    const float slope = MODE != PWR_BN ? a.slope[0] : 0.f;

    // output tile order: S3 needs rows c and c + 128 (mask real / imaginary part) back to back
    auto tile_of = [](int i) { return MODE != PWR_BN ? (i >> 1) + 4 * (i & 1) : i; };

    // weight rows of output tile ct: the k_pw16 image is [k chunk 8][hi|lo][256 co][32 k]; a (chunk, part, tile) piece is
    // 32 rows x 64 B = 2 KB contiguous.  2048 16-byte pieces per tile, 8 per thread.
    half8 pre[8];
    auto stage_load = [&](int ct) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;           // piece: [chunk 8][part 2][row 32][kq 4]
            const int kq = i & 3, row = (i >> 2) & 31, part = (i >> 7) & 1, chunk = i >> 8;
            pre[j] = W16[((size_t)(chunk * 2 + part) * 256 + ct * 32 + row) * 4 + kq];
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j;
            const int kq = i & 3, row = (i >> 2) & 31, part = (i >> 7) & 1, chunk = i >> 8;
            *reinterpret_cast<half8*>(Wb + buf * PWR_BUF + (part * 32 + row) * PWR_LDW + chunk * 32 + kq * 8) = pre[j];
        }
    };

    stage_load(tile_of(0));
    stage_write(0);
    int cur_b = -1;
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; ++it) {
        if (a.tile_ctr && tid == 0) s_next[it & 1] = (int)atomicAdd(a.tile_ctr, 1u) + (int)gridDim.x;
        const int b = tile / tps;
        const int p = (tile - b * tps) * PWR_PT + wave * 32 + r;
        const bool live = p < P;
        const unsigned pc = live ? p : P - 1;
        if (MODE == PWR_BN && b != cur_b) {  // block-uniform
            __syncthreads();                 // the previous tile's fragments are built
            gln_fold(a.stats + 2 * b, a.inv_count, a.gamma[tid], a.beta[tid], sc[tid], sh[tid]);
            cur_b = b;
        }
        __syncthreads();  // sc/sh visible; weight buffer 0 (written at the end of the previous tile) visible
        // ---- this wave's pixels: 16 k-steps of B fragments, hi and lo
        const float* __restrict__ Xb = X + (size_t)b * 256 * CS;
        const unsigned lofs = (unsigned)(8 * h) * CS + pc;
        half8 bh[16], bl[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Xb[lofs + (unsigned)(ks * 16 + j) * CS];
            float s[8], t[8];
            if (MODE == PWR_BN) {
                *reinterpret_cast<f32x4*>(s) = *reinterpret_cast<const f32x4*>(sc + ks * 16 + 8 * h);
                *reinterpret_cast<f32x4*>(s + 4) = *reinterpret_cast<const f32x4*>(sc + ks * 16 + 8 * h + 4);
                *reinterpret_cast<f32x4*>(t) = *reinterpret_cast<const f32x4*>(sh + ks * 16 + 8 * h);
                *reinterpret_cast<f32x4*>(t + 4) = *reinterpret_cast<const f32x4*>(sh + ks * 16 + 8 * h + 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = v[j];
                x = MODE == PWR_BN ? fmaxf(fmaf(x, s[j], t[j]), 0.f) : preluf_(x, slope);
                const _Float16 hi = (_Float16)x;
                bh[ks][j] = hi;
                bl[ks][j] = (_Float16)(x - (float)hi);
            }
        }
        // ---- 8 output tiles of 32 rows
        f32x16 keep;  // S3: the mask's real-part tile waits for its imaginary partner
        f32x16 acc2;  // S3T: decoder taps (32 rows, 18 live) x this wave's 32 pixels
        // S3T range safety: the separated spectrum (the taps GEMM's B operand) scales with the waveform's amplitude, and an f16 hi / lo
        // split keeps its low part only while |x| >= 2^-3 or so (f16 subnormals are flushed).  The encoder's statistics give the
        // mixture's rms(a0): the encoder rows are multiplied by the power of two that brings it to [1, 2) before the complex product and
        // the accumulator is scaled back - exact, and a quiet (x 1e-4) or hot (x 1e4) recording keeps the full split precision.
        float esc = 1.0f, eisc = WINV;
        if (MODE == PWR_S3T) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc2[q] = 0.f;
            if (a.stats) {
                // exponent arithmetic on the bits (wave-uniform, no transcendental, no extra vector registers: the first version used
                // log2f / exp2f here and pushed this register-bound kernel from 0.81 to 1.49 ms)
                const float ms = (float)(a.stats[2 * b + 1] * a.inv_count);  // mean square of a0 over the mixture
                const int eb = (int)((__float_as_uint(ms) >> 23) & 0xFF) - 127;  // floor(log2(ms)); ms = 0 or denormal -> -127
                int e = -(eb >> 1);                                               // ~ -log2(rms), within a factor 2 either way
                e = e < -40 ? -40 : (e > 40 ? 40 : e);
                esc = __uint_as_float((unsigned)(127 + e) << 23);
                eisc = __uint_as_float((unsigned)(127 - e - 8) << 23);
            }
        }
        float er[16], ei[16];
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const int ct = tile_of(i);
            stage_load(tile_of((i + 1) & 7));  // i == 7: tile 0 again, for the next pixel tile
            if (MODE == PWR_S3 && (i & 1)) {  // (S3T loads them in halves at the point of use: register budget)
                // encoder output rows of this pair, requested under the second tile's MFMAs
                const float* __restrict__ Ab = AUX + ((size_t)b * 256 + (ct - 4) * 32 + 4 * h) * CS + pc;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned ro = (unsigned)((q & 3) + 8 * (q >> 2)) * CS;
                    er[q] = Ab[ro];
                    ei[q] = Ab[ro + 128u * CS];
                }
            }
            const _Float16* wb = Wb + (i & 1) * PWR_BUF + r * PWR_LDW + 8 * h;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(wb + ks * 16);
                const half8 al = *reinterpret_cast<const half8*>(wb + 32 * PWR_LDW + ks * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
            }
            // accumulator register q: output row ct*32 + (q&3) + 8(q>>2) + 4h, pixel r
            if (MODE == PWR_BN) {
                float* __restrict__ Ob = OUT + ((size_t)b * 256 + ct * 32 + 4 * h) * CS + pc;
                if (live) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = (q & 3) + 8 * (q >> 2);
                        Ob[(unsigned)row * CS] = fmaf(acc[q], WINV, bs[ct * 32 + 4 * h + row]);
                    }
                }
            } else if (!(i & 1)) {
                keep = acc;
            } else if (MODE == PWR_S3T) {
                // separated spectrum rows c = m*32 + row (real) and c + 128 (imaginary) of this lane's pixel, eight rows (one
                // K step of the taps GEMM) at a time; they feed the matrix cores as they stand: the taps weight image
                // (packing.taps_perm_image) has its K axis in accumulator-register order
                const int m = ct - 4;
                const float* __restrict__ Ab = AUX + ((size_t)b * 256 + m * 32 + 4 * h) * CS + pc;
                const half8* __restrict__ TW = reinterpret_cast<const half8*>(a.w16b);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float er[8], ei[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned ro = (unsigned)((j & 3) + 8 * (2 * s2 + (j >> 2))) * CS;
                        er[j] = Ab[ro] * esc;
                        ei[j] = Ab[ro + 128u * CS] * esc;
                    }
                    half8 rh, rl, ih, il;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = 8 * s2 + j;
                        const int c = m * 32 + 4 * h + (q & 3) + 8 * (q >> 2);
                        const float mr = fmaxf(fmaf(keep[q], WINV, bs[c]), 0.f);
                        const float mi = fmaxf(fmaf(acc[q], WINV, bs[c + 128]), 0.f);
                        const float o_r = er[j] * mr - ei[j] * mi, o_i = er[j] * mi + ei[j] * mr;
                        const _Float16 a_ = (_Float16)o_r, b_ = (_Float16)o_i;
                        rh[j] = a_;
                        rl[j] = (_Float16)(o_r - (float)a_);
                        ih[j] = b_;
                        il[j] = (_Float16)(o_i - (float)b_);
                    }
                    // image: [m 4][part 2][s 2][hi|lo][32 taps][16 k] halfs -> 16-byte piece index ((idx*2 + hl)*32 + r)*2 + h
                    const int idx_r = (m * 2 + 0) * 2 + s2, idx_i = (m * 2 + 1) * 2 + s2;
                    const half8 trh = TW[((idx_r * 2 + 0) * 32 + r) * 2 + h], trl = TW[((idx_r * 2 + 1) * 32 + r) * 2 + h];
                    const half8 tih = TW[((idx_i * 2 + 0) * 32 + r) * 2 + h], til = TW[((idx_i * 2 + 1) * 32 + r) * 2 + h];
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(trh, rh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(trh, rl, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(trl, rh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tih, ih, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(tih, il, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(til, ih, acc2, 0, 0, 0);
                }
            } else {
                float* __restrict__ Ob = OUT + ((size_t)b * 256 + (ct - 4) * 32 + 4 * h) * CS + pc;
                if (live) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = (q & 3) + 8 * (q >> 2);
                        const int c = (ct - 4) * 32 + 4 * h + row;
                        const float mr = fmaxf(fmaf(keep[q], WINV, bs[c]), 0.f);
                        const float mi = fmaxf(fmaf(acc[q], WINV, bs[c + 128]), 0.f);
                        const unsigned ro = (unsigned)row * CS;
                        Ob[ro] = er[q] * mr - ei[q] * mi;
                        Ob[ro + 128u * CS] = er[q] * mi + ei[q] * mr;
                    }
                }
            }
            stage_write((i + 1) & 1);
            if (i < 7) __syncthreads();  // after i == 7 the barrier at the top of the next pixel tile orders buffer 0
        }
        if (MODE == PWR_S3T && live) {  // z (B, 18, P): tap row (q&3) + 8(q>>2) + 4h of the accumulator tile
            float* __restrict__ Zb = OUT + (size_t)b * a.cout_live * CS + pc;  // z (B, 18, cs)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int tap = (q & 3) + 8 * (q >> 2) + 4 * h;
                if (tap < a.cout_live) Zb[(unsigned)tap * CS] = acc2[q] * eisc;
            }
        }
        if (a.tile_ctr) {  // uniform (the barrier at the top of the next tile orders the s_next slot's reuse two tiles on)
            __syncthreads();
            tile = s_next[it & 1];
        } else {
            tile += gridDim.x;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void pwr_kernel(PwArgs a, int ntiles, int tps) {
    pwr_body<MODE>(a, a.x, a.aux, a.out, reinterpret_cast<const half8*>(a.w16), ntiles, tps);
}

template <int MODE>
int launch_pwr_t(const PwArgs& a_, int B, hipStream_t st) {
    if (a_.cs && a_.cs < a_.P) return RTFS_ERR_SHAPE;
    PwArgs a = a_;
    if (!a.cs) a.cs = a.P;
    if ((size_t)256 * a.cs >= ((size_t)1 << 31)) return RTFS_ERR_SHAPE;  // 32-bit element offsets inside a sample
    if (rtfs_set_max_lds((const void*)pwr_kernel<MODE>, PWR_LDS) != RTFS_OK) return RTFS_ERR_LAUNCH;
    const int tps = cdiv(a.P, PWR_PT), ntiles = tps * B;
    const int grid = ntiles < 512 ? ntiles : 512;  // persistent: 2 workgroups per CU
    hipLaunchKernelGGL((pwr_kernel<MODE>), dim3(grid), dim3(256), PWR_LDS, st, a, ntiles, tps);
    return rtfs_launch_status();
}

}  // namespace

int launch_pwr_audio_bn(const PwArgs& a, int B, hipStream_t st) { return launch_pwr_t<PWR_BN>(a, B, st); }
int launch_pwr_s3(const PwArgs& a, int B, hipStream_t st) { return launch_pwr_t<PWR_S3>(a, B, st); }
int launch_pwr_s3_taps(const PwArgs& a, int B, hipStream_t st) { return a.w16b ? launch_pwr_t<PWR_S3T>(a, B, st) : RTFS_ERR_ARG; }
