// Training side, file 1 of 4: the two GEMM kernels every module's forward-with-saved-state and backward are built on.
// (k_train_rnn.hip: SRU / LSTM / GRU scans + dual-path layout; k_train_conv.hip: ConvNormAct stages, depthwise convolutions, glue with
// adjoints; k_train_attn.hip: the two attention modules; entry points: api_train.hip.)
//
// Unlike the inference sweep (k_dualpath16.hip), the training path materialises U = x.W - the backward needs it and the weight gradient
// is a reduction over all rows anyway - so the structure is upstream sru's: GEMM, scan, GEMM; and with activations as rows (b, t, f) x C
// every 1x1 convolution and its two adjoints are these same two kernels.
//   bf16x3 split on the matrix cores (x = x1 + x2, both bf16: products x1.w1 + x2.w1 + x1.w2, error ~2^-17 per product).  bf16 keeps
//   f32's exponent range, so gradients of any magnitude need no scaling; fragments are converted in registers straight from the f32
//   operands in HBM/L2 (v_cvt_pk_bf16_f32).
#include "train_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
struct Frag {
    bf16x8 hi, lo;
};
__device__ __forceinline__ Frag split_bf16(const f32x4& a, const f32x4& b) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 h0 = (__bf16)a[j], h1 = (__bf16)b[j];
        f.hi[j] = h0;
        f.hi[4 + j] = h1;
        f.lo[j] = (__bf16)(a[j] - (float)h0);
        f.lo[4 + j] = (__bf16)(b[j] - (float)h1);
    }
    return f;
}
__device__ __forceinline__ void mfma3(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
}  // namespace

// ------------------------------------------------------------------------------------------------ C = A . Bt^T
// A (M, K) and Bt (N, K), both K-contiguous (lda, ldb multiples of 4 floats); C (M, N).  N % 64 == 0, K % 16 == 0, any M.
// Workgroup = 4 waves stacked along M (256 rows x 64 columns), wave tile 64 x 64.  Workgroups that share the A rows
// (the column blocks of one row block) share blockIdx % 8, i.e. one XCD's L2.
// MODE 0: C = ..., 1: C += ... (read-modify-write), 2: fold form - column block j lands j rows further down in a 64-wide C
// (C[(row + j) * ldc + col % 64] += ..., atomics: the adjoint of the unfold windows, see the dual-path backward).
template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
    g.A += (size_t)blockIdx.y * g.sA;
    g.B += (size_t)blockIdx.y * g.sB;
    g.C += (size_t)blockIdx.y * g.sC;
    const int id = blockIdx.x;
    const int rowblk = (id / (8 * g.ncb)) * 8 + (id & 7), colblk = (id >> 3) % g.ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int m0 = rowblk * 256 + wave * 64, n0 = colblk * 64;
    if (m0 >= g.M) return;
    const float *pa[2], *pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = min(m0 + 32 * i + r, g.M - 1);
        pa[i] = g.A + (size_t)row * g.lda + 8 * h;
        pb[i] = g.B + (size_t)(n0 + 32 * i + r) * g.ldb + 8 * h;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 ra[2][2], rb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ra[i][0] = *(const f32x4*)(pa[i]);
        ra[i][1] = *(const f32x4*)(pa[i] + 4);
        rb[i][0] = *(const f32x4*)(pb[i]);
        rb[i][1] = *(const f32x4*)(pb[i] + 4);
    }
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        const int kn = min(k0 + 16, g.K - 16);
        f32x4 na[2][2], nb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            na[i][0] = *(const f32x4*)(pa[i] + kn);
            na[i][1] = *(const f32x4*)(pa[i] + kn + 4);
            nb[i][0] = *(const f32x4*)(pb[i] + kn);
            nb[i][1] = *(const f32x4*)(pb[i] + kn + 4);
        }
        Frag fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = split_bf16(ra[i][0], ra[i][1]);
            fb[i] = split_bf16(rb[i][0], rb[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma3(acc[i][j], fa[i], fb[j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ra[i][x] = na[i][x];
                rb[i][x] = nb[i][x];
            }
    }
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (row < g.M) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) unsafeAtomicAdd(g.C + (size_t)(row + colblk) * g.ldc + 32 * j + r, acc[i][j][q]);
                }
            }
        return;
    }
    // Epilogue through LDS: the accumulator layout puts a lane's registers in different ROWS, so direct stores are dwords in 128-byte
    // runs (64 store instructions per tile, ~3 TB/s).  Each wave turns its tile, 32 rows at a time, into rows of 16 lanes x 16 bytes:
    // 8 fully coalesced 16-byte stores per half instead of 32 dword ones.
    __shared__ float tile[4][32][68];
    float (*t)[68] = tile[wave];
    const int cq = (lane & 15) * 4, r4 = lane >> 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b4 = *(const f32x4*)(g.bias + n0 + cq);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) t[(q & 3) + 8 * (q >> 2) + 4 * h][32 * j + r] = acc[i][j][q];
        // a wave only reads what it wrote itself: no workgroup barrier (LDS operations of a wave complete in order)
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int lr = r4 + 4 * p8, row = m0 + 32 * i + lr;
            if (row < g.M) {
                f32x4 v = *(const f32x4*)&t[lr][cq] + b4;
                f32x4* p = (f32x4*)(g.C + (size_t)row * g.ldc + n0 + cq);
                if (MODE == 1) v += *p;
                *p = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ C = A . Bt^T, 16x16x32 form
// The same product on v_mfma_f32_16x16x32_bf16 (MODE 0 / 1, K % 32 == 0).  Why a second form: with the 32x32x16 fragment a lane owns 8
// consecutive k of ONE of 32 rows, so a 16-byte load instruction touches 32 rows' lines for 32 bytes each - the kernel above is bound by the
// texture path's line look-ups as much as by HBM (0.36-0.43 of the HBM rate its traffic would allow).  In the 16x16x32 fragment four lanes
// (l >> 4 = 0 .. 3) own 4 x 8 consecutive k of one row: a load instruction touches 16 rows for a whole 128-byte line each - half the
// look-ups per byte - and the matrix pipe loses nothing (48 x 16 cycles per 32 k instead of 24 x 32).  Measured: training step 66.6 -> 66.0 ms
// at batch 16 (two A/B pairs on one box) - a small part of what holds these GEMMs back; the LDS-staged, pre-split-operand form is still open.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma3_16(f32x4_& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt16_kernel(GemmArgs g) {
    g.A += (size_t)blockIdx.y * g.sA;
    g.B += (size_t)blockIdx.y * g.sB;
    g.C += (size_t)blockIdx.y * g.sC;
    const int id = blockIdx.x;
    const int rowblk = (id / (8 * g.ncb)) * 8 + (id & 7), colblk = (id >> 3) % g.ncb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kg = lane >> 4;
    const int m0 = rowblk * 256 + wave * 64, n0 = colblk * 64;
    if (m0 >= g.M) return;
    const float *pa[4], *pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = min(m0 + 16 * i + r, g.M - 1);
        pa[i] = g.A + (size_t)row * g.lda + 8 * kg;
        pb[i] = g.B + (size_t)(n0 + 16 * i + r) * g.ldb + 8 * kg;
    }
    f32x4_ acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4][2], rb[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i][0] = *(const f32x4*)(pa[i]);
        ra[i][1] = *(const f32x4*)(pa[i] + 4);
        rb[i][0] = *(const f32x4*)(pb[i]);
        rb[i][1] = *(const f32x4*)(pb[i] + 4);
    }
    for (int k0 = 0; k0 < g.K; k0 += 32) {
        const int kn = min(k0 + 32, g.K - 32);
        f32x4 na[4][2], nb[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            na[i][0] = *(const f32x4*)(pa[i] + kn);
            na[i][1] = *(const f32x4*)(pa[i] + kn + 4);
            nb[i][0] = *(const f32x4*)(pb[i] + kn);
            nb[i][1] = *(const f32x4*)(pb[i] + kn + 4);
        }
        Frag fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = split_bf16(ra[i][0], ra[i][1]);
            fb[i] = split_bf16(rb[i][0], rb[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma3_16(acc[i][j], fa[i], fb[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                ra[i][x] = na[i][x];
                rb[i][x] = nb[i][x];
            }
    }
    // epilogue through LDS, 32 rows at a time (see gemm_nt_kernel): accumulator register q of tile (i, j) is row 16 i + 4 kg + q, column 16 j + r
    __shared__ float tile[4][32][68];
    float (*t)[68] = tile[wave];
    const int cq = (lane & 15) * 4, r4 = lane >> 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b4 = *(const f32x4*)(g.bias + n0 + cq);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) t[16 * ii + 4 * kg + q][16 * j + r] = acc[2 * half + ii][j][q];
        // a wave only reads what it wrote itself: no workgroup barrier (LDS operations of a wave complete in order)
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const int lr = r4 + 4 * p8, row = m0 + 32 * half + lr;
            if (row < g.M) {
                f32x4 v = *(const f32x4*)&t[lr][cq] + b4;
                f32x4* p = (f32x4*)(g.C + (size_t)row * g.ldc + n0 + cq);
                if (MODE == 1) v += *p;
                *p = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ C += A^T . B (split K)
// A (K, M), B (K, N), both with K as the slow axis; C (M, N) must hold the running sum (zeroed by the caller).
// M % 64 == 0, N % 64 == 0, any K.  One wave = one 64 x 64 tile of C over one K chunk; the four waves of a workgroup take four
// consecutive chunks of the same tile and fold their sums through LDS, then f32 atomics merge the workgroups (every atomic onto a
// contended address costs ~0.1 us: with one atomic pass per wave the weight gradients of the 1x1 convolutions - 4 tiles, 500 to 2000
// chunks - spent more time queueing at the L2 than streaming their operands).
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs g) {
    __shared__ float red[3][4096];
    g.A += (size_t)blockIdx.y * g.sA;
    g.B += (size_t)blockIdx.y * g.sB;
    g.C += (size_t)blockIdx.y * g.sC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_n = g.N >> 6, tiles = (g.M >> 6) * tiles_n;
    const int tile = (int)(blockIdx.x % (unsigned)tiles), kc = (int)(blockIdx.x / (unsigned)tiles) * 4 + wave;
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    const long kbeg = (long)kc * g.kchunk, kend = min((long)g.K, kbeg + g.kchunk);  // possibly empty (the last workgroup's spare waves)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const float* pa = g.A + m0 + r;
    const float* pb = g.B + n0 + r;
    for (long k0 = kbeg; k0 < kend; k0 += 16) {
        f32x4 va[2][2], vb[2][2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long kk = k0 + 8 * h + j;
            const bool ok = kk < kend;
            const long kr = ok ? kk : kend - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float av = pa[kr * g.lda + 32 * i], bv = pb[kr * g.ldb + 32 * i];
                va[i][j >> 2][j & 3] = ok ? av : 0.f;
                vb[i][j >> 2][j & 3] = ok ? bv : 0.f;
            }
        }
        Frag fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[i] = split_bf16(va[i][0], va[i][1]);
            fb[i] = split_bf16(vb[i][0], vb[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma3(acc[i][j], fa[i], fb[j]);
    }
    if (wave) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) red[wave - 1][((i * 2 + j) * 16 + q) * 64 + lane] = acc[i][j][q];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = ((i * 2 + j) * 16 + q) * 64 + lane;
                const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                unsafeAtomicAdd(g.C + (size_t)row * g.ldc + n0 + 32 * j + r, (acc[i][j][q] + red[0][e]) + (red[1][e] + red[2][e]));
            }
}

int launch_gemm_nt(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K, int mode,
                   hipStream_t st, const float* bias, int batch, size_t sA, size_t sB, size_t sC) {
    if (M < 1 || N < 64 || (N & 63) || K < 16 || (K & 15) || (lda & 3) || (ldb & 3)) return RTFS_ERR_SHAPE;
    if (mode != 2 && ((ldc & 3) || ((size_t)C & 15) || (bias && ((size_t)bias & 15)))) return RTFS_ERR_SHAPE;  // 16-byte row stores
    GemmArgs g;
    g.A = A; g.B = Bt; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias;
    g.sA = sA; g.sB = sB; g.sC = sC;
    g.ncb = N >> 6;
    g.nrb = cdiv(M, 256);
    const long grid = (long)cdiv(g.nrb, 8) * 8 * g.ncb;
    if (grid > 0x7fffffffL) return RTFS_ERR_SHAPE;
    const bool f16 = mode != 2 && (K & 31) == 0;  // the 16x16x32 form (fewer line look-ups per byte of A / Bt)
    if (mode == 2) hipLaunchKernelGGL(gemm_nt_kernel<2>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    else if (mode == 1 && f16) hipLaunchKernelGGL(gemm_nt16_kernel<1>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    else if (mode == 1) hipLaunchKernelGGL(gemm_nt_kernel<1>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    else if (f16) hipLaunchKernelGGL(gemm_nt16_kernel<0>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_nt_kernel<0>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    return rtfs_launch_status();
}

int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, hipStream_t st, int batch,
                   size_t sA, size_t sB, size_t sC) {
    if (M < 64 || (M & 63) || N < 64 || (N & 63) || K < 1 || K > 0x7fffffffL) return RTFS_ERR_SHAPE;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = (int)K;
    g.sA = sA; g.sB = sB; g.sC = sC;
    const int tiles = (M >> 6) * (N >> 6);
    // enough waves to fill the chip several times over, chunks of at least 256 k
    long splits = cdiv(8192, tiles);
    long kchunk = (cdiv((int)cdiv((int)K, (int)splits), 16)) * 16;
    if (kchunk < 256) kchunk = 256;
    splits = (K + kchunk - 1) / kchunk;
    g.kchunk = (int)kchunk;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((splits + 3) / 4 * tiles), batch), dim3(256), 0, st, g);
    return rtfs_launch_status();
}

