// Training-side kernels of the sru.SRU operator (the reference's only native seam,
// rnn_layers.py:99-105,150): a forward that keeps what the backward needs, and the backward itself.
//
// Unlike the inference sweep (k_dualpath16.hip), U = x.W is materialised here - the backward needs it and the weight
// gradient is a reduction over all L*N rows anyway - so the structure is the upstream one: GEMM, scan, GEMM.
//   GEMMs: bf16x3 split on the matrix cores (x = x1 + x2, both bf16: products x1.w1 + x2.w1 + x1.w2, error ~2^-17 per
//          product).  bf16 keeps f32's exponent range, so gradients of any magnitude need no scaling; fragments are
//          converted in registers straight from the f32 operands in HBM/L2 (v_cvt_pk_bf16_f32).
//   scans: one lane per (sequence, direction, unit); the forward scan is the transcendental chain of common.h's
//          sigmoid, the backward scan is FMA-only (f, r are recomputed from the saved cell states).
// Layouts: U (L, N, KC) with column m*64 + dir*32 + j (m = 0 candidate, 1 forget, 2 reset, 3 highway of layer 0), KC = 256
// for layer 0 and 192 above; c, h (L, N, 64) with column dir*32 + j.
#include "common.h"
#include "kernels.h"

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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (row < g.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (MODE == 2) {
                        unsafeAtomicAdd(g.C + (size_t)(row + colblk) * g.ldc + 32 * j + r, acc[i][j][q]);
                    } else {
                        float* p = g.C + (size_t)row * g.ldc + n0 + 32 * j + r;
                        const float v = acc[i][j][q] + (g.bias ? g.bias[n0 + 32 * j + r] : 0.f);
                        *p = MODE == 1 ? *p + v : v;
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ C += A^T . B (split K)
// A (K, M), B (K, N), both with K as the slow axis; C (M, N) must hold the running sum (zeroed by the caller).
// M % 64 == 0, N % 64 == 0, any K.  One wave = one 64 x 64 tile of C over one K chunk; f32 atomics merge the chunks.
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs g) {
    g.A += (size_t)blockIdx.y * g.sA;
    g.B += (size_t)blockIdx.y * g.sB;
    g.C += (size_t)blockIdx.y * g.sC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_n = g.N >> 6, tiles = (g.M >> 6) * tiles_n;
    const long gw = (long)blockIdx.x * 4 + wave;
    const int tile = (int)(gw % tiles), kc = (int)(gw / tiles);
    const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
    const long kbeg = (long)kc * g.kchunk, kend = min((long)g.K, kbeg + g.kchunk);
    if (kbeg >= kend) return;
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
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                unsafeAtomicAdd(g.C + (size_t)row * g.ldc + n0 + 32 * j + r, acc[i][j][q]);
            }
}

int launch_gemm_nt(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K, int mode,
                   hipStream_t st, const float* bias, int batch, size_t sA, size_t sB, size_t sC) {
    if (M < 1 || N < 64 || (N & 63) || K < 16 || (K & 15) || (lda & 3) || (ldb & 3)) return RTFS_ERR_SHAPE;
    GemmArgs g;
    g.A = A; g.B = Bt; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias;
    g.sA = sA; g.sB = sB; g.sC = sC;
    g.ncb = N >> 6;
    g.nrb = cdiv(M, 256);
    const long grid = (long)cdiv(g.nrb, 8) * 8 * g.ncb;
    if (grid > 0x7fffffffL) return RTFS_ERR_SHAPE;
    if (mode == 2) hipLaunchKernelGGL(gemm_nt_kernel<2>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
    else if (mode == 1) hipLaunchKernelGGL(gemm_nt_kernel<1>, dim3((unsigned)grid, batch), dim3(256), 0, st, g);
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
    const long waves = splits * tiles;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)((waves + 3) / 4), batch), dim3(256), 0, st, g);
    return rtfs_launch_status();
}

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

// LayerNorm over channels per position + layout change: x -> xn[n*Ls + s][c]
__global__ __launch_bounds__(256) void dp_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ xn, int R, int Ls) {
    extern __shared__ float tile[];  // [64][Ls + 1]
    __shared__ float mu[256], rs[256];
    const int n = blockIdx.x, tid = threadIdx.x, P = Ls + 1;
    const size_t base = seq_base(n, R, Ls), cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * Ls; idx += 256) {
        const int c = idx / Ls, s = idx - c * Ls;
        tile[c * P + s] = x[base + c * cs + s];
    }
    __syncthreads();
    for (int s = tid; s < Ls; s += 256) {
        float m = 0.f;
        for (int c = 0; c < 64; ++c) m += tile[c * P + s];
        m *= (1.f / 64);
        float v = 0.f;
        for (int c = 0; c < 64; ++c) {
            const float d = tile[c * P + s] - m;
            v = fmaf(d, d, v);
        }
        mu[s & 255] = m;
        rs[s & 255] = 1.0f / sqrtf(v * (1.f / 64) + RTFS_EPS);
        // Ls <= 256 is guaranteed by the launcher, so one round of this loop
    }
    __syncthreads();
    const int c = tid & 63;
    const float g = gamma[c], b = beta[c];
    for (int s = tid >> 6; s < Ls; s += 4) xn[((size_t)n * Ls + s) * 64 + c] = fmaf((tile[c * P + s] - mu[s]) * rs[s], g, b);
}

// out = y[n*Ls + s][c] + bias[c] + x  (back to the (B, 64, R, Ls) layout)
__global__ __launch_bounds__(256) void dp_out_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                     const float* __restrict__ x, float* __restrict__ out, int R, int Ls) {
    extern __shared__ float tile[];  // [Ls][65]
    const int n = blockIdx.x, tid = threadIdx.x;
    const size_t base = seq_base(n, R, Ls), cs = (size_t)R * Ls;
    for (int idx = tid; idx < 64 * Ls; idx += 256) tile[(idx >> 6) * 65 + (idx & 63)] = y[(size_t)n * Ls * 64 + idx] + bias[idx & 63];
    __syncthreads();
    for (int idx = tid; idx < 64 * Ls; idx += 256) {
        const int c = idx / Ls, s = idx - c * Ls;
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

namespace {
template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return RTFS_ERR_SHAPE;
    if (bytes > 48 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return RTFS_ERR_LAUNCH;
    return RTFS_OK;
}
}  // namespace

int launch_dp_ln_fwd(const float* x, const float* gamma, const float* beta, float* xn, int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8 || Ls > 256) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)64 * (Ls + 1) * sizeof(float);
    int rc = set_lds(dp_ln_fwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_ln_fwd_kernel, dim3(nseq), dim3(256), lds, st, x, gamma, beta, xn, R, Ls);
    return rtfs_launch_status();
}
int launch_dp_out(const float* y, const float* bias, const float* x, float* out, int nseq, int R, int Ls, hipStream_t st) {
    if (Ls < 8 || Ls > 256) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)Ls * 65 * sizeof(float);
    int rc = set_lds(dp_out_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_out_kernel, dim3(nseq), dim3(256), lds, st, y, bias, x, out, R, Ls);
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
__global__ __launch_bounds__(256) void cl_norm_act_fwd_kernel(ClStageArgs a) {
    const int b = blockIdx.y;
    float mean = 0.f, rstd = 1.f;
    if (a.norm == 1) stats_of(a.stats, b, 1.0 / (double)a.n, mean, rstd);
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const size_t base = (size_t)b * a.n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & (a.C - 1));
        float z = a.x[base + i];
        norm_of(a, b, c, mean, rstd);
        if (a.norm) z = fmaf((z - mean) * rstd, a.gamma[c], a.beta[c]);
        a.y[base + i] = act_fwd(z, a.act, slope);
    }
}

// reductions of the stage's backward: per sample S1 = sum da*gamma, S2 = sum da*gamma*xhat (f64 atomics into S[2b..], gLN only);
// per channel dgamma += sum da*xhat, dbeta += sum da; dslope.  The grid stride is a multiple of C, so a thread keeps one channel.
__global__ __launch_bounds__(256) void cl_norm_act_bwd_reduce_kernel(ClStageArgs a) {
    __shared__ double red[16];
    __shared__ float part[3][256];
    const int b = blockIdx.y, tid = threadIdx.x;
    float mean = 0.f, rstd = 1.f;
    if (a.norm == 1) stats_of(a.stats, b, 1.0 / (double)a.n, mean, rstd);
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const size_t base = (size_t)b * a.n;
    const int c = (int)(((size_t)blockIdx.x * 256 + tid) & (a.C - 1));
    norm_of(a, b, c, mean, rstd);
    const float g = a.norm ? a.gamma[c] : 1.f, be = a.norm ? a.beta[c] : 0.f;
    float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f, dsl = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < a.n; i += (size_t)gridDim.x * 256) {
        const float xv = a.x[base + i];
        const float xh = a.norm ? (xv - mean) * rstd : xv;
        const float z = a.norm ? fmaf(xh, g, be) : xv;
        const float da = act_bwd(z, a.dy[base + i], a.act, slope, dsl);
        dg = fmaf(da, xh, dg);
        db += da;
        s1 = fmaf(da, g, s1);
        s2 = fmaf(da * g, xh, s2);
    }
    if (a.norm == 1) block_stats_atomic_pair(s1, s2, red, a.S + 2 * b);
    part[0][tid] = dg;
    part[1][tid] = db;
    part[2][tid] = dsl;
    __syncthreads();
    if (a.norm) {
        if (a.C >= 256) {  // every thread of the workgroup has its own channel
            unsafeAtomicAdd(a.dgamma + c, dg);
            unsafeAtomicAdd(a.dbeta + c, db);
        } else if (tid < a.C) {
            float sg = 0.f, sb = 0.f;
            for (int j = tid; j < 256; j += a.C) {
                sg += part[0][j];
                sb += part[1][j];
            }
            unsafeAtomicAdd(a.dgamma + tid, sg);
            unsafeAtomicAdd(a.dbeta + tid, sb);
        }
    }
    if (a.act == 2 && tid < 64) {
        float v = part[2][tid] + part[2][tid + 64] + part[2][tid + 128] + part[2][tid + 192];
        v = wave_sum(v);
        if (tid == 0) unsafeAtomicAdd(a.dslope, v);
    }
}

__global__ __launch_bounds__(256) void cl_norm_act_bwd_apply_kernel(ClStageArgs a) {
    const int b = blockIdx.y;
    float mean = 0.f, rstd = 1.f, m1 = 0.f, m2 = 0.f;
    if (a.norm == 1) {
        stats_of(a.stats, b, 1.0 / (double)a.n, mean, rstd);
        m1 = (float)(a.S[2 * b] / (double)a.n);
        m2 = (float)(a.S[2 * b + 1] / (double)a.n);
    }
    const float slope = a.act == 2 ? a.slope[0] : 0.f;
    const size_t base = (size_t)b * a.n;
    float dummy = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & (a.C - 1));
        const float xv = a.x[base + i];
        norm_of(a, b, c, mean, rstd);
        const float g = a.norm ? a.gamma[c] : 1.f;
        const float xh = a.norm ? (xv - mean) * rstd : xv;
        const float z = a.norm ? fmaf(xh, g, a.beta[c]) : xv;
        const float da = act_bwd(z, a.dy[base + i], a.act, slope, dummy);
        float out = da;
        if (a.norm == 1) out = rstd * (da * g - m1 - xh * m2);
        else if (a.norm == 2) out = da * g * rstd;
        else if (a.norm == 3) out = g * rstd * (da - (float)a.inv_rows * (a.dbeta[c] + xh * a.dgamma[c]));
        a.dx[base + i] = out;
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

// out[c] += sum over rows of d[row][c]   (bias gradients); grid stride a multiple of C
__global__ __launch_bounds__(256) void cl_colsum_kernel(const float* __restrict__ d, float* __restrict__ out, size_t n, int C) {
    __shared__ float part[256];
    const int tid = threadIdx.x;
    float s = 0.f;
#pragma unroll 8
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) s += d[i];
    if (C >= 256) {
        unsafeAtomicAdd(out + (((size_t)blockIdx.x * 256 + tid) & (C - 1)), s);
        return;
    }
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float v = 0.f;
        for (int j = tid; j < 256; j += C) v += part[j];
        unsafeAtomicAdd(out + tid, v);
    }
}

// depthwise k x k convolution, channel-last: x (B, H, W, C), w (C, kh*kw), y (B, Ho, Wo, C); cross-correlation with
// top/left padding (pt, pl) and stride s (conv_layers.py:100-101: "same" k = 4 -> pt = pl = 1, stride 2 -> symmetric 1)
__global__ __launch_bounds__(256) void cl_dw_fwd_kernel(ClDwArgs a) {
    const unsigned total = (unsigned)a.B * a.Ho * a.Wo * a.C;  // < 2^31 (launcher)
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
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
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
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

// input gradient: dx[b,h,w,c] = sum over taps with (h + pt - ki) = ho*s, (w + pl - kj) = wo*s of w[c,ki,kj] * dy[b,ho,wo,c]
// (conditional loads on purpose: this kernel serves the stride-2 case, where three taps in four fail the parity test - loading them
// unconditionally costs 5x the traffic: 141 vs 50 us)
__global__ __launch_bounds__(256) void cl_dw_bwd_data_kernel(ClDwArgs a) {
    const unsigned total = (unsigned)a.B * a.H * a.W * a.C;  // < 2^31 (launcher)
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
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
    for (unsigned r = blockIdx.x * lanes + rl; r < rows; r += gridDim.x * lanes) {
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

__global__ __launch_bounds__(256) void cl_dw_wgrad_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int nwg, int taps,
                                                                 int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // (tap, c)
    if (i >= taps * C) return;
    float v = 0.f;
    for (int w = blockIdx.y; w < nwg; w += gridDim.y) v += scratch[(size_t)w * taps * C + i];
    unsafeAtomicAdd(dw + (i % C) * taps + i / C, v);
}

namespace {
inline unsigned grid_for(size_t n, unsigned cap = 8192) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

namespace {
inline bool cl_c_ok(int C) { return C >= 1 && C <= 1024 && !(C & (C - 1)); }
inline unsigned grid4(size_t n, unsigned cap) { return (grid_for(n, cap) + 3) / 4 * 4; }  // stride (grid * 256) % C == 0 for C <= 1024
}  // namespace
int launch_cl_norm_act_fwd(const ClStageArgs& a, int B, hipStream_t st) {
    if (!cl_c_ok(a.C)) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(cl_norm_act_fwd_kernel, dim3(grid_for(a.n, 2048), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}
// part 0: reduction + apply; 1: reduction only; 2: apply only (SyncBatchNorm all-reduces dgamma / dbeta in between)
int launch_cl_norm_act_bwd(const ClStageArgs& a, int B, hipStream_t st, int part) {
    if (!cl_c_ok(a.C)) return RTFS_ERR_SHAPE;
    if (part != 2 && (a.norm || a.act == 2)) {
        if (a.norm == 1 && hipMemsetAsync(a.S, 0, sizeof(double) * 2 * B, st) != hipSuccess) return RTFS_ERR_LAUNCH;
        hipLaunchKernelGGL(cl_norm_act_bwd_reduce_kernel, dim3(grid4(a.n, 256), B), dim3(256), 0, st, a);
    }
    if (part != 1) hipLaunchKernelGGL(cl_norm_act_bwd_apply_kernel, dim3(grid_for(a.n, 2048), B), dim3(256), 0, st, a);
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
int launch_cl_colsum(const float* d, float* out, size_t n, int C, hipStream_t st) {
    if (C >= 1 && !cl_c_ok(C)) {
        const size_t rows = n / C;
        const int chunk = 64;
        hipLaunchKernelGGL(cl_colsum_any_kernel, dim3((unsigned)((rows + chunk - 1) / chunk)), dim3(256), 0, st, d, out, rows, C, chunk);
        return rtfs_launch_status();
    }
    if (!cl_c_ok(C)) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(cl_colsum_kernel, dim3(grid4(n, 1024)), dim3(256), 0, st, d, out, n, C);
    return rtfs_launch_status();
}
namespace {
int launch_cl_dw_chunk(const ClDwArgs& a, int what, hipStream_t st) {
    const bool w4 = a.s == 1 && a.W >= 4 && a.Ho == a.H && a.Wo == a.W;
    const size_t n4 = (size_t)a.B * a.H * ((a.W + 3) / 4) * a.C;
    if (what == 0 && w4) hipLaunchKernelGGL(cl_dw_s1_w4_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, st, a);
    else if (what == 1 && w4) hipLaunchKernelGGL(cl_dw_s1_w4_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, st, a);
    else if (what == 0) hipLaunchKernelGGL(cl_dw_fwd_kernel, dim3(grid_for((size_t)a.B * a.Ho * a.Wo * a.C)), dim3(256), 0, st, a);
    else if (what == 1) hipLaunchKernelGGL(cl_dw_bwd_data_kernel, dim3(grid_for((size_t)a.B * a.H * a.W * a.C)), dim3(256), 0, st, a);
    else {
        const size_t rows = (size_t)a.B * a.Ho * a.Wo, per_wg = (size_t)(256 / a.C) * 8;
        size_t g = (rows + per_wg - 1) / per_wg;
        g = g < 1 ? 1 : (g > CL_DW_WGRAD_MAX_WG ? CL_DW_WGRAD_MAX_WG : g);
        if (!a.scratch) return RTFS_ERR_WORKSPACE;
        hipLaunchKernelGGL(cl_dw_wgrad_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
        hipLaunchKernelGGL(cl_dw_wgrad_reduce_kernel, dim3(cdiv(a.kh * a.kw * a.C, 256), g >= 64 ? 64 : (unsigned)g), dim3(256), 0, st,
                           a.scratch, a.dw, (int)g, a.kh * a.kw, a.C);
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

// ------------------------------------------------------------------------------------------------ TF attention, training side
#define LNG_BT 1  // (b,t) slices per workgroup of the LNG backward
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

__global__ __launch_bounds__(256) void att_lng_bwd_kernel(LngArgs a) {
    extern __shared__ float lds[];  // A [64][CZ+1] (activated input, later xhat), D [64][CZ+1] (gamma * dY)
    __shared__ float g1[16], g2[16], gsl[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, CZ = a.CZ, P = CZ + 1;
    float* A = lds;
    float* D = lds + 64 * P;
    // a workgroup walks LNG_BT consecutive (b,t) slices; each thread owns the same (f, c) elements in every slice, so the
    // (C, F) affine's gradients stay in registers and reach HBM as one atomic per element per workgroup
    float acc_g[32], acc_b[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc_g[k] = acc_b[k] = 0.f;
    if (tid < 16) gsl[tid] = 0.f;
    const int per = 64 * CZ / 256;  // 16 (CZ 64) or 32 (CZ 128) elements per thread
    for (int bt = blockIdx.x * LNG_BT; bt < min(a.nbt, (blockIdx.x + 1) * LNG_BT); ++bt) {
        const float* z = a.Z + (size_t)bt * 64 * CZ;
        const float* dy = a.dY + (size_t)bt * 64 * CZ;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k < per) {
                const int idx = tid + 256 * k;
                const int f = idx / CZ, c = idx - f * CZ, g = a.gof[c];
                const float v = z[idx];
                const float act = v >= 0.f ? v : a.slope[c] * v;
                float xh = 0.f, gd = 0.f;
                if (g < 16) {
                    xh = (act - a.stats[((size_t)bt * 16 + g) * 2]) * a.stats[((size_t)bt * 16 + g) * 2 + 1];
                    const float d = dy[idx];
                    gd = a.gamma[c * 64 + f] * d;
                    acc_g[k] = fmaf(d, xh, acc_g[k]);
                    acc_b[k] += d;
                }
                A[f * P + c] = xh;
                D[f * P + c] = gd;
            }
        }
        __syncthreads();
        for (int g = wave; g < a.ngroups; g += 4) {
            const int c0 = a.gstart[g], gs = a.gstart[g + 1] - c0, n = 64 * gs;
            float s1 = 0.f, s2 = 0.f;
            for (int i = lane; i < n; i += 64) {
                const int o = (i / gs) * P + c0 + i % gs;
                s1 += D[o];
                s2 = fmaf(D[o], A[o], s2);
            }
            s1 = wave_sum(s1) / n;
            s2 = wave_sum(s2) / n;
            if (lane == 0) {
                g1[g] = s1;
                g2[g] = s2;
            }
        }
        __syncthreads();
        float* dz = a.dZ + (size_t)bt * 64 * CZ;
        for (int idx = tid; idx < 64 * CZ; idx += 256) {
            const int f = idx / CZ, c = idx - f * CZ, g = a.gof[c];
            float out = 0.f;
            if (g < 16) {
                const float rstd = a.stats[((size_t)bt * 16 + g) * 2 + 1];
                const float dA = rstd * (D[f * P + c] - g1[g] - A[f * P + c] * g2[g]);
                const float v = z[idx];
                if (v >= 0.f) out = dA;
                else {
                    out = dA * a.slope[c];
                    atomicAdd(&gsl[g], dA * v);
                }
            }
            dz[idx] = out;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k < per) {
            const int idx = tid + 256 * k;
            const int f = idx / CZ, c = idx - f * CZ;
            // per-workgroup partial sums; att_lng_reduce_kernel adds them up (atomics from every workgroup onto the (C, F) affine's
            // 2 x 8192 addresses serialise: 346 -> 60 us)
            float* sc = a.scratch + (size_t)blockIdx.x * 2 * CZ * 64;
            sc[c * 64 + f] = a.gof[c] < 16 ? acc_g[k] : 0.f;
            sc[CZ * 64 + c * 64 + f] = a.gof[c] < 16 ? acc_b[k] : 0.f;
        }
    }
    if (tid < a.ngroups && gsl[tid] != 0.f) unsafeAtomicAdd(a.dslope + tid, gsl[tid]);
}

__global__ __launch_bounds__(256) void att_lng_reduce_kernel(const float* __restrict__ scratch, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int nwg, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // element of the (CZ, 64) affine
    if (i >= n) return;
    float g = 0.f, b = 0.f;
    for (int w = blockIdx.y; w < nwg; w += gridDim.y) {
        g += scratch[(size_t)w * 2 * n + i];
        b += scratch[(size_t)w * 2 * n + n + i];
    }
    unsafeAtomicAdd(dgamma + i, g);
    unsafeAtomicAdd(dbeta + i, b);
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

size_t att_lng_scratch_floats(int nbt) { return (size_t)cdiv(nbt, LNG_BT) * 2 * 128 * 64; }
int launch_att_lng(const LngArgs& a, int nbt, bool bwd, hipStream_t st) {
    if (a.CZ != 64 && a.CZ != 128) return RTFS_ERR_SHAPE;
    const size_t lds = (size_t)(bwd ? 2 : 1) * 64 * (a.CZ + 1) * sizeof(float);
    int rc = bwd ? set_lds(att_lng_bwd_kernel, lds) : set_lds(att_lng_fwd_kernel, lds);
    if (rc) return rc;
    if (bwd) {
        LngArgs b = a;
        b.nbt = nbt;
        if (!b.scratch) return RTFS_ERR_WORKSPACE;
        const int nwg = cdiv(nbt, LNG_BT);
        hipLaunchKernelGGL(att_lng_bwd_kernel, dim3(nwg), dim3(256), lds, st, b);
        hipLaunchKernelGGL(att_lng_reduce_kernel, dim3(cdiv(a.CZ * 64, 256), nwg >= 16 ? 16 : nwg), dim3(256), 0, st, b.scratch, a.dgamma, a.dbeta, nwg,
                           a.CZ * 64);
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
    if (T < 1 || Tp > 256) return RTFS_ERR_SHAPE;
    const size_t rows = (size_t)nbatch * T;
    hipLaunchKernelGGL(att_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, rows, T, Tp, scale, bwd ? 1 : 0);
    return rtfs_launch_status();
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
__global__ __launch_bounds__(256) void pool2d_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t N, int H, int W, int Ho,
                                                         int Wo, int C) {
    const size_t total = N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int w = (int)(p % W), h = (int)((p / W) % H);
        const size_t n = p / ((size_t)W * H);
        // candidate windows around floor(h * out / in): window starts are non-decreasing and each is at most in/out + 1 long
        const int hc = (int)(((long)h * Ho) / H), wc = (int)(((long)w * Wo) / W);
        float s = 0.f;
        for (int ho = max(hc - 1, 0); ho <= min(hc + 1, Ho - 1); ++ho) {
            const int h0 = win_lo(ho, H, Ho), h1 = win_hi(ho, H, Ho);
            if (h < h0 || h >= h1) continue;
            for (int wo = max(wc - 1, 0); wo <= min(wc + 1, Wo - 1); ++wo) {
                const int w0 = win_lo(wo, W, Wo), w1 = win_hi(wo, W, Wo);
                if (w < w0 || w >= w1) continue;
                s += dy[((n * Ho + ho) * Wo + wo) * C + c] / (float)((h1 - h0) * (w1 - w0));
            }
        }
        dx[i] = s;
    }
}
// InjectionMultiSum's last line (fusion.py:54-69): out = local * up(gate) + up(global), up = F.interpolate(mode="nearest")
__global__ __launch_bounds__(256) void tfar_combine_fwd_kernel(const float* __restrict__ le, const float* __restrict__ gate,
                                                               const float* __restrict__ ge, float* __restrict__ out, size_t N, int H, int W,
                                                               int Hg, int Wg, int C) {
    const size_t total = N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int w = (int)(p % W), h = (int)((p / W) % H);
        const size_t n = p / ((size_t)W * H);
        const size_t j = ((n * Hg + nearest_src(h, Hg, H)) * Wg + nearest_src(w, Wg, W)) * C + c;
        out[i] = fmaf(le[i], gate[j], ge[j]);
    }
}
__global__ __launch_bounds__(256) void tfar_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ le,
                                                               const float* __restrict__ gate, float* __restrict__ dle,
                                                               float* __restrict__ dgate, float* __restrict__ dge, size_t N, int H, int W, int Hg,
                                                               int Wg, int C) {
    // one thread per GLOBAL element: it owns the local pixels that read it (a contiguous block of rows x columns)
    const size_t total = N * Hg * Wg * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int wg = (int)(p % Wg), hg = (int)((p / Wg) % Hg);
        const size_t n = p / ((size_t)Wg * Hg);
        // local h reads global floor(h * Hg / H) == hg  <=>  h in [ceil(hg*H/Hg), ceil((hg+1)*H/Hg))
        const int h0 = (int)(((long)hg * H + Hg - 1) / Hg), h1 = min(H, (int)(((long)(hg + 1) * H + Hg - 1) / Hg));
        const int w0 = (int)(((long)wg * W + Wg - 1) / Wg), w1 = min(W, (int)(((long)(wg + 1) * W + Wg - 1) / Wg));
        const float g = gate[i];
        float sg = 0.f, se = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const size_t k = ((n * H + h) * W + w) * C + c;
                const float d = dout[k];
                dle[k] = d * g;
                sg = fmaf(d, le[k], sg);
                se += d;
            }
        dgate[i] = sg;
        dge[i] = se;
    }
}
int launch_pool2d(const float* x, float* y, size_t N, int H, int W, int Ho, int Wo, bool bwd, hipStream_t st, int C) {
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1 || Ho > H || Wo > W || C < 1) return RTFS_ERR_SHAPE;
    if (bwd) hipLaunchKernelGGL(pool2d_bwd_kernel, dim3(grid_for(N * H * W * C)), dim3(256), 0, st, x, y, N, H, W, Ho, Wo, C);
    else hipLaunchKernelGGL(pool2d_fwd_kernel, dim3(grid_for(N * Ho * Wo * C)), dim3(256), 0, st, x, y, N, H, W, Ho, Wo, C);
    return rtfs_launch_status();
}
int launch_tfar_combine(const float* le, const float* gate, const float* ge, float* out, size_t N, int H, int W, int Hg, int Wg, hipStream_t st,
                        int C) {
    if (Hg < 1 || Wg < 1 || Hg > H || Wg > W || C < 1) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(tfar_combine_fwd_kernel, dim3(grid_for(N * H * W * C)), dim3(256), 0, st, le, gate, ge, out, N, H, W, Hg, Wg, C);
    return rtfs_launch_status();
}
int launch_tfar_combine_bwd(const float* dout, const float* le, const float* gate, float* dle, float* dgate, float* dge, size_t N, int H, int W,
                            int Hg, int Wg, hipStream_t st, int C) {
    if (Hg < 1 || Wg < 1 || Hg > H || Wg > W || C < 1) return RTFS_ERR_SHAPE;
    hipLaunchKernelGGL(tfar_combine_bwd_kernel, dim3(grid_for(N * Hg * Wg * C)), dim3(256), 0, st, dout, le, gate, dle, dgate, dge, N, H, W, Hg, Wg, C);
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

// ------------------------------------------------------------------------------------------------ LSTM cell, training side
// nn.LSTM(512, 32, 4 layers, bidirectional) as DualPathRNN's other cell (rnn_layers.py:116-122); gates i, f, g, o.
// U = x . W_ih^T + (b_ih + b_hh) for both directions comes from the GEMM as rows x 256 (column dir*128 + gate*32 + j).  One wave per
// (sequence, direction): lane l < 32 owns gate rows i_j and g_j (j = l), lane l >= 32 rows f_j and o_j (j = l - 32), each with its
// 2 x 32 recurrent weights in registers; h_{t-1} is broadcast through LDS.  Saved for the backward: the four activated gates G
// (rows x 256), c, h (in the zero-padded slot layout the windows read) and h_{t-1} (rows x 64, for the W_hh gradient GEMM).
namespace {
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * sigmoidf_(2.0f * x) - 1.0f; }
}
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
