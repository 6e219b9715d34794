// Depthwise 4x4 convolution family + the G-level elementwise glue of the RTFS block.
// Reference modules restated here:
//   downsample_layers[0] (dw 4x4 s1 'same' +bias -> gLN)          separators/tdanet.py:59-74,110
//   downsample_layers[1] (dw 4x4 s2 pad 1 +bias -> gLN)           separators/tdanet.py:111-112
//   adaptive_avg_pool2d sum                                        separators/tdanet.py:115-116
//   InjectionMultiSum (TFAR): three dw 4x4 'same' no-bias + gLN    layers/fusion.py:24-69
// gLN (GroupNorm(1,C)) needs per-sample statistics of a conv's whole output, so every conv here
// writes (or only accumulates) its pre-norm output plus (sum, sumsq) in f64; the consumer folds
// mean/rstd/gamma/beta into one FMA at load time (gln_fold).  Zero padding is applied to the
// *normalised* input, i.e. out-of-image taps contribute 0, not `shift`.
#include "common.h"
#include "kernels.h"

// ---------------------------------------------------------------- stride-1 'same' 4x4 (pad lo 1, hi 2)
// One thread per (channel, column) walks a band of TH rows with a 4x4 register window (no LDS: the four
// overlapping row segments of neighbouring lanes are served by L1, HBM sees each element once per band).
// MODE 0: write pre-norm outputs + stats.  MODE 1: stats only.  MODE 2: TFAR apply:
//   out = gLN_loc(conv(x)) * sigmoid(gLN_gate(G_gate)^) + gLN_emb(G_emb)^ [+ gLN_add(addend)]
// where ^ is legacy nearest up-sampling from (Hg, Wg) to (H, W).
template <int NCONV, bool IN_AFFINE, int MODE>
__device__ __forceinline__ void dw_s1_body(const DwArgs& a, const float* __restrict__ X, const float* __restrict__ GATE,
                                           const float* __restrict__ EMB, const float* __restrict__ ADD, float* __restrict__ O0,
                                           float* __restrict__ O1, float* __restrict__ O2, float* __restrict__ O3) {
    __shared__ double red[8];
    const int H = a.H, W = a.W, C = a.C;
    const int b = blockIdx.z;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < C * W;
    const int c = live ? g / W : 0, f = live ? g - c * W : 0;
    const int r0 = blockIdx.y * a.TH, r1 = min(r0 + a.TH, H);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* __restrict__ xp = X + plane;
    float isc = 1.f, ish = 0.f;
    if (IN_AFFINE) gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    float wgt[NCONV][16], bia[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wgt[n][j] = a.w[n][c * 16 + j];
        bia[n] = a.bias[n] ? a.bias[n][c] : 0.f;
    }
    float lsc = 1.f, lsh = 0.f, gsc = 1.f, gsh = 0.f, esc = 1.f, esh = 0.f, asc = 1.f, ash = 0.f;
    size_t gplane = 0;
    int fg = 0;
    if (MODE == 2) {
        gln_fold(a.loc_stats + 2 * b, a.loc_inv_count, a.loc_gamma[c], a.loc_beta[c], lsc, lsh);
        gln_fold(a.gate_stats + 2 * b, a.g_inv_count, a.gate_gamma[c], a.gate_beta[c], gsc, gsh);
        gln_fold(a.emb_stats + 2 * b, a.g_inv_count, a.emb_gamma[c], a.emb_beta[c], esc, esh);
        if (a.addend) gln_fold(a.add_stats + 2 * b, a.add_inv_count, a.add_gamma[c], a.add_beta[c], asc, ash);
        gplane = ((size_t)b * C + c) * a.Hg * a.Wg;
        fg = nearest_src(f, a.Wg, W);
    }
    bool fok[4];
    int fcl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ff = f - 1 + j;
        fok[j] = live && ff >= 0 && ff < W;
        fcl[j] = ff < 0 ? 0 : (ff < W ? ff : W - 1);
    }
    // unconditional loads from clamped addresses, zero-padding applied by select (no branch + wait per tap)
    auto load_row = [&](int t, float (&row)[4]) {
        const bool tok = t >= 0 && t < H;
        const float* __restrict__ rp = xp + (size_t)(t < 0 ? 0 : (t < H ? t : H - 1)) * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = rp[fcl[j]];
            if (IN_AFFINE) v = fmaf(v, isc, ish);
            row[j] = (tok && fok[j]) ? v : 0.f;
        }
    };
    float win[4][4];
    load_row(r0 - 1, win[0]);
    load_row(r0, win[1]);
    load_row(r0 + 1, win[2]);
    float s[NCONV], ss[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) s[n] = ss[n] = 0.f;
#pragma unroll 4
    for (int t = r0; t < r1; ++t) {
        load_row(t + 2, win[3]);
        const size_t o = plane + (size_t)t * W + f;
#pragma unroll
        for (int n = 0; n < NCONV; ++n) {
            float acc = bia[n];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = fmaf(win[i][j], wgt[n][i * 4 + j], acc);
            if (live) {
                if (MODE == 0) (n == 0 ? O0 : n == 1 ? O1 : n == 2 ? O2 : O3)[o] = acc;
                if (MODE != 2) {
                    s[n] += acc;
                    ss[n] = fmaf(acc, acc, ss[n]);
                } else {
                    const int tg = nearest_src(t, a.Hg, H);
                    const size_t go = gplane + (size_t)tg * a.Wg + fg;
                    const float gate = sigmoidf_(fmaf(GATE[go], gsc, gsh));
                    const float emb = fmaf(EMB[go], esc, esh);
                    float y = fmaf(fmaf(acc, lsc, lsh), gate, emb);
                    if (ADD) y += fmaf(ADD[o], asc, ash);
                    O0[o] = y;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) win[i][j] = win[i + 1][j];
    }
    if (MODE != 2) {
#pragma unroll
        for (int n = 0; n < NCONV; ++n) {
            block_stats_atomic(s[n], ss[n], red, a.stats_out[n] + 2 * b);
            __syncthreads();
        }
    }
}

template <int NCONV, bool IN_AFFINE, int MODE>
__global__ __launch_bounds__(256) void dw_s1_kernel(DwArgs a) {
    dw_s1_body<NCONV, IN_AFFINE, MODE>(a, a.x, a.gate, a.emb, a.addend, a.out[0], a.out[1], a.out[2], a.out[3]);
}

// ---------------------------------------------------------------- packed variant for the full-resolution single-conv passes
// The depthwise passes are VALU-bound (16 FMAs per output), so this variant halves the FMA instruction count with
// v_pk_fma_f32: a thread owns TWO columns half a row apart (f and f + ceil(W/2)); their 4x4 windows are disjoint, so each
// load fills one half of an operand pair and both columns' accumulators advance with one packed FMA per tap.
// Zero padding: out-of-image COLUMNS are folded into per-lane weights (weight 0), out-of-image ROWS are selected to 0 only in
// the first / last row band; the input gLN fold is applied to the result: conv(pad0(s*x+b)) = s*conv(pad0(x)) + b*sum(valid w).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NCONV, bool IN_AFFINE, int MODE>
__device__ __forceinline__ void dw1p_body(const DwArgs& a, const float* __restrict__ X, const float* __restrict__ GATE,
                                          const float* __restrict__ EMB, const float* __restrict__ ADD, float* __restrict__ OUT,
                                          float* __restrict__ OUT1, float* __restrict__ OUT2, float* __restrict__ OUT3) {
    static_assert(NCONV == 1 || (MODE == 0 && !IN_AFFINE), "multi-conv: plain write + stats only");
    __shared__ double red[8];
    const int H = a.H, W = a.W, C = a.C;
    const int half = (W + 1) >> 1;
    const int b = blockIdx.z;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < C * half;
    const int c = live ? g / half : 0, fa = live ? g - c * half : 0, fb = fa + half;
    const bool liveb = live && fb < W;
    const int r0 = blockIdx.y * a.TH, r1 = min(r0 + a.TH, H);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* __restrict__ xp = X + plane;
    float isc = 1.f, ish = 0.f;
    if (IN_AFFINE) gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    // per-lane weights with the column padding folded in; clamped column offsets
    f32x2 wgt[NCONV][16];
    int ca[4], cb[4];
    f32x2 rowsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowsum[i] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xa = fa - 1 + j, xb = fb - 1 + j;
        const bool oka = live && xa >= 0 && xa < W, okb = liveb && xb >= 0 && xb < W;
        ca[j] = xa < 0 ? 0 : (xa < W ? xa : W - 1);
        cb[j] = xb < 0 ? 0 : (xb < W ? xb : W - 1);
#pragma unroll
        for (int n = 0; n < NCONV; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float w = a.w[n][c * 16 + i * 4 + j];
                wgt[n][i * 4 + j] = f32x2{oka ? w : 0.f, okb ? w : 0.f};
                if (n == 0) rowsum[i] += wgt[0][i * 4 + j];
            }
    }
    float bias[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) bias[n] = a.bias[n] ? a.bias[n][c] : 0.f;
    float lsc = 1.f, lsh = 0.f, gsc = 1.f, gsh = 0.f, esc = 1.f, esh = 0.f, asc = 1.f, ash = 0.f;
    size_t gplane = 0;
    int fga = 0, fgb = 0;
    if (MODE == 2) {
        gln_fold(a.loc_stats + 2 * b, a.loc_inv_count, a.loc_gamma[c], a.loc_beta[c], lsc, lsh);
        gln_fold(a.gate_stats + 2 * b, a.g_inv_count, a.gate_gamma[c], a.gate_beta[c], gsc, gsh);
        gln_fold(a.emb_stats + 2 * b, a.g_inv_count, a.emb_gamma[c], a.emb_beta[c], esc, esh);
        if (ADD) gln_fold(a.add_stats + 2 * b, a.add_inv_count, a.add_gamma[c], a.add_beta[c], asc, ash);
        gplane = ((size_t)b * C + c) * a.Hg * a.Wg;
        fga = nearest_src(fa, a.Wg, W);
        fgb = nearest_src(fb < W ? fb : W - 1, a.Wg, W);
    }
    const bool border = r0 == 0 || r1 + 2 > H;  // block-uniform: only these bands ever see an out-of-image row
    auto load_row = [&](int t, f32x2 (&row)[4]) {
        const int tc = t < 0 ? 0 : (t < H ? t : H - 1);
        const float* __restrict__ rp = xp + (size_t)tc * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) row[j] = f32x2{rp[ca[j]], rp[cb[j]]};
        if (border && (t < 0 || t >= H)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) row[j] = f32x2{0.f, 0.f};
        }
    };
    f32x2 win[4][4];
    load_row(r0 - 1, win[0]);
    load_row(r0, win[1]);
    load_row(r0 + 1, win[2]);
    f32x2 s2[NCONV], ss2[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) s2[n] = ss2[n] = f32x2{0.f, 0.f};
    const f32x2 wv_full = rowsum[0] + rowsum[1] + rowsum[2] + rowsum[3];
    const int fbc = fb < W ? fb : W - 1;
    // MODE 2: low-resolution source row floor(t Hg / H) as an incremental quotient / remainder (a 64-bit division per row
    // on the scalar unit costs more than the row's vector work)
    int tgq = 0, tgr = 0;
    if (MODE == 2) {
        tgq = (int)(((long long)r0 * a.Hg) / H);
        tgr = (int)(((long long)r0 * a.Hg) % H);
    }
#pragma unroll 2
    for (int t = r0; t < r1; ++t) {
        load_row(t + 2, win[3]);
        const size_t oa = plane + (size_t)t * W + fa, ob = plane + (size_t)t * W + fbc;
        if (NCONV > 1) {  // several convolutions of the same input (G-level TFAR embeddings / gates): write + stats
            const f32x2 m = {live ? 1.f : 0.f, liveb ? 1.f : 0.f};
#pragma unroll
            for (int n = 0; n < NCONV; ++n) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = win[i][j] * wgt[n][i * 4 + j] + acc;
                acc += bias[n];
                float* __restrict__ o_ = n == 0 ? OUT : n == 1 ? OUT1 : n == 2 ? OUT2 : OUT3;
                if (live) o_[oa] = acc.x;
                if (liveb) o_[ob] = acc.y;
                const f32x2 am = acc * m;
                s2[n] += am;
                ss2[n] = am * am + ss2[n];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) win[i][j] = win[i + 1][j];
            continue;
        }
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = win[i][j] * wgt[0][i * 4 + j] + acc;
        if (IN_AFFINE) {
            f32x2 wv = wv_full;
            if (border) {
                if (t - 1 < 0) wv -= rowsum[0];
                if (t + 1 >= H) wv -= rowsum[2];
                if (t + 2 >= H) wv -= rowsum[3];
            }
            acc = acc * isc + wv * ish;
        }
        acc += bias[0];
        if (MODE == 0) {
            if (live) OUT[oa] = acc.x;
            if (liveb) OUT[ob] = acc.y;
        }
        if (MODE != 2) {
            const f32x2 m = {live ? 1.f : 0.f, liveb ? 1.f : 0.f};
            const f32x2 am = acc * m;
            s2[0] += am;
            ss2[0] = am * am + ss2[0];
        } else {
            const int tg = tgq < a.Hg - 1 ? tgq : a.Hg - 1;
            tgr += a.Hg;
            while (tgr >= H) {  // one step when Hg <= H
                tgr -= H;
                ++tgq;
            }
            const size_t ga = gplane + (size_t)tg * a.Wg + fga, gb = gplane + (size_t)tg * a.Wg + fgb;
            const f32x2 gate = {sigmoidf_(fmaf(GATE[ga], gsc, gsh)), sigmoidf_(fmaf(GATE[gb], gsc, gsh))};
            const f32x2 emb = {fmaf(EMB[ga], esc, esh), fmaf(EMB[gb], esc, esh)};
            f32x2 y = (acc * lsc + lsh) * gate + emb;
            if (ADD) y += f32x2{fmaf(ADD[oa], asc, ash), fmaf(ADD[ob], asc, ash)};
            if (live) OUT[oa] = y.x;
            if (liveb) OUT[ob] = y.y;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) win[i][j] = win[i + 1][j];
    }
    if (MODE != 2) {
#pragma unroll
        for (int n = 0; n < NCONV; ++n) {
            block_stats_atomic(s2[n].x + s2[n].y, ss2[n].x + ss2[n].y, red, a.stats_out[n] + 2 * b);
            __syncthreads();
        }
    }
}

template <int NCONV, bool IN_AFFINE, int MODE>
__global__ __launch_bounds__(256) void dw1p_kernel(DwArgs a) {
    dw1p_body<NCONV, IN_AFFINE, MODE>(a, a.x, a.gate, a.emb, a.addend, a.out[0], a.out[1], a.out[2], a.out[3]);
}

// ---------------------------------------------------------------- stride-2 pad-1 4x4 + adaptive average pool
// Reads d0 = gLN(c0) through the fold; writes c1 (pre-norm conv output, +stats) and p0 = adaptive_avg_pool2d(d0)
// at the conv's output resolution (Ho = H/2, Wo = W/2).  The pool window of output (i,j) is
// rows [floor(i*H/Ho), ceil((i+1)*H/Ho)) which always lies inside the conv window [2i-1, 2i+3).
// One thread per (channel, output column) walks TH output rows, keeping the two rows shared by
// consecutive windows in registers.
__device__ __forceinline__ void dw_s2_pool_body(const DwArgs& a, const float* __restrict__ X, float* __restrict__ O0, float* __restrict__ O1) {
    __shared__ double red[8];
    const int H = a.H, W = a.W, C = a.C, Ho = a.Hg, Wo = a.Wg;
    const int b = blockIdx.z;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < C * Wo;
    const int c = live ? g / Wo : 0, j = live ? g - c * Wo : 0;
    const int i0 = blockIdx.y * a.TH, i1 = min(i0 + a.TH, Ho);
    const size_t plane = ((size_t)b * C + c) * H * W;
    const float* __restrict__ xp = X + plane;
    float isc, ish;
    gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    float wgt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wgt[k] = a.w[0][c * 16 + k];
    const float bia = a.bias[0][c];
    const int fs = (int)(((long long)j * W) / Wo), fe = (int)(((long long)(j + 1) * W + Wo - 1) / Wo);
    bool fok[4], fpool[4];
    int fcl[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int f = 2 * j - 1 + d;
        fok[d] = live && f >= 0 && f < W;
        fpool[d] = f >= fs && f < fe;
        fcl[d] = f < 0 ? 0 : (f < W ? f : W - 1);
    }
    auto load_row = [&](int t, float (&row)[4]) {
        const bool tok = t >= 0 && t < H;
        const float* __restrict__ rp = xp + (size_t)(t < 0 ? 0 : (t < H ? t : H - 1)) * W;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float v = fmaf(rp[fcl[d]], isc, ish);
            row[d] = (tok && fok[d]) ? v : 0.f;
        }
    };
    float win[4][4];
    load_row(2 * i0 - 1, win[0]);
    load_row(2 * i0, win[1]);
    float s = 0.f, ss = 0.f;
    const size_t oplane = ((size_t)b * C + c) * Ho * Wo;
#pragma unroll 2
    for (int i = i0; i < i1; ++i) {
        load_row(2 * i + 1, win[2]);
        load_row(2 * i + 2, win[3]);
        const int ts = (int)(((long long)i * H) / Ho), te = (int)(((long long)(i + 1) * H + Ho - 1) / Ho);
        float acc = bia, pool = 0.f;
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            const int t = 2 * i - 1 + di;
            const bool tp = t >= ts && t < te;
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) {
                acc = fmaf(win[di][dj], wgt[di * 4 + dj], acc);
                if (tp && fpool[dj]) pool += win[di][dj];
            }
        }
        pool /= (float)((te - ts) * (fe - fs));
        if (live) {
            const size_t o = oplane + (size_t)i * Wo + j;
            O0[o] = acc;
            O1[o] = pool;
            s += acc;
            ss = fmaf(acc, acc, ss);
        }
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) {
            win[0][dj] = win[2][dj];
            win[1][dj] = win[3][dj];
        }
    }
    block_stats_atomic(s, ss, red, a.stats_out[0] + 2 * b);
}

__global__ __launch_bounds__(256) void dw_s2_pool_kernel(DwArgs a) { dw_s2_pool_body(a, a.x, a.out[0], a.out[1]); }

// Packed variant of the kernel above (used when Wo >= 16): a thread owns the two output columns (j, j + ceil(Wo/2)) of one
// channel, so the 16-tap window lives in f32x2 registers and every multiply-add is a v_pk_fma_f32.  The convolution runs on
// the RAW input (column padding folded into per-lane weight pairs, out-of-range rows zeroed by a uniform branch) and the
// input fold is applied once per output: conv(isc x + ish) = isc conv(x) + ish * (sum of in-bounds weights) + bias; the
// pool likewise as isc * mean(x) + ish with the column-masked row sums carried in the rolling window.  The pool window's
// row range is kept as an incremental quotient / remainder (no per-row integer division on the scalar unit).
__device__ __forceinline__ void dw_s2p_body(const DwArgs& a, const float* __restrict__ X, float* __restrict__ O0, float* __restrict__ O1) {
    __shared__ double red[8];
    const int H = a.H, W = a.W, C = a.C, Ho = a.Hg, Wo = a.Wg;
    const int half = (Wo + 1) >> 1;
    const int b = blockIdx.z;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const bool live = g < C * half;
    const int c = live ? g / half : 0, ja = live ? g - c * half : 0;
    const int jb = ja + half;
    const bool liveb = live && jb < Wo;
    const int jbc = liveb ? jb : ja;
    const int i0 = blockIdx.y * a.TH, i1 = min(i0 + a.TH, Ho);
    const float* __restrict__ xp = X + ((size_t)b * C + c) * H * W;
    float isc, ish;
    gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    const float bia = a.bias[0][c];
    const int fsa = (int)(((long long)ja * W) / Wo), fea = (int)(((long long)(ja + 1) * W + Wo - 1) / Wo);
    const int fsb = (int)(((long long)jbc * W) / Wo), feb = (int)(((long long)(jbc + 1) * W + Wo - 1) / Wo);
    int ca[4], cb[4];
    f32x2 wgt[16], pm[4], wrow[4];
#pragma unroll
    for (int di = 0; di < 4; ++di) wrow[di] = f32x2{0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int fa = 2 * ja - 1 + d, fb = 2 * jbc - 1 + d;
        const bool oka = fa >= 0 && fa < W, okb = fb >= 0 && fb < W;
        ca[d] = fa < 0 ? 0 : (fa < W ? fa : W - 1);
        cb[d] = fb < 0 ? 0 : (fb < W ? fb : W - 1);
        pm[d] = f32x2{fa >= fsa && fa < fea ? 1.f : 0.f, fb >= fsb && fb < feb ? 1.f : 0.f};
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            const float w = a.w[0][c * 16 + di * 4 + d];
            wgt[di * 4 + d] = f32x2{oka ? w : 0.f, okb ? w : 0.f};
            wrow[di] += wgt[di * 4 + d];
        }
    }
    const f32x2 wall = wrow[0] + wrow[1] + wrow[2] + wrow[3];
    const f32x2 nf = {(float)(fea - fsa), (float)(feb - fsb)};
    auto load_row = [&](int t, f32x2 (&row)[4], f32x2& rs) {
        const float* __restrict__ rp = xp + (size_t)(t < 0 ? 0 : (t < H ? t : H - 1)) * W;
#pragma unroll
        for (int d = 0; d < 4; ++d) row[d] = f32x2{rp[ca[d]], rp[cb[d]]};
        if (t < 0 || t >= H) {  // uniform: only the first / last band row of the image
#pragma unroll
            for (int d = 0; d < 4; ++d) row[d] = f32x2{0.f, 0.f};
        }
        rs = row[0] * pm[0];
#pragma unroll
        for (int d = 1; d < 4; ++d) rs = row[d] * pm[d] + rs;
    };
    f32x2 win[4][4], rsw[4];
    load_row(2 * i0 - 1, win[0], rsw[0]);
    load_row(2 * i0, win[1], rsw[1]);
    // pool rows of output i: [floor(i H / Ho), ceil((i+1) H / Ho)); (q, rem) = divmod(i H, Ho) kept incrementally
    int q = (int)(((long long)i0 * H) / Ho), rem = (int)(((long long)i0 * H) % Ho);
    f32x2 s2 = {0.f, 0.f}, ss2 = {0.f, 0.f};
    const f32x2 m = {live ? 1.f : 0.f, liveb ? 1.f : 0.f};
    const size_t oplane = ((size_t)b * C + c) * Ho * Wo;
#pragma unroll 2
    for (int i = i0; i < i1; ++i) {
        load_row(2 * i + 1, win[2], rsw[2]);
        load_row(2 * i + 2, win[3], rsw[3]);
        const int ts = q;
        rem += H;
        while (rem >= Ho) {
            rem -= Ho;
            ++q;
        }
        const int te = q + (rem > 0 ? 1 : 0);
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int di = 0; di < 4; ++di)
#pragma unroll
            for (int dj = 0; dj < 4; ++dj) acc = win[di][dj] * wgt[di * 4 + dj] + acc;
        f32x2 bsum = wall;
        if (2 * i - 1 < 0 || 2 * i + 2 >= H) {  // uniform: band touches the top / bottom padding
            bsum = f32x2{0.f, 0.f};
#pragma unroll
            for (int di = 0; di < 4; ++di) {
                const int t = 2 * i - 1 + di;
                if (t >= 0 && t < H) bsum += wrow[di];
            }
        }
        acc = acc * isc + (bsum * ish + bia);
        f32x2 ps = {0.f, 0.f};
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            const int t = 2 * i - 1 + di;
            if (t >= ts && t < te) ps += rsw[di];  // uniform
        }
        const f32x2 cnt = nf * (float)(te - ts);
        const f32x2 pool = f32x2{ps.x / cnt.x, ps.y / cnt.y} * isc + ish;
        const size_t oa = oplane + (size_t)i * Wo + ja, ob = oplane + (size_t)i * Wo + jbc;
        if (live) {
            O0[oa] = acc.x;
            O1[oa] = pool.x;
        }
        if (liveb) {
            O0[ob] = acc.y;
            O1[ob] = pool.y;
        }
        const f32x2 am = acc * m;
        s2 += am;
        ss2 = am * am + ss2;
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) {
            win[0][dj] = win[2][dj];
            win[1][dj] = win[3][dj];
        }
        rsw[0] = rsw[2];
        rsw[1] = rsw[3];
    }
    block_stats_atomic(s2.x + s2.y, ss2.x + ss2.y, red, a.stats_out[0] + 2 * b);
}
__global__ __launch_bounds__(256) void dw_s2p_kernel(DwArgs a) { dw_s2p_body(a, a.x, a.out[0], a.out[1]); }

// ---------------------------------------------------------------- G-level elementwise glue
// g = p0 + gLN(c1)           (global pooling sum, tdanet.py:116)
__global__ __launch_bounds__(256) void g_form_kernel(const float* __restrict__ p0, const float* __restrict__ c1,
                                                     const double* __restrict__ st1, double inv_count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ g, int C, int HW) {
    const int c = blockIdx.y, b = blockIdx.z;
    float sc, sh;
    gln_fold(st1 + 2 * b, inv_count, gamma[c], beta[c], sc, sh);
    const size_t base = ((size_t)b * C + c) * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) g[base + i] = p0[base + i] + fmaf(c1[base + i], sc, sh);
}

// xf1 = gLN(l) * sigmoid(gLN(gate)) + gLN(emb)     (same-size InjectionMultiSum, fusion.py:62-67)
__global__ __launch_bounds__(256) void g_combine_kernel(GCombineArgs a) {
    const int c = blockIdx.y, b = blockIdx.z;
    float lsc, lsh, gsc, gsh, esc, esh;
    gln_fold(a.l_stats + 2 * b, a.inv_count, a.l_gamma[c], a.l_beta[c], lsc, lsh);
    gln_fold(a.gate_stats + 2 * b, a.inv_count, a.gate_gamma[c], a.gate_beta[c], gsc, gsh);
    gln_fold(a.emb_stats + 2 * b, a.inv_count, a.emb_gamma[c], a.emb_beta[c], esc, esh);
    const size_t base = ((size_t)b * a.C + c) * a.HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.HW; i += gridDim.x * 256)
        a.out[base + i] = fmaf(fmaf(a.l[base + i], lsc, lsh), sigmoidf_(fmaf(a.gate[base + i], gsc, gsh)), fmaf(a.emb[base + i], esc, esh));
}

// (N, H, W) -> (N, W, H) through a padded 32x32 LDS tile.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W) {
    __shared__ float t[32][33];
    const size_t base = (size_t)blockIdx.z * H * W;
    const int w0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (h0 + r < H && w0 + tx < W) t[r][tx] = x[base + (size_t)(h0 + r) * W + w0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (w0 + r < W && h0 + tx < H) y[base + (size_t)(w0 + r) * H + h0 + tx] = t[tx][r];
}

// stats of an arbitrary (B, N) tensor (used when a module is called stand-alone)
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ x, double* __restrict__ stats, size_t N) {
    __shared__ double red[8];
    const int b = blockIdx.y;
    float s = 0.f, ss = 0.f;
    const float* xb = x + (size_t)b * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const float v = xb[i];
        s += v;
        ss = fmaf(v, v, ss);
    }
    block_stats_atomic(s, ss, red, stats + 2 * b);
}

// ---------------------------------------------------------------- launchers
template <int NCONV, bool IN_AFFINE, int MODE>
static int launch_dw_s1_t(const DwArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL((dw_s1_kernel<NCONV, IN_AFFINE, MODE>), dim3(cdiv(a.C * a.W, 256), cdiv(a.H, a.TH), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

template <int NCONV, bool IN_AFFINE, int MODE>
static int launch_dw1p_t(const DwArgs& a, int B, hipStream_t st) {
    const int half = (a.W + 1) / 2;
    hipLaunchKernelGGL((dw1p_kernel<NCONV, IN_AFFINE, MODE>), dim3(cdiv(a.C * half, 256), cdiv(a.H, a.TH), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_dw_s1(const DwArgs& a, int nconv, bool in_affine, int mode, int B, hipStream_t st) {
    if (a.W >= 16) {  // packed two-column variant (v_pk_fma_f32); the scalar kernel below only serves very narrow inputs
        if (nconv == 1) {
            if (mode == 0) return in_affine ? launch_dw1p_t<1, true, 0>(a, B, st) : launch_dw1p_t<1, false, 0>(a, B, st);
            if (mode == 1) return in_affine ? launch_dw1p_t<1, true, 1>(a, B, st) : launch_dw1p_t<1, false, 1>(a, B, st);
            if (mode == 2) return in_affine ? launch_dw1p_t<1, true, 2>(a, B, st) : launch_dw1p_t<1, false, 2>(a, B, st);
        } else if (mode == 0 && !in_affine) {
            if (nconv == 2) return launch_dw1p_t<2, false, 0>(a, B, st);
            if (nconv == 4) {  // two 2-conv launches: the 4-conv kernel needs 256 VGPRs and runs slower than both together
                DwArgs b = a;
                for (int i = 0; i < 2; ++i) {
                    b.w[i] = a.w[2 + i]; b.bias[i] = a.bias[2 + i]; b.out[i] = a.out[2 + i]; b.stats_out[i] = a.stats_out[2 + i];
                }
                const int rc = launch_dw1p_t<2, false, 0>(a, B, st);
                return rc ? rc : launch_dw1p_t<2, false, 0>(b, B, st);
            }
        }
    }
    if (mode == 0 && nconv == 1 && !in_affine) return launch_dw_s1_t<1, false, 0>(a, B, st);
    if (mode == 0 && nconv == 1 && in_affine) return launch_dw_s1_t<1, true, 0>(a, B, st);
    if (mode == 0 && nconv == 2 && !in_affine) return launch_dw_s1_t<2, false, 0>(a, B, st);
    if (mode == 0 && nconv == 4 && !in_affine) return launch_dw_s1_t<4, false, 0>(a, B, st);
    if (mode == 0 && nconv == 3 && !in_affine) return launch_dw_s1_t<3, false, 0>(a, B, st);
    if (mode == 1 && nconv == 1 && in_affine) return launch_dw_s1_t<1, true, 1>(a, B, st);
    if (mode == 1 && nconv == 1 && !in_affine) return launch_dw_s1_t<1, false, 1>(a, B, st);
    if (mode == 2 && nconv == 1 && in_affine) return launch_dw_s1_t<1, true, 2>(a, B, st);
    if (mode == 2 && nconv == 1 && !in_affine) return launch_dw_s1_t<1, false, 2>(a, B, st);
    return RTFS_ERR_ARG;
}

int launch_dw_s2_pool(const DwArgs& a, int B, hipStream_t st) {
    if (a.Wg >= 16) {
        hipLaunchKernelGGL(dw_s2p_kernel, dim3(cdiv(a.C * ((a.Wg + 1) / 2), 256), cdiv(a.Hg, a.TH), B), dim3(256), 0, st, a);
        return rtfs_launch_status();
    }
    hipLaunchKernelGGL(dw_s2_pool_kernel, dim3(cdiv(a.C * a.Wg, 256), cdiv(a.Hg, a.TH), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_g_form(const float* p0, const float* c1, const double* st1, double inv_count, const float* gamma,
                  const float* beta, float* g, int B, int C, int HW, hipStream_t st) {
    hipLaunchKernelGGL(g_form_kernel, dim3(cdiv(HW, 256 * 4), C, B), dim3(256), 0, st, p0, c1, st1, inv_count, gamma, beta, g, C, HW);
    return rtfs_launch_status();
}

int launch_g_combine(const GCombineArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(g_combine_kernel, dim3(cdiv(a.HW, 256 * 4), a.C, B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_transpose(const float* x, float* y, int N, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(W, 32), cdiv(H, 32), N), dim3(256), 0, st, x, y, H, W);
    return rtfs_launch_status();
}

int launch_stats(const float* x, double* stats, int B, size_t N, hipStream_t st) {
    int gx = (int)((N + 256 * 16 - 1) / (256 * 16));
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(stats_kernel, dim3(gx, B), dim3(256), 0, st, x, stats, N);
    return rtfs_launch_status();
}
