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
    const size_t plane = ((size_t)b * C + c) * (size_t)a.cs;  // channel stride: H * W or padded (DwArgs.cs)
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

// ---------------------------------------------------------------- packed variant: adjacent column pairs + lane exchange
// Measured (PMC, profiles/): these passes are bound by the texture addresser, not by HBM or the VALU -- a dword load is
// issued at 4 lanes per clock (16 cycles per wave instruction whatever the element size), and a 4x4 window re-loads every
// input element four times (once per horizontal tap).  So here a thread owns the ADJACENT column pair (2p, 2p+1) of one
// channel, loads each input element exactly once, and takes the three neighbouring taps (2p-1, 2p+2, 2p+3) from the
// adjacent lanes' registers with v_mov_b32_dpp wave_shr:1 / wave_shl:1.  A wave covers 62 pairs plus one HALO lane on each
// side (lanes 0 and 63 only load; their outputs belong to the neighbouring waves): 3 % redundant loads instead of
// exec-masked edge loads, which measured 15 % (a masked dword load still occupies the addresser).  Channel boundaries need
// no care: a lane whose neighbour belongs to another channel wants a zero-padding column there, and its weight for that
// tap is 0.
// Both outputs of the pair advance with one v_pk_fma_f32 per tap.  Zero padding: out-of-image COLUMNS are folded into the
// per-lane weight pairs, out-of-image ROWS are zeroed only in the first / last row band; the input gLN fold is applied to
// the result: conv(pad0(s*x+b)) = s*conv(pad0(x)) + b*sum(valid w).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// XCD-contiguous block order.  The hardware deals the workgroups of a launch round-robin over the 8 XCDs (linear id mod 8), so neighbours in x -
// which share the cache lines their waves split - sat on different XCDs: each L2 fetched the shared lines again and wrote its part of a line
// back as a partial line.  The launch is 1-D (padded to a multiple of 8); XCD k takes the k-th contiguous eighth of the logical (x, y, z) grid.
#ifndef DW_XCD
#define DW_XCD 1
#endif
struct DwBlk { int x, y, z; bool ok; };
__device__ __forceinline__ DwBlk dw_block(const DwArgs& a) {
    int id = blockIdx.x;
    if (DW_XCD) {
        const int n8 = (int)(gridDim.x >> 3), q = id >> 3;
        id = (id & 7) * n8 + (a.rev ? n8 - 1 - q : q);  // rev: every XCD walks its eighth back to front
    }
    DwBlk k;
    if (a.job_stride) {  // two jobs per sample: [job 0's blocks | job 1's blocks] of sample 0, then of sample 1, ...
        const int z = id / a.job_stride, r = id - z * a.job_stride - a.job_off;
        k.ok = r >= 0 && r < a.gx * a.gy && z * a.gx * a.gy < a.nblk;
        const int rr = k.ok ? r : 0;
        k.x = rr % a.gx;
        k.y = rr / a.gx;
        k.z = k.ok ? z : 0;
        return k;
    }
    id -= a.blk0;
    k.ok = id >= 0 && id < a.nblk;
    id = k.ok ? id : 0;
    k.x = id % a.gx;
    const int t = id / a.gx;
    k.y = t % a.gy;
    k.z = t / a.gy;
    return k;
}
#define DW1P_PAIRS 62  // column pairs a wave produces (64 lanes - 2 halo lanes)

// element at (wave-uniform base) + (32-bit lane BYTE offset): the form global_load/store take as saddr + voffset
// (the empty asm keeps the zero-extension of the offset next to the access: hoisted out of the row loop it becomes a
// 64-bit register pair and every access pays a v_lshl_add_u64 instead)
__device__ __forceinline__ float ldo(const float* __restrict__ base, unsigned boff) {
    asm volatile("" : "+v"(boff));
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ void sto(float* __restrict__ base, unsigned boff, float v) {
    asm volatile("" : "+v"(boff));
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + boff) = v;
}
// two adjacent elements with one 8-byte access.  The address is only 4-byte aligned (odd row pitch): global loads / stores
// take that, and a dwordx2 stream runs at 7 TB/s where the dword stream tops out at 4.9 (tools/bench_stream.hip)
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ f32x2 ldo2(const float* __restrict__ base, unsigned boff) {
    asm volatile("" : "+v"(boff));
    const f32x2u v = *reinterpret_cast<const f32x2u*>(reinterpret_cast<const char*>(base) + boff);
    return f32x2{v.x, v.y};
}
__device__ __forceinline__ void sto2(float* __restrict__ base, unsigned boff, f32x2 v) {
    asm volatile("" : "+v"(boff));
    *reinterpret_cast<f32x2u*>(reinterpret_cast<char*>(base) + boff) = f32x2u{v.x, v.y};
}
// lane i <- lane i-1 (lane 0: 0) / lane i <- lane i+1 (lane 63: 0), whole wave
__device__ __forceinline__ float from_prev_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
}
__device__ __forceinline__ float from_next_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));  // wave_shl:1
}
// wave-uniform copies of a float (bit patterns: the builtins are integer ones)
__device__ __forceinline__ float first_lane_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float lane0_f(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)); }
// raw-buffer accesses: address = descriptor base (wave-uniform) + lane byte offset (VGPR) + row byte offset (SGPR)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ BufRsrc buf_rsrc(const float* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ f32x2 buf_ld2(BufRsrc rs, unsigned voff, unsigned soff) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0)); }
__device__ __forceinline__ float buf_ld1(BufRsrc rs, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0)); }
__device__ __forceinline__ void buf_st2(BufRsrc rs, unsigned voff, unsigned soff, f32x2 v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, v), rs, voff, soff, 0); }
__device__ __forceinline__ void buf_st1(BufRsrc rs, unsigned voff, unsigned soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0); }

// VAR (MODE 2 only): bit 0 = an addend tensor is present, bit 1 = the pair's two gate / embedding columns are shared with the next lane
// (2 Wg <= W).  Compile-time so that a row's loads are straight-line code: with these as run-time branches every row's loads sat in blocks
// of their own and the compiler flushed vmcnt(0) behind them.
//
// Round 3 rewrite of the inner loop.  Counters (tools/pmc_dw.sh) had these passes 50-66 % VALU-busy at 3-4 TB/s, and the ISA showed why: of
// ~60-85 vector instructions per output row only 18 were the packed FMAs - the rest built operand pairs (v_pk_fma_f32 wants even-aligned
// register pairs and the four taps of a row are OVERLAPPING pairs of five values: ~24 v_mov per row), zero-padded columns by select, copied
// lane offsets for every access and (scalar unit) recomputed 64-bit row pointers.  Now:
//   * a window row is kept as its four operand pairs (v0 v1)(v1 v2)(v2 v3)(v3 v4), built once when the row enters the window (it serves four
//     output rows): 2 selects + 4 DPP moves + 2 moves per row;
//   * out-of-image COLUMNS, halo lanes and lanes past the end are folded into per-lane weight PAIRS (one weight per output of the pair): no
//     selects, and a dead output is exactly 0, so the statistics need no mask;
//   * raw-buffer addressing: one lane offset per tensor for the whole kernel, the row offset is an SGPR (one s_mul per access).
template <int NCONV, bool IN_AFFINE, int MODE, int VAR = 0>
__device__ __forceinline__ void dw1p_body(const DwArgs& a, const float* __restrict__ X, const float* __restrict__ GATE,
                                          const float* __restrict__ EMB, const float* __restrict__ ADD_, float* __restrict__ OUT,
                                          float* __restrict__ OUT1, float* __restrict__ OUT2, float* __restrict__ OUT3) {
    static_assert(NCONV == 1 || (MODE == 0 && !IN_AFFINE), "multi-conv: plain write + stats only");
    __shared__ double red[8];
    // ROW (VAR bit 3; W == 129, the width every configuration of this model has): a wave is exactly ONE channel row - 64 lanes x 2 columns plus
    // column 128 as an extra value every lane carries (only lane 63's copy is ever stored) - so there are no halo lanes (the zero a DPP wave
    // shift delivers at the wave's ends IS the zero padding) and, the point of it, a wave's output over its band is ONE contiguous byte range of
    // the channel plane (516-byte rows).  Rows are appended to a per-wave LDS ring at their byte position and leave as 512-byte-ALIGNED chunks:
    // every store instruction writes four whole 128-byte lines.  Row-by-row stores at 516-byte pitch start and end inside a line every time;
    // tools/bench_rows.hip: the same read + write walk 3.8 TB/s with row stores, 4.7 staged, 5.1 with 512-byte rows.
    constexpr bool ROW = (VAR & 8) != 0;
    static_assert(!ROW || (NCONV == 1 && (MODE == 0 || MODE == 2) && !(VAR & 4)), "row variant: single conv, write or TFAR apply");
    // NOHALO (VAR bit 4; rows whose pair count divides 64, i.e. the 64-column low-resolution planes): a wave = 64 pairs = whole channel rows and
    // no halo lanes (a neighbour across a wave boundary is a neighbour across a channel boundary: its weight is 0 anyway), so every store
    // writes whole 256-byte rows instead of 496 bytes that begin and end inside a line.
    constexpr bool NOHALO = (VAR & 16) != 0;
    static_assert(!(ROW && NOHALO), "one mapping");
    constexpr int RING = 2048;
    __shared__ __attribute__((aligned(16))) unsigned char ring_all[ROW ? 4 : 1][ROW ? RING : 16];
    constexpr bool gshare = MODE == 2 && (VAR & 2);  // the launcher checks 2 Wg <= W
    constexpr bool HAS_ADD = MODE == 2 && (VAR & 1);
    constexpr bool COMBINE = MODE == 0 && (VAR & 4);  // input = TFAR combination of x / gate / emb, formed at load time (DwArgs.in_combine)
    const int H = a.H, W = a.W, C = a.C;
    const int NP = (W + 1) >> 1;  // column pairs per row
    const DwBlk blk = dw_block(a);
    if (!blk.ok) return;  // block-uniform (padding of the 1-D launch)
    const int b = blk.z;
    const int lane = threadIdx.x & 63;
    // flattened (channel, pair) index: wave-slot ws covers [62 ws - 1, 62 ws + 62]; lanes 0 and 63 are halo lanes
    const int ws = blk.x * 4 + (threadIdx.x >> 6);
    const int gi = ROW ? ws * NP + lane : NOHALO ? ws * 64 + lane : ws * DW1P_PAIRS - 1 + lane;
    const bool live = ROW || (NOHALO ? gi < C * NP : lane >= 1 && lane <= DW1P_PAIRS && gi < C * NP);  // gi >= 0 follows from lane >= 1
    const int gc = gi < 0 ? 0 : (gi < C * NP ? gi : C * NP - 1);      // halo / dead lanes still load real, finite data
    const int c = ROW ? ws : gc / NP, p = ROW ? lane : gc - c * NP;  // (ROW: the launcher checks C % 4 == 0)
    const int x0 = 2 * p, x1 = 2 * p + 1;
    const bool liveb = live && x1 < W;
    const int r0 = blk.y * a.TH, r1 = min(r0 + a.TH, H);
    const bool whole = x1 < W;  // the odd-width row's last pair (x1 == W) loads (x0 - 1, x0) instead and takes x0 from .y
    // addressing: one descriptor per tensor (base = this sample), one lane byte offset (channel plane + column), row offsets are scalars;
    // the launcher checks that C * cs * 4 fits 31 bits
    const unsigned pa = (unsigned)c * (unsigned)a.cs * 4u;
    const unsigned vld = pa + 4u * (unsigned)(whole ? x0 : x0 - 1);  // pair load position
    const unsigned vst = pa + 4u * (unsigned)x0;                     // store position
    const unsigned W4 = (unsigned)W * 4u;
    const size_t sample = (size_t)b * C * (size_t)a.cs;
    const BufRsrc xs = buf_rsrc(X + sample);
    BufRsrc os[NCONV];
    if (MODE != 1) {
#pragma unroll
        for (int n = 0; n < NCONV; ++n) os[n] = buf_rsrc((n == 0 ? OUT : n == 1 ? OUT1 : n == 2 ? OUT2 : OUT3) + sample);
    }
    // MODE 2: the gate / embedding tensors live at low resolution (Hg x Wg); output (t, x) reads ('nearest', legacy) row floor(t Hg / H),
    // column floor(x Wg / W).  When 2 Wg <= W the pair's two source columns are s0 and s0 or s0 + 1, and s0 + 1 is then exactly the NEXT
    // lane's s0: one load per lane and row, the neighbour's by DPP.
    unsigned fga = 0, fgb = 0;
    bool gnext = false;
    const unsigned vex = pa + 4u * (unsigned)(W - 1);  // ROW: column 128 of this wave's channel (one address for the whole wave)
    const size_t gsample = MODE == 2 ? (size_t)b * C * a.Hg * a.Wg : 0;
    const BufRsrc gs = buf_rsrc(MODE == 2 ? GATE + gsample : (COMBINE ? GATE + sample : X)), es = buf_rsrc(MODE == 2 ? EMB + gsample : (COMBINE ? EMB + sample : X));
    const BufRsrc as_ = buf_rsrc(HAS_ADD ? ADD_ + sample : X);
    if (MODE == 2) {
        const unsigned gpa = (unsigned)c * (unsigned)(a.Hg * a.Wg) * 4u;
        const unsigned xb_ = (unsigned)(x1 < W ? x1 : W - 1);
        const unsigned s0 = min((unsigned)x0 * (unsigned)a.Wg / (unsigned)W, (unsigned)a.Wg - 1u);
        const unsigned s1 = min(xb_ * (unsigned)a.Wg / (unsigned)W, (unsigned)a.Wg - 1u);
        fga = gpa + 4u * s0;
        fgb = gpa + 4u * s1;
        gnext = s1 != s0;
        // ROW: lane 0's two columns and lane 1's first share source column 0 (the launcher checks it), so lane 0 takes its value from lane 1 and
        // spends its own gather on the source column of column W - 1 instead - read back from lane 0 where it is needed
        if (ROW && lane == 0) fga = gpa + 4u * min((unsigned)(W - 1) * (unsigned)a.Wg / (unsigned)W, (unsigned)a.Wg - 1u);
    }
    const bool border = r0 == 0 || r1 + 3 > H;  // block-uniform: only these bands ever see an out-of-image row
    // MODE 2: low-resolution source row floor(t Hg / H) as an incremental quotient / remainder, advanced in LOAD order
    int tgq = 0, tgr = 0;
    if (MODE == 2) {
        tgq = (int)(((long long)r0 * a.Hg) / H);
        tgr = (int)(((long long)r0 * a.Hg) % H);
    }
    // A row in flight: own pair (raw, as loaded - the odd row end's select is applied by the consumer: a use next to the load would pull the
    // wait there) and, for MODE 2, the epilogue operands of OUTPUT row t - 2 (gate / embedding gathers, addend)
    struct Raw { f32x2 pr, ar, gr, er; float g0, g1, m0, m1, xe, ae; };  // xe, ae: column 128 (ROW)
    auto load_raw = [&](int t) {  // window row t, and the epilogue operands of OUTPUT row t - 2
        const int tc = t < 0 ? 0 : (t < H ? t : H - 1);
        Raw r;
        r.pr = buf_ld2(xs, vld, (unsigned)tc * W4);
        r.ar = r.gr = r.er = f32x2{0.f, 0.f};
        if (COMBINE) {
            r.gr = buf_ld2(gs, vld, (unsigned)tc * W4);
            r.er = buf_ld2(es, vld, (unsigned)tc * W4);
        }
        r.g0 = r.g1 = r.m0 = r.m1 = 0.f;
        r.xe = r.ae = 0.f;
        if (ROW) r.xe = buf_ld1(xs, vex, (unsigned)tc * W4);
        if (MODE == 2) {
            const int to = min(max(t - 2, r0), H - 1);  // output row served (clamped: the extra rows of the last trip are dropped)
            const int tg = tgq < a.Hg - 1 ? tgq : a.Hg - 1;
            {  // advance once per output row: scalar selects, no branch (Hg <= H: at most one step)
                const bool adv = t - 2 >= r0;
                tgr += adv ? a.Hg : 0;
                const bool wrap = tgr >= H;
                tgr -= wrap ? H : 0;
                tgq += wrap ? 1 : 0;
            }
            const unsigned grow = (unsigned)tg * (unsigned)a.Wg * 4u;
            r.g0 = buf_ld1(gs, fga, grow);  // every row (the source row repeats for two output rows: L1 hits): a branch around the gathers
            r.m0 = buf_ld1(es, fga, grow);  // would put a basic-block boundary in front of every wait
            if (!gshare) {
                r.g1 = buf_ld1(gs, fgb, grow);
                r.m1 = buf_ld1(es, fgb, grow);
            }
            if (HAS_ADD) r.ar = buf_ld2(as_, vld, (unsigned)to * W4);
            if (ROW && HAS_ADD) r.ae = buf_ld1(as_, vex, (unsigned)to * W4);
        }
        return r;
    };
    float lsc = 1.f, lsh = 0.f, gsc = 1.f, gsh = 0.f, esc = 1.f, esh = 0.f, asc = 1.f, ash = 0.f;  // set below, before the first complete()
    // a window row as the four operand pairs of its taps
    struct Row { f32x2 p[4]; float xe; };  // xe (ROW): column 128; output column 128 meets it and column 127 = p[1].y of lane 63
    auto complete = [&](int t, const Raw& r) {
        float v1 = whole ? r.pr.x : r.pr.y, v2 = r.pr.y;  // x1 == W: v2 only ever meets zero weights
        if (COMBINE) {
            const float g1_ = whole ? r.gr.x : r.gr.y, e1_ = whole ? r.er.x : r.er.y;
            v1 = fmaf(fmaf(v1, lsc, lsh), sigmoidf_(fmaf(g1_, gsc, gsh)), fmaf(e1_, esc, esh));
            v2 = fmaf(fmaf(v2, lsc, lsh), sigmoidf_(fmaf(r.gr.y, gsc, gsh)), fmaf(r.er.y, esc, esh));
        }
        float v0 = from_prev_lane(v2), v3 = from_next_lane(v1), v4 = from_next_lane(v2);
        if (ROW) v3 = lane == 63 ? r.xe : v3;  // the last pair's right neighbour is column 128
        Row w;
        w.xe = r.xe;
        w.p[0] = f32x2{v0, v1};
        w.p[1] = f32x2{v1, v2};
        w.p[2] = f32x2{v2, v3};
        w.p[3] = f32x2{v3, v4};
        if (border && (t < 0 || t >= H)) {  // uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) w.p[j] = f32x2{0.f, 0.f};
            w.xe = 0.f;
        }
        return w;
    };
    // ---- start-up: every load the first rows need is requested before the first wait and before the f64 gLN folds below
#ifndef DW1P_RQ
#define DW1P_RQ (MODE == 2 ? 4 : 8)
#endif
    constexpr int RQ = DW1P_RQ;  // rows in flight per wave = ring depth
    Raw st0 = load_raw(r0 - 1), st1 = load_raw(r0), st2 = load_raw(r0 + 1);
    Raw q[RQ];
#pragma unroll
    for (int k = 0; k < RQ; ++k) q[k] = load_raw(r0 + 2 + k);
    f32x4 wraw[NCONV][4];
#pragma unroll
    for (int n = 0; n < NCONV; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) wraw[n][i] = *reinterpret_cast<const f32x4*>(a.w[n] + c * 16 + i * 4);
    float isc = 1.f, ish = 0.f;
    if (IN_AFFINE) gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    if (MODE == 2 || COMBINE) {
        gln_fold(a.loc_stats + 2 * b, COMBINE ? a.g_inv_count : a.loc_inv_count, a.loc_gamma[c], a.loc_beta[c], lsc, lsh);
        gln_fold(a.gate_stats + 2 * b, a.g_inv_count, a.gate_gamma[c], a.gate_beta[c], gsc, gsh);
        gln_fold(a.emb_stats + 2 * b, a.g_inv_count, a.emb_gamma[c], a.emb_beta[c], esc, esh);
        if (HAS_ADD) gln_fold(a.add_stats + 2 * b, a.add_inv_count, a.add_gamma[c], a.add_beta[c], asc, ash);
        if (ROW) {  // one channel per wave: the folds that only meet plain FMAs live in scalar registers
            gsc = first_lane_f(gsc); gsh = first_lane_f(gsh);
            esc = first_lane_f(esc); esh = first_lane_f(esh);
            asc = first_lane_f(asc); ash = first_lane_f(ash);
        }
    }
    // weight pairs: tap (i, j) of output x0 meets column x0 - 1 + j, of output x1 column x1 - 1 + j; a weight whose tap column is outside the
    // image - or whose output does not exist (halo lane, lane past the end, x1 == W) - is 0
    bool okA[4], okB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xa = x0 - 1 + j, xb = x1 - 1 + j;
        okA[j] = live && xa >= 0 && xa < W;
        okB[j] = liveb && xb >= 0 && xb < W;
    }
    f32x2 wp[NCONV][16];
    f32x2 rowsum[4];
#pragma unroll
    for (int n = 0; n < NCONV; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) wp[n][i * 4 + j] = f32x2{okA[j] ? wraw[n][i][j] : 0.f, okB[j] ? wraw[n][i][j] : 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) rowsum[i] = (wp[0][i * 4] + wp[0][i * 4 + 1]) + (wp[0][i * 4 + 2] + wp[0][i * 4 + 3]);
    f32x2 bias[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) {
        const float bv = a.bias[n] ? a.bias[n][c] : 0.f;
        bias[n] = f32x2{live ? bv : 0.f, liveb ? bv : 0.f};
    }
    // ROW: output column W - 1 = 128 meets columns 127 and 128 only (taps 0 and 1 of every window row)
    // (its channel is the whole wave's: weights as scalars)
    float we0[4], we1[4], rse[4], bias_e = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        we0[i] = ROW ? first_lane_f(wraw[0][i][0]) : 0.f;
        we1[i] = ROW ? first_lane_f(wraw[0][i][1]) : 0.f;
        rse[i] = we0[i] + we1[i];
    }
    if (ROW) bias_e = a.bias[0] ? first_lane_f(a.bias[0][c]) : 0.f;
    const float wve_full = (rse[0] + rse[1]) + (rse[2] + rse[3]);
    // ROW: the staged store (see the head of this function).  Byte positions are relative to the channel plane; s0 / s1 = this band's range
    unsigned char* const ring = ring_all[ROW ? (threadIdx.x >> 6) : 0];
    const unsigned s0 = (unsigned)r0 * W4, s1 = (unsigned)r1 * W4;
    unsigned chunk = s0 >> 9;  // next 512-byte chunk to leave
    auto flush = [&](unsigned cidx) {
        const unsigned cb = cidx << 9, b0 = cb + 8u * (unsigned)lane;
        const f32x2 v = *reinterpret_cast<const f32x2*>(ring + (cb & (RING - 1)) + 8u * (unsigned)lane);
        if (cb >= s0 && cb + 512u <= s1) {  // uniform: the whole chunk is this band's
            buf_st2(os[0], pa + 8u * (unsigned)lane, cb, v);
        } else {  // a band's first / last chunk: the other bytes belong to the neighbouring band's wave
            if (b0 >= s0 && b0 + 4u <= s1) buf_st1(os[0], pa + 8u * (unsigned)lane, cb, v.x);
            if (b0 + 4u >= s0 && b0 + 8u <= s1) buf_st1(os[0], pa + 8u * (unsigned)lane + 4u, cb, v.y);
        }
    };
    auto stage = [&](int t, f32x2 y, float ye) {
        const unsigned pos = (unsigned)t * W4 + 8u * (unsigned)lane;
        *reinterpret_cast<float*>(ring + (pos & (RING - 1))) = y.x;
        *reinterpret_cast<float*>(ring + ((pos + 4u) & (RING - 1))) = y.y;
        if (lane == 63) *reinterpret_cast<float*>(ring + ((pos + 8u) & (RING - 1))) = ye;  // column 128 follows lane 63's pair
        const unsigned end = (unsigned)(t + 1) * W4;
        while (((chunk + 1u) << 9) <= end) {  // uniform
            flush(chunk);
            ++chunk;
        }
    };
    Row win[3];
    win[0] = complete(r0 - 1, st0);
    win[1] = complete(r0, st1);
    win[2] = complete(r0 + 1, st2);
    f32x2 s2[NCONV], ss2[NCONV];
#pragma unroll
    for (int n = 0; n < NCONV; ++n) s2[n] = ss2[n] = f32x2{0.f, 0.f};
    const f32x2 wv_full = (rowsum[0] + rowsum[1]) + (rowsum[2] + rowsum[3]);
    // one output row t (both columns of the pair) from window rows t-1 .. t+2; e = the raw record that came with row t+2
    auto do_row = [&](int t, const Row& w0, const Row& w1, const Row& w2, const Row& w3, const Raw& e) {
        const unsigned orow = (unsigned)t * W4;  // uniform
#pragma unroll
        for (int n = 0; n < NCONV; ++n) {
            // two independent chains (window rows 0-1 and 2-3): a single 16-deep chain of packed FMAs stalls on its own latency
            f32x2 accA = w0.p[0] * wp[n][0], accB = w2.p[0] * wp[n][8];
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                accA = w0.p[j] * wp[n][j] + accA;
                accB = w2.p[j] * wp[n][8 + j] + accB;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                accA = w1.p[j] * wp[n][4 + j] + accA;
                accB = w3.p[j] * wp[n][12 + j] + accB;
            }
            f32x2 acc = accA + accB;
            float acce = 0.f;
            if (ROW) {  // two chains, as above (every lane computes it, lane 63's is the one that counts)
                float ea = fmaf(w0.xe, we1[0], w0.p[1].y * we0[0]), eb = fmaf(w2.xe, we1[2], w2.p[1].y * we0[2]);
                ea = fmaf(w1.xe, we1[1], fmaf(w1.p[1].y, we0[1], ea));
                eb = fmaf(w3.xe, we1[3], fmaf(w3.p[1].y, we0[3], eb));
                acce = ea + eb;
            }
            if (IN_AFFINE) {  // conv(pad0(s x + b)) = s conv(pad0(x)) + b * (sum of the weights whose tap is inside the image)
                f32x2 wv = wv_full;
                float wve = wve_full;
                if (border) {
                    if (t - 1 < 0) { wv -= rowsum[0]; wve -= rse[0]; }
                    if (t + 1 >= H) { wv -= rowsum[2]; wve -= rse[2]; }
                    if (t + 2 >= H) { wv -= rowsum[3]; wve -= rse[3]; }
                }
                acc = acc * isc + wv * ish;
                if (ROW) acce = fmaf(acce, isc, wve * ish);
            }
            acc += bias[n];
            if (ROW) acce += bias_e;
            if (MODE == 0) {
                if (ROW) stage(t, acc, acce);
                else if (liveb) buf_st2(os[n], vst, orow, acc);
                else if (live) buf_st1(os[n], vst, orow, acc.x);
            }
            if (MODE != 2) {
                s2[n] += acc;  // dead outputs are exactly 0
                ss2[n] = acc * acc + ss2[n];
                if (ROW) {
                    const float em = lane == 63 ? acce : 0.f;
                    s2[n].x += em;
                    ss2[n].x = fmaf(em, em, ss2[n].x);
                }
            } else {
                float g0 = e.g0, m0 = e.m0, g1 = e.g1, m1 = e.m1;
                float ge = 0.f, me = 0.f;
                if (gshare) {
                    float gn = from_next_lane(e.g0), mn = from_next_lane(e.m0);
                    if (ROW) {
                        ge = lane0_f(e.g0);  // column 128's source column, gathered by lane 0 (see fga)
                        me = lane0_f(e.m0);
                        g0 = lane == 0 ? gn : g0;  // lane 0's own source column is lane 1's
                        m0 = lane == 0 ? mn : m0;
                        gn = lane == 63 ? ge : gn;  // the last pair's second source column is column 128's (the launcher checks both)
                        mn = lane == 63 ? me : mn;
                    }
                    g1 = gnext ? gn : g0;
                    m1 = gnext ? mn : m0;
                }
                const f32x2 gate = {sigmoidf_(fmaf(g0, gsc, gsh)), sigmoidf_(fmaf(g1, gsc, gsh))};
                const f32x2 emb = {fmaf(m0, esc, esh), fmaf(m1, esc, esh)};
                f32x2 y = (acc * lsc + lsh) * gate + emb;
                if (HAS_ADD) y += f32x2{fmaf(whole ? e.ar.x : e.ar.y, asc, ash), fmaf(e.ar.y, asc, ash)};
                if (ROW) {
                    float ye = fmaf(fmaf(acce, lsc, lsh), sigmoidf_(fmaf(ge, gsc, gsh)), fmaf(me, esc, esh));
                    if (HAS_ADD) ye += fmaf(e.ae, asc, ash);
                    stage(t, y, ye);
                } else if (liveb) buf_st2(os[0], vst, orow, y);
                else if (live) buf_st1(os[0], vst, orow, y.x);
            }
        }
    };
    // RQ output rows per trip, fully unrolled so that the ring index is static; the slot of row t + 2 + k is re-loaded (row t + 2 + RQ + k)
    // RQ rows before that value is needed
    for (int t = r0; t < r1; t += RQ) {
#pragma unroll
        for (int k = 0; k < RQ; ++k) {
            const Raw e = q[k];
            const Row n = complete(t + 2 + k, e);
            // MODE 2 re-loads the slot only after the row's last use of it (do_row reads the epilogue operands): while the old and the new
            // value of a slot overlap they live in different registers, and the copy at the loop's back edge then waits for EVERY load
            if (MODE != 2) q[k] = load_raw(t + 2 + RQ + k);
            if (t + k < r1) do_row(t + k, win[0], win[1], win[2], n, e);  // uniform
            if (MODE == 2) q[k] = load_raw(t + 2 + RQ + k);
            win[0] = win[1];
            win[1] = win[2];
            win[2] = n;
        }
    }
    if (ROW && (chunk << 9) < s1) flush(chunk);  // the band's last, partial chunk
    if (MODE != 2) {
#pragma unroll
        for (int n = 0; n < NCONV; ++n) {
            block_stats_atomic(s2[n].x + s2[n].y, ss2[n].x + ss2[n].y, red, a.stats_out[n] + 2 * b);
            __syncthreads();
        }
    }
}

// single-conv variants are held to 128 VGPRs (4 waves per SIMD): these passes live on bytes in flight
template <int NCONV, bool IN_AFFINE, int MODE, int VAR = 0>
__global__ __launch_bounds__(256, NCONV == 1 && MODE != 2 ? 4 : 2) void dw1p_kernel(DwArgs a) {
    dw1p_body<NCONV, IN_AFFINE, MODE, VAR>(a, a.x, a.gate, a.emb, a.addend, a.out[0], a.out[1], a.out[2], a.out[3]);
}

// three jobs in one launch (launch_dw_g3): block-uniform dispatch on the permuted block id
template <int V>
__global__ __launch_bounds__(256, 2) void dw_g3_kernel(DwArgs a0, DwArgs a1, DwArgs a2) {
    int id = blockIdx.x;
    if (DW_XCD) id = (id & 7) * (int)(gridDim.x >> 3) + (id >> 3);
    if (id < a1.blk0) dw1p_body<2, false, 0, V>(a0, a0.x, a0.gate, a0.emb, a0.addend, a0.out[0], a0.out[1], a0.out[2], a0.out[3]);
    else if (id < a2.blk0) dw1p_body<2, false, 0, V>(a1, a1.x, a1.gate, a1.emb, a1.addend, a1.out[0], a1.out[1], a1.out[2], a1.out[3]);
    else dw1p_body<1, true, 0, V>(a2, a2.x, a2.gate, a2.emb, a2.addend, a2.out[0], a2.out[1], a2.out[2], a2.out[3]);
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
    const size_t plane = ((size_t)b * C + c) * (size_t)a.cs;
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

// Lane-exchange variant of the stride-2 kernel (used when Wo >= 16; same idea as dw1p_kernel): a thread owns the INPUT column
// pair (2j, 2j+1) of one channel and produces output column j.  One 8-byte load per input row; the two outer taps (2j-1,
// 2j+2) come from the adjacent lanes by DPP; 62 pairs + 2 halo lanes per wave.  The convolution runs on the RAW input
// (column padding = zeroed window values, out-of-range rows zeroed by a uniform branch) and the input fold is applied once
// per output: conv(isc x + ish) = isc conv(x) + ish * (sum of in-bounds weights) + bias; the pool likewise as
// isc * mean(x) + ish with the column-masked row sums carried in the rolling window.  The pool window's row range is an
// incremental quotient / remainder (no per-row integer division on the scalar unit).
__device__ __forceinline__ void dw_s2x_body(const DwArgs& a, const float* __restrict__ X, float* __restrict__ O0, float* __restrict__ O1) {
    __shared__ double red[8];
    const int H = a.H, W = a.W, C = a.C, Ho = a.Hg, Wo = a.Wg;
    const int NP = (W + 1) >> 1;  // input column pairs per row (slot j >= Wo only loads)
    const DwBlk blk = dw_block(a);
    if (!blk.ok) return;
    const int b = blk.z;
    const int lane = threadIdx.x & 63;
    const int ws = blk.x * 4 + (threadIdx.x >> 6);
    const int gi = ws * DW1P_PAIRS - 1 + lane;
    const int gc = gi < 0 ? 0 : (gi < C * NP ? gi : C * NP - 1);
    const int c = gc / NP, j = gc - c * NP;
    const bool live = lane >= 1 && lane <= DW1P_PAIRS && gi < C * NP && j < Wo;
    const int x0 = 2 * j, x1 = 2 * j + 1;
    const int i0 = blk.y * a.TH, i1 = min(i0 + a.TH, Ho);
    const size_t sample = (size_t)b * C * (size_t)a.cs;
    const float* __restrict__ Xs_ = X + sample;
    const unsigned pa = (unsigned)c * (unsigned)a.cs * 4u;
    const bool whole = x1 < W;
    const unsigned o2 = pa + 4u * (unsigned)(whole ? x0 : x0 - 1);  // the odd-width row's last pair reads (x0-1, x0)
    auto load_raw = [&](int t) {
        const float* __restrict__ rp = Xs_ + (size_t)(t < 0 ? 0 : (t < H ? t : H - 1)) * W;  // uniform
        const f32x2 pr = ldo2(rp, o2);
        return f32x2{whole ? pr.x : pr.y, pr.y};
    };
    // ---- start-up loads first (rows 2 i0 - 1 .. 2 i0 + 4), then the fold and the weights
    f32x2 q[4];
    f32x2 st0 = load_raw(2 * i0 - 1), st1 = load_raw(2 * i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = load_raw(2 * i0 + 1 + k);
    f32x4 wraw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wraw[i] = *reinterpret_cast<const f32x4*>(a.w[0] + c * 16 + i * 4);
    float isc, ish;
    gln_fold(a.in_stats + 2 * b, a.in_inv_count, a.in_gamma[c], a.in_beta[c], isc, ish);
    const float bia = a.bias[0][c];
    const int fs = (int)(((long long)j * W) / Wo), fe = (int)(((long long)(min(j, Wo - 1) + 1) * W + Wo - 1) / Wo);
    float wgt[16], pm[4], wrow[4];
    bool okc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int f = 2 * j - 1 + d;
        okc[d] = f >= 0 && f < W;
        pm[d] = f >= fs && f < fe ? 1.f : 0.f;
    }
#pragma unroll
    for (int di = 0; di < 4; ++di) {
        wrow[di] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            wgt[di * 4 + d] = wraw[di][d];
            wrow[di] += okc[d] ? wraw[di][d] : 0.f;
        }
    }
    const float wall = wrow[0] + wrow[1] + wrow[2] + wrow[3];
    const float nf = (float)(fe - fs);
    // window row: the four tap columns 2j-1 .. 2j+2 (padding zeroed) and their pool-masked sum
    auto complete = [&](int t, f32x2 r, float (&v)[4], float& rs) {
        v[1] = r.x;
        v[2] = r.y;
        v[0] = from_prev_lane(r.y);
        v[3] = from_next_lane(r.x);
        if (!okc[0]) v[0] = 0.f;
        if (!okc[2]) v[2] = 0.f;
        if (!okc[3]) v[3] = 0.f;
        if (t < 0 || t >= H) {  // uniform: only the first / last band row of the image
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] = 0.f;
        }
        rs = v[0] * pm[0];
#pragma unroll
        for (int d = 1; d < 4; ++d) rs = fmaf(v[d], pm[d], rs);
    };
    float win[2][4], rsw[2];
    complete(2 * i0 - 1, st0, win[0], rsw[0]);
    complete(2 * i0, st1, win[1], rsw[1]);
    // pool rows of output i: [floor(i H / Ho), ceil((i+1) H / Ho)); (qq, rem) = divmod(i H, Ho) kept incrementally
    int qq = (int)(((long long)i0 * H) / Ho), rem = (int)(((long long)i0 * H) % Ho);
    float s1 = 0.f, ss1 = 0.f;
    const size_t osample = (size_t)b * C * Ho * Wo;
    const unsigned oo = ((unsigned)c * (unsigned)(Ho * Wo) + (unsigned)min(j, Wo - 1)) * 4u;
    auto do_row = [&](int i, const float (&w0)[4], const float (&w1)[4], const float (&w2)[4], const float (&w3)[4], float r0_, float r1_,
                      float r2_, float r3_) {
        const int ts = qq;
        rem += H;
        while (rem >= Ho) {
            rem -= Ho;
            ++qq;
        }
        const int te = qq + (rem > 0 ? 1 : 0);
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) acc = fmaf(w0[d], wgt[d], acc);
#pragma unroll
        for (int d = 0; d < 4; ++d) acc = fmaf(w1[d], wgt[4 + d], acc);
#pragma unroll
        for (int d = 0; d < 4; ++d) acc = fmaf(w2[d], wgt[8 + d], acc);
#pragma unroll
        for (int d = 0; d < 4; ++d) acc = fmaf(w3[d], wgt[12 + d], acc);
        float bsum = wall;
        if (2 * i - 1 < 0 || 2 * i + 2 >= H) {  // uniform: band touches the top / bottom padding
            bsum = 0.f;
#pragma unroll
            for (int di = 0; di < 4; ++di) {
                const int t = 2 * i - 1 + di;
                if (t >= 0 && t < H) bsum += wrow[di];
            }
        }
        acc = fmaf(acc, isc, fmaf(bsum, ish, bia));
        const float rr[4] = {r0_, r1_, r2_, r3_};
        float ps = 0.f;
#pragma unroll
        for (int di = 0; di < 4; ++di) {
            const int t = 2 * i - 1 + di;
            if (t >= ts && t < te) ps += rr[di];  // uniform
        }
        const float pool = fmaf(ps / (nf * (float)(te - ts)), isc, ish);
        const size_t orow = osample + (size_t)i * Wo;  // uniform
        if (live) {
            sto(O0 + orow, oo, acc);
            sto(O1 + orow, oo, pool);
            s1 += acc;
            ss1 = fmaf(acc, acc, ss1);
        }
    };
    // two output rows (four input rows) per trip; the next trip's rows are requested before this trip's arithmetic
    for (int i = i0; i < i1; i += 2) {
        float n[4][4], rn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) complete(2 * i + 1 + k, q[k], n[k], rn[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = load_raw(2 * i + 5 + k);
        do_row(i, win[0], win[1], n[0], n[1], rsw[0], rsw[1], rn[0], rn[1]);
        if (i + 1 < i1) do_row(i + 1, n[0], n[1], n[2], n[3], rn[0], rn[1], rn[2], rn[3]);  // uniform
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            win[0][d] = n[2][d];
            win[1][d] = n[3][d];
        }
        rsw[0] = rn[2];
        rsw[1] = rn[3];
    }
    block_stats_atomic(s1, ss1, red, a.stats_out[0] + 2 * b);
}
__global__ __launch_bounds__(256) void dw_s2x_kernel(DwArgs a) { dw_s2x_body(a, a.x, a.out[0], a.out[1]); }

// The stride-2 + pool pass (step 3) and the statistics pass of fusion 0's local convolution (step 14) read the same tensor, d0 = gLN(c0): one
// launch, the two jobs' workgroups interleaved per sample, so that the second job finds the sample's rows where the first one just left them
// (same XCD, memory-side cache) - and no side stream, no fork / join events, nothing left running beside the F sweep.
__global__ __launch_bounds__(256, 4) void dw_s2_stats_kernel(DwArgs a0, DwArgs a1) {
    int id = blockIdx.x;
    if (DW_XCD) {
        const int n8 = (int)(gridDim.x >> 3), q = id >> 3;
        id = (id & 7) * n8 + (a0.rev ? n8 - 1 - q : q);
    }
    const int r = id % a0.job_stride;  // block-uniform
    if (r < a1.job_off) dw_s2x_body(a0, a0.x, a0.out[0], a0.out[1]);
    else dw1p_body<1, true, 1>(a1, a1.x, a1.gate, a1.emb, a1.addend, a1.out[0], a1.out[1], a1.out[2], a1.out[3]);
}

// ---------------------------------------------------------------- G-level elementwise glue
// g = p0 + gLN(c1)           (global pooling sum, tdanet.py:116)
__global__ __launch_bounds__(256) void g_form_kernel(const float* __restrict__ p0, const float* __restrict__ c1,
                                                     const double* __restrict__ st1, double inv_count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ g, int C, int HW) {
    // one workgroup per (channel, sample) plane, 16-byte accesses: the f64 fold (~100 instructions) used to be paid per FOUR elements
    // (16 k workgroups of 1024 elements: 59 us for 196 MB of Infinity-Cache-resident data)
    const int c = blockIdx.y, b = blockIdx.z;
    float sc, sh;
    gln_fold(st1 + 2 * b, inv_count, gamma[c], beta[c], sc, sh);
    const size_t base = ((size_t)b * C + c) * HW;
    if ((HW & 3) == 0) {
        const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(p0 + base);
        const f32x4* __restrict__ c4 = reinterpret_cast<const f32x4*>(c1 + base);
        f32x4* __restrict__ g4 = reinterpret_cast<f32x4*>(g + base);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) g4[i] = p4[i] + (c4[i] * sc + sh);
        return;
    }
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
    if ((a.HW & 3) == 0) {  // one workgroup per plane, 16-byte accesses (see g_form_kernel)
        const f32x4* __restrict__ l4 = reinterpret_cast<const f32x4*>(a.l + base);
        const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(a.gate + base);
        const f32x4* __restrict__ e4 = reinterpret_cast<const f32x4*>(a.emb + base);
        f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(a.out + base);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.HW / 4; i += gridDim.x * 256) {
            const f32x4 lv = l4[i], gv = g4[i], ev = e4[i];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(fmaf(lv[k], lsc, lsh), sigmoidf_(fmaf(gv[k], gsc, gsh)), fmaf(ev[k], esc, esh));
            o4[i] = o;
        }
        return;
    }
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

// The same through a 64x64 tile with 16-byte accesses on both sides (4-byte-aligned multi-dword accesses run at full rate here, so odd
// row pitches such as T*F = 32379 keep the wide form): the (B, 256, T*F) <-> (B, T*F, 256) layout changes of the training step.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void transpose64_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W) {
    __shared__ float t[64][65];
    const size_t base = (size_t)blockIdx.z * H * W;
    // the last tile of a ragged axis is moved back so that it is whole (it overlaps its neighbour, which writes the same values): with
    // H = 125, W = 64 - the low-resolution planes around the time sweep - half of all tiles were ragged and took the dword path below
    int w0 = blockIdx.x * 64, h0 = blockIdx.y * 64;
    if (W >= 64 && w0 + 64 > W) w0 = W - 64;
    if (H >= 64 && h0 + 64 > H) h0 = H - 64;
    const int tx = (threadIdx.x & 15) * 4, ty = threadIdx.x >> 4;
    if (h0 + 64 <= H && w0 + 64 <= W) {
        f32x4u v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = *reinterpret_cast<const f32x4u*>(x + base + (size_t)(h0 + ty + 16 * p) * W + w0 + tx);
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int k = 0; k < 4; k++) t[ty + 16 * p][tx + k] = v[p][k];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int r = ty + 16 * p;
            f32x4u o = {t[tx][r], t[tx + 1][r], t[tx + 2][r], t[tx + 3][r]};
            *reinterpret_cast<f32x4u*>(y + base + (size_t)(w0 + r) * H + h0 + tx) = o;
        }
        return;
    }
    for (int r = ty; r < 64; r += 16)
        for (int k = 0; k < 4; k++)
            if (h0 + r < H && w0 + tx + k < W) t[r][tx + k] = x[base + (size_t)(h0 + r) * W + w0 + tx + k];
    __syncthreads();
    for (int r = ty; r < 64; r += 16)
        for (int k = 0; k < 4; k++)
            if (w0 + r < W && h0 + tx + k < H) y[base + (size_t)(w0 + r) * H + h0 + tx + k] = t[tx + k][r];
}

// stats of an arbitrary (B, N) tensor (used when a module is called stand-alone)
__global__ __launch_bounds__(256) void stats_kernel(const float* __restrict__ x, double* __restrict__ stats, size_t N) {
    __shared__ double red[8];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * N;
    const size_t n4 = N >> 2;
    f32x4u s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};  // 16-byte loads (a dword stream reads at 60 % of their rate)
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4u v = reinterpret_cast<const f32x4u*>(xb)[i];
        s4 += v;
        q4 += v * v;
    }
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]), ss = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    if (blockIdx.x == 0 && threadIdx.x < (N & 3)) {
        const float v = xb[(n4 << 2) + threadIdx.x];
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

// Band height for small batches: with 64-row bands a batch-1 launch is 68 workgroups that each walk 64 rows one after the other (13-35 us
// per kernel, ~15 of them per block).  Below two workgroups per CU the bands are shortened (multiples of the ring depth, >= 8 rows) until the
// launch has ~768 workgroups; the three halo rows a band re-reads do not matter at that size.  Batches >= 8 keep th.
static int band_rows(int th, int rows, int gx, int B, int target = 768) {
    if (gx * cdiv(rows, th) * B >= 512) return th;
    const int want = cdiv(target, gx * B);
    int t = cdiv(rows, want);
    t = (t + 7) / 8 * 8;
    return t < 8 ? 8 : (t > th ? th : t);
}

template <int NCONV, bool IN_AFFINE, int MODE, int VAR = 0>
static int launch_dw1p_t(const DwArgs& a_, int B, hipStream_t st) {
    DwArgs a = a_;
    const int half = (a.W + 1) / 2;
    a.gx = (VAR & 8) ? a.C / 4 : cdiv(cdiv(a.C * half, (VAR & 16) ? 64 : DW1P_PAIRS), 4);  // (row variant: a wave per channel)
    // (kernels that end in statistics atomics get fewer, longer workgroups: at batch 1 all of them add onto ONE pair of addresses)
    a.TH = band_rows(a.TH, a.H, a.gx, B, MODE == 2 ? 768 : 320);
    a.gy = cdiv(a.H, a.TH);
    a.nblk = a.gx * a.gy * B;
    hipLaunchKernelGGL((dw1p_kernel<NCONV, IN_AFFINE, MODE, VAR>), dim3((a.nblk + 7) / 8 * 8), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_dw_s1(const DwArgs& a_, int nconv, bool in_affine, int mode, int B, hipStream_t st) {
    if (a_.cs && a_.cs < a_.H * a_.W) return RTFS_ERR_SHAPE;
    DwArgs a = a_;
    if (!a.cs) a.cs = a.H * a.W;
    if ((size_t)a.C * a.cs * 4 >= ((size_t)1 << 31)) return RTFS_ERR_SHAPE;  // 32-bit lane offsets
    if (a.W >= 16) {  // packed two-column variant (v_pk_fma_f32); the scalar kernel below only serves very narrow inputs
        // row variant (a wave = one 129-column channel row, stores leave as aligned 512-byte chunks: dw1p_body): the three full-resolution
        // read + write passes of a block
        const bool row_ok = nconv == 1 && a.W == 129 && a.C % 4 == 0 && a.cs % 32 == 0 && a.out[0] && ((size_t)a.out[0] & 127) == 0;
        if (row_ok && mode == 0 && !in_affine) return launch_dw1p_t<1, false, 0, 8>(a, B, st);
        if (row_ok && mode == 2 && 2 * a.Wg <= a.W && (a.W - 1) * a.Wg / a.W == (a.W - 2) * a.Wg / a.W && 2 * a.Wg / a.W == 0) {
            if (in_affine && !a.addend) return launch_dw1p_t<1, true, 2, 10>(a, B, st);
            if (!in_affine && a.addend) return launch_dw1p_t<1, false, 2, 11>(a, B, st);
        }
        if (nconv == 1) {
            if (mode == 0) return in_affine ? launch_dw1p_t<1, true, 0>(a, B, st) : launch_dw1p_t<1, false, 0>(a, B, st);
            if (mode == 1) return in_affine ? launch_dw1p_t<1, true, 1>(a, B, st) : launch_dw1p_t<1, false, 1>(a, B, st);
            if (mode == 2) {
                const int var = (a.addend ? 1 : 0) | (2 * a.Wg <= a.W ? 2 : 0);
                if (in_affine) {
                    switch (var) {
                        case 0: return launch_dw1p_t<1, true, 2, 0>(a, B, st);
                        case 1: return launch_dw1p_t<1, true, 2, 1>(a, B, st);
                        case 2: return launch_dw1p_t<1, true, 2, 2>(a, B, st);
                        default: return launch_dw1p_t<1, true, 2, 3>(a, B, st);
                    }
                }
                switch (var) {
                    case 0: return launch_dw1p_t<1, false, 2, 0>(a, B, st);
                    case 1: return launch_dw1p_t<1, false, 2, 1>(a, B, st);
                    case 2: return launch_dw1p_t<1, false, 2, 2>(a, B, st);
                    default: return launch_dw1p_t<1, false, 2, 3>(a, B, st);
                }
            }
        } else if (mode == 0 && !in_affine) {
            if (nconv == 2 && a.in_combine && 64 % ((a.W + 1) / 2) == 0) return launch_dw1p_t<2, false, 0, 4 | 16>(a, B, st);
            if (nconv == 2) return a.in_combine ? launch_dw1p_t<2, false, 0, 4>(a, B, st) : launch_dw1p_t<2, false, 0>(a, B, st);
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

int launch_dw_g3(const DwArgs& conv4, const DwArgs& aff1, int B, hipStream_t st) {
    DwArgs j[3] = {conv4, conv4, aff1};
    const bool nohalo = 64 % ((conv4.W + 1) / 2) == 0 && conv4.W == aff1.W;  // whole rows per wave (dw1p_body NOHALO)
    for (int i = 0; i < 2; ++i) {
        j[1].w[i] = conv4.w[2 + i]; j[1].bias[i] = conv4.bias[2 + i]; j[1].out[i] = conv4.out[2 + i]; j[1].stats_out[i] = conv4.stats_out[2 + i];
    }
    int off = 0;
    for (int i = 0; i < 3; ++i) {
        DwArgs& a = j[i];
        if (a.W < 16 || (a.cs && a.cs < a.H * a.W)) return RTFS_ERR_ARG;  // the caller falls back to separate launches
        if (!a.cs) a.cs = a.H * a.W;
        if ((size_t)a.C * a.cs * 4 >= ((size_t)1 << 31)) return RTFS_ERR_ARG;
        a.gx = cdiv(cdiv(a.C * ((a.W + 1) / 2), nohalo ? 64 : DW1P_PAIRS), 4);
        a.TH = band_rows(a.TH, a.H, a.gx, 3 * B, 640);  // (three jobs share the launch)
        a.gy = cdiv(a.H, a.TH);
        a.nblk = a.gx * a.gy * B;
        a.blk0 = off;
        off += a.nblk;
    }
    if (nohalo) hipLaunchKernelGGL(dw_g3_kernel<16>, dim3((off + 7) / 8 * 8), dim3(256), 0, st, j[0], j[1], j[2]);
    else hipLaunchKernelGGL(dw_g3_kernel<0>, dim3((off + 7) / 8 * 8), dim3(256), 0, st, j[0], j[1], j[2]);
    return rtfs_launch_status();
}

int launch_dw_s2_stats(const DwArgs& s2_, const DwArgs& st_, int B, hipStream_t st) {
    DwArgs a = s2_, b = st_;
    if ((a.cs && a.cs < a.H * a.W) || a.x != b.x || a.cs != b.cs || a.H != b.H || a.W != b.W || a.C != b.C) return RTFS_ERR_ARG;
    if (!a.cs) a.cs = b.cs = a.H * a.W;
    if (!(a.Wg >= 16 && a.W >= 16 && (a.W + 1) / 2 >= a.Wg && (size_t)a.C * a.cs * 4 < ((size_t)1 << 31))) return RTFS_ERR_ARG;
    a.gx = b.gx = cdiv(cdiv(a.C * ((a.W + 1) / 2), DW1P_PAIRS), 4);
    a.TH = band_rows(a.TH, a.Hg, a.gx, B, 320);
    a.gy = cdiv(a.Hg, a.TH);
    b.TH = band_rows(b.TH, b.H, b.gx, B, 320);
    b.gy = cdiv(b.H, b.TH);
    const int ca = a.gx * a.gy, cb = b.gx * b.gy;
    a.nblk = ca * B;
    b.nblk = cb * B;
    a.job_stride = b.job_stride = ca + cb;
    a.job_off = 0;
    b.job_off = ca;
    b.rev = a.rev;
    const long total = (long)(ca + cb) * B;
    hipLaunchKernelGGL(dw_s2_stats_kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), 0, st, a, b);
    return rtfs_launch_status();
}

int launch_dw_s2_pool(const DwArgs& a_, int B, hipStream_t st) {
    if (a_.cs && a_.cs < a_.H * a_.W) return RTFS_ERR_SHAPE;
    DwArgs a = a_;
    if (!a.cs) a.cs = a.H * a.W;
    if (a.Wg >= 16 && a.W >= 2 && (a.W + 1) / 2 >= a.Wg && (size_t)a.C * a.cs * 4 < ((size_t)1 << 31)) {
        a.gx = cdiv(cdiv(a.C * ((a.W + 1) / 2), DW1P_PAIRS), 4);
        a.TH = band_rows(a.TH, a.Hg, a.gx, B, 320);
        a.gy = cdiv(a.Hg, a.TH);
        a.nblk = a.gx * a.gy * B;
        hipLaunchKernelGGL(dw_s2x_kernel, dim3((a.nblk + 7) / 8 * 8), dim3(256), 0, st, a);
        return rtfs_launch_status();
    }
    hipLaunchKernelGGL(dw_s2_pool_kernel, dim3(cdiv(a.C * a.Wg, 256), cdiv(a.Hg, a.TH), B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_g_form(const float* p0, const float* c1, const double* st1, double inv_count, const float* gamma,
                  const float* beta, float* g, int B, int C, int HW, hipStream_t st) {
    hipLaunchKernelGGL(g_form_kernel, dim3(B * C >= 1024 ? 1 : 4, C, B), dim3(256), 0, st, p0, c1, st1, inv_count, gamma, beta, g, C, HW);
    return rtfs_launch_status();
}

int launch_g_combine(const GCombineArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(g_combine_kernel, dim3(B * a.C >= 1024 ? 1 : 4, a.C, B), dim3(256), 0, st, a);
    return rtfs_launch_status();
}

int launch_transpose(const float* x, float* y, int N, int H, int W, hipStream_t st) {
    if (H >= 64 && W >= 64)
        hipLaunchKernelGGL(transpose64_kernel, dim3(cdiv(W, 64), cdiv(H, 64), N), dim3(256), 0, st, x, y, H, W);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(W, 32), cdiv(H, 32), N), dim3(256), 0, st, x, y, H, W);
    return rtfs_launch_status();
}

// Two-stage form for the training path (no memset, no contended f64 atomics: 256 workgroups per sample queue ~10 us at their sample's
// pair): every workgroup stores its (sum, sum of squares), a second kernel with one workgroup per sample folds and STORES stats[2b..].
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, double* __restrict__ part, size_t N) {
    __shared__ double red[8];
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * N;
    const size_t n4 = N >> 2;
    f32x4u s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4u v = reinterpret_cast<const f32x4u*>(xb)[i];
        s4 += v;
        q4 += v * v;
    }
    float s = (s4[0] + s4[1]) + (s4[2] + s4[3]), ss = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    if (blockIdx.x == 0 && threadIdx.x < (N & 3)) {
        const float v = xb[(n4 << 2) + threadIdx.x];
        s += v;
        ss = fmaf(v, v, ss);
    }
    const double ds = wave_sum_d((double)s), dss = wave_sum_d((double)ss);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[2 * w] = ds, red[2 * w + 1] = dss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = part + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        p[0] = (red[0] + red[2]) + (red[4] + red[6]);
        p[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}
__global__ __launch_bounds__(256) void stats_fold_kernel(const double* __restrict__ part, double* __restrict__ stats, int gx) {
    __shared__ double red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    double s = 0, ss = 0;
    for (int i = t; i < gx; i += 256) {
        s += part[((size_t)b * gx + i) * 2];
        ss += part[((size_t)b * gx + i) * 2 + 1];
    }
    s = wave_sum_d(s), ss = wave_sum_d(ss);
    if ((t & 63) == 0) red[2 * (t >> 6)] = s, red[2 * (t >> 6) + 1] = ss;
    __syncthreads();
    if (t == 0) {
        stats[2 * b] = (red[0] + red[2]) + (red[4] + red[6]);
        stats[2 * b + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}
// part: 2 * 256 * B doubles
int launch_stats2(const float* x, double* stats, int B, size_t N, double* part, hipStream_t st) {
    int gx = (int)((N / 4 + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(stats_partial_kernel, dim3(gx, B), dim3(256), 0, st, x, part, N);
    hipLaunchKernelGGL(stats_fold_kernel, dim3(B), dim3(256), 0, st, part, stats, gx);
    return rtfs_launch_status();
}

int launch_stats(const float* x, double* stats, int B, size_t N, hipStream_t st) {
    // every workgroup ends in two f64 atomics on its sample's pair: a few hundred long workgroups per sample beat thousands of short ones
    int gx = (int)((N / 4 + 256 * 8 - 1) / (256 * 8));
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(stats_kernel, dim3(gx, B), dim3(256), 0, st, x, stats, N);
    return rtfs_launch_status();
}
