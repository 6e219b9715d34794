// C-ABI layer: parameter-pack layouts, workspace carving and the launch sequences of each module.
// See include/rtfs_amd.h for the contract and the reference interfaces each entry point replaces.
#include "api_common.h"
#include <atomic>
#include <vector>

namespace {

// ---------------------------------------------------------------- parameter packs
// A pack is a flat float buffer; every tensor starts on a 64-float boundary.  The order below is the
// contract with rtfs-net_amd/packing.py.
struct Cursor {
    const float* p;
    size_t off = 0;
    explicit Cursor(const float* base) : p(base) {}
    const float* take(size_t n) {
        const float* r = p ? p + off : nullptr;
        off += (n + 63) / 64 * 64;
        return r;
    }
};

// A piece of work forked onto a library-internal side stream: begin() orders the side stream behind everything queued on `st` so far,
// join() (or the destructor, on EVERY exit path - also after a failed launch in between) orders `st` behind the side stream again, so
// the caller's stream never runs ahead of work this call queued elsewhere (include/rtfs_amd.h: all work of a call is ordered on `stream`).
struct Fork {
    hipStream_t st = nullptr;
    RtfsSide side{};
    bool open = false;
    int begin(hipStream_t owner, int slot) {
        st = owner;
        CHECK(rtfs_side_stream(owner, slot, &side));
        if (hipEventRecord(side.fork, st) != hipSuccess || hipStreamWaitEvent(side.stream, side.fork, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
        open = true;
        return RTFS_OK;
    }
    int join() {
        if (!open) return RTFS_OK;
        open = false;
        if (hipEventRecord(side.join, side.stream) != hipSuccess || hipStreamWaitEvent(st, side.join, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
        return RTFS_OK;
    }
    ~Fork() { (void)join(); }
};

struct EncPack {
    const float* w;  // (256, 18)
    explicit EncPack(Cursor& c) { w = c.take(CA * 18); }
};
struct BnPack {
    const float *gamma, *beta, *wt, *bias, *w16;
    explicit BnPack(Cursor& c) {
        gamma = c.take(CA);
        beta = c.take(CA);
        wt = c.take(CA * CA);
        bias = c.take(CA);
        w16 = c.take(CA * CA);
    }
};
struct DpPack {
    const float *ln_g, *ln_b, *W0, *Wl, *wc = nullptr, *bias, *Wt, *bt;
    const float *w16_l0 = nullptr, *w16_l = nullptr, *w16_ct = nullptr, *wc16 = nullptr, *bias16 = nullptr;  // f16x3 images (k_dualpath16.hip)
    const float *wf_l0 = nullptr, *wf_l = nullptr, *wf_ct = nullptr;  // the same images in fragment order (k_dualpath16s.hip)
    const float* whh = nullptr;  // LSTM cell only
    explicit DpPack(Cursor& c, int rnn_kind = 0) {
        ln_g = c.take(CH);
        ln_b = c.take(CH);
        if (rnn_kind == 1) {  // nn.LSTM(512, 32, 4 layers, bidirectional): order = packing._dualpath_lstm_parts
            W0 = c.take(512 * 256);
            Wl = c.take(3 * 64 * 256);
            bias = c.take(4 * 256);
            whh = c.take(4 * 2 * 32 * 128);
            Wt = c.take(512 * 64);
            bt = c.take(CH);
            return;
        }
        W0 = c.take(512 * 256);
        Wl = c.take(3 * 64 * 256);
        wc = c.take(4 * 128);
        bias = c.take(4 * 128);
        Wt = c.take(512 * 64);
        bt = c.take(CH);
        w16_l0 = c.take(512 * 256);
        w16_l = c.take(3 * 64 * 256);
        w16_ct = c.take(512 * 64);
        wc16 = c.take(4 * 128);
        bias16 = c.take(4 * 128);
        wf_l0 = c.take(512 * 256);
        wf_l = c.take(3 * 64 * 256);
        wf_ct = c.take(512 * 64);
    }
};
struct AttnPack {
    const float *qkv_wt, *qkv_b, *qkv_slope, *qkv_gamma, *qkv_beta, *proj_wt, *proj_b, *proj_slope, *proj_gamma, *proj_beta;
    explicit AttnPack(Cursor& c) {
        qkv_wt = c.take(64 * 96);
        qkv_b = c.take(96);
        qkv_slope = c.take(12);
        qkv_gamma = c.take(96 * FQ);
        qkv_beta = c.take(96 * FQ);
        proj_wt = c.take(64 * 64);
        proj_b = c.take(64);
        proj_slope = c.take(1);
        proj_gamma = c.take(64 * FQ);
        proj_beta = c.take(64 * FQ);
    }
};
struct TfarPack {  // InjectionMultiSum: local_embedding, global_embedding, global_gate
    const float *loc_w, *loc_g, *loc_b, *emb_w, *emb_g, *emb_b, *gate_w, *gate_g, *gate_b;
    explicit TfarPack(Cursor& c) {
        loc_w = c.take(CH * 16);
        loc_g = c.take(CH);
        loc_b = c.take(CH);
        emb_w = c.take(CH * 16);
        emb_g = c.take(CH);
        emb_b = c.take(CH);
        gate_w = c.take(CH * 16);
        gate_g = c.take(CH);
        gate_b = c.take(CH);
    }
};
struct BlockPack {
    const float *gw, *gb, *gslope, *proj_wt, *proj_b;
    const float *ds0_w, *ds0_b, *ds0_g, *ds0_be, *ds1_w, *ds1_b, *ds1_g, *ds1_be;
    DpPack dpF, dpT;
    AttnPack attn;
    TfarPack fus0, fus1, cat0;
    const float *res_wt, *res_b, *proj_w16, *res_w16, *proj_w16_perm;
    static BlockPack make(Cursor& c, int rnn_kind = 0) {
        const float* gw = c.take(CA);
        const float* gb = c.take(CA);
        const float* gs = c.take(1);
        const float* pw = c.take(CA * CH);
        const float* pb = c.take(CH);
        const float* d[8];
        for (int i = 0; i < 2; ++i) {
            d[4 * i] = c.take(CH * 16);
            d[4 * i + 1] = c.take(CH);
            d[4 * i + 2] = c.take(CH);
            d[4 * i + 3] = c.take(CH);
        }
        DpPack f(c, rnn_kind), t(c, rnn_kind);
        AttnPack at(c);
        TfarPack f0(c), f1(c), c0(c);
        const float* rw = c.take(CH * CA);
        const float* rb = c.take(CA);
        const float* p16 = c.take(CA * CH);
        const float* r16 = c.take(CH * CA);
        const float* pp16 = c.take(CA * CH);
        return BlockPack{gw, gb, gs, pw, pb, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], f, t, at, f0, f1, c0, rw, rb, p16, r16, pp16};
    }
};
struct CafPack {
    const float *w_key, *bn_key, *w_val, *bn_val, *w_att, *b_att, *g_att, *be_att, *w_resize, *b_resize, *g_resize, *be_resize;
    explicit CafPack(Cursor& c) {
        w_key = c.take(CA);
        bn_key = c.take(4 * CA);
        w_val = c.take(CA);
        bn_val = c.take(4 * CA);
        w_att = c.take(1024 * 2);
        b_att = c.take(1024);
        g_att = c.take(1024);
        be_att = c.take(1024);
        w_resize = c.take(CA * 2);
        b_resize = c.take(CA);
        g_resize = c.take(CA);
        be_resize = c.take(CA);
    }
};
struct S3Pack {
    const float *slope, *wt, *bias, *w16;
    explicit S3Pack(Cursor& c) {
        slope = c.take(1);
        wt = c.take(CA * CA);
        bias = c.take(CA);
        w16 = c.take(CA * CA);
    }
};
struct DecPack {
    const float *wt, *w16;  // (256, 32): 18 tap maps (o*3+dt)*3+df, zero padded to 32; and its f16 split image
    const float* w16p;      // the same maps with K in accumulator-register order (S3 + taps kernel)
    explicit DecPack(Cursor& c) {
        wt = c.take(CA * 32);
        w16 = c.take(CA * 32);
        w16p = c.take(8192);
    }
};

// RTFS_GEMM_F32=1 selects the exact-f32 MFMA kernels instead of the f16x3 split-precision ones (A/B + debugging).
inline bool gemm_f32() {
    static const bool v = [] {
        const char* e = getenv("RTFS_GEMM_F32");
        return e && e[0] == '1';
    }();
    return v;
}

// ---------------------------------------------------------------- dual path
DpArgs dp_args(const DpPack& p, const float* x, float* out, int R, int Ls, size_t bstride, size_t rstride, size_t cstride) {
    DpArgs a;
    a.x = x;
    a.out = out;
    a.R = R;
    a.Ls = Ls;
    a.bstride = bstride;
    a.rstride = rstride;
    a.cstride = cstride;
    a.ln_gamma = p.ln_g;
    a.ln_beta = p.ln_b;
    a.W0 = p.W0;
    a.Wl = p.Wl;
    a.wc = p.wc;
    a.bias = p.bias;
    a.Wt = p.Wt;
    a.bt = p.bt;
    a.whh = p.whh;
    return a;
}

Dp16Args dp16_args(const DpPack& p, const float* x, float* out, int nseq, int R, int Ls, size_t bstride, size_t rstride, size_t cstride) {
    Dp16Args a;
    a.x = x;
    a.out = out;
    a.nseq = nseq;
    a.R = R;
    a.Ls = Ls;
    a.bstride = bstride;
    a.rstride = rstride;
    a.cstride = cstride;
    a.ln_gamma = p.ln_g;
    a.ln_beta = p.ln_b;
    a.w16_l0 = reinterpret_cast<const half8*>(p.w16_l0);
    a.w16_l = reinterpret_cast<const half8*>(p.w16_l);
    a.w16_ct = reinterpret_cast<const half8*>(p.w16_ct);
    a.wf_l0 = reinterpret_cast<const half8*>(p.wf_l0);
    a.wf_l = reinterpret_cast<const half8*>(p.wf_l);
    a.wf_ct = reinterpret_cast<const half8*>(p.wf_ct);
    a.wc16 = p.wc16;
    a.bias16 = p.bias16;
    a.bt = p.bt;
    return a;
}

// x, out (B,64,T,F).  dim 4: sequences along F, rows (b,t).  dim 3: along T via two tiled transposes.
int dualpath(const DpPack& p, const float* x, float* out, int B, int T, int F, int dim, float* tA, float* tB, hipStream_t st) {
    const size_t plane = (size_t)T * F;
    if (dim == 4) {
        RTFS_RETURN_IF(F < 8 || F > 250, RTFS_ERR_SHAPE);
        if (!gemm_f32() && !p.whh) return launch_dualpath16(dp16_args(p, x, out, B * T, T, F, CH * plane, F, plane), st);
        return launch_dualpath(dp_args(p, x, out, T, F, CH * plane, F, plane), B * T, st);
    }
    RTFS_RETURN_IF(T < 8 || T > 250, RTFS_ERR_SHAPE);
    CHECK(launch_transpose(x, tA, B * CH, T, F, st));
    if (!gemm_f32() && !p.whh) {
        CHECK(launch_dualpath16(dp16_args(p, tA, tB, B * F, F, T, CH * plane, T, plane), st));
    } else {
        CHECK(launch_dualpath(dp_args(p, tA, tB, F, T, CH * plane, T, plane), B * F, st));
    }
    return launch_transpose(tB, out, B * CH, F, T, st);
}

// ---------------------------------------------------------------- attention
int attention(const AttnPack& p, const float* x, float* out, int B, int T, float* q, float* k, float* v, float* o, hipStream_t st) {
    RowCanArgs a;
    a.x = x;
    a.wt = p.qkv_wt;
    a.bias = p.qkv_b;
    a.slope = p.qkv_slope;
    a.gamma = p.qkv_gamma;
    a.beta = p.qkv_beta;
    a.ngroups = 12;
    // channel order: Q_h (4 each), K_h (4 each), V_h (16 each)
    for (int g = 0; g <= 12; ++g) a.group_start[g] = g <= 8 ? 4 * g : 32 + 16 * (g - 8);
    for (int g = 0; g < 12; ++g)
        for (int o_ = a.group_start[g]; o_ < a.group_start[g + 1]; ++o_) a.group_of[o_] = (unsigned char)g;
    a.T = T;
    a.q = q;
    a.k = k;
    a.v = v;
    CHECK(launch_row_can_qkv(a, B, st));
    AttnArgs c;
    c.q = q;
    c.k = k;
    c.v = v;
    c.out = o;
    c.T = T;
    c.scale = 1.0f / 16.0f;  // 1/sqrt(E*F) = 1/sqrt(4*64)
    CHECK(launch_attn_core(c, B, st));
    RowCanArgs r;
    r.x = o;
    r.wt = p.proj_wt;
    r.bias = p.proj_b;
    r.slope = p.proj_slope;
    r.gamma = p.proj_gamma;
    r.beta = p.proj_beta;
    r.ngroups = 1;
    r.group_start[0] = 0;
    r.group_start[1] = 64;
    r.T = T;
    r.res = x;
    r.out = out;
    return launch_row_can_proj(r, B, st);
}

// ---------------------------------------------------------------- RTFS block
struct BlockWs {
    float *residual, *x_enc, *c0, *xf0, *expanded;                       // full resolution
    float *c1, *p0, *g, *gF, *tA, *tB, *gT, *gA, *q, *k, *v, *o;         // compressed resolution
    float *E0, *G0, *E1, *G1, *L1, *xf1, *E2, *G2;
    double* stats;  // 11 slots x (B,2)
    int Bfull = 0;  // batch size the statistic slots are laid out for
    int cs = 0;     // channel stride (floats) of the full-resolution tensors: T * F, or padded to whole 128-byte lines (pitch())
    static constexpr int NSTAT = 11;
    enum { S_C0, S_C1, S_E0, S_G0, S_E1, S_G1, S_L1, S_E2, S_G2, S_L0, S_L2 };
    BlockWs(Arena& a, int B, int T, int F, int cs_ = 0) {
        const size_t Pg = (size_t)(T / 2) * (F / 2);
        cs = cs_ ? cs_ : T * F;
        const size_t P = (size_t)cs;
        Bfull = B;
        residual = a.take<float>(B * CA * P);
        x_enc = a.take<float>(B * CH * P);
        c0 = a.take<float>(B * CH * P);
        xf0 = a.take<float>(B * CH * P);
        expanded = a.take<float>(B * CH * P);
        float** gs[] = {&c1, &p0, &g, &gF, &tA, &tB, &gT, &gA, &v, &o, &E0, &G0, &E1, &G1, &L1, &xf1, &E2, &G2};
        for (float** s : gs) *s = a.take<float>(B * CH * Pg);
        q = a.take<float>(B * CH * Pg / 4);
        k = a.take<float>(B * CH * Pg / 4);
        stats = a.take<double>((size_t)NSTAT * B * 2);
    }
    double* st(int slot, int /*B*/) const { return stats ? stats + (size_t)slot * Bfull * 2 : nullptr; }
};

int block_head(const BlockPack& p, const float* x, const float* x_res, int B, int T, int F, BlockWs& w, hipStream_t st, const CafArgs* caf, unsigned* ctr = nullptr) {
    const int P = T * F;
    {  // 1. gateway (dw 1x1 + PReLU) -> residual; projection 1x1 256->64 -> x_enc          tdanet.py:106-107
        PwArgs a;
        a.x = x;
        a.x2 = x_res;
        a.res_out = w.residual;
        a.wt = p.proj_wt;
        a.w16 = p.proj_w16;
        a.bias = p.proj_b;
        a.out = w.x_enc;
        a.gw = p.gw;
        a.gb = p.gb;
        a.slope = p.gslope;
        a.P = P;
        a.cs = w.cs;
        a.tile_ctr = ctr;
        if (caf && !gemm_f32()) {  // block input = CAF(x, video) + x_res, applied while streaming x (fused separator path)
            a.caf_r = caf->r_out; a.caf_att = caf->att_out;
            a.caf_w_key = caf->w_key; a.caf_bn_key = caf->bn_key; a.caf_w_val = caf->w_val; a.caf_bn_val = caf->bn_val;
            a.caf_T = caf->T; a.caf_F = caf->F; a.caf_Tv = caf->Tv;
        }
        if (gemm_f32()) return launch_pw_gateway_proj(a, B, st);
        const int rc = launch_pws_head4(a, B, st);  // padded rows, plain input: the ring kernel
        if (rc == RTFS_ERR_ARG) CHECK(launch_pws_gateway_proj(a, B, st));
        else CHECK(rc);
    }
    return RTFS_OK;
}

// steps 2-17: everything between the projection (x_enc, residual in the workspace) and `expanded`
int block_body(const BlockPack& p, int B, int T, int F, const BlockWs& w, hipStream_t st, bool zero_stats = true) {
    const int Tp = T / 2, Fp = F / 2;
    const int P = T * F, Pg = Tp * Fp;
    const double icF = 1.0 / ((double)CH * P), icG = 1.0 / ((double)CH * Pg);
    typedef BlockWs W;
    // Band heights (rows a workgroup walks).  The read + write passes at full resolution take 24-row bands: 64-row bands are 2176 workgroups for
    // 1024 resident ones, i.e. a third round of 128 workgroups on an otherwise idle chip (same-box A/B: step 2 162 -> 153 us, 15 172 -> 162,
    // 17 213 -> 204, stride-2 pass 118 -> 103 with 12 output rows; 10.51 -> 10.35 ms per forward).  The statistics passes (one pair of f64 atomics
    // per workgroup) and the low-resolution launches are better off with 64 (measured: 62 -> 65 / 70 us and 122 -> 131 us with shorter bands).
    constexpr int TH_A = 24, TH_B = 24, TH_C = 12, TH_G = 64, TH_S = 64;
    // (the separator gives every block application its own, already zeroed slots: one memset per call instead of one per block - 6 us each on
    // the chain, 1 % of a batch-1 forward)
    if (zero_stats && hipMemsetAsync(w.stats, 0, sizeof(double) * W::NSTAT * w.Bfull * 2, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    {  // 2. downsample[0]: dw 4x4 s1 + bias -> c0 (pre-gLN) + stats                         tdanet.py:110
        DwArgs a;
        a.x = w.x_enc;
        a.w[0] = p.ds0_w;
        a.bias[0] = p.ds0_b;
        a.out[0] = w.c0;
        a.stats_out[0] = w.st(W::S_C0, B);
        a.C = CH; a.H = T; a.W = F; a.TH = TH_A; a.cs = w.cs;
        CHECK(launch_dw_s1(a, 1, false, 0, B, st));
    }
    // Step 14 (statistics of fusion 0's local conv on d0: one full-resolution read, HBM-bound) needs only c0 and its statistics.  Round 2 ran it
    // on a side stream beside steps 3-13 (14.43 -> 14.25 ms per forward); since round 3 it shares step 3's launch (below).
    DwArgs s14;   // step 14: statistics of fusion 0's local conv on d0
    s14.x = w.c0;
    s14.in_stats = w.st(W::S_C0, B); s14.in_inv_count = icF; s14.in_gamma = p.ds0_g; s14.in_beta = p.ds0_be;
    s14.C = CH; s14.H = T; s14.W = F; s14.TH = 64; s14.Hg = Tp; s14.Wg = Fp; s14.cs = w.cs;
    s14.w[0] = p.fus0.loc_w;
    s14.stats_out[0] = w.st(W::S_L0, B);
    s14.rev = 1;  // (as step 3, which it runs beside)
    DwArgs s3;  // 3. downsample[1] on d0 = gLN(c0): dw 4x4 s2 -> c1 + stats; p0 = adaptive_avg_pool2d(d0)   tdanet.py:111-116
    s3.x = w.c0;
    s3.in_stats = w.st(W::S_C0, B); s3.in_inv_count = icF; s3.in_gamma = p.ds0_g; s3.in_beta = p.ds0_be;
    s3.w[0] = p.ds1_w;
    s3.bias[0] = p.ds1_b;
    s3.out[0] = w.c1;
    s3.out[1] = w.p0;
    s3.stats_out[0] = w.st(W::S_C1, B);
    s3.C = CH; s3.H = T; s3.W = F; s3.Hg = Tp; s3.Wg = Fp; s3.TH = TH_C; s3.cs = w.cs;
    // Traversal order against the memory-side cache (256 MB; a 64-channel tensor is 265 MB at batch 32): step 2 wrote c0 front to back, so
    // its END is what is still on the die - this pass walks back to front (tools/bench_mall.hip: a read-only consumer of a 265 MB tensor
    // 59 -> 44 us; here 125 -> 112 us).  Likewise step 16 after 15 (79 -> 62 us) and step 17 after 16.  10.39 -> 10.17 ms per forward.
    s3.rev = 1;
    // Steps 3 and 14 read the same tensor: one launch, the two jobs interleaved per sample (launch_dw_s2_stats; the block's frequency axis is
    // tied to 2 x 64 by the attention's LayerNorm, so the launcher's width conditions always hold here).
    // (same-box A/B against the side-stream form: batch 32 9.81 -> 9.74 ms with the F sweep 295 -> 289 us - nothing runs beside it any more -
    // and `g_form` 66 -> 33 us; batch 1 1.46 -> 1.42 ms, two event pairs per block fewer)
    {
        const int rc = launch_dw_s2_stats(s3, s14, B, st);
        if (rc != RTFS_OK) return rc == RTFS_ERR_ARG ? RTFS_ERR_SHAPE : rc;
    }
    // 4. g = pool(d0) + d1
    CHECK(launch_g_form(w.p0, w.c1, w.st(W::S_C1, B), icG, p.ds1_g, p.ds1_be, w.g, B, CH, Pg, st));
    // 5-8. dual-path sweeps along F then T                                                  yaml layer_1 / layer_2
    CHECK(dualpath(p.dpF, w.g, w.gF, B, Tp, Fp, 4, w.tA, w.tB, st));
    CHECK(dualpath(p.dpT, w.gF, w.gT, B, Tp, Fp, 3, w.tA, w.tB, st));
    // 9. TF self-attention                                                                  yaml layer_3
    CHECK(attention(p.attn, w.gT, w.gA, B, Tp, w.q, w.k, w.v, w.o, st));
    {  // 10. the four G-level convs on the attention output (fusion 0/1: global_embedding, global_gate)
       // 11. fusion 1 local_embedding on d1 = gLN(c1) - independent of step 10: one launch for the three jobs
        DwArgs a;
        a.x = w.gA;
        a.w[0] = p.fus0.emb_w; a.w[1] = p.fus0.gate_w; a.w[2] = p.fus1.emb_w; a.w[3] = p.fus1.gate_w;
        a.out[0] = w.E0; a.out[1] = w.G0; a.out[2] = w.E1; a.out[3] = w.G1;
        a.stats_out[0] = w.st(W::S_E0, B); a.stats_out[1] = w.st(W::S_G0, B);
        a.stats_out[2] = w.st(W::S_E1, B); a.stats_out[3] = w.st(W::S_G1, B);
        a.C = CH; a.H = Tp; a.W = Fp; a.TH = TH_G;
        DwArgs l;
        l.x = w.c1;
        l.in_stats = w.st(W::S_C1, B); l.in_inv_count = icG; l.in_gamma = p.ds1_g; l.in_beta = p.ds1_be;
        l.w[0] = p.fus1.loc_w;
        l.out[0] = w.L1;
        l.stats_out[0] = w.st(W::S_L1, B);
        l.C = CH; l.H = Tp; l.W = Fp; l.TH = TH_G;
        const int rc = launch_dw_g3(a, l, B, st);
        if (rc == RTFS_ERR_ARG) {
            CHECK(launch_dw_s1(a, 4, false, 0, B, st));
            CHECK(launch_dw_s1(l, 1, true, 0, B, st));
        } else {
            CHECK(rc);
        }
    }
    {  // 12 + 13. xf1 = gLN(L1) * sigmoid(gLN(G1)) + gLN(E1) (fusion.py:62-67) is only ever read by the concat layer's two global convs:
       // they form it at load time, xf1 itself is never written
        DwArgs a;
        a.x = w.L1; a.gate = w.G1; a.emb = w.E1;
        a.in_combine = 1;
        a.loc_stats = w.st(W::S_L1, B); a.gate_stats = w.st(W::S_G1, B); a.emb_stats = w.st(W::S_E1, B);
        a.loc_gamma = p.fus1.loc_g; a.loc_beta = p.fus1.loc_b;
        a.gate_gamma = p.fus1.gate_g; a.gate_beta = p.fus1.gate_b;
        a.emb_gamma = p.fus1.emb_g; a.emb_beta = p.fus1.emb_b;
        a.g_inv_count = icG;
        a.w[0] = p.cat0.emb_w; a.w[1] = p.cat0.gate_w;
        a.out[0] = w.E2; a.out[1] = w.G2;
        a.stats_out[0] = w.st(W::S_E2, B); a.stats_out[1] = w.st(W::S_G2, B);
        a.C = CH; a.H = Tp; a.W = Fp; a.TH = TH_G;
        if (Fp >= 16) {
            CHECK(launch_dw_s1(a, 2, false, 0, B, st));
        } else {  // (narrow inputs run the scalar kernels: separate combination pass)
            GCombineArgs g;
            g.l = w.L1; g.gate = w.G1; g.emb = w.E1;
            g.l_stats = a.loc_stats; g.gate_stats = a.gate_stats; g.emb_stats = a.emb_stats;
            g.l_gamma = a.loc_gamma; g.l_beta = a.loc_beta; g.gate_gamma = a.gate_gamma; g.gate_beta = a.gate_beta; g.emb_gamma = a.emb_gamma; g.emb_beta = a.emb_beta;
            g.inv_count = icG;
            g.out = w.xf1;
            g.C = CH; g.HW = Pg;
            CHECK(launch_g_combine(g, B, st));
            a.x = w.xf1; a.gate = nullptr; a.emb = nullptr; a.in_combine = 0;
            CHECK(launch_dw_s1(a, 2, false, 0, B, st));
        }
    }
    DwArgs d0in;  // common: read d0 = gLN(c0) at full resolution
    d0in.x = w.c0;
    d0in.in_stats = w.st(W::S_C0, B); d0in.in_inv_count = icF; d0in.in_gamma = p.ds0_g; d0in.in_beta = p.ds0_be;
    d0in.C = CH; d0in.H = T; d0in.W = F; d0in.TH = 64; d0in.Hg = Tp; d0in.Wg = Fp; d0in.cs = w.cs;
    {  // 15. xf0 = gLN(conv(d0)) * sigmoid(gLN(G0))^ + gLN(E0)^                              fusion.py:58-67
        DwArgs a = d0in;
        a.w[0] = p.fus0.loc_w;
        a.loc_stats = w.st(W::S_L0, B); a.loc_inv_count = icF; a.loc_gamma = p.fus0.loc_g; a.loc_beta = p.fus0.loc_b;
        a.gate = w.G0; a.gate_stats = w.st(W::S_G0, B); a.gate_gamma = p.fus0.gate_g; a.gate_beta = p.fus0.gate_b;
        a.emb = w.E0; a.emb_stats = w.st(W::S_E0, B); a.emb_gamma = p.fus0.emb_g; a.emb_beta = p.fus0.emb_b;
        a.g_inv_count = icG;
        a.out[0] = w.xf0;
        a.TH = TH_B;
        CHECK(launch_dw_s1(a, 1, true, 2, B, st));
    }
    DwArgs xin;  // common: read xf0
    xin.x = w.xf0;
    xin.C = CH; xin.H = T; xin.W = F; xin.TH = 64; xin.Hg = Tp; xin.Wg = Fp; xin.cs = w.cs;
    {  // 16. concat layer local_embedding conv on xf0: statistics only
        DwArgs a = xin;
        a.w[0] = p.cat0.loc_w;
        a.stats_out[0] = w.st(W::S_L2, B);
        a.rev = 1;  // xf0 was just written front to back by step 15 (see step 3)
        a.TH = TH_S;
        CHECK(launch_dw_s1(a, 1, false, 1, B, st));
    }
    {  // 17. expanded = InjectionMultiSum(xf0, xf1) + d0                                     tdanet.py:125
        DwArgs a = xin;
        a.w[0] = p.cat0.loc_w;
        a.loc_stats = w.st(W::S_L2, B); a.loc_inv_count = icF; a.loc_gamma = p.cat0.loc_g; a.loc_beta = p.cat0.loc_b;
        a.gate = w.G2; a.gate_stats = w.st(W::S_G2, B); a.gate_gamma = p.cat0.gate_g; a.gate_beta = p.cat0.gate_b;
        a.emb = w.E2; a.emb_stats = w.st(W::S_E2, B); a.emb_gamma = p.cat0.emb_g; a.emb_beta = p.cat0.emb_b;
        a.g_inv_count = icG;
        a.addend = w.c0; a.add_stats = w.st(W::S_C0, B); a.add_inv_count = icF; a.add_gamma = p.ds0_g; a.add_beta = p.ds0_be;
        a.out[0] = w.expanded;
        a.rev = 1;
        a.TH = TH_B;
        CHECK(launch_dw_s1(a, 1, false, 2, B, st));
    }
    return RTFS_OK;
}

int block_tail(const BlockPack& p, float* out, int B, int T, int F, BlockWs& w, hipStream_t st, unsigned* ctr = nullptr) {
    const int P = T * F;
    {  // 18. out = residual_conv(expanded) + residual                                        tdanet.py:129
        PwArgs a;
        a.x = w.expanded;
        a.wt = p.res_wt;
        a.w16 = p.res_w16;
        a.bias = p.res_b;
        a.aux = w.residual;
        a.out = out;
        a.P = P;
        a.cs = w.cs;
        a.tile_ctr = ctr;
        CHECK(gemm_f32() ? launch_pw_residual(a, B, st) : launch_pws_residual(a, B, st));
    }
    return RTFS_OK;
}

int block_forward(const BlockPack& p, const float* x, const float* x_res, float* out, int B, int T, int F, BlockWs& w, hipStream_t st,
                  const CafArgs* caf = nullptr) {
    CHECK(block_head(p, x, x_res, B, T, F, w, st, caf));
    CHECK(block_body(p, B, T, F, w, st));
    return block_tail(p, out, B, T, F, w, st);
}

// block boundary of the fused separator: residual_conv(i) + [CAF] + a1 + gateway + projection(i+1) in one kernel
// res_has_a1 (in / out): whether w.residual currently holds residual + a1.  A boundary that has to read a1 anyway also adds it to the residual
// it writes (unless it is the last one: the tail wants the plain residual), and the boundary after it then does not read a1 at all.
int block_boundary(const BlockPack& p, const float* a1, int B, int T, int F, BlockWs& w, hipStream_t st, const CafArgs* caf, unsigned* ctr = nullptr,
                   bool* res_has_a1 = nullptr, bool last = true, bool res_from_a1 = false) {
    B2bArgs a;
    a.x = w.expanded;
    a.res = w.residual;
    a.a1 = a1;
    a.xenc = w.x_enc;
    a.w1_16 = p.res_w16;
    a.b1 = p.res_b;
    a.w2_16 = p.proj_w16_perm;
    a.bp = p.proj_b;
    a.gw = p.gw;
    a.gb = p.gb;
    a.slope = p.gslope;
    a.P = T * F;
    a.cs = w.cs;
    if (caf) {
        a.caf_r = caf->r_out; a.caf_att = caf->att_out;
        a.caf_rt = caf->r_t; a.caf_attt = caf->att_t;
        a.caf_w_key = caf->w_key; a.caf_bn_key = caf->bn_key; a.caf_w_val = caf->w_val; a.caf_bn_val = caf->bn_val;
        a.caf_T = caf->T; a.caf_F = caf->F; a.caf_Tv = caf->Tv;
    }
    if (res_has_a1 && launch_pws_b2b4_qualifies(a)) {
        a.a1_mode = *res_has_a1 ? 0 : (last ? 1 : 3);
        *res_has_a1 = a.a1_mode == 3;
        if (res_from_a1 && caf) a.a1_mode |= 4;  // residual_0 was not written by the head: it is the gateway of a1
    } else if (res_from_a1) {
        return RTFS_ERR_ARG;  // (cannot happen: the head kernel that skipped the residual qualifies on the same conditions)
    }
    const int rc = launch_pws_b2b4(a, B, ctr, st);  // padded rows: the pipelined kernel (k_b2b.hip)
    return rc == RTFS_ERR_ARG ? launch_pws_b2b(a, B, st) : rc;
}

int audio_bn(const BnPack& p, const float* x, const double* stats, float* out, int B, int P, hipStream_t st, int cs = 0, unsigned* ctr = nullptr) {
    PwArgs a;
    a.x = x;
    a.wt = p.wt;
    a.bias = p.bias;
    a.out = out;
    a.stats = stats;
    a.inv_count = 1.0 / ((double)CA * P);
    a.gamma = p.gamma;
    a.beta = p.beta;
    a.P = P;
    a.cs = cs;
    a.tile_ctr = ctr;
    a.w16 = p.w16;
    return gemm_f32() ? launch_pw_audio_bn(a, B, st) : launch_pwr_audio_bn(a, B, st);
}

CafArgs caf_args(const CafPack& p, const float* audio, const float* video, float* out, float* r, float* att, int T, int F, int Tv) {
    CafArgs a;
    a.audio = audio; a.video = video; a.out = out; a.r_out = r; a.att_out = att;
    a.T = T; a.F = F; a.Tv = Tv;
    a.w_key = p.w_key; a.bn_key = p.bn_key; a.w_val = p.w_val; a.bn_val = p.bn_val;
    a.w_att = p.w_att; a.b_att = p.b_att; a.g_att = p.g_att; a.be_att = p.be_att;
    a.w_resize = p.w_resize; a.b_resize = p.b_resize; a.g_resize = p.g_resize; a.be_resize = p.be_resize;
    return a;
}

int s3_mask(const S3Pack& p, const float* refined, const float* a0, float* out, int B, int P, hipStream_t st) {
    PwArgs a;
    a.x = refined;
    a.wt = p.wt;
    a.bias = p.bias;
    a.aux = a0;
    a.out = out;
    a.slope = p.slope;
    a.P = P;
    a.w16 = p.w16;
    return gemm_f32() ? launch_pw_s3(a, B, st) : launch_pwr_s3(a, B, st);
}

int decoder(const DecPack& p, const float* x, float* wav, float* z, int B, int T, int L, hipStream_t st) {
    PwArgs a;
    a.x = x;
    a.wt = p.wt;
    a.out = z;
    a.P = T * NF;
    a.cout_live = 18;
    a.w16 = p.w16;
    CHECK(gemm_f32() ? launch_pw_dec_taps(a, B, st) : launch_pw16_dec_taps(a, B, st));
    return launch_dec_istft(z, wav, B, T, NF, L, (size_t)T * NF, (size_t)18 * T * NF, st);
}

template <class PackT>
size_t pack_size() {
    Cursor c(nullptr);
    PackT p(c);
    (void)p;
    return c.off;
}

bool shape_ok_block(int B, int T, int F) { return B >= 1 && F / 2 == FQ && T / 2 >= 8 && T / 2 <= 250; }

}  // namespace

extern "C" {

const char* rtfs_version(void) { return "rtfs_amd 0.1 gfx950"; }

int rtfs_num_frames(int L) { return 1 + L / 128; }

size_t rtfs_pack_floats(int kind) {
    switch (kind) {
        case RTFS_PACK_ENCODER: return pack_size<EncPack>();
        case RTFS_PACK_AUDIO_BN: return pack_size<BnPack>();
        case RTFS_PACK_BLOCK: {
            Cursor c(nullptr);
            BlockPack::make(c);
            return c.off;
        }
        case RTFS_PACK_DUALPATH: return pack_size<DpPack>();
        case RTFS_PACK_DUALPATH_LSTM: {
            Cursor c(nullptr);
            DpPack p(c, 1);
            (void)p;
            return c.off;
        }
        case RTFS_PACK_BLOCK_LSTM: {
            Cursor c(nullptr);
            BlockPack::make(c, 1);
            return c.off;
        }
        case RTFS_PACK_ATTENTION: return pack_size<AttnPack>();
        case RTFS_PACK_TFAR: return pack_size<TfarPack>();
        case RTFS_PACK_CAF: return pack_size<CafPack>();
        case RTFS_PACK_S3: return pack_size<S3Pack>();
        case RTFS_PACK_DECODER: return pack_size<DecPack>();
    }
    return 0;
}

// ------------------------------------------------------------ encoder
size_t rtfs_stft_encoder_workspace_bytes(int B, int L) { return (size_t)B * 2 * rtfs_num_frames(L) * NF * sizeof(float) + 256; }

int rtfs_stft_encoder_f32(const float* wav, const float* pack, float* a0, double* stats, int B, int L, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!wav || !pack || !a0 || B < 1 || L <= 128, RTFS_ERR_ARG);
    const int T = rtfs_num_frames(L);
    Arena ar(ws, ws_bytes);
    float* spec = ar.take<float>((size_t)B * 2 * T * NF);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    EncPack p(c);
    if (stats && hipMemsetAsync(stats, 0, sizeof(double) * 2 * B, S(stream)) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_stft(wav, spec, B, L, T, S(stream)));
    return launch_enc_conv(spec, p.w, a0, stats, B, CA, T, NF, (size_t)T * NF, (size_t)CA * T * NF, S(stream));
}

// ------------------------------------------------------------ audio bottleneck
size_t rtfs_audio_bottleneck_workspace_bytes(int B) { return sizeof(double) * 2 * B + 256; }

int rtfs_audio_bottleneck_f32(const float* x, const double* stats, const float* pack, float* out, int B, int T, int F, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !out || B < 1 || T < 1 || F < 1, RTFS_ERR_ARG);
    Cursor c(pack);
    BnPack p(c);
    if (!stats) {
        Arena ar(ws, ws_bytes);
        double* s = ar.take<double>(2 * B);
        RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
        if (hipMemsetAsync(s, 0, sizeof(double) * 2 * B, S(stream)) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_stats(x, s, B, (size_t)CA * T * F, S(stream)));
        stats = s;
    }
    return audio_bn(p, x, stats, out, B, T * F, S(stream));
}

// ------------------------------------------------------------ RTFS block
size_t rtfs_block_workspace_bytes(int B, int T, int F) {
    Arena ar(nullptr, 0);
    BlockWs w(ar, B, T, F);
    return ar.off + 256;
}

int rtfs_block_f32(const float* x, const float* x_res, const float* pack, float* out, int B, int T, int F, void* ws, size_t ws_bytes, void* stream, int rnn_kind) {
    RTFS_RETURN_IF(!x || !pack || !out, RTFS_ERR_ARG);
    RTFS_RETURN_IF(!shape_ok_block(B, T, F), RTFS_ERR_SHAPE);
    Arena ar(ws, ws_bytes);
    BlockWs w(ar, B, T, F);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    RTFS_RETURN_IF(rnn_kind != 0 && rnn_kind != 1, RTFS_ERR_ARG);
    Cursor c(pack);
    BlockPack p = BlockPack::make(c, rnn_kind);
    return block_forward(p, x, x_res, out, B, T, F, w, S(stream));
}

// ------------------------------------------------------------ dual path
size_t rtfs_dualpath_workspace_bytes(int B, int T, int F) { return 2 * ((size_t)B * CH * T * F * sizeof(float) + 256); }

int rtfs_dualpath_lstm_f32(const float* x, const float* pack, float* out, int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !out || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    Arena ar(ws, ws_bytes);
    float* tA = ar.take<float>((size_t)B * CH * T * F);
    float* tB = ar.take<float>((size_t)B * CH * T * F);
    RTFS_RETURN_IF(dim == 3 && (!ws || !ar.ok()), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    DpPack p(c, 1);
    return dualpath(p, x, out, B, T, F, dim, tA, tB, S(stream));
}

int rtfs_dualpath_sru_f32(const float* x, const float* pack, float* out, int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !out || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    Arena ar(ws, ws_bytes);
    float* tA = ar.take<float>((size_t)B * CH * T * F);
    float* tB = ar.take<float>((size_t)B * CH * T * F);
    RTFS_RETURN_IF(dim == 3 && (!ws || !ar.ok()), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    DpPack p(c);
    return dualpath(p, x, out, B, T, F, dim, tA, tB, S(stream));
}

// ------------------------------------------------------------ TF attention
size_t rtfs_tf_attention_workspace_bytes(int B, int T) { return (size_t)B * T * (4 * 256 + 4 * 256 + 4 * 1024 + 64 * FQ) * sizeof(float) + 4 * 256; }

int rtfs_tf_attention_f32(const float* x, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !out || B < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(T < 1 || T > 256, RTFS_ERR_SHAPE);
    Arena ar(ws, ws_bytes);
    float* q = ar.take<float>((size_t)B * 4 * T * 256);
    float* k = ar.take<float>((size_t)B * 4 * T * 256);
    float* v = ar.take<float>((size_t)B * 4 * T * 1024);
    float* o = ar.take<float>((size_t)B * 64 * T * FQ);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    AttnPack p(c);
    return attention(p, x, out, B, T, q, k, v, o, S(stream));
}

// ------------------------------------------------------------ TFAR
size_t rtfs_tfar_workspace_bytes(int B, int H, int W, int Hg, int Wg) {
    return 3 * ((size_t)B * CH * (size_t)(H * W > Hg * Wg ? Hg * Wg : H * W) * sizeof(float) + 256) + 3 * 2 * B * sizeof(double) + 256;
}

int rtfs_tfar_f32(const float* local, const float* global, const float* pack, float* out, int B, int H, int W, int Hg, int Wg, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!local || !global || !pack || !out || B < 1, RTFS_ERR_ARG);
    const bool up = (size_t)H * W > (size_t)Hg * Wg;
    RTFS_RETURN_IF(!up && (H != Hg || W != Wg), RTFS_ERR_SHAPE);  // the path only ever has equal-size or larger local maps
    Arena ar(ws, ws_bytes);
    const size_t ng = (size_t)B * CH * Hg * Wg;
    float* E = ar.take<float>(ng);
    float* G = ar.take<float>(ng);
    float* Lc = ar.take<float>(ng);
    double* st = ar.take<double>(3 * 2 * B);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    TfarPack p(c);
    hipStream_t s = S(stream);
    if (hipMemsetAsync(st, 0, sizeof(double) * 6 * B, s) != hipSuccess) return RTFS_ERR_LAUNCH;
    double *stE = st, *stG = st + 2 * B, *stL = st + 4 * B;
    const double icG = 1.0 / ((double)CH * Hg * Wg), icL = 1.0 / ((double)CH * H * W);
    {
        DwArgs a;
        a.x = global;
        a.w[0] = p.emb_w; a.w[1] = p.gate_w;
        a.out[0] = E; a.out[1] = G;
        a.stats_out[0] = stE; a.stats_out[1] = stG;
        a.C = CH; a.H = Hg; a.W = Wg; a.TH = 64;
        CHECK(launch_dw_s1(a, 2, false, 0, B, s));
    }
    if (up) {
        DwArgs a;
        a.x = local;
        a.w[0] = p.loc_w;
        a.C = CH; a.H = H; a.W = W; a.TH = 64; a.Hg = Hg; a.Wg = Wg;
        a.stats_out[0] = stL;
        CHECK(launch_dw_s1(a, 1, false, 1, B, s));
        a.stats_out[0] = nullptr;
        a.loc_stats = stL; a.loc_inv_count = icL; a.loc_gamma = p.loc_g; a.loc_beta = p.loc_b;
        a.gate = G; a.gate_stats = stG; a.gate_gamma = p.gate_g; a.gate_beta = p.gate_b;
        a.emb = E; a.emb_stats = stE; a.emb_gamma = p.emb_g; a.emb_beta = p.emb_b;
        a.g_inv_count = icG;
        a.out[0] = out;
        return launch_dw_s1(a, 1, false, 2, B, s);
    }
    {
        DwArgs a;
        a.x = local;
        a.w[0] = p.loc_w;
        a.out[0] = Lc;
        a.stats_out[0] = stL;
        a.C = CH; a.H = H; a.W = W; a.TH = 64;
        CHECK(launch_dw_s1(a, 1, false, 0, B, s));
    }
    GCombineArgs a;
    a.l = Lc; a.gate = G; a.emb = E;
    a.l_stats = stL; a.gate_stats = stG; a.emb_stats = stE;
    a.l_gamma = p.loc_g; a.l_beta = p.loc_b; a.gate_gamma = p.gate_g; a.gate_beta = p.gate_b; a.emb_gamma = p.emb_g; a.emb_beta = p.emb_b;
    a.inv_count = icG;
    a.out = out;
    a.C = CH; a.HW = H * W;
    return launch_g_combine(a, B, s);
}

// ------------------------------------------------------------ CAF
size_t rtfs_caf_workspace_bytes(int B, int Tv) { return 2 * ((size_t)B * CA * Tv * sizeof(float) + 256); }

int rtfs_caf_f32(const float* audio, const float* video, const float* pack, float* out, int B, int T, int F, int Tv, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!audio || !video || !pack || !out || B < 1 || T < 1 || F < 1 || Tv < 1, RTFS_ERR_ARG);
    Arena ar(ws, ws_bytes);
    float* r = ar.take<float>((size_t)B * CA * Tv);
    float* att = ar.take<float>((size_t)B * CA * Tv);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    CafPack p(c);
    CafArgs a = caf_args(p, audio, video, out, r, att, T, F, Tv);
    CHECK(launch_caf_video(a, B, S(stream)));
    return launch_caf_apply(a, B, S(stream));
}

// ------------------------------------------------------------ VP block
size_t rtfs_vp_pack_floats(void) {
    size_t n = 0;
    auto t = [&](size_t k) { n += (k + 63) / 64 * 64; };
    t(512); t(512); t(1); t(512 * 64); t(64);
    for (int i = 0; i < 4; ++i) { t(192); t(64); t(64); }
    t(64); t(64); t(16 * 64); t(192 * 64); t(192); t(64 * 64); t(64); t(64); t(64);
    t(128 * 64); t(128); t(128); t(128 * 3); t(128); t(64 * 128); t(64); t(64);
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 3; ++j) { t(192); t(64); t(64); }
    t(64 * 512); t(512);
    return n;
}

int rtfs_vp_block_f32(const float* video, const float* pack, float* out, int B, int Tv, void* stream) {
    RTFS_RETURN_IF(!video || !pack || !out || B < 1, RTFS_ERR_ARG);
    return launch_vp_block(video, pack, out, B, Tv, S(stream));
}

// ------------------------------------------------------------ S^3
int rtfs_s3_mask_f32(const float* refined, const float* a0, const float* pack, float* out, int B, int T, int F, void* stream) {
    RTFS_RETURN_IF(!refined || !a0 || !pack || !out || B < 1 || T < 1 || F < 1, RTFS_ERR_ARG);
    Cursor c(pack);
    S3Pack p(c);
    return s3_mask(p, refined, a0, out, B, T * F, S(stream));
}

// ------------------------------------------------------------ decoder
size_t rtfs_istft_decoder_workspace_bytes(int B, int T) { return (size_t)B * 18 * T * NF * sizeof(float) + 256; }

int rtfs_istft_decoder_f32(const float* x, const float* pack, float* wav, int B, int T, int L, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !wav || B < 1 || T < 1 || L < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(L > 128 * T + 127, RTFS_ERR_SHAPE);  // every output sample needs at least one frame
    Arena ar(ws, ws_bytes);
    float* z = ar.take<float>((size_t)B * 18 * T * NF);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    Cursor c(pack);
    DecPack p(c);
    return decoder(p, x, wav, z, B, T, L, S(stream));
}

// ------------------------------------------------------------ whole separator
namespace {
// Channel stride of the separator's full-resolution tensors.  T * F = T * 129 floats is odd, so in a contiguous (B, C, T, F) tensor every
// channel row starts somewhere inside a 128-byte line and every 64-pixel wave segment of the pointwise kernels writes two PARTIAL lines;
// partial lines whose halves come from workgroups on different XCDs cannot merge in an L2 and reach memory as masked writes.  Measured
// (tools/bench_stream5.hip, row-walk pattern of the pointwise kernels): stores 3.35 TB/s at pitch 32379, 5.5 TB/s at 32384; two reads +
// one write 4.55 -> 5.35 TB/s.  All of these tensors are workspace (never seen by the caller), so their rows are padded - to a multiple of 64 floats, so that
// every 64-pixel wave segment also lies INSIDE its row's allocation and the padded-row kernels need no bounds predicates (k_b2b.hip).
inline int pitch(int P) { return (P + 63) / 64 * 64; }

struct SepWs {
    float *spec, *a0, *a1, *cur, *nxt, *r, *att, *rt, *attt, *z;
    float* encimg;  // encoder f16x3 fragment image (32 KB, written by enc_stats_kernel of every call)
    float* wpad;    // mask-conv weight image with 72-byte rows (LDS-DMA source of the tail kernel, 288 KB; written by the same kernel)
    float* wpad0;   // the same for the bottleneck weights (head kernel)
    double* st0;
    unsigned* ctr;  // 64 tile counters (one per persistent launch of the call), zeroed together with st0
    double* bstats;  // statistic slots of up to STAT_APPS block applications (BlockWs::NSTAT x (B, 2) each), zeroed together with st0
    static constexpr int STAT_APPS = 16;
    int B_ = 0;
    int cs;
    BlockWs blk;
    SepWs(Arena& a, int B, int T, int Tv, int cs_)
        : spec(a.take<float>((size_t)B * 2 * T * NF)),
          a0(a.take<float>((size_t)B * CA * cs_)),
          a1(a.take<float>((size_t)B * CA * cs_)),
          cur(a.take<float>((size_t)B * CA * cs_)),
          nxt(a.take<float>((size_t)B * CA * cs_)),
          r(a.take<float>((size_t)B * CA * Tv)),
          att(a.take<float>((size_t)B * CA * Tv)),
          rt(a.take<float>((size_t)B * CA * Tv)),
          attt(a.take<float>((size_t)B * CA * Tv)),
          z(a.take<float>((size_t)B * 18 * cs_)),
          encimg(a.take<float>(8192)),
          wpad(a.take<float>(73728)),
          wpad0(a.take<float>(73728)),
          st0(a.take<double>(2 * B + 32)),
          ctr(reinterpret_cast<unsigned*>(st0 ? st0 + 2 * B : nullptr)),
          bstats(a.take<double>((size_t)STAT_APPS * BlockWs::NSTAT * B * 2)),
          B_(B),
          cs(cs_),
          blk(a, B, T, NF, cs_) {}
};
}  // namespace

namespace {
struct SepPacks {
    EncPack pe;
    BnPack pb;
    BlockPack pk;
    CafPack pc;
    S3Pack ps;
    DecPack pd;
};

// the whole chain for mixtures [first, first + B) of the call: wav / video_vp / out already point at mixture `first`
int separator_part(const SepPacks& k, const float* wav, const float* video_vp, float* out, int B, int L, int T, int Tv, int repeats, SepWs& w,
                   hipStream_t st, void* video_ready) {
    const EncPack& pe = k.pe;
    const BnPack& pb = k.pb;
    const BlockPack& pk = k.pk;
    const CafPack& pc = k.pc;
    const S3Pack& ps = k.ps;
    const DecPack& pd = k.pd;
    const int P = T * NF, cs = w.cs;
    // one memset per call: the bottleneck statistics, the tile counters and the statistic slots of every block application (SepWs::bstats)
    const size_t app_stats = (size_t)BlockWs::NSTAT * B * 2;
    const size_t zero_bytes = (size_t)(reinterpret_cast<char*>(w.bstats + SepWs::STAT_APPS * app_stats) - reinterpret_cast<char*>(w.st0));
    if (hipMemsetAsync(w.st0, 0, zero_bytes, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    int nctr = 0;
    CHECK(launch_stft(wav, w.spec, B, L, T, st));
    // fused path: the encoder output a0 is never written.  Its gLN statistics come from the spectrogram (enc_stats_kernel); the bottleneck + first
    // block head kernel (k_bnh.hip) and the tail kernel (k_s3f.hip) rebuild the a0 tiles they need on the matrix cores.
    bool head_done = false;  // = "a0 does not exist"
    if (!gemm_f32() && repeats > 1) {
        BnHeadArgs f;
        f.spec = w.spec; f.enc_img = w.encimg; f.T = T; f.F = NF;
        f.a1 = w.a1; f.res = nullptr; f.xenc = w.blk.x_enc;  // residual_0 is formed by the first boundary from a1 (B2bArgs::a1_mode bit 2)
        f.stats = w.st0; f.inv_count = 1.0 / ((double)CA * P); f.gamma = pb.gamma; f.beta = pb.beta;
        f.w16 = w.wpad0; f.bias = pb.bias;
        f.gw = pk.gw; f.gb = pk.gb; f.slope = pk.gslope; f.w2_16 = pk.proj_w16_perm; f.bp = pk.proj_b;
        f.P = P; f.cs = cs;
        f.tile_ctr = w.ctr + nctr;
        if (launch_bn_head_qualifies(f)) {
            EncPadJobs pad;
            pad.src[0] = pb.w16; pad.dst[0] = w.wpad0;  // bottleneck: the head kernel's LDS-DMA source
            pad.src[1] = ps.w16; pad.dst[1] = w.wpad;   // mask conv: the tail kernel's
            CHECK(launch_enc_stats(w.spec, pe.w, w.st0, w.encimg, pad, B, T, NF, st));
            CHECK(launch_bn_head(f, B, st));
            ++nctr;
            head_done = true;
        }
    }
    if (!head_done) CHECK(launch_enc_conv(w.spec, pe.w, w.a0, w.st0, B, CA, T, NF, (size_t)cs, (size_t)CA * cs, st));
    if (!head_done) CHECK(audio_bn(pb, w.a0, w.st0, w.a1, B, P, st, cs, w.ctr + nctr++));
    // refinement_module.py:45-62: block(a1); CAF; then (repeats-1) x block(audio + a1), shared weights
    // refinement_module.py:45-62: block(a1); CAF; then (repeats-1) x block(audio + a1), shared weights.
    CafArgs ca = caf_args(pc, w.cur, video_vp, w.nxt, w.r, w.att, T, NF, Tv);
    ca.r_t = w.rt;
    ca.att_t = w.attt;
    float *cur = w.cur, *nxt = w.nxt;
    if (gemm_f32() || repeats == 1) {  // unfused reference sequence (A/B path); contiguous tensors (cs == P, see rtfs_separator_forward_f32)
        RTFS_RETURN_IF(cs != P, RTFS_ERR_ARG);
        CHECK(block_forward(pk, w.a1, nullptr, w.cur, B, T, NF, w.blk, st));
        if (video_ready && hipStreamWaitEvent(st, (hipEvent_t)video_ready, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_caf_video(ca, B, st));
        CHECK(launch_caf_apply(ca, B, st));
        cur = w.nxt;
        nxt = w.cur;
        for (int i = 1; i < repeats; ++i) {
            CHECK(block_forward(pk, cur, w.a1, nxt, B, T, NF, w.blk, st));
            float* t = cur;
            cur = nxt;
            nxt = t;
        }
    } else {
        // block outputs between applications never reach HBM: the residual conv of block i, the CAF (after block 0),
        // the "+ a1" and the gateway + projection of block i+1 run back to back in one kernel (block_boundary).
        // the video side of the CAF (one workgroup per mixture, ~85 us) needs only the VP block's output: side stream, joined before the
        // first block boundary
        Fork cafv;
        CHECK(cafv.begin(st, 15));
        if (video_ready && hipStreamWaitEvent(cafv.side.stream, (hipEvent_t)video_ready, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_caf_video(ca, B, cafv.side.stream));
        if (!head_done) CHECK(block_head(pk, w.a1, nullptr, B, T, NF, w.blk, st, nullptr, w.ctr + nctr++));
        bool res_has_a1 = false;
        const bool own_slots = repeats <= SepWs::STAT_APPS;  // (more applications than slots: every block zeroes and reuses the first)
        for (int i = 0; i < repeats; ++i) {
            if (own_slots) w.blk.stats = w.bstats + (size_t)i * app_stats;
            CHECK(block_body(pk, B, T, NF, w.blk, st, !own_slots));
            if (i == 0) CHECK(cafv.join());
            if (i + 1 < repeats) CHECK(block_boundary(pk, w.a1, B, T, NF, w.blk, st, i == 0 ? &ca : nullptr, nctr < 64 ? w.ctr + nctr++ : nullptr, &res_has_a1, i + 2 == repeats, head_done && i == 0));
            else {
                // last application: residual conv + S3 mask + complex product + decoder taps in one kernel (k_s3f.hip); `refined` never exists
                TailS3Args f;
                f.x = w.blk.expanded; f.res = w.blk.residual; f.spec = w.spec; f.enc_img = w.encimg; f.T = T; f.F = NF; f.z = w.z;
                f.w1_16 = pk.res_w16; f.b1 = pk.res_b; f.w16 = head_done ? (const void*)w.wpad : (const void*)ps.w16; f.bias = ps.bias; f.slope = ps.slope; f.w16b = pd.w16p;
                f.stats = w.st0; f.inv_count = 1.0 / ((double)CA * P);
                f.P = P; f.cs = cs; f.cout_live = 18;
                f.tile_ctr = nctr < 64 ? w.ctr + nctr++ : nullptr;
                if (head_done) {  // (the two launches qualify together: same P, same pitch)
                    CHECK(launch_tail_s3t(f, B, st));
                    return launch_dec_istft(w.z, out, B, T, NF, L, (size_t)cs, (size_t)18 * cs, st);
                }
                CHECK(block_tail(pk, cur, B, T, NF, w.blk, st, f.tile_ctr));
            }
        }
    }
    if (!gemm_f32()) {  // S3 + decoder taps in one kernel: the separated spectrum never goes to HBM
        PwArgs a;
        a.x = cur; a.bias = ps.bias; a.aux = w.a0; a.out = w.z; a.slope = ps.slope; a.P = P; a.cs = cs; a.w16 = ps.w16; a.w16b = pd.w16p; a.cout_live = 18;
        a.tile_ctr = nctr < 64 ? w.ctr + nctr++ : nullptr;
        a.stats = w.st0; a.inv_count = 1.0 / ((double)CA * P);  // rms(a0) per mixture: the amplitude the taps GEMM's operand is normalised by
        CHECK(launch_pwr_s3_taps(a, B, st));
        return launch_dec_istft(w.z, out, B, T, NF, L, (size_t)cs, (size_t)18 * cs, st);
    }
    CHECK(s3_mask(ps, cur, w.a0, nxt, B, P, st));
    return decoder(pd, nxt, out, w.z, B, T, L, st);
}
}  // namespace

// How many parts the batch is cut into (default 1).  Two half batches on two streams overlap the HBM-bound streaming kernels of one half with
// the latency-bound sweeps / attention of the other (the sweep kernels keep HBM at < 10 %, the streaming kernels keep the matrix cores at
// < 10 %): 14.2 -> 13.0 ms at batch 32, three parts 13.4 on a box where two gave 13.5, four lose again (the quarter-batch launches are too
// small); tools/probe_two_streams.py measured the same from two host-side calls before this was built in.  It is NOT the default because
// every kernel then shares the chip with a kernel of the other half: a sweep launch takes 0.236 ms for half the sequences instead of 0.31 for
// all of them, so per-kernel durations (and the roofline figure bench.py derives from them) stop describing the kernel.
// rtfs_set_batch_split(n) or RTFS_SPLIT=n select it; a part is never smaller than 8 mixtures.
static std::atomic<int> g_split{0};
int rtfs_set_batch_split(int n) {
    if (n < 0 || n > 8) return RTFS_ERR_ARG;
    g_split.store(n);
    return RTFS_OK;
}
static int separator_parts(int B, int split = 0) {  // split > 0: this call's own setting; 0: the process default (setter, else RTFS_SPLIT, else 1)
    static const int env = getenv("RTFS_SPLIT") ? atoi(getenv("RTFS_SPLIT")) : 0;
    int n = split > 0 ? split : g_split.load();
    if (n <= 0) n = env > 0 ? env : 1;
    while (n > 1 && B / n < 8) --n;
    return n;
}

size_t rtfs_separator_workspace_bytes(int B, int L, int Tv) { return rtfs_separator_workspace_bytes_ex(B, L, Tv, 0); }

size_t rtfs_separator_workspace_bytes_ex(int B, int L, int Tv, int split) {
    if (split < 0 || split > 8) return 0;
    Arena ar(nullptr, 0);
    const int T = rtfs_num_frames(L), np = separator_parts(B, split);
    for (int i = 0; i < np; ++i) SepWs w(ar, (B * (i + 1)) / np - (B * i) / np, T, Tv, pitch(T * NF));  // upper bound: the unfused path carves less
    return ar.off + 256;
}

int rtfs_separator_forward_f32(const float* wav, const float* video_vp, const float* pack_enc, const float* pack_bn, const float* pack_block,
                               const float* pack_caf, const float* pack_s3, const float* pack_dec, float* out, int B, int L, int Tv,
                               int repeats, void* ws, size_t ws_bytes, void* stream, void* video_ready, int rnn_kind) {
    return rtfs_separator_forward_ex_f32(wav, video_vp, pack_enc, pack_bn, pack_block, pack_caf, pack_s3, pack_dec, out, B, L, Tv, repeats, ws, ws_bytes,
                                         stream, video_ready, rnn_kind, 0);
}

int rtfs_separator_forward_ex_f32(const float* wav, const float* video_vp, const float* pack_enc, const float* pack_bn, const float* pack_block,
                                  const float* pack_caf, const float* pack_s3, const float* pack_dec, float* out, int B, int L, int Tv,
                                  int repeats, void* ws, size_t ws_bytes, void* stream, void* video_ready, int rnn_kind, int split) {
    RTFS_RETURN_IF(split < 0 || split > 8, RTFS_ERR_ARG);
    RTFS_RETURN_IF(!wav || !video_vp || !pack_enc || !pack_bn || !pack_block || !pack_caf || !pack_s3 || !pack_dec || !out, RTFS_ERR_ARG);
    RTFS_RETURN_IF(B < 1 || L <= 128 || Tv < 1 || repeats < 1, RTFS_ERR_ARG);
    const int T = rtfs_num_frames(L);
    RTFS_RETURN_IF(!shape_ok_block(B, T, NF), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(rnn_kind != 0 && rnn_kind != 1, RTFS_ERR_ARG);
    const int np = separator_parts(B, split);
    Arena ar(ws, ws_bytes);
    std::vector<SepWs> parts;
    parts.reserve(np);
    // padded channel rows on the fused path; the unfused A/B sequence (exact-f32 GEMMs, or a single repeat) runs kernels that know only
    // contiguous tensors
    const int cstride = (gemm_f32() || repeats == 1) ? T * NF : pitch(T * NF);
    for (int i = 0; i < np; ++i) parts.emplace_back(ar, (B * (i + 1)) / np - (B * i) / np, T, Tv, cstride);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    Cursor ce(pack_enc), cb(pack_bn), ck(pack_block), cc(pack_caf), cs(pack_s3), cd(pack_dec);
    const SepPacks k{EncPack(ce), BnPack(cb), BlockPack::make(ck, rnn_kind), CafPack(cc), S3Pack(cs), DecPack(cd)};
    if (np == 1) return separator_part(k, wav, video_vp, out, B, L, T, Tv, repeats, parts[0], st, video_ready);
    // part 0 on the caller's stream, parts 1.. on side streams forked from it and joined back into it
    std::vector<RtfsSide> side(np);
    for (int i = 1; i < np; ++i) {
        CHECK(rtfs_side_stream(st, i, &side[i]));
        if (hipEventRecord(side[i].fork, st) != hipSuccess || hipStreamWaitEvent(side[i].stream, side[i].fork, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
    }
    int rc = RTFS_OK;
    for (int i = 0; i < np && rc == RTFS_OK; ++i) {
        const int first = (B * i) / np, nb = (B * (i + 1)) / np - first;
        rc = separator_part(k, wav + (size_t)first * L, video_vp + (size_t)first * 512 * Tv, out + (size_t)first * L, nb, L, T, Tv, repeats, parts[i],
                            i == 0 ? st : side[i].stream, video_ready);
    }
    for (int i = 1; i < np; ++i)  // join even after a failed launch: the caller's stream must not run ahead of work already queued
        if (hipEventRecord(side[i].join, side[i].stream) != hipSuccess || hipStreamWaitEvent(st, side[i].join, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
    return rc;
}

// ------------------------------------------------------------ stand-alone SRU operator
size_t rtfs_sru_workspace_bytes(int L, int N) {
    (void)L;
    (void)N;
    return 256;
}

int rtfs_sru_f32(const float* x, const float* pack, float* h, int L, int N, void* ws, size_t ws_bytes, void* stream) {
    (void)ws;
    (void)ws_bytes;
    RTFS_RETURN_IF(!x || !pack || !h || L < 1 || N < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(L > 243, RTFS_ERR_SHAPE);
    Cursor c(pack);
    DpPack p(c);
    return launch_sru_standalone(x, h, L, N, p.W0, p.Wl, p.wc, p.bias, S(stream));
}

// ------------------------------------------------------------ diagnostics
// Runs the dual-path sweep (dim 4: along F, dim 3: x is already (B,64,F,T) transposed) in the phase-stamped
// diagnostic build; stamps: DEVICE buffer of ceil(nseq/NSEQ) x 16 u64.  Not part of the product path.
int rtfs_debug_sweep_stamps(const float* x, const float* pack, float* out, int B, int R, int Ls, unsigned long long* stamps, void* stream) {
    RTFS_RETURN_IF(!x || !pack || !out || !stamps, RTFS_ERR_ARG);
    Cursor c(pack);
    DpPack p(c);
    const size_t plane = (size_t)R * Ls;
    Dp16Args a = dp16_args(p, x, out, B * R, R, Ls, CH * plane, Ls, plane);
    a.stamps = stamps;
    return launch_dualpath16(a, S(stream));
}

// ------------------------------------------------------------ self test
int rtfs_selftest_mfma_f16(const float* A, const float* B, float* D, void* stream) {
    RTFS_RETURN_IF(!A || !B || !D, RTFS_ERR_ARG);
    return launch_mfma_f16_selftest(A, B, D, S(stream));
}

// ------------------------------------------------------------ measurement hook
int rtfs_sweep_timing_enable(int on) { return dualpath_timing_enable(on); }

int rtfs_sweep_timing_collect(float* ms, int* seq_len, int* n_seq, int cap) {
    RTFS_RETURN_IF(!ms || !seq_len || !n_seq || cap < 1, RTFS_ERR_ARG);
    return dualpath_timing_collect(ms, seq_len, n_seq, cap);
}

int rtfs_pit_pairwise_sdr_f32(const float* ests, const float* targets, int B, int n_src, int L, int sdr_type, int zero_mean,
                              int take_log, float* pw_loss, float* min_loss, int* perm, void* stream) {
    if (!ests || !targets || !pw_loss || !min_loss || !perm) return RTFS_ERR_ARG;
    return launch_pit_pairwise(ests, targets, B, n_src, L, sdr_type, zero_mean, take_log, pw_loss, min_loss, perm, (hipStream_t)stream);
}

size_t rtfs_video_pack_floats(void) { return video_pack_floats(); }
size_t rtfs_video_workspace_bytes(int B, int T) { return video_workspace_bytes(B, T); }
int rtfs_video_frontend_f32(const float* lips, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!lips || !pack || !out || !ws) return RTFS_ERR_ARG;
    return video_frontend(lips, pack, out, B, T, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
