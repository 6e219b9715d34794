// C-ABI layer: parameter-pack layouts, workspace carving and the launch sequences of each module.
// See include/rtfs_amd.h for the contract and the reference interfaces each entry point replaces.
#include "../../include/rtfs_amd.h"
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

namespace {

constexpr int NF = 129;   // STFT bins (n_fft 256)
constexpr int CA = 256;   // audio feature channels
constexpr int CH = 64;    // block hidden channels
constexpr int FQ = 64;    // compressed frequency bins (n_freqs)

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

// ---------------------------------------------------------------- workspace carving
struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(void* b, size_t c) : base((char*)b), cap(c) {}
    template <class T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

#define CHECK(expr)                  \
    do {                             \
        int _e = (expr);             \
        if (_e != RTFS_OK) return _e; \
    } while (0)

inline hipStream_t S(void* s) { return (hipStream_t)s; }

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
    static constexpr int NSTAT = 11;
    enum { S_C0, S_C1, S_E0, S_G0, S_E1, S_G1, S_L1, S_E2, S_G2, S_L0, S_L2 };
    BlockWs(Arena& a, int B, int T, int F) {
        const size_t P = (size_t)T * F, Pg = (size_t)(T / 2) * (F / 2);
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
    double* st(int slot, int B) const { return stats ? stats + (size_t)slot * B * 2 : nullptr; }
};

int block_head(const BlockPack& p, const float* x, const float* x_res, int B, int T, int F, BlockWs& w, hipStream_t st, const CafArgs* caf) {
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
        if (caf && !gemm_f32()) {  // block input = CAF(x, video) + x_res, applied while streaming x (fused separator path)
            a.caf_r = caf->r_out; a.caf_att = caf->att_out;
            a.caf_w_key = caf->w_key; a.caf_bn_key = caf->bn_key; a.caf_w_val = caf->w_val; a.caf_bn_val = caf->bn_val;
            a.caf_T = caf->T; a.caf_F = caf->F; a.caf_Tv = caf->Tv;
        }
        CHECK(gemm_f32() ? launch_pw_gateway_proj(a, B, st) : launch_pws_gateway_proj(a, B, st));
    }
    return RTFS_OK;
}

// steps 2-17: everything between the projection (x_enc, residual in the workspace) and `expanded`
int block_body(const BlockPack& p, int B, int T, int F, BlockWs& w, hipStream_t st) {
    const int Tp = T / 2, Fp = F / 2;
    const int P = T * F, Pg = Tp * Fp;
    const double icF = 1.0 / ((double)CH * P), icG = 1.0 / ((double)CH * Pg);
    typedef BlockWs W;
    if (hipMemsetAsync(w.stats, 0, sizeof(double) * W::NSTAT * B * 2, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    {  // 2. downsample[0]: dw 4x4 s1 + bias -> c0 (pre-gLN) + stats                         tdanet.py:110
        DwArgs a;
        a.x = w.x_enc;
        a.w[0] = p.ds0_w;
        a.bias[0] = p.ds0_b;
        a.out[0] = w.c0;
        a.stats_out[0] = w.st(W::S_C0, B);
        a.C = CH; a.H = T; a.W = F; a.TH = 64;
        CHECK(launch_dw_s1(a, 1, false, 0, B, st));
    }
    {  // 3. downsample[1] on d0 = gLN(c0): dw 4x4 s2 -> c1 + stats; p0 = adaptive_avg_pool2d(d0)   tdanet.py:111-116
        DwArgs a;
        a.x = w.c0;
        a.in_stats = w.st(W::S_C0, B); a.in_inv_count = icF; a.in_gamma = p.ds0_g; a.in_beta = p.ds0_be;
        a.w[0] = p.ds1_w;
        a.bias[0] = p.ds1_b;
        a.out[0] = w.c1;
        a.out[1] = w.p0;
        a.stats_out[0] = w.st(W::S_C1, B);
        a.C = CH; a.H = T; a.W = F; a.Hg = Tp; a.Wg = Fp; a.TH = 64;
        CHECK(launch_dw_s2_pool(a, B, st));
    }
    // 4. g = pool(d0) + d1
    CHECK(launch_g_form(w.p0, w.c1, w.st(W::S_C1, B), icG, p.ds1_g, p.ds1_be, w.g, B, CH, Pg, st));
    // 5-8. dual-path sweeps along F then T                                                  yaml layer_1 / layer_2
    CHECK(dualpath(p.dpF, w.g, w.gF, B, Tp, Fp, 4, w.tA, w.tB, st));
    CHECK(dualpath(p.dpT, w.gF, w.gT, B, Tp, Fp, 3, w.tA, w.tB, st));
    // 9. TF self-attention                                                                  yaml layer_3
    CHECK(attention(p.attn, w.gT, w.gA, B, Tp, w.q, w.k, w.v, w.o, st));
    {  // 10. the four G-level convs on the attention output (fusion 0/1: global_embedding, global_gate)
        DwArgs a;
        a.x = w.gA;
        a.w[0] = p.fus0.emb_w; a.w[1] = p.fus0.gate_w; a.w[2] = p.fus1.emb_w; a.w[3] = p.fus1.gate_w;
        a.out[0] = w.E0; a.out[1] = w.G0; a.out[2] = w.E1; a.out[3] = w.G1;
        a.stats_out[0] = w.st(W::S_E0, B); a.stats_out[1] = w.st(W::S_G0, B);
        a.stats_out[2] = w.st(W::S_E1, B); a.stats_out[3] = w.st(W::S_G1, B);
        a.C = CH; a.H = Tp; a.W = Fp; a.TH = 64;
        CHECK(launch_dw_s1(a, 4, false, 0, B, st));
    }
    {  // 11. fusion 1 local_embedding on d1 = gLN(c1)
        DwArgs a;
        a.x = w.c1;
        a.in_stats = w.st(W::S_C1, B); a.in_inv_count = icG; a.in_gamma = p.ds1_g; a.in_beta = p.ds1_be;
        a.w[0] = p.fus1.loc_w;
        a.out[0] = w.L1;
        a.stats_out[0] = w.st(W::S_L1, B);
        a.C = CH; a.H = Tp; a.W = Fp; a.TH = 64;
        CHECK(launch_dw_s1(a, 1, true, 0, B, st));
    }
    {  // 12. xf1 = gLN(L1) * sigmoid(gLN(G1)) + gLN(E1)                                      fusion.py:62-67
        GCombineArgs a;
        a.l = w.L1; a.gate = w.G1; a.emb = w.E1;
        a.l_stats = w.st(W::S_L1, B); a.gate_stats = w.st(W::S_G1, B); a.emb_stats = w.st(W::S_E1, B);
        a.l_gamma = p.fus1.loc_g; a.l_beta = p.fus1.loc_b;
        a.gate_gamma = p.fus1.gate_g; a.gate_beta = p.fus1.gate_b;
        a.emb_gamma = p.fus1.emb_g; a.emb_beta = p.fus1.emb_b;
        a.inv_count = icG;
        a.out = w.xf1;
        a.C = CH; a.HW = Pg;
        CHECK(launch_g_combine(a, B, st));
    }
    {  // 13. concat layer's global convs on xf1
        DwArgs a;
        a.x = w.xf1;
        a.w[0] = p.cat0.emb_w; a.w[1] = p.cat0.gate_w;
        a.out[0] = w.E2; a.out[1] = w.G2;
        a.stats_out[0] = w.st(W::S_E2, B); a.stats_out[1] = w.st(W::S_G2, B);
        a.C = CH; a.H = Tp; a.W = Fp; a.TH = 64;
        CHECK(launch_dw_s1(a, 2, false, 0, B, st));
    }
    DwArgs d0in;  // common: read d0 = gLN(c0) at full resolution
    d0in.x = w.c0;
    d0in.in_stats = w.st(W::S_C0, B); d0in.in_inv_count = icF; d0in.in_gamma = p.ds0_g; d0in.in_beta = p.ds0_be;
    d0in.C = CH; d0in.H = T; d0in.W = F; d0in.TH = 64; d0in.Hg = Tp; d0in.Wg = Fp;
    {  // 14. fusion 0 local_embedding conv on d0: statistics only
        DwArgs a = d0in;
        a.w[0] = p.fus0.loc_w;
        a.stats_out[0] = w.st(W::S_L0, B);
        CHECK(launch_dw_s1(a, 1, true, 1, B, st));
    }
    {  // 15. xf0 = gLN(conv(d0)) * sigmoid(gLN(G0))^ + gLN(E0)^                              fusion.py:58-67
        DwArgs a = d0in;
        a.w[0] = p.fus0.loc_w;
        a.loc_stats = w.st(W::S_L0, B); a.loc_inv_count = icF; a.loc_gamma = p.fus0.loc_g; a.loc_beta = p.fus0.loc_b;
        a.gate = w.G0; a.gate_stats = w.st(W::S_G0, B); a.gate_gamma = p.fus0.gate_g; a.gate_beta = p.fus0.gate_b;
        a.emb = w.E0; a.emb_stats = w.st(W::S_E0, B); a.emb_gamma = p.fus0.emb_g; a.emb_beta = p.fus0.emb_b;
        a.g_inv_count = icG;
        a.out[0] = w.xf0;
        CHECK(launch_dw_s1(a, 1, true, 2, B, st));
    }
    DwArgs xin;  // common: read xf0
    xin.x = w.xf0;
    xin.C = CH; xin.H = T; xin.W = F; xin.TH = 64; xin.Hg = Tp; xin.Wg = Fp;
    {  // 16. concat layer local_embedding conv on xf0: statistics only
        DwArgs a = xin;
        a.w[0] = p.cat0.loc_w;
        a.stats_out[0] = w.st(W::S_L2, B);
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
        CHECK(launch_dw_s1(a, 1, false, 2, B, st));
    }
    return RTFS_OK;
}

int block_tail(const BlockPack& p, float* out, int B, int T, int F, BlockWs& w, hipStream_t st) {
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
int block_boundary(const BlockPack& p, const float* a1, int B, int T, int F, BlockWs& w, hipStream_t st, const CafArgs* caf) {
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
    if (caf) {
        a.caf_r = caf->r_out; a.caf_att = caf->att_out;
        a.caf_w_key = caf->w_key; a.caf_bn_key = caf->bn_key; a.caf_w_val = caf->w_val; a.caf_bn_val = caf->bn_val;
        a.caf_T = caf->T; a.caf_F = caf->F; a.caf_Tv = caf->Tv;
    }
    return launch_pws_b2b(a, B, st);
}

int audio_bn(const BnPack& p, const float* x, const double* stats, float* out, int B, int P, hipStream_t st) {
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
struct SepWs {
    float *spec, *a0, *a1, *cur, *nxt, *r, *att, *z;
    double* st0;
    BlockWs blk;
    SepWs(Arena& a, int B, int T, int Tv)
        : spec(a.take<float>((size_t)B * 2 * T * NF)),
          a0(a.take<float>((size_t)B * CA * T * NF)),
          a1(a.take<float>((size_t)B * CA * T * NF)),
          cur(a.take<float>((size_t)B * CA * T * NF)),
          nxt(a.take<float>((size_t)B * CA * T * NF)),
          r(a.take<float>((size_t)B * CA * Tv)),
          att(a.take<float>((size_t)B * CA * Tv)),
          z(a.take<float>((size_t)B * 18 * T * NF)),
          st0(a.take<double>(2 * B)),
          blk(a, B, T, NF) {}
};
}  // namespace

size_t rtfs_separator_workspace_bytes(int B, int L, int Tv) {
    Arena ar(nullptr, 0);
    SepWs w(ar, B, rtfs_num_frames(L), Tv);
    return ar.off + 256;
}

int rtfs_separator_forward_f32(const float* wav, const float* video_vp, const float* pack_enc, const float* pack_bn, const float* pack_block,
                               const float* pack_caf, const float* pack_s3, const float* pack_dec, float* out, int B, int L, int Tv,
                               int repeats, void* ws, size_t ws_bytes, void* stream, void* video_ready, int rnn_kind) {
    RTFS_RETURN_IF(!wav || !video_vp || !pack_enc || !pack_bn || !pack_block || !pack_caf || !pack_s3 || !pack_dec || !out, RTFS_ERR_ARG);
    RTFS_RETURN_IF(B < 1 || L <= 128 || Tv < 1 || repeats < 1, RTFS_ERR_ARG);
    const int T = rtfs_num_frames(L);
    RTFS_RETURN_IF(!shape_ok_block(B, T, NF), RTFS_ERR_SHAPE);
    Arena ar(ws, ws_bytes);
    SepWs w(ar, B, T, Tv);
    RTFS_RETURN_IF(!ws || !ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    Cursor ce(pack_enc), cb(pack_bn), ck(pack_block), cc(pack_caf), cs(pack_s3), cd(pack_dec);
    EncPack pe(ce);
    BnPack pb(cb);
    RTFS_RETURN_IF(rnn_kind != 0 && rnn_kind != 1, RTFS_ERR_ARG);
    BlockPack pk = BlockPack::make(ck, rnn_kind);
    CafPack pc(cc);
    S3Pack ps(cs);
    DecPack pd(cd);
    const int P = T * NF;
    if (hipMemsetAsync(w.st0, 0, sizeof(double) * 2 * B, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_stft(wav, w.spec, B, L, T, st));
    CHECK(launch_enc_conv(w.spec, pe.w, w.a0, w.st0, B, CA, T, NF, (size_t)P, (size_t)CA * P, st));
    CHECK(audio_bn(pb, w.a0, w.st0, w.a1, B, P, st));
    // refinement_module.py:45-62: block(a1); CAF; then (repeats-1) x block(audio + a1), shared weights
    // refinement_module.py:45-62: block(a1); CAF; then (repeats-1) x block(audio + a1), shared weights.
    CafArgs ca = caf_args(pc, w.cur, video_vp, w.nxt, w.r, w.att, T, NF, Tv);
    float *cur = w.cur, *nxt = w.nxt;
    if (gemm_f32() || repeats == 1) {  // unfused reference sequence (A/B path)
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
        CHECK(block_head(pk, w.a1, nullptr, B, T, NF, w.blk, st, nullptr));
        for (int i = 0; i < repeats; ++i) {
            CHECK(block_body(pk, B, T, NF, w.blk, st));
            if (i == 0) {
                if (video_ready && hipStreamWaitEvent(st, (hipEvent_t)video_ready, 0) != hipSuccess) return RTFS_ERR_LAUNCH;
                CHECK(launch_caf_video(ca, B, st));
            }
            if (i + 1 < repeats) CHECK(block_boundary(pk, w.a1, B, T, NF, w.blk, st, i == 0 ? &ca : nullptr));
            else CHECK(block_tail(pk, cur, B, T, NF, w.blk, st));
        }
    }
    if (!gemm_f32() && !getenv("RTFS_NO_S3T")) {  // S3 + decoder taps in one kernel: the separated spectrum never goes to HBM
        PwArgs a;
        a.x = cur; a.bias = ps.bias; a.aux = w.a0; a.out = w.z; a.slope = ps.slope; a.P = P; a.w16 = ps.w16; a.w16b = pd.w16p; a.cout_live = 18;
        CHECK(launch_pwr_s3_taps(a, B, st));
        return launch_dec_istft(w.z, out, B, T, NF, L, (size_t)T * NF, (size_t)18 * T * NF, st);
    }
    CHECK(s3_mask(ps, cur, w.a0, nxt, B, P, st));
    return decoder(pd, nxt, out, w.z, B, T, L, st);
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

// ------------------------------------------------------------ SRU operator, training side (k_train.hip)
namespace {
constexpr size_t TP_WT0 = 0, TP_WTL = TP_WT0 + 256 * 512, TP_WP0 = TP_WTL + 3 * 192 * 64, TP_WPL = TP_WP0 + 512 * 256,
                 TP_WC = TP_WPL + 3 * 64 * 192, TP_BIAS = TP_WC + 512, TP_END = TP_BIAS + 512;
constexpr size_t GP_W0 = 0, GP_WL = 512 * 256, GP_WC = GP_WL + 3 * 64 * 192, GP_BIAS = GP_WC + 512, GP_END = GP_BIAS + 512;
struct SruSaved {  // views into the saved-state buffer of one forward
    float *U[4], *c[4], *h[3];
    SruSaved(float* p, size_t LN) {
        U[0] = p; p += LN * 256;
        for (int l = 1; l < 4; ++l) { U[l] = p; p += LN * 192; }
        for (int l = 0; l < 4; ++l) { c[l] = p; p += LN * 64; }
        for (int l = 0; l < 3; ++l) { h[l] = p; p += LN * 64; }
    }
};
}  // namespace

size_t rtfs_sru_train_pack_floats(void) { return TP_END; }
size_t rtfs_sru_grad_floats(void) { return GP_END; }
size_t rtfs_sru_saved_floats(int L, int N) { return (size_t)L * N * (256 + 3 * 192 + 4 * 64 + 3 * 64); }
size_t rtfs_sru_backward_workspace_bytes(int L, int N) { return (size_t)L * N * (256 + 2 * 64) * sizeof(float) + 256; }

int rtfs_sru_forward_train_f32(const float* x, const float* tpack, float* h, float* saved, int L, int N, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !h || !saved || L < 1 || N < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF((size_t)L * N > 0x7fffffffu / 512, RTFS_ERR_SHAPE);
    const int LN = L * N;
    SruSaved sv(saved, (size_t)LN);
    hipStream_t st = S(stream);
    for (int l = 0; l < 4; ++l) {
        const float* xin = l == 0 ? x : sv.h[l - 1];
        const int K = l == 0 ? 512 : 64, KC = l == 0 ? 256 : 192;
        const float* Wt = l == 0 ? tpack + TP_WT0 : tpack + TP_WTL + (size_t)(l - 1) * 192 * 64;
        CHECK(launch_gemm_nt(xin, K, Wt, K, sv.U[l], KC, LN, KC, K, 0, st));
        SruScanArgs a;
        a.U = sv.U[l]; a.xin = l == 0 ? nullptr : xin; a.wc = tpack + TP_WC + 128 * l; a.bias = tpack + TP_BIAS + 128 * l;
        a.h = l == 3 ? h : sv.h[l]; a.c = sv.c[l]; a.L = L; a.N = N; a.KC = KC; a.ts = N; a.ns = 1;
        CHECK(launch_sru_scan_fwd(a, st));
    }
    return RTFS_OK;
}

int rtfs_sru_backward_f32(const float* x, const float* tpack, const float* saved, const float* dh, float* dx, float* dparams, int L,
                          int N, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !saved || !dh || !dx || !dparams || L < 1 || N < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF((size_t)L * N > 0x7fffffffu / 512, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_sru_backward_workspace_bytes(L, N), RTFS_ERR_WORKSPACE);
    const int LN = L * N;
    SruSaved sv(const_cast<float*>(saved), (size_t)LN);
    hipStream_t st = S(stream);
    float* dU = (float*)ws;
    float* gbuf[2] = {dU + (size_t)LN * 256, dU + (size_t)LN * 320};
    if (hipMemsetAsync(dparams, 0, GP_END * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    const float* g = dh;
    for (int l = 3; l >= 0; --l) {
        const int K = l == 0 ? 512 : 64, KC = l == 0 ? 256 : 192;
        const float* xin = l == 0 ? x : sv.h[l - 1];
        float* gnext = gbuf[l & 1];
        SruScanArgs a;
        a.U = sv.U[l]; a.xin = l == 0 ? nullptr : xin; a.wc = tpack + TP_WC + 128 * l; a.bias = tpack + TP_BIAS + 128 * l;
        a.c = sv.c[l]; a.g = g; a.dU = dU; a.dxp = l == 0 ? nullptr : gnext; a.dwc = dparams + GP_WC + 128 * l;
        a.dbias = dparams + GP_BIAS + 128 * l; a.L = L; a.N = N; a.KC = KC; a.ts = N; a.ns = 1;
        CHECK(launch_sru_scan_bwd(a, st));
        const float* Wp = l == 0 ? tpack + TP_WP0 : tpack + TP_WPL + (size_t)(l - 1) * 64 * 192;
        float* dWp = l == 0 ? dparams + GP_W0 : dparams + GP_WL + (size_t)(l - 1) * 64 * 192;
        // input gradient: dU . W^T (+ the highway term the scan already wrote for layers 1-3)
        CHECK(launch_gemm_nt(dU, KC, Wp, KC, l == 0 ? dx : gnext, K, LN, K, KC, l != 0 ? 1 : 0, st));
        CHECK(launch_gemm_tn(xin, K, dU, KC, dWp, KC, K, KC, (long)LN, st));
        g = gnext;
    }
    return RTFS_OK;
}

// ------------------------------------------------------------ DualPathRNN (SRU cell), training side
namespace {
constexpr size_t DT_G = 0, DT_B = 64, DT_SRU = 128, DT_WCF = DT_SRU + TP_END, DT_WCB = DT_WCF + 64 * 512, DT_BT = DT_WCB + 64 * 512,
                 DT_END = DT_BT + 64;
constexpr size_t DG_G = 0, DG_B = 64, DG_SRU = 128, DG_WCT = DG_SRU + GP_END, DG_BT = DG_WCT + 512 * 64, DG_END = DG_BT + 64;
struct DpSaved {  // sequence-major training layout, see k_train.hip
    float *xn, *U[4], *c[4], *hpad[4];
    size_t floats;
    DpSaved(float* p, size_t rows) {
        float* p0 = p;
        xn = p; p += (rows + 8) * 64;
        U[0] = p; p += rows * 256;
        for (int l = 1; l < 4; ++l) { U[l] = p; p += rows * 192; }
        for (int l = 0; l < 4; ++l) { c[l] = p; p += rows * 64; }
        for (int l = 0; l < 4; ++l) { hpad[l] = p; p += (rows + 8) * 64; }
        floats = (size_t)(p - p0);
    }
};
struct DpGeom {
    int nseq, R, Ls, L;
    size_t rows, elems;
    DpGeom(int B, int T, int F, int dim) {
        R = dim == 4 ? T : F;
        Ls = dim == 4 ? F : T;
        L = Ls - 7;
        nseq = B * R;
        rows = (size_t)nseq * Ls;
        elems = (size_t)B * CH * T * F;
    }
    bool ok() const { return Ls >= 8 && Ls <= 256 && rows * 512 < 0x7fffffffu; }
};
}  // namespace

size_t rtfs_dualpath_train_pack_floats(void) { return DT_END; }
size_t rtfs_dualpath_grad_floats(void) { return DG_END; }
size_t rtfs_dualpath_saved_floats(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    return DpSaved(nullptr, g.rows).floats;
}
size_t rtfs_dualpath_train_workspace_bytes(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    // forward: xt, out_t, y; backward: xt, dout_t, dx_t, dy, dU, 2 x g, dxn  (the larger of the two, plus alignment slack)
    const size_t fwd = 2 * g.elems + g.rows * 64, bwd = 3 * g.elems + (g.rows + 8) * 64 * 2 + g.rows * (256 + 128);
    return (fwd > bwd ? fwd : bwd) * sizeof(float) + 16 * 256;
}

int rtfs_dualpath_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim, void* ws,
                                    size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !out || !saved || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* ot = ar.take<float>(g.elems);
    float* y = ar.take<float>(g.rows * 64);
    DpSaved sv(saved, g.rows);
    hipStream_t st = S(stream);
    const int M = (int)g.rows;
    const float* src = x;
    if (dim == 3) {
        CHECK(launch_transpose(x, xt, B * CH, T, F, st));
        src = xt;
    }
    // rows past the last slot are read by the last windows: keep them zero
    if (hipMemsetAsync(sv.xn + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    for (int l = 0; l < 4; ++l)
        if (hipMemsetAsync(sv.hpad[l] + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_dp_ln_fwd(src, tpack + DT_G, tpack + DT_B, sv.xn, g.nseq, g.R, g.Ls, st));
    const float* sp = tpack + DT_SRU;
    for (int l = 0; l < 4; ++l) {
        const float* xin = l == 0 ? sv.xn : sv.hpad[l - 1] + 7 * 64;
        const int K = l == 0 ? 512 : 64, KC = l == 0 ? 256 : 192;
        const float* Wt = l == 0 ? sp + TP_WT0 : sp + TP_WTL + (size_t)(l - 1) * 192 * 64;
        CHECK(launch_gemm_nt(xin, 64, Wt, K, sv.U[l], KC, M, KC, K, 0, st));
        SruScanArgs a;
        a.U = sv.U[l]; a.xin = l == 0 ? nullptr : xin; a.wc = sp + TP_WC + 128 * l; a.bias = sp + TP_BIAS + 128 * l;
        a.h = sv.hpad[l] + 7 * 64; a.c = sv.c[l]; a.L = g.L; a.N = g.nseq; a.KC = KC; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_sru_scan_fwd(a, st));
    }
    // ConvTranspose1d as a GEMM over the 8-row windows of the zero-padded hidden sequence (rnn_layers.py:129,153)
    CHECK(launch_gemm_nt(sv.hpad[3], 64, tpack + DT_WCF, 512, y, 64, M, 64, 512, 0, st));
    CHECK(launch_dp_out(y, tpack + DT_BT, src, dim == 4 ? out : ot, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(ot, out, B * CH, F, T, st));
    return RTFS_OK;
}

int rtfs_dualpath_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams, int B,
                               int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !saved || !dout || !dx || !dparams || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* dt = ar.take<float>(g.elems);
    float* dxt = ar.take<float>(g.elems);
    float* dy = ar.take<float>((g.rows + 8) * 64);
    float* dxn = ar.take<float>((g.rows + 8) * 64);
    float* dU = ar.take<float>(g.rows * 256);
    float* gbuf[2] = {ar.take<float>(g.rows * 64), ar.take<float>(g.rows * 64)};
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    DpSaved sv(const_cast<float*>(saved), g.rows);
    hipStream_t st = S(stream);
    const int M = (int)g.rows;
    const float *srcx = x, *srcd = dout;
    if (dim == 3) {
        CHECK(launch_transpose(x, xt, B * CH, T, F, st));
        CHECK(launch_transpose(dout, dt, B * CH, T, F, st));
        srcx = xt;
        srcd = dt;
    }
    if (hipMemsetAsync(dparams, 0, DG_END * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dy + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dxn, 0, (g.rows + 8) * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_dp_dy(srcd, dy, dparams + DG_BT, g.nseq, g.R, g.Ls, st));
    // ConvTranspose1d: weight gradient = (windows of h)^T . dy, input gradient = windows of dy . W
    CHECK(launch_gemm_tn(sv.hpad[3], 64, dy, 64, dparams + DG_WCT, 64, 512, 64, (long)M, st));
    CHECK(launch_gemm_nt(dy, 64, tpack + DT_WCB, 512, gbuf[0], 64, M, 64, 512, 0, st));
    const float* gcur = gbuf[0];
    const float* sp = tpack + DT_SRU;
    float* gp = dparams + DG_SRU;
    for (int l = 3; l >= 0; --l) {
        const int K = l == 0 ? 512 : 64, KC = l == 0 ? 256 : 192;
        const float* xin = l == 0 ? sv.xn : sv.hpad[l - 1] + 7 * 64;
        float* gnext = gcur == gbuf[0] ? gbuf[1] : gbuf[0];
        SruScanArgs a;
        a.U = sv.U[l]; a.xin = l == 0 ? nullptr : xin; a.wc = sp + TP_WC + 128 * l; a.bias = sp + TP_BIAS + 128 * l;
        a.c = sv.c[l]; a.g = gcur; a.dU = dU; a.dxp = l == 0 ? nullptr : gnext; a.dwc = gp + GP_WC + 128 * l;
        a.dbias = gp + GP_BIAS + 128 * l; a.L = g.L; a.N = g.nseq; a.KC = KC; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_sru_scan_bwd(a, st));
        const float* Wp = l == 0 ? sp + TP_WP0 : sp + TP_WPL + (size_t)(l - 1) * 64 * 192;
        float* dWp = l == 0 ? gp + GP_W0 : gp + GP_WL + (size_t)(l - 1) * 64 * 192;
        if (l == 0) CHECK(launch_gemm_nt(dU, KC, Wp, KC, dxn, 64, M, 512, KC, 2, st));  // fold: adjoint of the unfold windows
        else CHECK(launch_gemm_nt(dU, KC, Wp, KC, gnext, 64, M, 64, KC, 1, st));
        CHECK(launch_gemm_tn(xin, 64, dU, KC, dWp, KC, K, KC, (long)M, st));
        gcur = gnext;
    }
    CHECK(launch_dp_ln_bwd(srcx, dxn, srcd, tpack + DT_G, dim == 4 ? dx : dxt, dparams + DG_G, dparams + DG_B, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(dxt, dx, B * CH, F, T, st));
    return RTFS_OK;
}

// ------------------------------------------------------------ DualPathRNN with the LSTM cell, training side
namespace {
// per-layer slots of the LSTM training pack / gradient buffer (Din = 512 for layer 0, 64 above)
struct LstmLayout {
    size_t wih[4], wiht[4], bias[4], whh[4], wcf, wcb, bt, end;   // pack
    size_t g_wih[4], g_bias[4], g_whh[4], g_wct, g_bt, g_end;     // gradients
    LstmLayout() {
        size_t o = 128;
        for (int l = 0; l < 4; ++l) {
            const size_t din = l == 0 ? 512 : 64;
            wih[l] = o; o += 256 * din;
            wiht[l] = o; o += din * 256;
            bias[l] = o; o += 256;
            whh[l] = o; o += 2 * 128 * 32;
        }
        wcf = o; o += 64 * 512; wcb = o; o += 64 * 512; bt = o; o += 64; end = o;
        o = 128;
        for (int l = 0; l < 4; ++l) {
            const size_t din = l == 0 ? 512 : 64;
            g_wih[l] = o; o += 256 * din;
            g_bias[l] = o; o += 256;
            g_whh[l] = o; o += 2 * 128 * 32;
        }
        g_wct = o; o += 512 * 64; g_bt = o; o += 64; g_end = o;
    }
};
struct LstmSaved {
    float *xn, *G[4], *c[4], *hpad[4], *hprev[4];
    size_t floats;
    LstmSaved(float* p, size_t rows) {
        float* p0 = p;
        xn = p; p += (rows + 8) * 64;
        for (int l = 0; l < 4; ++l) { G[l] = p; p += rows * 256; }
        for (int l = 0; l < 4; ++l) { c[l] = p; p += rows * 64; }
        for (int l = 0; l < 4; ++l) { hpad[l] = p; p += (rows + 8) * 64; }
        for (int l = 0; l < 4; ++l) { hprev[l] = p; p += rows * 64; }
        floats = (size_t)(p - p0);
    }
};
}  // namespace

size_t rtfs_dualpath_lstm_train_pack_floats(void) { return LstmLayout().end; }
size_t rtfs_dualpath_lstm_grad_floats(void) { return LstmLayout().g_end; }
size_t rtfs_dualpath_lstm_saved_floats(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    return LstmSaved(nullptr, g.rows).floats;
}
size_t rtfs_dualpath_lstm_train_workspace_bytes(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    return rtfs_dualpath_train_workspace_bytes(B, T, F, dim) + (g.rows * 256 + 256 * 64) * sizeof(float) + 4 * 256;
}

int rtfs_dualpath_lstm_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim, void* ws,
                                         size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !out || !saved || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_lstm_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* ot = ar.take<float>(g.elems);
    float* y = ar.take<float>(g.rows * 64);
    float* U = ar.take<float>(g.rows * 256);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    LstmSaved sv(saved, g.rows);
    LstmLayout lo;
    hipStream_t st = S(stream);
    const int M = (int)g.rows;
    const float* src = x;
    if (dim == 3) {
        CHECK(launch_transpose(x, xt, B * CH, T, F, st));
        src = xt;
    }
    if (hipMemsetAsync(sv.xn + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    for (int l = 0; l < 4; ++l) {
        if (hipMemsetAsync(sv.hpad[l] + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
        if (hipMemsetAsync(sv.hprev[l], 0, g.rows * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;  // non-step rows feed a GEMM
    }
    CHECK(launch_dp_ln_fwd(src, tpack + DT_G, tpack + DT_B, sv.xn, g.nseq, g.R, g.Ls, st));
    for (int l = 0; l < 4; ++l) {
        const float* xin = l == 0 ? sv.xn : sv.hpad[l - 1] + 7 * 64;
        const int K = l == 0 ? 512 : 64;
        CHECK(launch_gemm_nt(xin, 64, tpack + lo.wih[l], K, U, 256, M, 256, K, 0, st, tpack + lo.bias[l]));
        LstmScanArgs a;
        a.U = U; a.whh = tpack + lo.whh[l]; a.G = sv.G[l]; a.c = sv.c[l]; a.h = sv.hpad[l] + 7 * 64; a.hprev = sv.hprev[l];
        a.L = g.L; a.N = g.nseq; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_lstm_scan(a, false, st));
    }
    CHECK(launch_gemm_nt(sv.hpad[3], 64, tpack + lo.wcf, 512, y, 64, M, 64, 512, 0, st));
    CHECK(launch_dp_out(y, tpack + lo.bt, src, dim == 4 ? out : ot, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(ot, out, B * CH, F, T, st));
    return RTFS_OK;
}

int rtfs_dualpath_lstm_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams,
                                    int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !saved || !dout || !dx || !dparams || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_lstm_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* dt = ar.take<float>(g.elems);
    float* dxt = ar.take<float>(g.elems);
    float* dy = ar.take<float>((g.rows + 8) * 64);
    float* dxn = ar.take<float>((g.rows + 8) * 64);
    float* dU = ar.take<float>(g.rows * 256);
    float* gbuf[2] = {ar.take<float>(g.rows * 64), ar.take<float>(g.rows * 64)};
    float* whh64 = ar.take<float>(256 * 64);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    LstmSaved sv(const_cast<float*>(saved), g.rows);
    LstmLayout lo;
    hipStream_t st = S(stream);
    const int M = (int)g.rows;
    const float *srcx = x, *srcd = dout;
    if (dim == 3) {
        CHECK(launch_transpose(x, xt, B * CH, T, F, st));
        CHECK(launch_transpose(dout, dt, B * CH, T, F, st));
        srcx = xt;
        srcd = dt;
    }
    if (hipMemsetAsync(dparams, 0, lo.g_end * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dy + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dxn, 0, (g.rows + 8) * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_dp_dy(srcd, dy, dparams + lo.g_bt, g.nseq, g.R, g.Ls, st));
    CHECK(launch_gemm_tn(sv.hpad[3], 64, dy, 64, dparams + lo.g_wct, 64, 512, 64, (long)M, st));
    CHECK(launch_gemm_nt(dy, 64, tpack + lo.wcb, 512, gbuf[0], 64, M, 64, 512, 0, st));
    const float* gcur = gbuf[0];
    for (int l = 3; l >= 0; --l) {
        const int K = l == 0 ? 512 : 64;
        const float* xin = l == 0 ? sv.xn : sv.hpad[l - 1] + 7 * 64;
        float* gnext = gcur == gbuf[0] ? gbuf[1] : gbuf[0];
        LstmScanArgs a;
        a.whh = tpack + lo.whh[l]; a.G = sv.G[l]; a.c = sv.c[l]; a.g = gcur; a.dU = dU; a.L = g.L; a.N = g.nseq; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_lstm_scan(a, true, st));
        CHECK(launch_cl_colsum(dU, dparams + lo.g_bias[l], g.rows * 256, 256, st));
        // recurrent weights: (dU^T . h_{t-1}) is (256, 64); the two (128, 32) diagonal blocks are the two directions
        if (hipMemsetAsync(whh64, 0, 256 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_gemm_tn(dU, 256, sv.hprev[l], 64, whh64, 64, 256, 64, (long)M, st));
        for (int d = 0; d < 2; ++d)
            if (hipMemcpy2DAsync(dparams + lo.g_whh[l] + (size_t)d * 128 * 32, 32 * sizeof(float), whh64 + (size_t)d * 128 * 64 + d * 32,
                                 64 * sizeof(float), 32 * sizeof(float), 128, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return RTFS_ERR_LAUNCH;
        if (l == 0) CHECK(launch_gemm_nt(dU, 256, tpack + lo.wiht[l], 256, dxn, 64, M, 512, 256, 2, st));
        else CHECK(launch_gemm_nt(dU, 256, tpack + lo.wiht[l], 256, gnext, 64, M, 64, 256, 0, st));
        CHECK(launch_gemm_tn(dU, 256, xin, 64, dparams + lo.g_wih[l], K, 256, K, (long)M, st));
        gcur = gnext;
    }
    CHECK(launch_dp_ln_bwd(srcx, dxn, srcd, tpack + DT_G, dim == 4 ? dx : dxt, dparams + DG_G, dparams + DG_B, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(dxt, dx, B * CH, F, T, st));
    return RTFS_OK;
}

// ------------------------------------------------------------ ConvNormAct, training side (channel-last rows inside)
namespace {
struct CnaCfg {
    int Cin, Cout, k, stride, depthwise, pre_norm, pre_act, norm, act, has_bias, is2d, phase, world;
    int kh, kw, pt, pl, H, W, Ho, Wo, B;
    size_t rows_in, rows_out;
    // parameter / gradient layout (floats)
    size_t o_pg, o_pb, o_ps, o_w, o_wt, o_b, o_g, o_be, o_s, o_rm, o_rv, p_end;  // params
    size_t g_pg, g_pb, g_ps, g_w, g_b, g_g, g_be, g_s, g_end;            // grads
    bool ok;
    CnaCfg(const int* c, int B_, int H_, int W_) {
        Cin = c[0]; Cout = c[1]; k = c[2]; stride = c[3]; depthwise = c[4]; pre_norm = c[5]; pre_act = c[6]; norm = c[7]; act = c[8];
        has_bias = c[9]; is2d = c[10]; phase = c[11]; world = c[12] < 1 ? 1 : c[12];
        B = B_; H = H_; W = W_;
        kh = is2d ? k : 1;
        kw = k;
        const int p = stride > 1 ? (k - 1) / 2 : (k - 1) / 2;  // stride 1: "same" puts the smaller half first; stride > 1: symmetric
        pt = is2d ? p : 0;
        pl = p;
        if (stride == 1) { Ho = H; Wo = W; }
        else { Ho = is2d ? (H + 2 * p - k) / stride + 1 : 1; Wo = (W + 2 * p - k) / stride + 1; }
        rows_in = (size_t)B * H * W;
        rows_out = (size_t)B * Ho * Wo;
        auto pad64 = [](size_t n) { return (n + 63) / 64 * 64; };
        const size_t wn = depthwise ? (size_t)Cout * kh * kw : (size_t)Cout * Cin;
        size_t o = 0;
        o_pg = o; o += pad64(Cin); o_pb = o; o += pad64(Cin); o_ps = o; o += 64;
        o_w = o; o += pad64(wn); o_wt = o; o += depthwise ? 0 : pad64(wn);
        o_b = o; o += pad64(Cout); o_g = o; o += pad64(Cout); o_be = o; o += pad64(Cout); o_s = o; o += 64;
        o_rm = o; o += pad64(Cout); o_rv = o; o += pad64(Cout); p_end = o;
        o = 0;
        g_pg = o; o += pad64(Cin); g_pb = o; o += pad64(Cin); g_ps = o; o += 64; g_w = o; o += pad64(wn);
        g_b = o; o += pad64(Cout); g_g = o; o += pad64(Cout); g_be = o; o += pad64(Cout); g_s = o; o += 64; g_end = o;
        auto pow2 = [](int v, int cap) { return v >= 1 && v <= cap && !(v & (v - 1)); };
        ok = pow2(Cin, 1024) && pow2(Cout, 1024) && k >= 1 && kh <= 4 && kw <= 5 &&
             (stride == 1 || stride == 2) && Ho >= 1 && Wo >= 1 &&
             (depthwise ? Cin == Cout : (k == 1 && stride == 1 && Cin % 16 == 0 && Cout % 64 == 0 && Cin % 64 == 0)) &&
             pre_norm >= 0 && pre_norm <= 1 && norm >= 0 && norm <= 3 && phase >= 0 && phase <= 2 && (phase == 0 || norm == 3) && pre_act >= 0 && pre_act <= 3 && act >= 0 && act <= 3 &&
             rows_in * (size_t)(Cin > Cout ? Cin : Cout) < 0x7fffffffu;
    }
    bool pre() const { return pre_norm || pre_act; }
    bool post() const { return norm || act; }
};
struct CnaSaved {
    float *r0, *r2, *r3;
    double *st0, *st3, *cst;
    size_t floats;
    CnaSaved(float* p, const CnaCfg& c) {
        float* p0 = p;
        r0 = p; p += c.rows_in * c.Cin;
        r2 = p; p += c.rows_in * c.Cin;
        r3 = p; p += c.rows_out * c.Cout;
        st0 = (double*)p; p += 4 * c.B;
        st3 = (double*)p; p += 4 * c.B;
        cst = (double*)p; p += 4 * c.Cout;  // BatchNorm batch statistics (norm 3)
        floats = (size_t)(p - p0);
    }
};
}  // namespace

size_t rtfs_cna_param_floats(const int* cfg) { return CnaCfg(cfg, 1, 8, 8).p_end; }
size_t rtfs_cna_grad_floats(const int* cfg) { return CnaCfg(cfg, 1, 8, 8).g_end; }
size_t rtfs_cna_saved_floats(const int* cfg, int B, int H, int W) {
    CnaCfg c(cfg, B, H, W);
    return CnaSaved(nullptr, c).floats + 64;
}
size_t rtfs_cna_workspace_bytes(const int* cfg, int B, int H, int W) {
    CnaCfg c(cfg, B, H, W);
    return (2 * c.rows_out * c.Cout + 2 * c.rows_in * c.Cin + 4 * (size_t)B + (size_t)CL_DW_WGRAD_MAX_WG * 20 * 256) * sizeof(float) + 8 * 256;
}
// float offset, inside `saved`, of the 2 * Cout doubles (sum, sum of squares per channel) a norm = 3 forward accumulates; the gradient
// buffer's dgamma / dbeta float offsets for the matching exchange in the backward
size_t rtfs_cna_saved_stats_offset(const int* cfg, int B, int H, int W) {
    CnaCfg c(cfg, B, H, W);
    CnaSaved sv(nullptr, c);
    return (size_t)((float*)sv.cst - (float*)nullptr);
}
void rtfs_cna_grad_norm_offsets(const int* cfg, size_t* dgamma, size_t* dbeta) {
    CnaCfg c(cfg, 1, 8, 8);
    *dgamma = c.g_g;
    *dbeta = c.g_be;
}
void rtfs_cna_out_shape(const int* cfg, int H, int W, int* Ho, int* Wo) {
    CnaCfg c(cfg, 1, H, W);
    *Ho = c.Ho;
    *Wo = c.Wo;
}

int rtfs_cna_forward_train_f32(const float* x, const float* params, float* out, float* saved, const int* cfg, int B, int H, int W,
                               void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !params || !out || !saved || !cfg || B < 1 || H < 1 || W < 1, RTFS_ERR_ARG);
    CnaCfg c(cfg, B, H, W);
    RTFS_RETURN_IF(!c.ok, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_cna_workspace_bytes(cfg, B, H, W), RTFS_ERR_WORKSPACE);
    CnaSaved sv((float*)align_up((size_t)saved, 16), c);
    Arena ar(ws, ws_bytes);
    float* r5 = ar.take<float>(c.rows_out * c.Cout);
    hipStream_t st = S(stream);
    const size_t n_in = (size_t)H * W * c.Cin, n_out = (size_t)c.Ho * c.Wo * c.Cout;
    // phase 1 stops once the BatchNorm batch statistics are in `saved`; phase 2 resumes there (the caller all-reduced them in between)
    if (c.phase != 2) {
    CHECK(launch_transpose(x, sv.r0, B, c.Cin, H * W, st));  // (B, C, P) -> (B, P, C)
    const float* conv_in = sv.r0;
    if (c.pre()) {
        ClStageArgs a;
        a.x = sv.r0; a.y = sv.r2; a.n = n_in; a.C = c.Cin; a.norm = c.pre_norm; a.act = c.pre_act;
        a.gamma = params + c.o_pg; a.beta = params + c.o_pb; a.slope = params + c.o_ps; a.stats = sv.st0;
        if (c.pre_norm) {
            if (hipMemsetAsync(sv.st0, 0, sizeof(double) * 2 * B, st) != hipSuccess) return RTFS_ERR_LAUNCH;
            CHECK(launch_stats(sv.r0, sv.st0, B, n_in, st));
        }
        CHECK(launch_cl_norm_act_fwd(a, B, st));
        conv_in = sv.r2;
    }
    float* conv_out = c.post() ? sv.r3 : r5;
    if (c.depthwise) {
        ClDwArgs d;
        d.x = conv_in; d.w = params + c.o_w; d.bias = c.has_bias ? params + c.o_b : nullptr; d.y = conv_out;
        d.B = B; d.H = H; d.W = W; d.C = c.Cin; d.Ho = c.Ho; d.Wo = c.Wo; d.kh = c.kh; d.kw = c.kw; d.s = c.stride; d.pt = c.pt; d.pl = c.pl;
        CHECK(launch_cl_dw(d, 0, st));
    } else {
        CHECK(launch_gemm_nt(conv_in, c.Cin, params + c.o_w, c.Cin, conv_out, c.Cout, (int)c.rows_in, c.Cout, c.Cin, 0, st,
                             c.has_bias ? params + c.o_b : nullptr));
    }
    if (c.norm == 3) {
        if (hipMemsetAsync(sv.cst, 0, sizeof(double) * 2 * c.Cout, st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_cl_chan_stats(sv.r3, sv.cst, c.rows_out * c.Cout, c.Cout, st));
    }
    }  // phase != 2
    if (c.phase == 1) return RTFS_OK;
    if (c.post()) {
        ClStageArgs a;
        a.x = sv.r3; a.y = r5; a.n = n_out; a.C = c.Cout; a.norm = c.norm; a.act = c.act;
        a.gamma = params + c.o_g; a.beta = params + c.o_be; a.slope = params + c.o_s; a.stats = sv.st3;
        a.rmean = params + c.o_rm; a.rvar = params + c.o_rv; a.cstats = sv.cst; a.inv_rows = 1.0 / ((double)c.rows_out * c.world);
        if (c.norm == 1) {
            if (hipMemsetAsync(sv.st3, 0, sizeof(double) * 2 * B, st) != hipSuccess) return RTFS_ERR_LAUNCH;
            CHECK(launch_stats(sv.r3, sv.st3, B, n_out, st));
        }
        CHECK(launch_cl_norm_act_fwd(a, B, st));
    }
    return launch_transpose(r5, out, B, c.Ho * c.Wo, c.Cout, st);  // (B, P, C) -> (B, C, P)
}

// nn.BatchNorm's running-statistics update for a forward that ran with norm = 3 (train-mode BatchNorm): reads the batch statistics the
// forward left in `saved`, updates running_mean / running_var (DEVICE pointers to the module's buffers) in place
int rtfs_cna_bn_update_f32(const float* saved, const int* cfg, int B, int H, int W, float* running_mean, float* running_var, float momentum,
                           void* stream) {
    RTFS_RETURN_IF(!saved || !cfg || !running_mean || !running_var, RTFS_ERR_ARG);
    CnaCfg c(cfg, B, H, W);
    RTFS_RETURN_IF(!c.ok || c.norm != 3, RTFS_ERR_SHAPE);
    CnaSaved sv((float*)align_up((size_t)saved, 16), c);
    return launch_bn_update(sv.cst, running_mean, running_var, c.Cout, (double)c.rows_out * c.world, momentum, S(stream));
}

int rtfs_cna_backward_f32(const float* params, const float* saved, const float* dout, float* dx, float* dparams, const int* cfg, int B,
                          int H, int W, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!params || !saved || !dout || !dx || !dparams || !cfg || B < 1, RTFS_ERR_ARG);
    CnaCfg c(cfg, B, H, W);
    RTFS_RETURN_IF(!c.ok, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_cna_workspace_bytes(cfg, B, H, W), RTFS_ERR_WORKSPACE);
    CnaSaved sv((float*)align_up((size_t)saved, 16), c);
    Arena ar(ws, ws_bytes);
    float* d5 = ar.take<float>(c.rows_out * c.Cout);
    float* d3b = ar.take<float>(c.rows_out * c.Cout);
    float* d2 = ar.take<float>(c.rows_in * c.Cin);
    float* d0 = ar.take<float>(c.rows_in * c.Cin);
    double* Sb = ar.take<double>(2 * (size_t)B);
    float* wg_scratch = ar.take<float>(c.depthwise ? (size_t)CL_DW_WGRAD_MAX_WG * c.kh * c.kw * c.Cin : 0);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    const size_t n_in = (size_t)H * W * c.Cin, n_out = (size_t)c.Ho * c.Wo * c.Cout;
    // phase 1 (SyncBatchNorm): stop after the post-stage's reduction (dgamma / dbeta in dparams); phase 2: resume with the apply pass.
    // The workspace must be the same buffer in both calls (d5 lives there).
    if (c.phase != 2) {
        if (hipMemsetAsync(dparams, 0, c.g_end * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_transpose(dout, d5, B, c.Cout, c.Ho * c.Wo, st));
    }
    const float* d3 = d5;
    if (c.post()) {
        ClStageArgs a;
        a.x = sv.r3; a.dy = d5; a.dx = d3b; a.n = n_out; a.C = c.Cout; a.norm = c.norm; a.act = c.act;
        a.gamma = params + c.o_g; a.beta = params + c.o_be; a.slope = params + c.o_s; a.stats = sv.st3; a.S = Sb;
        a.rmean = params + c.o_rm; a.rvar = params + c.o_rv; a.cstats = sv.cst; a.inv_rows = 1.0 / ((double)c.rows_out * c.world);
        a.dgamma = dparams + c.g_g; a.dbeta = dparams + c.g_be; a.dslope = dparams + c.g_s;
        CHECK(launch_cl_norm_act_bwd(a, B, st, c.phase));
        if (c.phase == 1) return RTFS_OK;
        d3 = d3b;
    }
    const float* conv_in = c.pre() ? sv.r2 : sv.r0;
    if (c.has_bias) CHECK(launch_cl_colsum(d3, dparams + c.g_b, c.rows_out * c.Cout, c.Cout, st));
    if (c.depthwise) {
        ClDwArgs d;
        d.x = conv_in; d.w = params + c.o_w; d.dy = d3; d.dx = d2; d.dw = dparams + c.g_w; d.scratch = wg_scratch;
        d.B = B; d.H = H; d.W = W; d.C = c.Cin; d.Ho = c.Ho; d.Wo = c.Wo; d.kh = c.kh; d.kw = c.kw; d.s = c.stride; d.pt = c.pt; d.pl = c.pl;
        CHECK(launch_cl_dw(d, 1, st));
        CHECK(launch_cl_dw(d, 2, st));
    } else {
        CHECK(launch_gemm_nt(d3, c.Cout, params + c.o_wt, c.Cout, d2, c.Cin, (int)c.rows_in, c.Cin, c.Cout, 0, st));
        CHECK(launch_gemm_tn(d3, c.Cout, conv_in, c.Cin, dparams + c.g_w, c.Cin, c.Cout, c.Cin, (long)c.rows_in, st));
    }
    const float* dfirst = d2;
    if (c.pre()) {
        ClStageArgs a;
        a.x = sv.r0; a.dy = d2; a.dx = d0; a.n = n_in; a.C = c.Cin; a.norm = c.pre_norm; a.act = c.pre_act;
        a.gamma = params + c.o_pg; a.beta = params + c.o_pb; a.slope = params + c.o_ps; a.stats = sv.st0; a.S = Sb;
        a.dgamma = dparams + c.g_pg; a.dbeta = dparams + c.g_pb; a.dslope = dparams + c.g_ps;
        CHECK(launch_cl_norm_act_bwd(a, B, st));
        dfirst = d0;
    }
    return launch_transpose(dfirst, dx, B, H * W, c.Cin, st);
}

// ------------------------------------------------------------ MultiHeadSelfAttention2D, training side
namespace {
// parameter slots (floats)
constexpr size_t AT_W = 0, AT_WT = AT_W + 128 * 64, AT_B = AT_WT + 64 * 128, AT_SL = AT_B + 128, AT_G = AT_SL + 128, AT_BE = AT_G + 128 * 64,
                 AT_WP = AT_BE + 128 * 64, AT_WPT = AT_WP + 64 * 64, AT_BP = AT_WPT + 64 * 64, AT_SLP = AT_BP + 64, AT_GP = AT_SLP + 64,
                 AT_BEP = AT_GP + 64 * 64, AT_END = AT_BEP + 64 * 64;
constexpr size_t AG_W = 0, AG_B = AG_W + 128 * 64, AG_SL = AG_B + 128, AG_G = AG_SL + 64, AG_BE = AG_G + 128 * 64, AG_WP = AG_BE + 128 * 64,
                 AG_BP = AG_WP + 64 * 64, AG_SLP = AG_BP + 64, AG_GP = AG_SLP + 64, AG_BEP = AG_GP + 64 * 64, AG_END = AG_BEP + 64 * 64;
struct AttGeom {
    int B, T, Tp, nb;
    size_t R, qk, v, sc;
    AttGeom(int B_, int T_) : B(B_), T(T_) {
        Tp = (T + 63) / 64 * 64;
        nb = 4 * B;
        R = (size_t)B * T * 64;
        qk = (size_t)nb * Tp * 256;
        v = (size_t)nb * Tp * 1024;
        sc = (size_t)nb * Tp * Tp;
    }
};
struct AttSaved {
    float *r0, *Z, *st, *Qp, *Kp, *Vp, *P, *ratt, *Z2, *st2;
    size_t floats;
    AttSaved(float* p, const AttGeom& g) {
        float* p0 = p;
        r0 = p; p += g.R * 64;
        Z = p; p += g.R * 128;
        st = p; p += (size_t)g.B * g.T * 32;
        Qp = p; p += g.qk;
        Kp = p; p += g.qk;
        Vp = p; p += g.v;
        P = p; p += g.sc;
        ratt = p; p += g.R * 64;
        Z2 = p; p += g.R * 64;
        st2 = p; p += (size_t)g.B * g.T * 32;
        floats = (size_t)(p - p0);
    }
};
void att_groups(LngArgs& a, bool qkv) {
    if (qkv) {
        a.CZ = 128; a.ngroups = 12;
        for (int g = 0; g < 8; ++g) a.gstart[g] = 4 * g;
        for (int g = 8; g <= 12; ++g) a.gstart[g] = 32 + 16 * (g - 8);
        for (int c = 0; c < 128; ++c) a.gof[c] = c < 32 ? c / 4 : (c < 96 ? 8 + (c - 32) / 16 : 255);
    } else {
        a.CZ = 64; a.ngroups = 1; a.gstart[0] = 0; a.gstart[1] = 64;
        for (int c = 0; c < 64; ++c) a.gof[c] = 0;
    }
}
}  // namespace

size_t rtfs_tf_attention_train_pack_floats(void) { return AT_END; }
size_t rtfs_tf_attention_grad_floats(void) { return AG_END; }
size_t rtfs_tf_attention_saved_floats(int B, int T) { return AttSaved(nullptr, AttGeom(B, T)).floats; }
size_t rtfs_tf_attention_train_workspace_bytes(int B, int T) {
    AttGeom g(B, T);
    // backward is the larger: d rows (64), dZ2/dratt (64), dY/dZ (128 x 2), dO (v), dVp (v), dP (sc), Kt (qk), dQp, dKp (qk x 2)
    return (g.R * (64 + 64 + 64 + 128 + 128) + 2 * g.v + g.sc + 3 * g.qk) * sizeof(float) + 16 * 256;
}

int rtfs_tf_attention_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, void* ws,
                                        size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !out || !saved || B < 1 || T < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(T > 256 || (size_t)B * T * 64 * 128 >= 0x7fffffffu, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_tf_attention_train_workspace_bytes(B, T), RTFS_ERR_WORKSPACE);
    AttGeom g(B, T);
    AttSaved sv(saved, g);
    Arena ar(ws, ws_bytes);
    float* Y = ar.take<float>(g.R * 128);
    float* Vt = ar.take<float>(g.v);
    float* Op = ar.take<float>(g.v);
    float* rout = ar.take<float>(g.R * 64);
    hipStream_t st = S(stream);
    const int R = (int)g.R;
    CHECK(launch_transpose(x, sv.r0, B, 64, T * 64, st));
    CHECK(launch_gemm_nt(sv.r0, 64, tpack + AT_W, 64, sv.Z, 128, R, 128, 64, 0, st, tpack + AT_B));
    LngArgs a;
    att_groups(a, true);
    a.Z = sv.Z; a.Y = Y; a.stats = sv.st; a.slope = tpack + AT_SL; a.gamma = tpack + AT_G; a.beta = tpack + AT_BE;
    CHECK(launch_att_lng(a, B * T, false, st));
    if (hipMemsetAsync(sv.Qp, 0, (2 * g.qk + g.v) * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;  // Qp, Kp, Vp are adjacent
    CHECK(launch_att_pack_qkv(Y, sv.Qp, sv.Kp, sv.Vp, B, T, g.Tp, 0, st));
    // scores = Q K^T / sqrt(E * F) (attention.py:169-172), softmax over keys
    CHECK(launch_gemm_nt(sv.Qp, 256, sv.Kp, 256, sv.P, g.Tp, T, g.Tp, 256, 0, st, nullptr, g.nb, (size_t)g.Tp * 256, (size_t)g.Tp * 256,
                         (size_t)g.Tp * g.Tp));
    CHECK(launch_att_softmax(sv.P, nullptr, g.nb, T, g.Tp, 1.0f / 16.0f, false, st));
    CHECK(launch_transpose(sv.Vp, Vt, g.nb, g.Tp, 1024, st));
    CHECK(launch_gemm_nt(sv.P, g.Tp, Vt, g.Tp, Op, 1024, T, 1024, g.Tp, 0, st, nullptr, g.nb, (size_t)g.Tp * g.Tp, (size_t)1024 * g.Tp,
                         (size_t)g.Tp * 1024));
    CHECK(launch_att_pack_o(sv.ratt, Op, B, T, g.Tp, 0, st));
    CHECK(launch_gemm_nt(sv.ratt, 64, tpack + AT_WP, 64, sv.Z2, 64, R, 64, 64, 0, st, tpack + AT_BP));
    LngArgs b;
    att_groups(b, false);
    b.Z = sv.Z2; b.Y = rout; b.res = sv.r0; b.stats = sv.st2; b.slope = tpack + AT_SLP; b.gamma = tpack + AT_GP; b.beta = tpack + AT_BEP;
    CHECK(launch_att_lng(b, B * T, false, st));
    return launch_transpose(rout, out, B, T * 64, 64, st);
}

int rtfs_tf_attention_backward_f32(const float* tpack, const float* saved, const float* dout, float* dx, float* dparams, int B, int T,
                                   void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!tpack || !saved || !dout || !dx || !dparams || B < 1 || T < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(T > 256 || (size_t)B * T * 64 * 128 >= 0x7fffffffu, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_tf_attention_train_workspace_bytes(B, T), RTFS_ERR_WORKSPACE);
    AttGeom g(B, T);
    AttSaved sv(const_cast<float*>(saved), g);
    Arena ar(ws, ws_bytes);
    float* drow = ar.take<float>(g.R * 64);   // d(out rows), becomes d r0
    float* dZ2 = ar.take<float>(g.R * 64);
    float* dratt = ar.take<float>(g.R * 64);
    float* dY = ar.take<float>(g.R * 128);
    float* dZ = ar.take<float>(g.R * 128);
    float* dO = ar.take<float>(g.v);
    float* dVp = ar.take<float>(g.v);
    float* dP = ar.take<float>(g.sc);
    float* Kt = ar.take<float>(g.qk);
    float* dQp = ar.take<float>(g.qk);
    float* dKp = ar.take<float>(g.qk);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    const int R = (int)g.R;
    const size_t sQ = (size_t)g.Tp * 256, sV = (size_t)g.Tp * 1024, sS = (size_t)g.Tp * g.Tp;
    if (hipMemsetAsync(dparams, 0, AG_END * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_transpose(dout, drow, B, 64, T * 64, st));
    // concat projection ConvActNorm: LNG, then the 1x1 convolution
    LngArgs b;
    att_groups(b, false);
    b.Z = sv.Z2; b.stats = sv.st2; b.slope = tpack + AT_SLP; b.gamma = tpack + AT_GP; b.beta = tpack + AT_BEP; b.dY = drow; b.dZ = dZ2;
    b.dgamma = dparams + AG_GP; b.dbeta = dparams + AG_BEP; b.dslope = dparams + AG_SLP;
    CHECK(launch_att_lng(b, B * T, true, st));
    CHECK(launch_cl_colsum(dZ2, dparams + AG_BP, g.R * 64, 64, st));
    CHECK(launch_gemm_tn(dZ2, 64, sv.ratt, 64, dparams + AG_WP, 64, 64, 64, (long)R, st));
    CHECK(launch_gemm_nt(dZ2, 64, tpack + AT_WPT, 64, dratt, 64, R, 64, 64, 0, st));
    // attention core
    // the split-K GEMMs accumulate into their outputs; padding rows (t >= T) of dO / dQp are never read
    if (hipMemsetAsync(dVp, 0, g.v * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dKp, 0, g.qk * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_att_pack_o(dratt, dO, B, T, g.Tp, 1, st));
    CHECK(launch_gemm_nt(dO, 1024, sv.Vp, 1024, dP, g.Tp, T, g.Tp, 1024, 0, st, nullptr, g.nb, sV, sV, sS));       // dP = dO V^T
    CHECK(launch_gemm_tn(sv.P, g.Tp, dO, 1024, dVp, 1024, g.Tp, 1024, (long)T, st, g.nb, sS, sV, sV));              // dV = P^T dO
    CHECK(launch_att_softmax(dP, sv.P, g.nb, T, g.Tp, 1.0f / 16.0f, true, st));                                    // dP -> dS
    CHECK(launch_transpose(sv.Kp, Kt, g.nb, g.Tp, 256, st));
    CHECK(launch_gemm_nt(dP, g.Tp, Kt, g.Tp, dQp, 256, T, 256, g.Tp, 0, st, nullptr, g.nb, sS, sQ, sQ));            // dQ = dS K
    CHECK(launch_gemm_tn(dP, g.Tp, sv.Qp, 256, dKp, 256, g.Tp, 256, (long)T, st, g.nb, sS, sQ, sQ));                // dK = dS^T Q
    CHECK(launch_att_pack_qkv(dY, dQp, dKp, dVp, B, T, g.Tp, 1, st));
    // the twelve Q/K/V ConvActNorms
    LngArgs a;
    att_groups(a, true);
    a.Z = sv.Z; a.stats = sv.st; a.slope = tpack + AT_SL; a.gamma = tpack + AT_G; a.beta = tpack + AT_BE; a.dY = dY; a.dZ = dZ;
    a.dgamma = dparams + AG_G; a.dbeta = dparams + AG_BE; a.dslope = dparams + AG_SL;
    CHECK(launch_att_lng(a, B * T, true, st));
    CHECK(launch_cl_colsum(dZ, dparams + AG_B, g.R * 128, 128, st));
    CHECK(launch_gemm_tn(dZ, 128, sv.r0, 64, dparams + AG_W, 64, 128, 64, (long)R, st));
    CHECK(launch_gemm_nt(dZ, 128, tpack + AT_WT, 128, drow, 64, R, 64, 128, 1, st));  // + the residual's gradient already in drow
    return launch_transpose(drow, dx, B, T * 64, 64, st);
}

// ------------------------------------------------------------ block glue with gradients: pooling, TFAR combine
int rtfs_adaptive_avg_pool2d_f32(const float* x, float* y, int N, int H, int W, int Ho, int Wo, void* stream) {
    RTFS_RETURN_IF(!x || !y || N < 1, RTFS_ERR_ARG);
    return launch_pool2d(x, y, (size_t)N, H, W, Ho, Wo, false, S(stream));
}
int rtfs_adaptive_avg_pool2d_backward_f32(const float* dy, float* dx, int N, int H, int W, int Ho, int Wo, void* stream) {
    RTFS_RETURN_IF(!dy || !dx || N < 1, RTFS_ERR_ARG);
    return launch_pool2d(dy, dx, (size_t)N, H, W, Ho, Wo, true, S(stream));
}
int rtfs_tfar_combine_f32(const float* local, const float* gate, const float* glob, float* out, int N, int H, int W, int Hg, int Wg,
                          void* stream) {
    RTFS_RETURN_IF(!local || !gate || !glob || !out || N < 1, RTFS_ERR_ARG);
    return launch_tfar_combine(local, gate, glob, out, (size_t)N, H, W, Hg, Wg, S(stream));
}
int rtfs_tfar_combine_backward_f32(const float* dout, const float* local, const float* gate, float* dlocal, float* dgate, float* dglob, int N,
                                   int H, int W, int Hg, int Wg, void* stream) {
    RTFS_RETURN_IF(!dout || !local || !gate || !dlocal || !dgate || !dglob || N < 1, RTFS_ERR_ARG);
    return launch_tfar_combine_bwd(dout, local, gate, dlocal, dgate, dglob, (size_t)N, H, W, Hg, Wg, S(stream));
}

// ------------------------------------------------------------ encoder / decoder / S^3, training side
namespace {
// dW (256, 18) = big (B,256,T,F)^T . patches(z (B,2,T,F)); ws needs rows_big (R x 256) + patch rows (R x 64) + dw64 (256 x 64)
int wgrad_3x3(const float* big, const float* z, float* dw, int B, int T, float* rows_big, float* prow, float* dw64, hipStream_t st) {
    const size_t R = (size_t)B * T * NF;
    CHECK(launch_transpose(big, rows_big, B, CA, T * NF, st));
    CHECK(launch_patch3x3_rows(z, prow, B, T, NF, st));
    if (hipMemsetAsync(dw64, 0, sizeof(float) * CA * 64, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_gemm_tn(rows_big, CA, prow, 64, dw64, 64, CA, 64, (long)R, st));
    if (hipMemcpy2DAsync(dw, 18 * sizeof(float), dw64, 64 * sizeof(float), 18 * sizeof(float), CA, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return RTFS_ERR_LAUNCH;
    return RTFS_OK;
}
size_t wgrad_3x3_floats(int B, int T) { return (size_t)B * T * NF * (CA + 64) + CA * 64 + 256; }
}  // namespace

size_t rtfs_stft_encoder_backward_workspace_bytes(int B, int L) {
    const int T = rtfs_num_frames(L);
    return ((size_t)B * 2 * T * NF + wgrad_3x3_floats(B, T)) * sizeof(float) + 8 * 256;
}
int rtfs_stft_encoder_backward_f32(const float* wav, const float* da0, float* dw, int B, int L, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!wav || !da0 || !dw || B < 1 || L <= 128, RTFS_ERR_ARG);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_stft_encoder_backward_workspace_bytes(B, L), RTFS_ERR_WORKSPACE);
    const int T = rtfs_num_frames(L);
    Arena ar(ws, ws_bytes);
    float* spec = ar.take<float>((size_t)B * 2 * T * NF);
    float* rows = ar.take<float>((size_t)B * T * NF * CA);
    float* prow = ar.take<float>((size_t)B * T * NF * 64);
    float* dw64 = ar.take<float>(CA * 64);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    CHECK(launch_stft(wav, spec, B, L, T, S(stream)));
    return wgrad_3x3(da0, spec, dw, B, T, rows, prow, dw64, S(stream));
}

size_t rtfs_istft_decoder_backward_workspace_bytes(int B, int T) {
    return ((size_t)B * 2 * T * NF + wgrad_3x3_floats(B, T)) * sizeof(float) + 8 * 256;
}
// x (B,256,T,129) decoder input, w = ConvTranspose2d weight (256,2,3,3) as stored, dwav (B,L) -> dx (B,256,T,129), dw (256,2,3,3)
int rtfs_istft_decoder_backward_f32(const float* x, const float* w, const float* dwav, float* dx, float* dw, int B, int T, int L, void* ws,
                                    size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !w || !dwav || !dx || !dw || B < 1 || T < 1 || L < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(L > 128 * T + 127, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_istft_decoder_backward_workspace_bytes(B, T), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* dspec = ar.take<float>((size_t)B * 2 * T * NF);
    float* rows = ar.take<float>((size_t)B * T * NF * CA);
    float* prow = ar.take<float>((size_t)B * T * NF * 64);
    float* dw64 = ar.take<float>(CA * 64);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    CHECK(launch_istft_adjoint(dwav, dspec, B, T, L, st));
    // the adjoint of ConvTranspose2d(256 -> 2, 3x3, pad 1) is Conv2d(2 -> 256, 3x3, pad 1) with the same weight tensor
    CHECK(launch_enc_conv(dspec, w, dx, nullptr, B, CA, T, NF, (size_t)T * NF, (size_t)CA * T * NF, st));
    return wgrad_3x3(x, dspec, dw, B, T, rows, prow, dw64, st);
}

// S^3: out = e (x) m on (B, [re 128 | im 128], P) maps (mask_generator.py:71-82); conj_first != 0 gives conj(a) (x) b, the adjoint
// with respect to the other factor (dm = conj(e) (x) dout, de = conj(m) (x) dout)
int rtfs_s3_cmul_f32(const float* a, const float* b, float* out, int B, int P, int conj_first, void* stream) {
    RTFS_RETURN_IF(!a || !b || !out || B < 1 || P < 1, RTFS_ERR_ARG);
    return launch_cmul(a, b, out, B, (size_t)128 * P, conj_first, S(stream));
}

// ------------------------------------------------------------ CAF glue with adjoints
int rtfs_caf_attention_f32(const float* att_embed, float* att, int B, int C, int Tv, void* stream) {
    RTFS_RETURN_IF(!att_embed || !att || B < 1 || C < 1, RTFS_ERR_ARG);
    return launch_caf_att(att_embed, att, nullptr, nullptr, B * C, Tv, false, S(stream));
}
int rtfs_caf_attention_backward_f32(const float* att, const float* datt, float* datt_embed, int B, int C, int Tv, void* stream) {
    RTFS_RETURN_IF(!att || !datt || !datt_embed || B < 1 || C < 1, RTFS_ERR_ARG);
    return launch_caf_att(nullptr, const_cast<float*>(att), datt, datt_embed, B * C, Tv, true, S(stream));
}
int rtfs_caf_combine_f32(const float* key, const float* value, const float* resized, const float* att, float* out, int N, int T, int F, int Tv,
                         void* stream) {
    RTFS_RETURN_IF(!key || !value || !resized || !att || !out || N < 1, RTFS_ERR_ARG);
    return launch_caf_combine(key, value, resized, att, out, (size_t)N, T, F, Tv, S(stream));
}
int rtfs_caf_combine_backward_f32(const float* dout, const float* key, const float* value, const float* resized, const float* att, float* dkey,
                                  float* dvalue, float* dresized, float* datt, int N, int T, int F, int Tv, void* stream) {
    RTFS_RETURN_IF(!dout || !key || !value || !resized || !att || !dkey || !dvalue || !dresized || !datt || N < 1, RTFS_ERR_ARG);
    return launch_caf_combine_bwd(dout, key, value, resized, att, dkey, dvalue, dresized, datt, (size_t)N, T, F, Tv, S(stream));
}

// ------------------------------------------------------------ PIT loss gradient
int rtfs_pit_sdr_backward_f32(const float* ests, const float* targets, const int* perm, const float* dmin_loss, float* dests, int B, int n_src,
                              int L, int sdr_type, int zero_mean, int take_log, void* stream) {
    RTFS_RETURN_IF(!ests || !targets || !perm || !dmin_loss || !dests || B < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(sdr_type < 0 || sdr_type > 2, RTFS_ERR_ARG);
    return launch_pit_sdr_bwd(ests, targets, perm, dmin_loss, dests, B, n_src, L, sdr_type, zero_mean, take_log, S(stream));
}

// ------------------------------------------------------------ video-side attention pieces (rows = (b, t), channels last)
// nn.LayerNorm over the last axis of (N, C) rows
int rtfs_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float* y, int N, int C, void* stream) {
    RTFS_RETURN_IF(!x || !gamma || !beta || !y || N < 1, RTFS_ERR_ARG);
    return launch_ln_rows(x, gamma, beta, y, nullptr, nullptr, nullptr, nullptr, (size_t)N, C, false, S(stream));
}
int rtfs_layernorm_rows_backward_f32(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C,
                                     void* stream) {
    RTFS_RETURN_IF(!x || !gamma || !dy || !dx || !dgamma || !dbeta || N < 1, RTFS_ERR_ARG);
    hipStream_t st = S(stream);
    if (hipMemsetAsync(dgamma, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, st) != hipSuccess)
        return RTFS_ERR_LAUNCH;
    return launch_ln_rows(x, gamma, nullptr, nullptr, dy, dx, dgamma, dbeta, (size_t)N, C, true, st);
}
// nn.Linear on rows: y (M,N) = x (M,K) . W (N,K)^T + bias;  backward: dx = dy . W, dW = dy^T . x, dbias = column sums of dy.
// N % 64 == 0, K % 64 == 0.  ws (backward): N*K floats for W^T.
int rtfs_linear_rows_f32(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, void* stream) {
    RTFS_RETURN_IF(!x || !W || !y || M < 1, RTFS_ERR_ARG);
    return launch_gemm_nt(x, K, W, K, y, N, M, N, K, 0, S(stream), bias);
}
int rtfs_linear_rows_backward_f32(const float* x, const float* W, const float* dy, float* dx, float* dW, float* dbias, int M, int N, int K,
                                  void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !W || !dy || !dx || !dW || M < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF((N & 63) || (K & 63), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < (size_t)N * K * sizeof(float), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    float* Wt = (float*)ws;
    CHECK(launch_transpose(W, Wt, 1, N, K, st));  // (N, K) -> (K, N)
    CHECK(launch_gemm_nt(dy, N, Wt, N, dx, K, M, K, N, 0, st));
    if (hipMemsetAsync(dW, 0, sizeof(float) * N * K, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    CHECK(launch_gemm_tn(dy, N, x, K, dW, K, N, K, (long)M, st));
    if (dbias) {
        if (hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_cl_colsum(dy, dbias, (size_t)M * N, N, st));
    }
    return RTFS_OK;
}
// softmax(q k^T / sqrt(hd)) v per (batch, head) on packed projections qkv (B*T, 3*nh*hd) -> o (B*T, nh*hd); pmask optional
int rtfs_mha_core_f32(const float* qkv, const float* pmask, float* o, int B, int T, int n_head, int head_dim, void* stream) {
    RTFS_RETURN_IF(!qkv || !o || B < 1, RTFS_ERR_ARG);
    return launch_mha_core(qkv, pmask, o, nullptr, nullptr, B, T, n_head, head_dim, false, S(stream));
}
int rtfs_mha_core_backward_f32(const float* qkv, const float* pmask, const float* dout, float* dqkv, int B, int T, int n_head, int head_dim,
                               void* stream) {
    RTFS_RETURN_IF(!qkv || !dout || !dqkv || B < 1, RTFS_ERR_ARG);
    return launch_mha_core(qkv, pmask, nullptr, dout, dqkv, B, T, n_head, head_dim, true, S(stream));
}

// C = A . Bt^T (kind 0; accumulate adds to C) or C += A^T . B (kind 1): the two GEMM forms of the training path, exposed for tests
int rtfs_debug_gemm_f32(int kind, const float* A, const float* B, float* C, int M, int N, int K, int accumulate, void* stream) {
    RTFS_RETURN_IF(!A || !B || !C, RTFS_ERR_ARG);
    if (kind == 0) return launch_gemm_nt(A, K, B, K, C, N, M, N, K, accumulate != 0 ? 1 : 0, S(stream));
    if (kind == 1) return launch_gemm_tn(A, M, B, N, C, N, M, N, (long)K, S(stream));
    return RTFS_ERR_ARG;
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
