// C-ABI layer, training side: forward-with-saved-state / backward entry points of every module, the packs and
// saved-state layouts they use, and the launch sequences over the kernels of k_train_{gemm,rnn,conv,attn}.hip / k_loss.hip.
// See include/rtfs_amd.h for the contract and the reference interfaces each entry point replaces.
#include "api_common.h"

// ------------------------------------------------------------ SRU operator, training side (k_train_gemm.hip, k_train_rnn.hip)
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
struct DpSaved {  // sequence-major training layout, see k_train_rnn.hip
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
    // the forward takes any sweep length (layout kernels walk 256-position chunks, the scans are loops); the BACKWARD's layout kernels
    // (dp_dy, dp_ln_bwd) still hold a whole sequence in LDS: training segments are limited to Ls <= 256 (4 s), inference is not
    bool ok() const { return Ls >= 8 && rows * 512 < 0x7fffffffu; }
    bool ok_backward() const { return ok() && Ls <= 256; }
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

// dim 3 / 4: x, out (B,64,T,F) as in the reference; dim 13 / 14: the same sweeps on rows (B,T,F,64) (what the training kernels of a block
// hand each other): the F-sweep's slots then ARE the rows, the T-sweep goes through a row permutation instead of two tiled transposes.
int rtfs_dualpath_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim, void* ws,
                                    size_t ws_bytes, void* stream) {
    const bool rows = dim >= 10;
    dim = rows ? dim - 10 : dim;
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
        if (rows) CHECK(launch_rows_permute(x, xt, B, T, F, CH, st));
        else CHECK(launch_transpose(x, xt, B * CH, T, F, st));
        src = xt;
    }
    // rows past the last slot are read by the last windows: keep them zero
    if (hipMemsetAsync(sv.xn + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    for (int l = 0; l < 4; ++l)
        if (hipMemsetAsync(sv.hpad[l] + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (rows) CHECK(launch_ln_rows(src, tpack + DT_G, tpack + DT_B, sv.xn, nullptr, nullptr, nullptr, nullptr, g.rows, CH, false, st));
    else CHECK(launch_dp_ln_fwd(src, tpack + DT_G, tpack + DT_B, sv.xn, g.nseq, g.R, g.Ls, st));
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
    if (rows) {
        CHECK(launch_rows_bias_res(y, tpack + DT_BT, src, dim == 4 ? out : ot, g.rows * 64, CH, st));
        if (dim == 3) CHECK(launch_rows_permute(ot, out, B, F, T, CH, st));
        return RTFS_OK;
    }
    CHECK(launch_dp_out(y, tpack + DT_BT, src, dim == 4 ? out : ot, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(ot, out, B * CH, F, T, st));
    return RTFS_OK;
}

int rtfs_dualpath_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams, int B,
                               int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    const bool rows = dim >= 10;
    dim = rows ? dim - 10 : dim;
    RTFS_RETURN_IF(!x || !tpack || !saved || !dout || !dx || !dparams || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok_backward(), RTFS_ERR_SHAPE);
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
        if (rows) {
            CHECK(launch_rows_permute(x, xt, B, T, F, CH, st));
            CHECK(launch_rows_permute(dout, dt, B, T, F, CH, st));
        } else {
            CHECK(launch_transpose(x, xt, B * CH, T, F, st));
            CHECK(launch_transpose(dout, dt, B * CH, T, F, st));
        }
        srcx = xt;
        srcd = dt;
    }
    if (hipMemsetAsync(dparams, 0, DG_END * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dy + g.rows * 64, 0, 8 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (hipMemsetAsync(dxn, 0, (g.rows + 8) * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    if (rows) {  // the gradient already is rows: copy it next to its zero tail (the windows read 7 rows past the end), bias gradient = column sums
        if (hipMemcpyAsync(dy, srcd, g.rows * 64 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_cl_colsum(srcd, dparams + DG_BT, g.rows * 64, 64, st));
    } else {
        CHECK(launch_dp_dy(srcd, dy, dparams + DG_BT, g.nseq, g.R, g.Ls, st));
    }
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
    if (rows) {
        CHECK(launch_ln_rows(srcx, tpack + DT_G, nullptr, nullptr, dxn, dim == 4 ? dx : dxt, dparams + DG_G, dparams + DG_B, g.rows, CH, true, st, srcd));
        if (dim == 3) CHECK(launch_rows_permute(dxt, dx, B, F, T, CH, st));
        return RTFS_OK;
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
    RTFS_RETURN_IF(!g.ok_backward(), RTFS_ERR_SHAPE);
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

// ------------------------------------------------------------ DualPathRNN with the GRU cell (forward with saved state + backward)
namespace {
struct GruLayout {
    size_t wih[4], wiht[4], bih[4], whh[4], bhh[4], wcf, wcb, bt, end;   // pack
    size_t g_wih[4], g_bih[4], g_whh[4], g_bhh[4], g_wct, g_bt, g_end;   // gradients
    GruLayout() {
        size_t o = 128;
        for (int l = 0; l < 4; ++l) {
            const size_t din = l == 0 ? 512 : 64;
            wih[l] = o; o += 192 * din;
            wiht[l] = o; o += din * 192;
            bih[l] = o; o += 192;
            whh[l] = o; o += 2 * 96 * 32;
            bhh[l] = o; o += 192;
        }
        wcf = o; o += 64 * 512; wcb = o; o += 64 * 512; bt = o; o += 64; end = o;
        o = 128;
        for (int l = 0; l < 4; ++l) {
            const size_t din = l == 0 ? 512 : 64;
            g_wih[l] = o; o += 192 * din;
            g_bih[l] = o; o += 192;
            g_whh[l] = o; o += 2 * 96 * 32;
            g_bhh[l] = o; o += 192;
        }
        g_wct = o; o += 512 * 64; g_bt = o; o += 64; g_end = o;
    }
};
struct GruSaved {
    float *xn, *Sg[4], *hpad[4], *hprev[4];
    size_t floats;
    GruSaved(float* p, size_t rows) {
        float* p0 = p;
        xn = p; p += (rows + 8) * 64;
        for (int l = 0; l < 4; ++l) { Sg[l] = p; p += rows * 256; }
        for (int l = 0; l < 4; ++l) { hpad[l] = p; p += (rows + 8) * 64; }
        for (int l = 0; l < 4; ++l) { hprev[l] = p; p += rows * 64; }
        floats = (size_t)(p - p0);
    }
};
}  // namespace

size_t rtfs_dualpath_gru_train_pack_floats(void) { return GruLayout().end; }
size_t rtfs_dualpath_gru_grad_floats(void) { return GruLayout().g_end; }
size_t rtfs_dualpath_gru_saved_floats(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    return GruSaved(nullptr, g.rows).floats;
}
size_t rtfs_dualpath_gru_train_workspace_bytes(int B, int T, int F, int dim) {
    DpGeom g(B, T, F, dim);
    return rtfs_dualpath_train_workspace_bytes(B, T, F, dim) + (g.rows * 2 * 192 + 192 * 64) * sizeof(float) + 4 * 256;
}

int rtfs_dualpath_gru_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim, void* ws,
                                        size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !out || !saved || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_gru_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* ot = ar.take<float>(g.elems);
    float* y = ar.take<float>(g.rows * 64);
    float* U = ar.take<float>(g.rows * 192);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    GruSaved sv(saved, g.rows);
    GruLayout lo;
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
        if (hipMemsetAsync(sv.hprev[l], 0, g.rows * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    }
    CHECK(launch_dp_ln_fwd(src, tpack + DT_G, tpack + DT_B, sv.xn, g.nseq, g.R, g.Ls, st));
    for (int l = 0; l < 4; ++l) {
        const float* xin = l == 0 ? sv.xn : sv.hpad[l - 1] + 7 * 64;
        const int K = l == 0 ? 512 : 64;
        CHECK(launch_gemm_nt(xin, 64, tpack + lo.wih[l], K, U, 192, M, 192, K, 0, st, tpack + lo.bih[l]));
        GruScanArgs a;
        a.U = U; a.whh = tpack + lo.whh[l]; a.bhh = tpack + lo.bhh[l]; a.S = sv.Sg[l]; a.h = sv.hpad[l] + 7 * 64; a.hprev = sv.hprev[l];
        a.L = g.L; a.N = g.nseq; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_gru_scan(a, false, st));
    }
    CHECK(launch_gemm_nt(sv.hpad[3], 64, tpack + lo.wcf, 512, y, 64, M, 64, 512, 0, st));
    CHECK(launch_dp_out(y, tpack + lo.bt, src, dim == 4 ? out : ot, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(ot, out, B * CH, F, T, st));
    return RTFS_OK;
}

int rtfs_dualpath_gru_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams,
                                   int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !saved || !dout || !dx || !dparams || B < 1 || (dim != 3 && dim != 4), RTFS_ERR_ARG);
    DpGeom g(B, T, F, dim);
    RTFS_RETURN_IF(!g.ok_backward(), RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_dualpath_gru_train_workspace_bytes(B, T, F, dim), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    float* xt = ar.take<float>(g.elems);
    float* dt = ar.take<float>(g.elems);
    float* dxt = ar.take<float>(g.elems);
    float* dy = ar.take<float>((g.rows + 8) * 64);
    float* dxn = ar.take<float>((g.rows + 8) * 64);
    float* dU = ar.take<float>(g.rows * 192);
    float* dHR = ar.take<float>(g.rows * 192);
    float* gbuf[2] = {ar.take<float>(g.rows * 64), ar.take<float>(g.rows * 64)};
    float* whh64 = ar.take<float>(192 * 64);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    GruSaved sv(const_cast<float*>(saved), g.rows);
    GruLayout lo;
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
        GruScanArgs a;
        a.whh = tpack + lo.whh[l]; a.bhh = tpack + lo.bhh[l]; a.S = sv.Sg[l]; a.hprev = sv.hprev[l]; a.g = gcur; a.dU = dU; a.dHR = dHR;
        a.L = g.L; a.N = g.nseq; a.ts = 1; a.ns = g.Ls; a.pad = 1;
        CHECK(launch_gru_scan(a, true, st));
        CHECK(launch_cl_colsum(dU, dparams + lo.g_bih[l], g.rows * 192, 192, st));
        CHECK(launch_cl_colsum(dHR, dparams + lo.g_bhh[l], g.rows * 192, 192, st));
        if (hipMemsetAsync(whh64, 0, 192 * 64 * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
        CHECK(launch_gemm_tn(dHR, 192, sv.hprev[l], 64, whh64, 64, 192, 64, (long)M, st));
        for (int d = 0; d < 2; ++d)
            if (hipMemcpy2DAsync(dparams + lo.g_whh[l] + (size_t)d * 96 * 32, 32 * sizeof(float), whh64 + (size_t)d * 96 * 64 + d * 32,
                                 64 * sizeof(float), 32 * sizeof(float), 96, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return RTFS_ERR_LAUNCH;
        if (l == 0) CHECK(launch_gemm_nt(dU, 192, tpack + lo.wiht[l], 192, dxn, 64, M, 512, 192, 2, st));
        else CHECK(launch_gemm_nt(dU, 192, tpack + lo.wiht[l], 192, gnext, 64, M, 64, 192, 0, st));
        CHECK(launch_gemm_tn(dU, 192, xin, 64, dparams + lo.g_wih[l], K, 192, K, (long)M, st));
        gcur = gnext;
    }
    CHECK(launch_dp_ln_bwd(srcx, dxn, srcd, tpack + DT_G, dim == 4 ? dx : dxt, dparams + DG_G, dparams + DG_B, g.nseq, g.R, g.Ls, st));
    if (dim == 3) CHECK(launch_transpose(dxt, dx, B * CH, F, T, st));
    return RTFS_OK;
}

// ------------------------------------------------------------ ConvNormAct, training side (channel-last rows inside)
namespace {
struct CnaCfg {
    int Cin, Cout, k, stride, depthwise, pre_norm, pre_act, norm, act, has_bias, is2d, phase, world, in_rows, out_rows;
    int kh, kw, pt, pl, H, W, Ho, Wo, B;
    size_t rows_in, rows_out;
    // parameter / gradient layout (floats)
    size_t o_pg, o_pb, o_ps, o_w, o_wt, o_b, o_g, o_be, o_s, o_rm, o_rv, p_end;  // params
    size_t g_pg, g_pb, g_ps, g_w, g_b, g_g, g_be, g_s, g_end;            // grads
    bool ok;
    CnaCfg(const int* c, int B_, int H_, int W_) {
        Cin = c[0]; Cout = c[1]; k = c[2]; stride = c[3]; depthwise = c[4]; pre_norm = c[5]; pre_act = c[6]; norm = c[7]; act = c[8];
        has_bias = c[9]; is2d = c[10]; phase = c[11]; world = c[12] < 1 ? 1 : c[12]; in_rows = c[13]; out_rows = c[14];
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
    return (2 * c.rows_out * c.Cout + 2 * c.rows_in * c.Cin + 4 * (size_t)B + (size_t)CL_DW_WGRAD_MAX_WG * 20 * 256 +
            cl_stage_partial_floats(B, c.Cin > c.Cout ? c.Cin : c.Cout) + 1024 * (size_t)B) * sizeof(float) + 10 * 256;
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
    double* stat_part = ar.take<double>(512 * (size_t)B);
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    const size_t n_in = (size_t)H * W * c.Cin, n_out = (size_t)c.Ho * c.Wo * c.Cout;
    // phase 1 stops once the BatchNorm batch statistics are in `saved`; phase 2 resumes there (the caller all-reduced them in between)
    if (c.phase != 2) {
    // in_rows: x already is (B, P, C) rows (the producer was another training kernel); it is then read in place, forward and backward
    const float* r0 = c.in_rows ? x : sv.r0;
    if (!c.in_rows) CHECK(launch_transpose(x, sv.r0, B, c.Cin, H * W, st));  // (B, C, P) -> (B, P, C)
    const float* conv_in = r0;
    if (c.pre()) {
        ClStageArgs a;
        a.x = r0; a.y = sv.r2; a.n = n_in; a.C = c.Cin; a.norm = c.pre_norm; a.act = c.pre_act;
        a.gamma = params + c.o_pg; a.beta = params + c.o_pb; a.slope = params + c.o_ps; a.stats = sv.st0;
        if (c.pre_norm) {
            CHECK(launch_stats2(r0, sv.st0, B, n_in, stat_part, st));
        }
        CHECK(launch_cl_norm_act_fwd(a, B, st));
        conv_in = sv.r2;
    }
    float* conv_out = c.post() ? sv.r3 : (c.out_rows ? out : r5);
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
        a.x = sv.r3; a.y = c.out_rows ? out : r5; a.n = n_out; a.C = c.Cout; a.norm = c.norm; a.act = c.act;
        a.gamma = params + c.o_g; a.beta = params + c.o_be; a.slope = params + c.o_s; a.stats = sv.st3;
        a.rmean = params + c.o_rm; a.rvar = params + c.o_rv; a.cstats = sv.cst; a.inv_rows = 1.0 / ((double)c.rows_out * c.world);
        if (c.norm == 1) {
            CHECK(launch_stats2(sv.r3, sv.st3, B, n_out, stat_part, st));
        }
        CHECK(launch_cl_norm_act_fwd(a, B, st));
    }
    if (c.out_rows) return RTFS_OK;
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

int rtfs_cna_backward_f32(const float* x, const float* params, const float* saved, const float* dout, float* dx, float* dparams,
                          const int* cfg, int B, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!params || !saved || !dout || !dx || !dparams || !cfg || B < 1, RTFS_ERR_ARG);
    CnaCfg c(cfg, B, H, W);
    RTFS_RETURN_IF(!c.ok, RTFS_ERR_SHAPE);
    RTFS_RETURN_IF(c.in_rows && !x, RTFS_ERR_ARG);  // rows input is not copied into `saved`: the caller hands it back
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_cna_workspace_bytes(cfg, B, H, W), RTFS_ERR_WORKSPACE);
    CnaSaved sv((float*)align_up((size_t)saved, 16), c);
    Arena ar(ws, ws_bytes);
    float* d5 = ar.take<float>(c.rows_out * c.Cout);
    float* d3b = ar.take<float>(c.rows_out * c.Cout);
    float* d2 = ar.take<float>(c.rows_in * c.Cin);
    float* d0 = ar.take<float>(c.rows_in * c.Cin);
    double* Sb = ar.take<double>(2 * (size_t)B);
    float* wg_scratch = ar.take<float>(c.depthwise ? (size_t)CL_DW_WGRAD_MAX_WG * c.kh * c.kw * c.Cin : 0);
    float* stage_partial = ar.take<float>(cl_stage_partial_floats(B, c.Cin > c.Cout ? c.Cin : c.Cout));
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    const size_t n_in = (size_t)H * W * c.Cin, n_out = (size_t)c.Ho * c.Wo * c.Cout;
    // phase 1 (SyncBatchNorm): stop after the post-stage's reduction (dgamma / dbeta in dparams); phase 2: resume with the apply pass.
    // The workspace must be the same buffer in both calls (d5 lives there).
    const float* r0 = c.in_rows ? x : sv.r0;
    if (c.out_rows) d5 = const_cast<float*>(dout);  // gradient arrives as rows
    if (c.phase != 2) {
        if (hipMemsetAsync(dparams, 0, c.g_end * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
        if (!c.out_rows) CHECK(launch_transpose(dout, d5, B, c.Cout, c.Ho * c.Wo, st));
    }
    const float* d3 = d5;
    if (c.post()) {
        ClStageArgs a;
        a.x = sv.r3; a.dy = d5; a.dx = d3b; a.n = n_out; a.C = c.Cout; a.norm = c.norm; a.act = c.act;
        a.gamma = params + c.o_g; a.beta = params + c.o_be; a.slope = params + c.o_s; a.stats = sv.st3; a.S = Sb;
        a.rmean = params + c.o_rm; a.rvar = params + c.o_rv; a.cstats = sv.cst; a.inv_rows = 1.0 / ((double)c.rows_out * c.world);
        a.dgamma = dparams + c.g_g; a.dbeta = dparams + c.g_be; a.dslope = dparams + c.g_s; a.partial = stage_partial;
        CHECK(launch_cl_norm_act_bwd(a, B, st, c.phase));
        if (c.phase == 1) return RTFS_OK;
        d3 = d3b;
    }
    const float* conv_in = c.pre() ? sv.r2 : r0;
    if (!c.pre() && c.in_rows) d2 = dx;  // the convolution's input gradient is the module's
    if (c.has_bias) CHECK(launch_cl_colsum(d3, dparams + c.g_b, c.rows_out * c.Cout, c.Cout, st, stage_partial));
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
        if (c.in_rows) d0 = dx;
        a.x = r0; a.dy = d2; a.dx = d0; a.n = n_in; a.C = c.Cin; a.norm = c.pre_norm; a.act = c.pre_act;
        a.gamma = params + c.o_pg; a.beta = params + c.o_pb; a.slope = params + c.o_ps; a.stats = sv.st0; a.S = Sb;
        a.dgamma = dparams + c.g_pg; a.dbeta = dparams + c.g_pb; a.dslope = dparams + c.g_ps; a.partial = stage_partial;
        CHECK(launch_cl_norm_act_bwd(a, B, st));
        dfirst = d0;
    }
    if (c.in_rows) return RTFS_OK;
    return launch_transpose(dfirst, dx, B, H * W, c.Cin, st);
}

// CAF combine on rows: key, value, out (B, T, F, C); resized, att (B, C, Tv)
int rtfs_caf_combine_rows_f32(const float* key, const float* value, const float* resized, const float* att, float* out, int B, int T, int F,
                              int C, int Tv, void* stream) {
    RTFS_RETURN_IF(!key || !value || !resized || !att || !out || B < 1, RTFS_ERR_ARG);
    return launch_caf_combine_rows(key, value, resized, att, out, B, T, F, C, Tv, S(stream));
}
int rtfs_caf_combine_rows_backward_f32(const float* dout, const float* key, const float* value, const float* resized, const float* att,
                                       float* dkey, float* dvalue, float* dresized, float* datt, int B, int T, int F, int C, int Tv,
                                       void* stream) {
    RTFS_RETURN_IF(!dout || !key || !value || !resized || !att || !dkey || !dvalue || !dresized || !datt || B < 1, RTFS_ERR_ARG);
    return launch_caf_combine_rows_bwd(dout, key, value, resized, att, dkey, dvalue, dresized, datt, B, T, F, C, Tv, S(stream));
}

// ------------------------------------------------------------ RTFS block gateway (depthwise 1x1 + PReLU on x + x_res), training side
// gradient buffer: [dw C | db C | dslope 1], C and 2C + 1 rounded up to 64
size_t rtfs_gateway_grad_floats(int C) { return 2 * (size_t)align_up((size_t)C, 64) + 64; }
size_t rtfs_gateway_workspace_bytes(int C) { return cl_stage_partial_floats(2, C) * sizeof(float) + 256; }
int rtfs_gateway_forward_train_f32(const float* x, const float* x_res, const float* w, const float* b, const float* slope, float* out,
                                   size_t rows, int C, void* stream) {
    RTFS_RETURN_IF(!x || !w || !b || !slope || !out || rows < 1, RTFS_ERR_ARG);
    GatewayArgs a;
    a.x = x; a.xr = x_res; a.w = w; a.b = b; a.slope = slope; a.y = out; a.n4 = rows * (size_t)C / 4; a.C = C;
    return launch_gateway(a, false, nullptr, nullptr, nullptr, S(stream));
}
int rtfs_gateway_backward_f32(const float* x, const float* x_res, const float* w, const float* b, const float* slope, const float* dout,
                              float* dx, float* dparams, size_t rows, int C, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !w || !b || !slope || !dout || !dx || !dparams || rows < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(!ws || ws_bytes < rtfs_gateway_workspace_bytes(C), RTFS_ERR_WORKSPACE);
    Arena ar(ws, ws_bytes);
    GatewayArgs a;
    a.partial = ar.take<float>(cl_stage_partial_floats(2, C));
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    if (hipMemsetAsync(dparams, 0, rtfs_gateway_grad_floats(C) * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    a.x = x; a.xr = x_res; a.w = w; a.b = b; a.slope = slope; a.dy = dout; a.dx = dx; a.n4 = rows * (size_t)C / 4; a.C = C;
    const size_t Cp = align_up((size_t)C, 64);
    return launch_gateway(a, true, dparams, dparams + Cp, dparams + 2 * Cp, st);
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
    return (g.R * (64 + 64 + 64 + 128 + 128) + 2 * g.v + g.sc + 3 * g.qk + att_lng_scratch_floats(B * T)) * sizeof(float) + 16 * 256;
}

// rows != 0: x / out (and their gradients) are rows (B, T, 64 f, 64 c) instead of (B, 64, T, 64); the rows input is then used in place
// and handed to the backward again as `x`.
int rtfs_tf_attention_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int rows, void* ws,
                                        size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!x || !tpack || !out || !saved || B < 1 || T < 1, RTFS_ERR_ARG);
    RTFS_RETURN_IF(T > 4096 || (size_t)B * T * 64 * 128 >= 0x7fffffffu, RTFS_ERR_SHAPE);  // scores are (4B, Tp, Tp) floats in `saved`
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
    const float* r0 = rows ? x : sv.r0;
    if (!rows) CHECK(launch_transpose(x, sv.r0, B, 64, T * 64, st));
    CHECK(launch_gemm_nt(r0, 64, tpack + AT_W, 64, sv.Z, 128, R, 128, 64, 0, st, tpack + AT_B));
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
    b.Z = sv.Z2; b.Y = rows ? out : rout; b.res = r0; b.stats = sv.st2; b.slope = tpack + AT_SLP; b.gamma = tpack + AT_GP; b.beta = tpack + AT_BEP;
    CHECK(launch_att_lng(b, B * T, false, st));
    if (rows) return RTFS_OK;
    return launch_transpose(rout, out, B, T * 64, 64, st);
}

int rtfs_tf_attention_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams, int B,
                                   int T, int rows, void* ws, size_t ws_bytes, void* stream) {
    RTFS_RETURN_IF(!tpack || !saved || !dout || !dx || !dparams || B < 1 || T < 1 || (rows && !x), RTFS_ERR_ARG);
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
    float* lng_scratch = ar.take<float>(att_lng_scratch_floats(B * T));
    RTFS_RETURN_IF(!ar.ok(), RTFS_ERR_WORKSPACE);
    hipStream_t st = S(stream);
    const int R = (int)g.R;
    const size_t sQ = (size_t)g.Tp * 256, sV = (size_t)g.Tp * 1024, sS = (size_t)g.Tp * g.Tp;
    if (hipMemsetAsync(dparams, 0, AG_END * sizeof(float), st) != hipSuccess) return RTFS_ERR_LAUNCH;
    const float* r0 = rows ? x : sv.r0;
    if (rows) {
        if (hipMemcpyAsync(drow, dout, g.R * 64 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return RTFS_ERR_LAUNCH;
    } else {
        CHECK(launch_transpose(dout, drow, B, 64, T * 64, st));
    }
    // concat projection ConvActNorm: LNG, then the 1x1 convolution
    LngArgs b;
    att_groups(b, false);
    b.Z = sv.Z2; b.stats = sv.st2; b.slope = tpack + AT_SLP; b.gamma = tpack + AT_GP; b.beta = tpack + AT_BEP; b.dY = drow; b.dZ = dZ2;
    b.dgamma = dparams + AG_GP; b.dbeta = dparams + AG_BEP; b.dslope = dparams + AG_SLP; b.scratch = lng_scratch;
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
    a.dgamma = dparams + AG_G; a.dbeta = dparams + AG_BE; a.dslope = dparams + AG_SL; a.scratch = lng_scratch;
    CHECK(launch_att_lng(a, B * T, true, st));
    CHECK(launch_cl_colsum(dZ, dparams + AG_B, g.R * 128, 128, st));
    CHECK(launch_gemm_tn(dZ, 128, r0, 64, dparams + AG_W, 64, 128, 64, (long)R, st));
    CHECK(launch_gemm_nt(dZ, 128, tpack + AT_WT, 128, drow, 64, R, 64, 128, 1, st));  // + the residual's gradient already in drow
    if (rows) return hipMemcpyAsync(dx, drow, g.R * 64 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? RTFS_OK : RTFS_ERR_LAUNCH;
    return launch_transpose(drow, dx, B, T * 64, 64, st);
}

// ------------------------------------------------------------ layout change between (B, C, P) and rows (B, P, C)
int rtfs_layout_f32(const float* x, float* y, int B, int C, int P, int to_rows, void* stream) {
    RTFS_RETURN_IF(!x || !y || B < 1 || C < 1 || P < 1, RTFS_ERR_ARG);
    return to_rows ? launch_transpose(x, y, B, C, P, S(stream)) : launch_transpose(x, y, B, P, C, S(stream));
}

// ------------------------------------------------------------ block glue with gradients: pooling, TFAR combine
int rtfs_adaptive_avg_pool2d_f32(const float* x, float* y, int N, int H, int W, int Ho, int Wo, int inner, void* stream) {
    RTFS_RETURN_IF(!x || !y || N < 1, RTFS_ERR_ARG);
    return launch_pool2d(x, y, (size_t)N, H, W, Ho, Wo, false, S(stream), inner);
}
int rtfs_adaptive_avg_pool2d_backward_f32(const float* dy, float* dx, int N, int H, int W, int Ho, int Wo, int inner, void* stream) {
    RTFS_RETURN_IF(!dy || !dx || N < 1, RTFS_ERR_ARG);
    return launch_pool2d(dy, dx, (size_t)N, H, W, Ho, Wo, true, S(stream), inner);
}
int rtfs_tfar_combine_f32(const float* local, const float* gate, const float* glob, float* out, int N, int H, int W, int Hg, int Wg,
                          int inner, void* stream) {
    RTFS_RETURN_IF(!local || !gate || !glob || !out || N < 1, RTFS_ERR_ARG);
    return launch_tfar_combine(local, gate, glob, out, (size_t)N, H, W, Hg, Wg, S(stream), inner);
}
int rtfs_tfar_combine_backward_f32(const float* dout, const float* local, const float* gate, float* dlocal, float* dgate, float* dglob, int N,
                                   int H, int W, int Hg, int Wg, int inner, void* stream) {
    RTFS_RETURN_IF(!dout || !local || !gate || !dlocal || !dgate || !dglob || N < 1, RTFS_ERR_ARG);
    return launch_tfar_combine_bwd(dout, local, gate, dlocal, dgate, dglob, (size_t)N, H, W, Hg, Wg, S(stream), inner);
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

