// Video front-end ("the step before the path"): FRCNNVideoModel with the ResNet-18 trunk and PReLU,
// eval mode (reference src/models/videomodels/frcnn_videomodel.py:16-72, resnet.py:23-118).
//
//   vid_pad_kernel      lips (B,1,T,88,88) -> zero-padded volume (B, T+4, 94, 94)            (Conv3d padding (2,3,3))
//   vid_conv_kernel     implicit-GEMM convolution on the f16 matrix cores, 3-term hi/lo split (x = xh + xl,
//                       256 w = wh + wl; same arithmetic as k_pw16.hip), eval BatchNorm folded into weights + bias:
//                         STEM  Conv3d(1,64,(5,7,7),s(1,2,2)) + BN3d + PReLU(64)              frcnn_videomodel.py:42-52
//                         C3    Conv2d 3x3 (stride 1|2, pad 1) + BN [+ residual] [+ PReLU]    resnet.py:51-66
//                         C1    Conv2d 1x1 stride 2 + BN  (downsample branch)                 resnet.py:10-14
//   vid_maxpool_kernel  MaxPool3d((1,3,3), s(1,2,2), p(0,1,1))                               frcnn_videomodel.py:53
//   vid_avgpool_kernel  AdaptiveAvgPool2d(1) + view + transpose -> (B,512,T)                 resnet.py:116-118, frcnn:70
//
// Activations between convolutions are stored per frame CHANNEL-LAST as (n, H+2, W+2, C) with a zero border of one pixel:
// the 3x3 gathers need no bounds checks (kernels write interiors only; the borders are zeroed once per call), the 8
// consecutive channels of a B fragment are two 16-byte loads (a dword gather per channel kept the texture addresser, not the
// matrix cores, busy), and an accumulator tile's 4-channel groups are 16-byte stores.
// GEMM view: D[co][pixel] = sum_k W[co][k] X[k][pixel], k = tap * Cin + ci (tap-major: a 32-deep K chunk lies inside one
// tap since Cin % 32 == 0; the stem has Cin = 1 and k = tap, 245 padded to 256, gathered through an offset table).
// Workgroup = 4 waves = 128 output pixels x 64 output channels; weights stream through a double-buffered LDS image
// [hi|lo][64 co][32 k] per K chunk (register prefetch, one barrier per chunk); each lane gathers the 8 k-values of its
// pixel's B fragment straight from global memory.
#include "common.h"
#include "kernels.h"

namespace {

enum { VM_STEM = 0, VM_C3 = 1, VM_C1 = 2 };
constexpr int V_LDW = 40;  // staged weight row: 32 k + 8 pad halfs

struct VidConvArgs {
    const float* x;       // input activations (padded channel-last layout) or the padded volume (stem)
    const half8* w16;     // [Cout/64][K/32][hi|lo][64][32] halfs
    const float* bias;    // (Cout) folded BatchNorm shift
    const float* slope;   // (Cout) PReLU slopes or null
    const float* res;     // residual (padded layout of the OUTPUT geometry) or null
    float* out;           // padded layout (n, Ho+2, Wo+2, Cout); stem: unpadded (n, Ho, Wo, 64)
    int N, Cin, Cout, Hi, Wi, Ho, Wo, stride;  // Hi, Wi: input interior size (stem: 88); N frames
    int T;                // stem: frames per clip
};

// MT = 32-row output tiles per wave (a workgroup covers 32 MT output channels); PT = 32-pixel tiles per wave (a workgroup
// covers 128 PT pixels; PT = 2 measured no faster: the 3x3 gathers re-read the input nine times through L2, which is what
// bounds these kernels -- an LDS-resident input tile reused across the taps is the next step)
// CG = output-channel groups of waves (threads = 256 CG): CG = 2 gives a 128-pixel x 256-channel workgroup tile, halving the
// L2 traffic of both operands for the wide layers (these kernels are L2-bound: each tap re-reads the input)
template <int MODE, int MT, int PT = 1, int CG = 1>
__global__ __launch_bounds__(256 * CG) void vid_conv_kernel(VidConvArgs a) {
    constexpr int CO = 32 * MT * CG, NT = 256 * CG;
    __shared__ __attribute__((aligned(16))) _Float16 Ws[2][2 * CO * V_LDW];
    __shared__ int tab[MODE == VM_STEM ? 256 : 1];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_ & 3, wcg = wave_ >> 2;     // pixel tile of the wave, output-channel group of the wave
    // XCD-aware block order (workgroup L runs on XCD L % 8): the output-channel blocks of one pixel tile share its gathered
    // input, so they get the same residue and consecutive slots -> the input is fetched into ONE L2, once
    const int ncb = a.Cout / CO;
    const int slot = blockIdx.x >> 3;
    const int ptile = (slot / ncb) * 8 + (blockIdx.x & 7);
    const int cb = slot - (slot / ncb) * ncb;          // block of CO output channels
    if (ptile * 128 * PT >= a.N * a.Ho * a.Wo) return;  // uniform; before any barrier
    const int npix = a.N * a.Ho * a.Wo;
    const int K = MODE == VM_STEM ? 256 : (MODE == VM_C3 ? 9 : 1) * a.Cin;
    const int nchunk = K / 32;
    const int Hp = a.Hi + 2, Wp = a.Wi + 2;  // padded input plane (2-D modes)
    if (MODE == VM_STEM && tid < 256) {  // (CG == 1 for the stem)
        const int k = tid, dt = k / 49, rem = k - dt * 49, dy = rem / 7, dx = rem - dy * 7;
        tab[tid] = k < 245 ? (dt * 94 + dy) * 94 + dx : 0;  // offset inside the padded volume
    }
    // this lane's output pixels (one per pixel tile) and the base of their receptive fields in the input
    bool live[PT];
    int pcs[PT], ns[PT], ys[PT], xs[PT];
    size_t base[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int pix = (ptile * PT + t) * 128 + wave * 32 + r;
        live[t] = pix < npix;
        const int pc = live[t] ? pix : npix - 1;
        const int n = pc / (a.Ho * a.Wo), yx = pc - n * (a.Ho * a.Wo), y = yx / a.Wo, x = yx - y * a.Wo;
        pcs[t] = pc; ns[t] = n; ys[t] = y; xs[t] = x;
        if (MODE == VM_STEM) {
            const int b = n / a.T, tt = n - b * a.T;
            base[t] = ((size_t)(b * (a.T + 4) + tt) * 94 + 2 * y) * 94 + 2 * x;
        } else {
            const int off = MODE == VM_C3 ? 0 : 1;  // 1x1: no padding -> interior starts at (1,1)
            base[t] = (((size_t)n * Hp + (y * a.stride + off)) * Wp + (x * a.stride + off)) * a.Cin;
        }
    }
    // image: [Cout/64][K/32][hi|lo][64][32]; a block of CO channels = CO/64 consecutive 64-blocks
    const half8* wimg = a.w16 + (size_t)cb * (CO / 64) * nchunk * 512;
    constexpr int NPRE = 2 * CO * 4 / NT;  // 16-byte pieces per thread and chunk
    half8 pre[NPRE];
    // pieces of a chunk: [sub-block CO/64][part 2][co 64][kq 4] -> 512 per sub-block
    auto stage_load = [&](int c) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + NT * j, sb = i >> 9, w = i & 511;
            pre[j] = wimg[((size_t)sb * nchunk + c) * 512 + w];
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + NT * j, sb = i >> 9, w = i & 511;
            const int part = w >> 8, co = sb * 64 + ((w >> 2) & 63);
            *reinterpret_cast<half8*>(&Ws[buf][(part * CO + co) * V_LDW + (w & 3) * 8]) = pre[j];
        }
    };
    f32x16 acc[PT][MT];
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][m][q] = 0.f;
    auto gather = [&](int c, float (&v)[PT][2][8]) {
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (MODE == VM_STEM) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[t][ks][j] = a.x[base[t] + tab[c * 32 + ks * 16 + 8 * h + j]];
            } else {
                const int k0 = c * 32;
                const int tap = k0 / a.Cin, ci0 = k0 - tap * a.Cin;  // uniform
                const int dy = MODE == VM_C3 ? tap / 3 : 0, dx = MODE == VM_C3 ? tap - 3 * dy : 0;
                const float* xp = a.x + base[t] + ((size_t)dy * Wp + dx) * a.Cin + ci0 + 8 * h;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    *reinterpret_cast<f32x4*>(&v[t][ks][0]) = *reinterpret_cast<const f32x4*>(xp + ks * 16);
                    *reinterpret_cast<f32x4*>(&v[t][ks][4]) = *reinterpret_cast<const f32x4*>(xp + ks * 16 + 4);
                }
            }
        }
    };
    stage_load(0);
    stage_write(0);
    __syncthreads();  // also orders the stem's offset table
    float v[PT][2][8];
    gather(0, v);
    for (int c = 0; c < nchunk; ++c) {
        const int cn = c + 1 < nchunk ? c + 1 : c;
        stage_load(cn);
        half8 bh[PT][2], bl[PT][2];
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const _Float16 hi = (_Float16)v[t][ks][j];
                    bh[t][ks][j] = hi;
                    bl[t][ks][j] = (_Float16)(v[t][ks][j] - (float)hi);
                }
        gather(cn, v);  // the next chunk's activations arrive under this chunk's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const half8 ah = *reinterpret_cast<const half8*>(&Ws[c & 1][((wcg * MT + m) * 32 + r) * V_LDW + ks * 16 + 8 * h]);
                const half8 al = *reinterpret_cast<const half8*>(&Ws[c & 1][(CO + (wcg * MT + m) * 32 + r) * V_LDW + ks * 16 + 8 * h]);
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t][ks], acc[t][m], 0, 0, 0);
                    acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t][ks], acc[t][m], 0, 0, 0);
                    acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t][ks], acc[t][m], 0, 0, 0);
                }
            }
        if (c + 1 < nchunk) stage_write((c + 1) & 1);
        __syncthreads();
    }
    constexpr float WINV = 1.0f / 256.0f;
    const int Hop = a.Ho + 2, Wop = a.Wo + 2;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!live[t]) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {  // accumulator registers 4 q4 .. 4 q4 + 3 = 4 consecutive output channels
                const int co = cb * CO + (wcg * MT + m) * 32 + 8 * q4 + 4 * h;
                const size_t o = (MODE == VM_STEM ? (size_t)pcs[t] : ((size_t)ns[t] * Hop + (ys[t] + 1)) * Wop + (xs[t] + 1)) * a.Cout + co;
                f32x4 val;
#pragma unroll
                for (int i = 0; i < 4; ++i) val[i] = fmaf(acc[t][m][4 * q4 + i], WINV, a.bias[co + i]);
                if (a.res) val += *reinterpret_cast<const f32x4*>(a.res + o);
                if (a.slope) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) val[i] = preluf_(val[i], a.slope[co + i]);
                }
                *reinterpret_cast<f32x4*>(a.out + o) = val;
            }
    }
}

// ---------------------------------------------------------------- stride-1 3x3 convolution with an LDS-resident input tile
// The gather kernel above re-reads the input once per tap (9x the tensor through L2, which is what bounds it).  Here a
// workgroup owns up to 256 output pixels -- G whole frames, or a band of RH rows of one frame (layer 1: 11 of 22 rows) --
// and loads their input footprint (G x (RH+2) x (W+2) padded pixels x 32 channels) into LDS once per 32-channel chunk;
// the nine taps then read their B fragments from LDS (two ds_read_b128 per fragment; 144-byte pixel rows, conflict-free).
// 4 waves x 2 pixel tiles x MT output tiles; the weight chunks stream through a double-buffered LDS image as above.
constexpr int VL_PIXLD = 36;  // floats per staged pixel (32 channels + 4 pad)

template <int MT>
__global__ __launch_bounds__(256) void vid_conv3l_kernel(VidConvArgs a, int G, int RH, int bands) {
    constexpr int CO = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char vsm[];
    _Float16* Ws = reinterpret_cast<_Float16*>(vsm);                       // [2][2 * CO * V_LDW]
    float* Xs = reinterpret_cast<float*>(vsm + (size_t)2 * 2 * CO * V_LDW * 2);  // [G][(RH+2)][Wp][VL_PIXLD]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = blockIdx.y;
    const int H = a.Ho, W = a.Wo, Wp = W + 2, Hp = H + 2;  // stride 1: input and output geometry coincide
    const int grp = blockIdx.x / bands, band = blockIdx.x - grp * bands;
    const int n0 = grp * G, y0 = band * RH;
    const int nfr = min(G, a.N - n0), rows = min(RH, H - y0);
    const int npx = nfr * rows * W;         // live output pixels of this workgroup (<= 256)
    const int tile_px = nfr * (rows + 2) * Wp;  // staged input pixels
    // this lane's two output pixels: slot -> (frame in group, row in band, column); staged-pixel index of tap (0,0)
    bool live[2];
    int sbase[2], on[2], oy[2], ox[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int slot = (wave * 2 + t) * 32 + r;
        live[t] = slot < npx;
        const int sc = live[t] ? slot : 0;
        const int gf = sc / (rows * W), rem = sc - gf * (rows * W), y = rem / W, x = rem - y * W;
        on[t] = n0 + gf; oy[t] = y0 + y; ox[t] = x;
        sbase[t] = ((gf * (rows + 2) + y) * Wp + x) * VL_PIXLD + 8 * h;
    }
    const int nci = a.Cin / 32;
    const half8* wimg = a.w16 + (size_t)cb * (CO / 64) * (9 * nci) * 512;
    half8 pre[2 * (CO / 64)];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int sb = 0; sb < CO / 64; ++sb)
#pragma unroll
            for (int j = 0; j < 2; ++j) pre[sb * 2 + j] = wimg[((size_t)sb * (9 * nci) + c) * 512 + tid + 256 * j];
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int sb = 0; sb < CO / 64; ++sb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j;
                const int part = i >> 8, co = sb * 64 + ((i >> 2) & 63);
                *reinterpret_cast<half8*>(Ws + buf * (2 * CO * V_LDW) + (part * CO + co) * V_LDW + (i & 3) * 8) = pre[sb * 2 + j];
            }
    };
    f32x16 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][m][q] = 0.f;
    // weight chunk order: for each 32-channel chunk cic, the nine taps (image chunk index = tap * nci + cic)
    stage_load(0);
    int wbuf = 0;
    for (int cic = 0; cic < nci; ++cic) {
        __syncthreads();  // every wave has finished reading the previous input tile (and the last weight buffer)
        // ---- input tile of this channel chunk: global (n, Hp, Wp, Cin) channel-last -> LDS [pixel][32 + pad]
        for (int i = tid; i < tile_px * 8; i += 256) {
            const int px = i >> 3, piece = i & 7;
            const int gf = px / ((rows + 2) * Wp), rem = px - gf * ((rows + 2) * Wp), yy = rem / Wp, xx = rem - yy * Wp;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)(n0 + gf) * Hp + (y0 + yy)) * Wp + xx) * a.Cin + cic * 32 + piece * 4);
            *reinterpret_cast<f32x4*>(Xs + px * VL_PIXLD + piece * 4) = v;
        }
        stage_write(wbuf);
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int cnext = tap < 8 ? (tap + 1) * nci + cic : (cic + 1 < nci ? cic + 1 : cic);  // next chunk in this order
            stage_load(cnext);
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int toff = (dy * Wp + dx) * VL_PIXLD;
            const _Float16* wb = Ws + wbuf * (2 * CO * V_LDW);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 bh[2], bl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[8];
                    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(Xs + sbase[t] + toff + ks * 16);
                    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(Xs + sbase[t] + toff + ks * 16 + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const _Float16 hi = (_Float16)v[j];
                        bh[t][j] = hi;
                        bl[t][j] = (_Float16)(v[j] - (float)hi);
                    }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const half8 ah = *reinterpret_cast<const half8*>(wb + (m * 32 + r) * V_LDW + ks * 16 + 8 * h);
                    const half8 al = *reinterpret_cast<const half8*>(wb + (CO + m * 32 + r) * V_LDW + ks * 16 + 8 * h);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t], acc[t][m], 0, 0, 0);
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t], acc[t][m], 0, 0, 0);
                    }
                }
            }
            if (tap < 8) {  // the next tap's weights go to the other buffer; the chunk after tap 8 is written after the tile swap
                stage_write(wbuf ^ 1);
                __syncthreads();
                wbuf ^= 1;
            }
        }
        wbuf ^= 1;  // tap 8's prefetch (first chunk of the next cic) is written at the top of the next iteration
    }
    constexpr float WINV = 1.0f / 256.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (!live[t]) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int co = cb * CO + m * 32 + 8 * q4 + 4 * h;
                const size_t o = (((size_t)on[t] * Hp + (oy[t] + 1)) * Wp + (ox[t] + 1)) * a.Cout + co;
                f32x4 val;
#pragma unroll
                for (int i = 0; i < 4; ++i) val[i] = fmaf(acc[t][m][4 * q4 + i], WINV, a.bias[co + i]);
                if (a.res) val += *reinterpret_cast<const f32x4*>(a.res + o);
                if (a.slope) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) val[i] = preluf_(val[i], a.slope[co + i]);
                }
                *reinterpret_cast<f32x4*>(a.out + o) = val;
            }
    }
}

__global__ __launch_bounds__(256) void vid_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int B, int T) {
    // one thread per element of the padded volume (B, T+4, 94, 94)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * (T + 4) * 94 * 94;
    if (i >= total) return;
    const int xx = (int)(i % 94), yy = (int)((i / 94) % 94), tt = (int)((i / (94 * 94)) % (T + 4)), b = (int)(i / ((size_t)94 * 94 * (T + 4)));
    const int t = tt - 2, y = yy - 3, xq = xx - 3;
    const bool in = t >= 0 && t < T && y >= 0 && y < 88 && xq >= 0 && xq < 88;
    xp[i] = in ? x[(((size_t)b * T + t) * 88 + y) * 88 + xq] : 0.f;
}

// channel-last (n, Hi, Hi, C) -> padded (n, Ho+2, Ho+2, C) interior: max over rows 2y-1..2y+1, cols 2x-1..2x+1 in the image;
// one thread per (pixel, 4 channels): 16-byte accesses
__global__ __launch_bounds__(256) void vid_maxpool_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int Hi, int Ho) {
    const int c4 = C / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * Ho * Ho * c4) return;
    const int c = (int)(i % c4) * 4, x = (int)((i / c4) % Ho), y = (int)((i / ((size_t)c4 * Ho)) % Ho), n = (int)(i / ((size_t)c4 * Ho * Ho));
    const float* p = in + (size_t)n * Hi * Hi * C + c;
    f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = 2 * y + dy, xx = 2 * x + dx;
            if (yy >= 0 && yy < Hi && xx >= 0 && xx < Hi) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + ((size_t)yy * Hi + xx) * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
    *reinterpret_cast<f32x4*>(out + (((size_t)n * (Ho + 2) + (y + 1)) * (Ho + 2) + (x + 1)) * C + c) = m;
}

// padded channel-last (n = b*T + t, H+2, H+2, C) -> (B, C, T) mean over the H x H interior
__global__ __launch_bounds__(256) void vid_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int T, int C, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T * C) return;
    const int c = i % C, n = i / C, b = n / T, t = n - b * T;
    const float* p = in + (size_t)n * (H + 2) * (H + 2) * C + c;
    float s = 0.f;
    for (int y = 1; y <= H; ++y)
        for (int x = 1; x <= H; ++x) s += p[((size_t)y * (H + 2) + x) * C];
    out[((size_t)b * C + c) * T + t] = s / (float)(H * H);
}

template <int MT>
int conv3l_launch(const VidConvArgs& a, hipStream_t st) {
    // frames per workgroup / row band so that a workgroup owns <= 256 output pixels
    const int hw = a.Ho * a.Wo;
    int G = 1, RH = a.Ho, bands = 1;
    if (hw > 256) {
        bands = cdiv(hw, 256);
        RH = cdiv(a.Ho, bands);
        bands = cdiv(a.Ho, RH);
        if (RH * a.Wo > 256) return RTFS_ERR_SHAPE;
    } else {
        G = 256 / hw;
    }
    const size_t lds = (size_t)2 * 2 * (32 * MT) * V_LDW * 2 + (size_t)G * (RH + 2) * (a.Wo + 2) * VL_PIXLD * 4;
    if (lds > 160 * 1024) return RTFS_ERR_SHAPE;
    if (rtfs_set_max_lds((const void*)vid_conv3l_kernel<MT>, lds) != RTFS_OK) return RTFS_ERR_LAUNCH;
    hipLaunchKernelGGL((vid_conv3l_kernel<MT>), dim3(cdiv(a.N, G) * bands, a.Cout / (32 * MT)), dim3(256), lds, st, a, G, RH, bands);
    return rtfs_launch_status();
}

// zero the one-pixel border of a padded channel-last activation buffer (n, H+2, H+2, C): 4 (H+1) border pixels per frame
__global__ __launch_bounds__(256) void vid_border_kernel(float* __restrict__ buf, int N, int H, int C) {
    const int c4 = C / 4, nb = 4 * (H + 1);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * nb * c4) return;
    const int cc = (int)(i % c4), k = (int)((i / c4) % nb), n = (int)(i / ((size_t)c4 * nb));
    const int Hp = H + 2, side = k / (H + 1), j = k - side * (H + 1);
    // side 0: top row, columns 0..H; 1: right column, rows 0..H; 2: bottom row, columns 1..H+1; 3: left column, rows 1..H+1
    const int y = side == 0 ? 0 : side == 1 ? j : side == 2 ? Hp - 1 : j + 1;
    const int x = side == 0 ? j : side == 1 ? Hp - 1 : side == 2 ? j + 1 : 0;
    *reinterpret_cast<f32x4*>(buf + (((size_t)n * Hp + y) * Hp + x) * C + cc * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

int conv_launch(int mode, const VidConvArgs& a, hipStream_t st) {
    const int px = cdiv(a.N * a.Ho * a.Wo, 128);
    // LDS-resident input tile (nine taps per load): pays off for the large-plane 64-channel layer only -- the deeper layers'
    // small planes give too few, too LDS-heavy workgroups (measured 376 us against 243 for the gather kernel)
    if (mode == VM_C3 && a.stride == 1 && a.Hi == a.Ho && a.Wi == a.Wo && a.Cout == 64) return conv3l_launch<2>(a, st);
    auto grid1 = [&](int ncb) { return dim3((unsigned)(cdiv(px, 8) * 8 * ncb)); };
    if (mode == VM_STEM) hipLaunchKernelGGL((vid_conv_kernel<VM_STEM, 2>), grid1(1), dim3(256), 0, st, a);
    else if (a.Cout % 128 == 0) {  // (CG = 2, a 256-channel tile on 8 waves, measured slower: 309 us against 243)  // 128 output channels per workgroup: the gathered activations are reused twice as often
        if (mode == VM_C3) hipLaunchKernelGGL((vid_conv_kernel<VM_C3, 4>), grid1(a.Cout / 128), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((vid_conv_kernel<VM_C1, 4>), grid1(a.Cout / 128), dim3(256), 0, st, a);
    } else {
        if (mode == VM_C3) hipLaunchKernelGGL((vid_conv_kernel<VM_C3, 2>), grid1(a.Cout / 64), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((vid_conv_kernel<VM_C1, 2>), grid1(a.Cout / 64), dim3(256), 0, st, a);
    }
    return rtfs_launch_status();
}

}  // namespace

// Parameter pack (rtfs-net_amd/packing.py:pack_video): for the stem and then for every convolution of the trunk in
// forward order (conv1, [downsample], conv2 per block): image (Cout * Kpad floats), bias (Cout), slope (Cout; zeros = none)
size_t video_pack_floats() {
    size_t n = 0;
    auto add = [&](size_t cout, size_t k) { n += (cout * k + 63) / 64 * 64 + 2 * ((cout + 63) / 64 * 64); };
    add(64, 256);
    int inpl = 64;
    const int planes[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < 2; ++bi) {
            const int cin = bi == 0 ? inpl : planes[li];
            add(planes[li], 9 * cin);
            if (bi == 0 && li > 0) add(planes[li], cin);
            add(planes[li], 9 * planes[li]);
            if (bi == 1) inpl = planes[li];
        }
    return n;
}

size_t video_workspace_bytes(int B, int T) {
    const size_t N = (size_t)B * T;
    size_t n = (size_t)B * (T + 4) * 94 * 94 + N * 64 * 44 * 44;
    const int planes[4] = {64, 128, 256, 512}, hw[4] = {24, 13, 8, 5};
    for (int li = 0; li < 4; ++li) n += 3 * N * planes[li] * hw[li] * hw[li];
    return n * sizeof(float);
}

int video_frontend(const float* lips, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes, hipStream_t st) {
    if (B < 1 || T < 1) return RTFS_ERR_SHAPE;
    if (ws_bytes < video_workspace_bytes(B, T)) return RTFS_ERR_WORKSPACE;
    const int N = B * T;
    float* w = reinterpret_cast<float*>(ws);
    float* xp = w;
    w += (size_t)B * (T + 4) * 94 * 94;
    float* y44 = w;
    w += (size_t)N * 64 * 44 * 44;
    const int planes[4] = {64, 128, 256, 512}, hw[4] = {24, 13, 8, 5};
    float* buf[4][3];
    float* act0 = w;
    for (int li = 0; li < 4; ++li)
        for (int k = 0; k < 3; ++k) {
            buf[li][k] = w;
            w += (size_t)N * planes[li] * hw[li] * hw[li];
        }
    (void)act0;
    for (int li = 0; li < 4; ++li)  // zero borders (interiors are always written before they are read)
        for (int k = 0; k < 3; ++k) {
            const size_t cnt = (size_t)N * 4 * (hw[li] - 1) * (planes[li] / 4);
            hipLaunchKernelGGL(vid_border_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, buf[li][k], N, hw[li] - 2, planes[li]);
        }
    if (rtfs_launch_status()) return RTFS_ERR_LAUNCH;
    size_t off = 0;
    auto take = [&](size_t n) {
        const float* r = pack + off;
        off += (n + 63) / 64 * 64;
        return r;
    };
    {
        const size_t total = (size_t)B * (T + 4) * 94 * 94;
        hipLaunchKernelGGL(vid_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, lips, xp, B, T);
        if (rtfs_launch_status()) return RTFS_ERR_LAUNCH;
    }
    {  // stem conv + BN + PReLU -> (N,64,44,44); max pool -> buf[0][0] (N,64,24,24 padded)
        VidConvArgs a{};
        a.x = xp; a.w16 = reinterpret_cast<const half8*>(take(64 * 256)); a.bias = take(64); a.slope = take(64);
        a.res = nullptr; a.out = y44; a.N = N; a.Cin = 1; a.Cout = 64; a.Hi = 88; a.Wi = 88; a.Ho = 44; a.Wo = 44; a.stride = 2; a.T = T;
        if (int rc = conv_launch(VM_STEM, a, st)) return rc;
        hipLaunchKernelGGL(vid_maxpool_kernel, dim3((unsigned)(((size_t)N * 16 * 22 * 22 + 255) / 256)), dim3(256), 0, st, y44, buf[0][0], N, 64, 44, 22);
        if (rtfs_launch_status()) return RTFS_ERR_LAUNCH;
    }
    const float* x = buf[0][0];
    int xi = 0;      // which of the 3 buffers of the current level holds x
    int inpl = 64, Hin = 22;
    for (int li = 0; li < 4; ++li) {
        const int pl = planes[li], Hout = hw[li] - 2;
        for (int bi = 0; bi < 2; ++bi) {
            const int stride = (bi == 0 && li > 0) ? 2 : 1;
            const int cin = bi == 0 ? inpl : pl;
            // pick two scratch buffers of this level different from the one holding x (x may live on the previous level)
            int s0 = 0, s1 = 1;
            if (bi == 1 || li == 0) {
                s0 = (xi + 1) % 3;
                s1 = (xi + 2) % 3;
            }
            float* hbuf = buf[li][s0];
            float* obuf = buf[li][s1];
            VidConvArgs c1{};
            c1.x = x; c1.w16 = reinterpret_cast<const half8*>(take((size_t)pl * 9 * cin)); c1.bias = take(pl); c1.slope = take(pl);
            c1.res = nullptr; c1.out = hbuf; c1.N = N; c1.Cin = cin; c1.Cout = pl; c1.Hi = Hin; c1.Wi = Hin; c1.Ho = Hout; c1.Wo = Hout; c1.stride = stride;
            if (int rc = conv_launch(VM_C3, c1, st)) return rc;
            const float* resid = x;
            if (bi == 0 && li > 0) {  // downsample branch: 1x1 stride 2 + BN -> third buffer of this level
                float* rbuf = buf[li][2];
                VidConvArgs d{};
                d.x = x; d.w16 = reinterpret_cast<const half8*>(take((size_t)pl * cin)); d.bias = take(pl); (void)take(pl); d.slope = nullptr;
                d.res = nullptr; d.out = rbuf; d.N = N; d.Cin = cin; d.Cout = pl; d.Hi = Hin; d.Wi = Hin; d.Ho = Hout; d.Wo = Hout; d.stride = 2;
                if (int rc = conv_launch(VM_C1, d, st)) return rc;
                resid = rbuf;
            }
            VidConvArgs c2{};
            c2.x = hbuf; c2.w16 = reinterpret_cast<const half8*>(take((size_t)pl * 9 * pl)); c2.bias = take(pl); c2.slope = take(pl);
            c2.res = resid; c2.out = obuf; c2.N = N; c2.Cin = pl; c2.Cout = pl; c2.Hi = Hout; c2.Wi = Hout; c2.Ho = Hout; c2.Wo = Hout; c2.stride = 1;
            if (int rc = conv_launch(VM_C3, c2, st)) return rc;
            x = obuf;
            xi = s1;
            Hin = Hout;
        }
        inpl = pl;
    }
    hipLaunchKernelGGL(vid_avgpool_kernel, dim3(cdiv(N * 512, 256)), dim3(256), 0, st, x, out, B, T, 512, 3);
    return rtfs_launch_status();
}
