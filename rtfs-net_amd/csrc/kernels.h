// Internal (C++) launch interface between the C-ABI layer (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "common.h"

struct PwArgs {
    const float* x = nullptr;     // (B, CIN, P)
    const float* x2 = nullptr;    // optional second addend (gateway: the bottleneck residual a1)
    float* res_out = nullptr;     // gateway: PReLU(dw1x1(x [+x2])) written through, (B, CIN, P)
    const float* wt = nullptr;    // (CIN, COUT) transposed 1x1 weight (exact-f32 kernels)
    const void* w16 = nullptr;    // [CIN/32][hi|lo][COUT][32] f16 split image of 256*W (f16x3 kernels)
    const void* w16b = nullptr;   // S3 + decoder taps: K-permuted taps image (packing.taps_perm_image)
    const float* bias = nullptr;  // (COUT)
    const float* aux = nullptr;   // EPI_BIAS_RES: residual (B,COUT,P); EPI_S3: encoder output a0 (B,COUT,P)
    float* out = nullptr;         // (B, COUT, P)
    const double* stats = nullptr;  // PRO_GLN_RELU: (B,2) sum / sumsq of x
    double inv_count = 0;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const float* gw = nullptr;  // gateway depthwise scale (CIN)
    const float* gb = nullptr;  // gateway depthwise bias (CIN)
    const float* slope = nullptr;  // PReLU slope (1)
    int P = 0;
    int cs = 0;         // channel stride (floats) of every (B, C, P) tensor of the call; 0 = P (contiguous).  The fused separator pads it
                        // to a multiple of 32 floats so that every 64-pixel wave segment is whole 128-byte lines ("pitch")
    int cout_live = 0;  // EPI_TAPS: number of real output channels
    unsigned* tile_ctr = nullptr;  // persistent kernels: zeroed counter word that hands out tiles (null: static stride)
    // optional CAF prologue of the gateway kernel (fused separator path): x <- CAF(x, video) before the residual add
    const float* caf_r = nullptr;    // (B,256,Tv) resize(video)
    const float* caf_att = nullptr;  // (B,256,Tv) softmax attention
    const float *caf_w_key = nullptr, *caf_bn_key = nullptr, *caf_w_val = nullptr, *caf_bn_val = nullptr;
    int caf_T = 0, caf_F = 0, caf_Tv = 0;
};

int launch_stft(const float* wav, float* spec, int B, int L, int T, hipStream_t st);
int launch_enc_conv(const float* spec, const float* w, float* a0, double* stats, int B, int C, int T, int F, size_t cs,
                    size_t bs, hipStream_t st);
// gLN statistics of the encoder output computed from the spectrogram (a0 is not formed) + the encoder's f16x3 fragment image (32 KB at img)
// + (one extra row of workgroups) up to two 256 -> 256 f16x3 weight images [8 chunks][hi|lo][256 rows][64 B] copied with their rows padded to
// 72 bytes (294912 B each): the LDS image of the head / tail kernels, which they fetch by LDS-DMA - a DMA writes LDS linearly, so the
// conflict-free row padding has to exist in memory
struct EncPadJobs {
    const void* src[2] = {nullptr, nullptr};
    void* dst[2] = {nullptr, nullptr};
};
int launch_enc_stats(const float* spec, const float* w, double* stats, void* img, const EncPadJobs& pad, int B, int T, int F, hipStream_t st);
int launch_dec_istft(const float* z, float* wav, int B, int T, int F, int L, size_t zcs, size_t zbs, hipStream_t st);

int launch_pw_audio_bn(const PwArgs& a, int B, hipStream_t st);
int launch_pw_gateway_proj(const PwArgs& a, int B, hipStream_t st);
int launch_pw_residual(const PwArgs& a, int B, hipStream_t st);
int launch_pw_s3(const PwArgs& a, int B, hipStream_t st);
int launch_pw_dec_taps(const PwArgs& a, int B, hipStream_t st);
int launch_pw16_audio_bn(const PwArgs& a, int B, hipStream_t st);
int launch_pw16_gateway_proj(const PwArgs& a, int B, hipStream_t st);
int launch_pw16_residual(const PwArgs& a, int B, hipStream_t st);
int launch_pw16_s3(const PwArgs& a, int B, hipStream_t st);
int launch_pw16_dec_taps(const PwArgs& a, int B, hipStream_t st);
// 256 -> 256 with register-resident pixels (k_pwr.hip)
int launch_pwr_audio_bn(const PwArgs& a, int B, hipStream_t st);
int launch_pwr_s3(const PwArgs& a, int B, hipStream_t st);
int launch_pwr_s3_taps(const PwArgs& a, int B, hipStream_t st);  // S3 + decoder taps: writes z (B,18,P), not the separated spectrum
// block boundary: residual_conv(i) + gateway/projection(i+1) back to back (k_pws.hip)
struct B2bArgs {
    const float* x = nullptr;     // expanded_i (B,64,P)
    float* res = nullptr;         // in: residual_i, out: residual_{i+1} (B,256,P), rewritten in place
    const float* a1 = nullptr;    // bottleneck output (B,256,P), RefinementModule's residual
    float* xenc = nullptr;        // x_enc_{i+1} (B,64,P)
    const void* w1_16 = nullptr;  // residual_conv f16x3 image [2][hi|lo][256][32]
    const float* b1 = nullptr;    // (256)
    const void* w2_16 = nullptr;  // projection f16x3 image [8][hi|lo][64][32] with the K axis in accumulator order
    const float* bp = nullptr;    // (64)
    const float *gw = nullptr, *gb = nullptr, *slope = nullptr;
    int P = 0;
    int cs = 0;  // channel stride of x / res / a1 / xenc (floats); 0 = P
    const float *caf_r = nullptr, *caf_att = nullptr, *caf_w_key = nullptr, *caf_bn_key = nullptr, *caf_w_val = nullptr, *caf_bn_val = nullptr;
    const float *caf_rt = nullptr, *caf_attt = nullptr;  // (B, Tv, 256) transposed copies of caf_r / caf_att (k_b2b.hip)
    int caf_T = 0, caf_F = 0, caf_Tv = 0;
    // k_b2b.hip only: bit 0 = `res` does not contain a1 yet (read a1, add it to the block input); bit 1 = write the new residual WITH a1 added, so
    // that the next boundary runs with bit 0 clear and never reads a1.  1 = the reference sequence as written (launch_pws_b2b knows no other).
    // bit 2 (with a CAF, i.e. the first boundary): `res` is not read - residual_0 = PReLU(gw a1 + gb) is formed from the a1 rows (the head kernel
    // then need not write it).
    int a1_mode = 1;
};
int launch_pws_b2b(const B2bArgs& a, int B, hipStream_t st);
bool launch_pws_b2b4_qualifies(const B2bArgs& a);
// second generation (k_b2b.hip): padded channel rows (cs % 64 == 0), no CAF; ctr = zeroed tile counter of this launch or null.
// RTFS_ERR_ARG = call does not qualify, use launch_pws_b2b
int launch_pws_b2b4(const B2bArgs& a, int B, unsigned* ctr, hipStream_t st);
// audio bottleneck (gLN -> ReLU -> 1x1 256->256) + first block head (gateway + projection) in one kernel (k_bnh.hip): padded rows only,
// RTFS_ERR_ARG = call does not qualify, use launch_pwr_audio_bn + launch_pws_head4
struct BnHeadArgs {
    const float* spec = nullptr;    // spectrogram (B,2,T,F): the encoder output a0 is rebuilt on the fly, not read
    const void* enc_img = nullptr;  // encoder f16x3 fragment image (enc_stats_kernel)
    int T = 0, F = 0;
    float* a1 = nullptr;            // bottleneck output (B,256,cs)
    float* res = nullptr;           // gateway output (B,256,cs); null = not written (the first boundary forms it from a1: B2bArgs::a1_mode bit 2)
    float* xenc = nullptr;          // projection output (B,64,cs)
    const double* stats = nullptr;  // (B,2) sum / sumsq of a0
    double inv_count = 0;
    const float *gamma = nullptr, *beta = nullptr;  // gLN (256)
    const void* w16 = nullptr;      // bottleneck f16x3 image [8][hi|lo][256][32]
    const float* bias = nullptr;    // (256)
    const float *gw = nullptr, *gb = nullptr, *slope = nullptr;  // gateway dw 1x1 (256), PReLU (1)
    const void* w2_16 = nullptr;    // projection f16x3 image [8][hi|lo][64][32], K in accumulator order
    const float* bp = nullptr;      // (64)
    int P = 0, cs = 0;
    int stagger = 2;                // start offset between the four workgroup groups, in units of 8128 cycles (s_sleep 127); 0 = none
    unsigned* tile_ctr = nullptr;
};
bool launch_bn_head_qualifies(const BnHeadArgs& a);  // same conditions as launch_tail_s3t (same P, same pitch)
int launch_bn_head(const BnHeadArgs& a, int B, hipStream_t st);
// residual conv of the last block application + S3 mask + complex product + decoder taps in one kernel (k_s3f.hip): padded rows only,
// RTFS_ERR_ARG = call does not qualify, use launch_pws_residual + launch_pwr_s3_taps
struct TailS3Args {
    const float* x = nullptr;     // expanded (B,64,cs)
    const float* res = nullptr;   // residual (B,256,cs)
    const float* spec = nullptr;  // spectrogram (B,2,T,F): the encoder output a0 is rebuilt on the fly, not read
    const void* enc_img = nullptr;  // encoder f16x3 fragment image (enc_stats_kernel)
    int T = 0, F = 0;
    float* z = nullptr;           // decoder taps (B,cout_live,cs)
    const void* w1_16 = nullptr;  // residual_conv f16x3 image [2][hi|lo][256][32]
    const float* b1 = nullptr;    // (256)
    const void* w16 = nullptr;    // mask conv f16x3 image [8][hi|lo][256 rows of 72 bytes: 32 halfs + 8 bytes of padding] (EncPadJobs)
    const float* bias = nullptr;  // (256)
    const float* slope = nullptr; // mask head PReLU (1)
    const void* w16b = nullptr;   // K-permuted taps image (packing.taps_perm_image)
    const double* stats = nullptr;  // (B,2) sum / sumsq of a0 (range normalisation of the taps operand), may be null
    double inv_count = 0;
    int P = 0, cs = 0, cout_live = 0;
    unsigned* tile_ctr = nullptr;
};
int launch_tail_s3t(const TailS3Args& a, int B, hipStream_t st);
int launch_pws_head4(const PwArgs& a, int B, hipStream_t st);  // block head on padded rows (k_b2b.hip); RTFS_ERR_ARG = use launch_pws_gateway_proj
int launch_pws_gateway_proj(const PwArgs& a, int B, hipStream_t st);
int launch_pws_residual(const PwArgs& a, int B, hipStream_t st);
int launch_mfma_f16_selftest(const float* A, const float* B, float* D, hipStream_t st);

// Depthwise 4x4 family.  Tensors are (B, C, H, W) contiguous.
struct DwArgs {
    const float* x = nullptr;
    // input fold (IN_AFFINE): x is a pre-norm conv output, its gLN is applied at load time
    const double* in_stats = nullptr;
    double in_inv_count = 0;
    const float* in_gamma = nullptr;
    const float* in_beta = nullptr;
    const float* w[4] = {nullptr, nullptr, nullptr, nullptr};     // (C,16) each
    const float* bias[4] = {nullptr, nullptr, nullptr, nullptr};  // (C) or null
    float* out[4] = {nullptr, nullptr, nullptr, nullptr};
    double* stats_out[4] = {nullptr, nullptr, nullptr, nullptr};  // (B,2) each
    int C = 0, H = 0, W = 0, TH = 8;
    int cs = 0;  // channel stride (floats) of the (B, C, H, W) tensors x / out[] / addend; 0 = H * W.  gate / emb (Hg x Wg) stay contiguous
    int in_combine = 0;  // lane-exchange kernels, MODE 0: the input is the same-size TFAR combination gLN(x) * sigmoid(gLN(gate)) + gLN(emb) of three
                         // pre-norm tensors (folds: loc_* / gate_* / emb_*, all with g_inv_count), formed at load time (fusion.py:62-67)
    int rev = 0;  // lane-exchange kernels: walk the tensor back to front (a consumer that starts where its producer stopped finds that end in the
                  // memory-side cache: tools/bench_mall.hip)
    int job_stride = 0, job_off = 0;  // two jobs interleaved per sample in one launch (launch_dw_s2_stats): logical block id -> sample id / job_stride,
                                      // this job's block (id % job_stride) - job_off if that lies in [0, gx * gy); 0 = one job (id -> x, y, z)
    int gx = 0, gy = 0, nblk = 0, blk0 = 0;  // set by the launchers of the lane-exchange kernels: logical grid (gx, gy, B) behind a 1-D launch
                                              // (XCD order); blk0 = first block of this job when several jobs share one launch
    // MODE 2 (TFAR apply) / stride-2 kernel: the low-resolution side
    int Hg = 0, Wg = 0;
    const double* loc_stats = nullptr;
    double loc_inv_count = 0;
    const float* loc_gamma = nullptr;
    const float* loc_beta = nullptr;
    const float* gate = nullptr;
    const double* gate_stats = nullptr;
    const float* gate_gamma = nullptr;
    const float* gate_beta = nullptr;
    const float* emb = nullptr;
    const double* emb_stats = nullptr;
    const float* emb_gamma = nullptr;
    const float* emb_beta = nullptr;
    double g_inv_count = 0;
    const float* addend = nullptr;  // optional "+ d0" (pre-norm c0 with its fold)
    const double* add_stats = nullptr;
    double add_inv_count = 0;
    const float* add_gamma = nullptr;
    const float* add_beta = nullptr;
};

struct GCombineArgs {
    const float *l, *gate, *emb;
    const double *l_stats, *gate_stats, *emb_stats;
    const float *l_gamma, *l_beta, *gate_gamma, *gate_beta, *emb_gamma, *emb_beta;
    double inv_count;
    float* out;
    int C, HW;
};

int launch_dw_s1(const DwArgs& a, int nconv, bool in_affine, int mode, int B, hipStream_t st);
// One launch for the block's three independent low-resolution conv jobs (steps 10 and 11: four plain convs on one input as two 2-conv
// jobs + one conv on a gLN-folded input): each alone is 576 workgroups, too few to fill the chip
int launch_dw_g3(const DwArgs& conv4, const DwArgs& aff1, int B, hipStream_t st);
int launch_dw_s2_pool(const DwArgs& a, int B, hipStream_t st);
// the stride-2 + pool pass and a statistics-only pass (MODE 1, input fold) over the SAME input in one launch; RTFS_ERR_ARG = not eligible
int launch_dw_s2_stats(const DwArgs& s2, const DwArgs& stats, int B, hipStream_t st);
int launch_g_form(const float* p0, const float* c1, const double* st1, double inv_count, const float* gamma,
                  const float* beta, float* g, int B, int C, int HW, hipStream_t st);
int launch_g_combine(const GCombineArgs& a, int B, hipStream_t st);
int launch_transpose(const float* x, float* y, int N, int H, int W, hipStream_t st);
int launch_stats(const float* x, double* stats, int B, size_t N, hipStream_t st);
int launch_stats2(const float* x, double* stats, int B, size_t N, double* part, hipStream_t st);  // part: 512 * B doubles

// Fused dual-path SRU sweep.  Sequence n = (b, row): element (c, s) at
//   x[(n / R) * bstride + (n % R) * rstride + c * cstride + s],  s = 0..Ls-1 contiguous.
struct DpArgs {
    const float* x = nullptr;
    float* out = nullptr;
    int R = 0, Ls = 0;
    size_t bstride = 0, rstride = 0, cstride = 0;
    const float* ln_gamma = nullptr;  // (64)
    const float* ln_beta = nullptr;   // (64)
    const float* W0 = nullptr;        // (512, 256)  layer-0 projection, column (dir*32+j)*4+m
    const float* Wl = nullptr;        // 3 x (64, 256) layers 1-3, k=3 columns padded to 4 (m=3 zero)
    const float* wc = nullptr;        // 4 x (128)   [v_f(dir,j) | v_r(dir,j)]
    const float* bias = nullptr;      // 4 x (128)   [b_f | b_r]
    const float* Wt = nullptr;        // (512 = kk*64+ci, 64 co)  ConvTranspose1d weight, re-ordered
    const float* bt = nullptr;        // (64)
    const float* whh = nullptr;       // LSTM cell: 4 x 2 x (32 k, 128 = gate*32 + j) recurrent weights; then W0/Wl columns are
                                      // dir*128 + gate*32 + j (gates i,f,g,o) and bias is 4 x (256) = b_ih + b_hh
};
size_t dualpath_lds_bytes(int Ls);
int launch_dualpath(const DpArgs& a, int nseq, hipStream_t st);
int launch_sru_standalone(const float* x, float* h, int L, int N, const float* W0, const float* Wl, const float* wc,
                          const float* bias, hipStream_t st);

// TF attention
struct RowCanArgs {
    const float* x = nullptr;      // (B,64,T,64)
    const float* wt = nullptr;     // (64, NOUT) transposed 1x1 weights of every ConvActNorm in the call
    const float* bias = nullptr;   // (NOUT)
    const float* slope = nullptr;  // (ngroups) PReLU slope of each ConvActNorm
    const float* gamma = nullptr;  // (NOUT, 64)
    const float* beta = nullptr;   // (NOUT, 64)
    int ngroups = 0;
    int group_start[13] = {0};     // channel range of each LayerNorm group
    unsigned char group_of[96] = {0};
    int T = 0;
    int rpw = 0;  // frames per workgroup (set by the launcher: 8, fewer for small batches)
    float *q = nullptr, *k = nullptr, *v = nullptr;  // NOUT == 96
    const float* res = nullptr;                       // NOUT == 64: residual
    float* out = nullptr;
};
struct AttnArgs {
    const float *q = nullptr, *k = nullptr, *v = nullptr;
    float* out = nullptr;  // (B,64,T,64), channel = head*16 + c
    int T = 0;
    float scale = 0.f;
    int npairs = 0;  // B * heads, set by the launcher
};
int launch_row_can_qkv(const RowCanArgs& a, int B, hipStream_t st);
int launch_row_can_proj(const RowCanArgs& a, int B, hipStream_t st);
int launch_attn_core(const AttnArgs& a, int B, hipStream_t st);
size_t attn_core_lds_bytes(int T);

// CAF
struct CafArgs {
    const float* audio = nullptr;  // (B,256,T,F)
    const float* video = nullptr;  // (B,512,Tv)
    float* out = nullptr;          // (B,256,T,F)
    float* r_out = nullptr;        // (B,256,Tv) workspace: resize(video)
    float* att_out = nullptr;      // (B,256,Tv) workspace: softmax attention
    float *r_t = nullptr, *att_t = nullptr;  // optional (B,Tv,256) transposed copies of the two (fused separator)
    int T = 0, F = 0, Tv = 0;
    const float *w_key = nullptr, *bn_key = nullptr;  // (256), (4,256) = [weight | bias | running_mean | running_var]
    const float *w_val = nullptr, *bn_val = nullptr;
    const float *w_att = nullptr, *b_att = nullptr, *g_att = nullptr, *be_att = nullptr;          // (1024,2),(1024)x3
    const float *w_resize = nullptr, *b_resize = nullptr, *g_resize = nullptr, *be_resize = nullptr;  // (256,2),(256)x3
};
int launch_caf_video(const CafArgs& a, int B, hipStream_t st);
int launch_caf_apply(const CafArgs& a, int B, hipStream_t st);
int dualpath_timing_enable(int on);
int dualpath_timing_collect(float* ms, int* ls, int* nseq, int cap);

// Fused dual-path SRU sweep, f16x3 generation (k_dualpath16.hip).  Same sequence addressing as DpArgs.
struct Dp16Args {
    const float* x = nullptr;
    float* out = nullptr;
    int nseq = 0, R = 0, Ls = 0;
    size_t bstride = 0, rstride = 0, cstride = 0;
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    const half8* w16_l0 = nullptr;  // [16 chunks][hi|lo][256 cols = dir*128 + gate*32 + j][32 k'], k' = kk*64 + c
    const half8* w16_l = nullptr;   // 3 x [2 chunks][hi|lo][256][32]; gate 3 = identity block (highway input)
    const half8* w16_ct = nullptr;  // [8 chunks][hi|lo][64 co][64 k'], k' = kk*64 + ci
    // the same three images in MFMA fragment order (generation 3, k_dualpath16s.hip): [K step 16][dir][gate tile m][hi|lo][lane] x half8 with
    // lane (h, r) = W[k = 16 step + 8 h + j][col = dir*128 + m*32 + r]; conv-transpose: [tap][co tile][ks][hi|lo][lane], W[co = 32 tile + r][k = 16 ks + 8 h + j]
    const half8 *wf_l0 = nullptr, *wf_l = nullptr, *wf_ct = nullptr;
    const float* wc16 = nullptr;    // 4 x (128): v_f, v_r scaled by -log2(e)
    const float* bias16 = nullptr;  // 4 x (128): b_f, b_r scaled by -log2(e)
    const float* bt = nullptr;      // (64)
    unsigned long long* stamps = nullptr;  // diagnostic build only: [workgroups][16] s_memtime stamps
};
size_t dp16_lds_bytes(int Ls, int nseq_per_wg);
int launch_dualpath16(const Dp16Args& a, hipStream_t st);
// generation 3 (k_dualpath16s.hip): 256-thread workgroups, two per CU, L <= 128; launch_dualpath16 routes to it
size_t dp16s_lds_bytes(int Ls, int nseq_per_wg);
int launch_dualpath16s(const Dp16Args& a, hipStream_t st);
void* dualpath_timing_begin(int Ls, int nseq, hipStream_t st);
void dualpath_timing_end(void* slot, hipStream_t st);

// VP block (video 1-D TDANetBlock)
size_t vp_lds_bytes(int Tv);
int launch_vp_block(const float* video, const float* pack, float* out, int B, int Tv, hipStream_t st);

// evaluation-side loss path (k_loss.hip): pairwise negative SNR / SI-SDR / SD-SDR + best permutation, n_src <= 4
int launch_pit_pairwise(const float* est, const float* tgt, int B, int n, int L, int kind, int zero_mean, int take_log, float* pw,
                        float* min_loss, int* perm, hipStream_t st);

// video front-end (k_video.hip): FRCNNVideoModel, ResNet-18 trunk, PReLU, eval
size_t video_pack_floats();
size_t video_workspace_bytes(int B, int T);
int video_frontend(const float* lips, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes, hipStream_t st);

// training-side GEMMs and SRU scans (k_train_gemm.hip, k_train_rnn.hip)
struct GemmArgs {
    const float *A = nullptr, *B = nullptr;
    float* C = nullptr;
    int lda = 0, ldb = 0, ldc = 0, M = 0, N = 0, K = 0;
    int ncb = 0, nrb = 0;  // NT: column / row blocks
    int kchunk = 0;        // TN: k range of one wave
    const float* bias = nullptr;  // NT modes 0/1: added per column
    size_t sA = 0, sB = 0, sC = 0;  // batch strides (blockIdx.y)
};
// C (M,N) [+]= A (M,K) . Bt (N,K)^T  (both K-contiguous; N % 64 == 0, K % 16 == 0)
// mode 0: C = ; 1: C += ; 2: C (rows x 64) [(row + colblk) * ldc + col % 64] += (atomics; the fold of unfold windows)
int launch_gemm_nt(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K, int mode,
                   hipStream_t st, const float* bias = nullptr, int batch = 1, size_t sA = 0, size_t sB = 0, size_t sC = 0);
// C (M,N) += A (K,M)^T . B (K,N)  (split-K with f32 atomics; M, N % 64 == 0)
int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, long K, hipStream_t st, int batch = 1,
                   size_t sA = 0, size_t sB = 0, size_t sC = 0);
struct SruScanArgs {
    const float* U = nullptr;    // (L, N, KC), column m*64 + dir*32 + j
    const float* xin = nullptr;  // (L, N, 64) highway input of layers 1-3; nullptr: U's m = 3 block (layer 0)
    const float* wc = nullptr;   // (128) v_f | v_r
    const float* bias = nullptr; // (128) b_f | b_r
    float* h = nullptr;          // forward: (L, N, 64)
    float* c = nullptr;          // forward: written; backward: read
    const float* g = nullptr;    // backward: dL/dh (L, N, 64)
    float* dU = nullptr;         // backward: (L, N, KC)
    float* dxp = nullptr;        // backward, layers 1-3: highway gradient (L, N, 64)
    float* dwc = nullptr;        // backward: (128) running sums
    float* dbias = nullptr;      // backward: (128)
    int L = 0, N = 0, KC = 0;
    long ts = 0, ns = 0;  // row of (step t, sequence n) = t*ts + n*ns
    int pad = 0;          // sequence-major dual-path layout: zero the 7 non-step rows of each slot
};
int launch_sru_scan_fwd(const SruScanArgs& a, hipStream_t st);
int launch_sru_scan_bwd(const SruScanArgs& a, hipStream_t st);
// dual-path training layout kernels: x (B, 64, R, Ls) <-> rows [n*Ls + s][64]
int launch_dp_ln_fwd(const float* x, const float* gamma, const float* beta, float* xn, int nseq, int R, int Ls, hipStream_t st);
int launch_dp_out(const float* y, const float* bias, const float* x, float* out, int nseq, int R, int Ls, hipStream_t st);
int launch_dp_dy(const float* dout, float* dy, float* dbias, int nseq, int R, int Ls, hipStream_t st);
int launch_dp_ln_bwd(const float* x, const float* dxn, const float* dout, const float* gamma, float* dx, float* dgamma, float* dbeta,
                     int nseq, int R, int Ls, hipStream_t st);
// channel-last ConvNormAct training kernels (k_train_conv.hip)
struct ClStageArgs {  // y = act(gLN(x)) over rows x C, per-sample statistics
    const float* x = nullptr;
    float* y = nullptr;
    const double* stats = nullptr;  // (B, 2) sum, sum of squares of x per sample (norm only)
    const float *gamma = nullptr, *beta = nullptr, *slope = nullptr;
    const float *rmean = nullptr, *rvar = nullptr;  // norm 2: BatchNorm running statistics (per channel)
    const double* cstats = nullptr;                 // norm 3: per-channel (sum, sum of squares) of this batch
    double inv_rows = 0;                            // norm 3: 1 / (rows over the whole batch)
    const float* dy = nullptr;  // backward
    float* dx = nullptr;
    double* S = nullptr;        // backward scratch (B, 2)
    float* partial = nullptr;   // backward: cl_stage_partial_floats(B, C) floats of per-workgroup sums (two-stage reduction)
    float *dgamma = nullptr, *dbeta = nullptr, *dslope = nullptr;
    size_t n = 0;               // elements per sample = rows per sample * C
    int C = 0, norm = 0, act = 0;
};
#define CL_DW_WGRAD_MAX_WG 2048
struct ClDwArgs {
    const float *x = nullptr, *w = nullptr, *bias = nullptr, *dy = nullptr;
    float *y = nullptr, *dx = nullptr, *dw = nullptr;
    float* scratch = nullptr;  // wgrad: CL_DW_WGRAD_MAX_WG x kh*kw x C per-workgroup partial sums
    int B = 0, H = 0, W = 0, C = 0, Ho = 0, Wo = 0, kh = 0, kw = 0, s = 1, pt = 0, pl = 0;
    int Cp = 0;  // row pitch in channels (set by the launcher; C is the slice a launch covers)
};
struct GatewayArgs {  // y = PReLU(w_c * (x + xr) + b_c) over n4 float4 of rows x C (C fastest)
    const float *x = nullptr, *xr = nullptr, *w = nullptr, *b = nullptr, *slope = nullptr, *dy = nullptr;
    float *y = nullptr, *dx = nullptr, *partial = nullptr;
    size_t n4 = 0;
    int C = 0;
};
int launch_gateway(const GatewayArgs& a, bool bwd, float* dw, float* db, float* dslope, hipStream_t st);
size_t cl_stage_partial_floats(int B, int C);
int launch_cl_norm_act_fwd(const ClStageArgs& a, int B, hipStream_t st);
int launch_cl_norm_act_bwd(const ClStageArgs& a, int B, hipStream_t st, int part = 0);
int launch_cl_colsum(const float* d, float* out, size_t n, int C, hipStream_t st, float* partial = nullptr);
int launch_cl_chan_stats(const float* x, double* stats, size_t n, int C, hipStream_t st);
int launch_bn_update(const double* stats, float* rmean, float* rvar, int C, double rows, float momentum, hipStream_t st);
int launch_cl_dw(const ClDwArgs& a, int what /* 0 fwd, 1 bwd data, 2 wgrad */, hipStream_t st);
// TF attention training kernels (k_train_attn.hip)
struct LngArgs {  // PReLU + LayerNormalization4D((C_group, 64)) over rows (b,t,f) x CZ
    const float* Z = nullptr;      // pre-activation rows
    float* Y = nullptr;            // forward output rows
    const float* res = nullptr;    // forward: optional residual rows added to Y (same shape)
    float* stats = nullptr;        // (B*T, 16, 2) mean, rstd per group (written forward, read backward)
    const float *slope = nullptr, *gamma = nullptr, *beta = nullptr;  // (CZ), (CZ,64), (CZ,64)
    const float* dY = nullptr;     // backward
    float* dZ = nullptr;
    float *dgamma = nullptr, *dbeta = nullptr, *dslope = nullptr;      // (CZ,64), (CZ,64), (ngroups)
    float* scratch = nullptr;      // backward: (workgroups, 2, CZ*64) partial sums of dgamma | dbeta
    int CZ = 0, ngroups = 0, nbt = 0;
    int gstart[17] = {0};
    unsigned char gof[128] = {0};  // channel -> group, 255 = padding channel
};
int launch_att_lng(const LngArgs& a, int nbt, bool bwd, hipStream_t st);
int launch_att_pack_qkv(float* rows, float* Qp, float* Kp, float* Vp, int B, int T, int Tp, int dir, hipStream_t st);
int launch_att_pack_o(float* rows, float* Op, int B, int T, int Tp, int dir, hipStream_t st);
int launch_att_softmax(float* S, const float* P, int nbatch, int T, int Tp, float scale, bool bwd, hipStream_t st);
int launch_pool2d(const float* x, float* y, size_t N, int H, int W, int Ho, int Wo, bool bwd, hipStream_t st, int C = 1);
int launch_tfar_combine(const float* le, const float* gate, const float* ge, float* out, size_t N, int H, int W, int Hg, int Wg, hipStream_t st,
                        int C = 1);
int launch_tfar_combine_bwd(const float* dout, const float* le, const float* gate, float* dle, float* dgate, float* dge, size_t N, int H, int W,
                            int Hg, int Wg, hipStream_t st, int C = 1);
int launch_patch3x3_rows(const float* z, float* rows, int B, int T, int F, hipStream_t st);
int launch_istft_adjoint(const float* dwav, float* dspec, int B, int T, int L, hipStream_t st);
int launch_cmul(const float* a, const float* b, float* out, int B, size_t half, int conj_a, hipStream_t st);
int launch_caf_att(const float* in, float* out, const float* dout, float* din, int nbc, int Tv, bool bwd, hipStream_t st);
int launch_caf_combine(const float* key, const float* value, const float* r, const float* att, float* out, size_t N, int T, int F, int Tv,
                       hipStream_t st);
int launch_caf_combine_rows(const float* key, const float* value, const float* r, const float* att, float* out, int B, int T, int F, int C, int Tv,
                            hipStream_t st);
int launch_caf_combine_rows_bwd(const float* dout, const float* key, const float* value, const float* r, const float* att, float* dkey,
                                float* dvalue, float* dr, float* datt, int B, int T, int F, int C, int Tv, hipStream_t st);
int launch_caf_combine_bwd(const float* dout, const float* key, const float* value, const float* r, const float* att, float* dkey,
                           float* dvalue, float* dr, float* datt, size_t N, int T, int F, int Tv, hipStream_t st);
int launch_pit_sdr_bwd(const float* est, const float* tgt, const int* perm, const float* dmin, float* dest, int B, int n, int L, int kind,
                       int zero_mean, int take_log, hipStream_t st);
int launch_ln_rows(const float* x, const float* gamma, const float* beta, float* y, const float* dy, float* dx, float* dgamma, float* dbeta,
                   size_t N, int C, bool bwd, hipStream_t st, const float* res = nullptr);
int launch_mha_core(const float* qkv, const float* pmask, float* o, const float* dout, float* dqkv, int B, int T, int nh, int hd, bool bwd,
                    hipStream_t st);
struct LstmScanArgs {
    const float* U = nullptr;     // forward: (rows, 256) pre-activations incl. both biases, column dir*128 + gate*32 + j
    const float* whh = nullptr;   // (2, 128, 32)
    float *G = nullptr, *c = nullptr, *h = nullptr, *hprev = nullptr;  // forward: written; backward: G, c read
    const float* g = nullptr;     // backward: dL/dh (rows, 64)
    float* dU = nullptr;          // backward: (rows, 256)
    int L = 0, N = 0, pad = 0;
    long ts = 0, ns = 0;
};
int launch_lstm_scan(const LstmScanArgs& a, bool bwd, hipStream_t st);
int launch_rows_bias_res(const float* y, const float* bias, const float* x, float* out, size_t n, int C, hipStream_t st);
int launch_rows_permute(const float* x, float* y, int B, int H, int W, int C, hipStream_t st);
size_t att_lng_scratch_floats(int nbt);
struct GruScanArgs {
    const float* U = nullptr;     // forward: (rows, 192) = W_ih x + b_ih, column dir*96 + gate*32 + j (gates r, z, n)
    const float* whh = nullptr;   // (2, 96, 32)
    const float* bhh = nullptr;   // (2, 96)
    float *S = nullptr, *h = nullptr, *hprev = nullptr;  // forward: written; backward: S, hprev read
    const float* g = nullptr;     // backward: dL/dh (rows, 64)
    float *dU = nullptr, *dHR = nullptr;  // backward: (rows, 192) each
    int L = 0, N = 0, pad = 0;
    long ts = 0, ns = 0;
};
int launch_gru_scan(const GruScanArgs& a, bool bwd, hipStream_t st);
