/* rtfs_amd.h -- C ABI of the MI355X-native RTFS-Net separator forward pass (librtfs_amd.so).
 *
 * The reference (SutirthaChakraborty/RTFS-Net) has no FFI of its own: its hot path is the Python
 * nn.Module tree under AVNet.forward (src/models/tdavnet.py:86-97) plus ONE third-party native
 * operator, sru.SRU (src/models/layers/rnn_layers.py:99-105,150).  Each entry point below is what a
 * binding for one of those modules / that operator calls; the comment on each names the reference
 * interface it replaces.  INTEGRATION.md shows the ctypes stubs a maintainer would add.
 *
 * Conventions (all entry points):
 *   - raw DEVICE pointers, float32, contiguous row-major, reference layouts:
 *       spectrogram-shaped tensors (B, C, T, F) with F fastest; waveforms (B, L); lip embedding (B, 512, Tv)
 *   - `pack` = that module's parameters as ONE contiguous device buffer laid out as documented in
 *     rtfs-net_amd/packing.py (every tensor padded to a multiple of 64 floats, 1x1 weights stored
 *     transposed [cin][cout]); rtfs_pack_floats(kind) returns the expected length
 *   - caller allocates outputs and the workspace (size from the matching *_workspace_bytes query);
 *     kernels are enqueued on `stream` (a hipStream_t passed as void*), never synchronise, never
 *     allocate, never call back; workspace contents are scratch
 *   - return 0 on success; <0 on error: -1 bad shape, -2 workspace too small, -3 launch failure, -4 bad argument
 *   - eval-mode semantics (BatchNorm running statistics, no dropout); re-entrant: the only host-side state is a mutex-guarded cache
 *     of per-(device, kernel) launch attributes, the per-(device, caller stream) internal side streams of
 *     rtfs_separator_forward_f32 (forked from `stream` by an event and joined back into it inside the call: from outside all work of a call
 *     is ordered on `stream`), the process-wide DEFAULT of the batch split (rtfs_set_batch_split; the _ex entry points take it per call) and
 *     the diagnostic sweep-timing log (off by default), so the
 *     library may be driven from several host threads / devices in one process (one thread per stream).
 *   - length limits of the FUSED entry points (they keep a whole sweep / score row / video pyramid on chip and return -1 beyond):
 *       sweep axis of rtfs_dualpath_* / rtfs_block_f32 / rtfs_separator_forward_f32   <= 250 positions (T/2 <= 250: 4 s of audio)
 *       keys of rtfs_tf_attention_f32                                                 <= 256
 *       video frames of rtfs_vp_block_f32                                             <= 120
 *     The reference has no length limit (rnn_layers.py:136-162, attention.py:149-189; infer_any_video.py:86 feeds whole files): longer
 *     inputs go through the UNFUSED entry points below (rtfs_*_forward_train_f32 and friends: GEMM + scan + GEMM sweeps, batched-GEMM
 *     attention, per-layer video block), which take any length; rtfs-net_amd/{models,layers}.py route by length (FUSED_MAX_*).
 * F must satisfy F/2 == 64 wherever a block / attention is involved (the reference's n_freqs: 64 ties
 * LayerNormalization4D's parameters to 64 compressed frequency bins, config/lrs2_RTFSNet_4_layer.yaml:68).
 */
#ifndef RTFS_AMD_H
#define RTFS_AMD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum rtfs_pack_kind {
    RTFS_PACK_ENCODER = 0,
    RTFS_PACK_AUDIO_BN = 1,
    RTFS_PACK_BLOCK = 2,
    RTFS_PACK_DUALPATH = 3,
    RTFS_PACK_ATTENTION = 4,
    RTFS_PACK_TFAR = 5,
    RTFS_PACK_CAF = 6,
    RTFS_PACK_S3 = 7,
    RTFS_PACK_DECODER = 8,
    RTFS_PACK_BLOCK_LSTM = 9,    /* RTFS block whose two DualPathRNN layers use rnn_type LSTM */
    RTFS_PACK_DUALPATH_LSTM = 10
};

/* library / build identification: returns "rtfs_amd <version> gfx950" */
const char* rtfs_version(void);
/* number of floats in a parameter pack of the given kind */
size_t rtfs_pack_floats(int kind);
/* frames of the STFT for a waveform of L samples: 1 + L/128 */
int rtfs_num_frames(int L);

/* STFTEncoder.forward (src/models/TDAVNet/encoder.py:161-175): wav (B,L) -> a0 (B,256,T,129).
 * workspace holds the (B,2,T,129) spectrogram.  If stats != NULL it receives (B,2) doubles
 * = (sum, sum of squares) of a0 per sample (consumed by the audio bottleneck's gLN). */
size_t rtfs_stft_encoder_workspace_bytes(int B, int L);
int rtfs_stft_encoder_f32(const float* wav, const float* pack, float* a0, double* stats, int B, int L, void* ws,
                          size_t ws_bytes, void* stream);

/* audio_bottleneck = ConvNormAct(gLN -> ReLU -> Conv2d 1x1 256->256) (tdavnet.py:59,89;
 * layers/conv_layers.py:65-129).  stats: (B,2) doubles of x as produced by the encoder, or NULL to
 * have them computed here (needs the workspace). */
size_t rtfs_audio_bottleneck_workspace_bytes(int B);
int rtfs_audio_bottleneck_f32(const float* x, const double* stats, const float* pack, float* out, int B, int T, int F,
                              void* ws, size_t ws_bytes, void* stream);

/* RTFS block = TDANetBlock.forward, is2d, upsampling_depth 2, globalatt = [DualPathRNN(F), DualPathRNN(T),
 * MultiHeadSelfAttention2D] (src/models/separators/tdanet.py:104-131).  x_res may be NULL; otherwise the block
 * input is x + x_res (RefinementModule's residual, refinement_module.py:51,60). */
size_t rtfs_block_workspace_bytes(int B, int T, int F);
int rtfs_block_f32(const float* x, const float* x_res, const float* pack, float* out, int B, int T, int F, void* ws,
                   size_t ws_bytes, void* stream, int rnn_kind /* 0 = SRU pack, 1 = LSTM pack */);

/* DualPathRNN.forward with rnn_type SRU, kernel 8, stride 1, 4 layers, bidirectional, hidden 32
 * (src/models/layers/rnn_layers.py:136-162).  x, out (B,64,T,F); dim = 4 sweeps along F, 3 along T. */
size_t rtfs_dualpath_workspace_bytes(int B, int T, int F);
int rtfs_dualpath_sru_f32(const float* x, const float* pack, float* out, int B, int T, int F, int dim, void* ws,
                          size_t ws_bytes, void* stream);

/* The same module with rnn_type LSTM = nn.LSTM(512, 32, num_layers 4, bidirectional) (rnn_layers.py:116-122), the
 * reference's stock-torch alternative cell; exact-f32 MFMA GEMMs + per-step recurrent term.  pack kind DUALPATH_LSTM. */
int rtfs_dualpath_lstm_f32(const float* x, const float* pack, float* out, int B, int T, int F, int dim, void* ws,
                           size_t ws_bytes, void* stream);

/* MultiHeadSelfAttention2D.forward, 4 heads, hid_chan 4, dim 3 (src/models/layers/attention.py:149-189). x (B,64,T,64). */
size_t rtfs_tf_attention_workspace_bytes(int B, int T);
int rtfs_tf_attention_f32(const float* x, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes,
                          void* stream);

/* InjectionMultiSum.forward (TFAR, src/models/layers/fusion.py:54-69): local (B,64,H,W), global (B,64,Hg,Wg). */
size_t rtfs_tfar_workspace_bytes(int B, int H, int W, int Hg, int Wg);
int rtfs_tfar_f32(const float* local, const float* global, const float* pack, float* out, int B, int H, int W, int Hg,
                  int Wg, void* ws, size_t ws_bytes, void* stream);

/* CAF = ATTNFusion.forward with video_fusion False -> ATTNFusionCell (src/models/TDAVNet/fusion.py:204-212,
 * src/models/layers/fusion.py:252-274): audio (B,256,T,F), video (B,512,Tv) -> fused audio. */
size_t rtfs_caf_workspace_bytes(int B, int Tv);
int rtfs_caf_f32(const float* audio, const float* video, const float* pack, float* out, int B, int T, int F, int Tv,
                 void* ws, size_t ws_bytes, void* stream);

/* VP block = the video-side 1-D TDANetBlock.forward (upsampling_depth 4, kernel 3, BatchNorm1d, GlobalAttention;
 * src/models/separators/tdanet.py:104-131 with yaml video_params): video (B,512,Tv) -> (B,512,Tv), Tv <= 120 (see the conventions).
 * pack = rtfs-net_amd/packing.py:pack_vp (eval BatchNorm folded); rtfs_vp_pack_floats() returns its length. */
size_t rtfs_vp_pack_floats(void);
int rtfs_vp_block_f32(const float* video, const float* pack, float* out, int B, int Tv, void* stream);

/* S^3 = MaskGenerator.forward with RI_split, n_src 1 (src/models/TDAVNet/mask_generator.py:67-99):
 * refined, a0 (B,256,T,F) -> separated embedding (B,1,256,T,F). */
int rtfs_s3_mask_f32(const float* refined, const float* a0, const float* pack, float* out, int B, int T, int F,
                     void* stream);

/* STFTDecoder.forward (src/models/TDAVNet/decoder.py:110-132): x (B,1,256,T,129) -> wav (B,1,L). */
size_t rtfs_istft_decoder_workspace_bytes(int B, int T);
int rtfs_istft_decoder_f32(const float* x, const float* pack, float* wav, int B, int T, int L, void* ws, size_t ws_bytes,
                           void* stream);

/* The whole separator, AVNet.forward minus the (tiny, 0.004 GMAC) video-side VP block whose output the caller
 * passes in (src/models/tdavnet.py:86-97, refinement_module.py:45-62):
 *   encoder -> audio bottleneck -> block -> CAF(video_vp) -> (repeats-1) x block(+a1) -> S^3 -> decoder.
 * packs: encoder, audio_bn, block, caf, s3, decoder.  wav (B,L), video_vp (B,512,Tv) -> out (B,1,L).
 * video_ready: optional hipEvent_t (as void*, may be NULL) recorded by the caller after video_vp was produced on
 * ANOTHER stream; the library makes `stream` wait for it right before the CAF block, so the VP block overlaps the
 * encoder and the first RTFS block. */
size_t rtfs_separator_workspace_bytes(int B, int L, int Tv);
/* Throughput option of rtfs_separator_forward_f32 (process-wide, default 1 = off; 0 restores the default / RTFS_SPLIT): the batch is cut
 * into n parts (each >= 8 mixtures) that run as independent chains on internal side streams forked from and joined back into `stream`, so the
 * HBM-bound kernels of one part run beside the latency-bound sweeps of another (batch 32: 14.2 -> 13.0 ms with n = 2).  Results per mixture
 * do not depend on it.  Call it before rtfs_separator_workspace_bytes: the workspace layout follows the setting. */
int rtfs_set_batch_split(int n);
/* The same option PER CALL (re-entrant: two host threads can pick different schedules): split = 1 .. 8 parts for this call, 0 = the process
 * default above.  The workspace query and the forward call must be given the same value. */
size_t rtfs_separator_workspace_bytes_ex(int B, int L, int Tv, int split);
int rtfs_separator_forward_ex_f32(const float* wav, const float* video_vp, const float* pack_enc, const float* pack_bn,
                                  const float* pack_block, const float* pack_caf, const float* pack_s3,
                                  const float* pack_dec, float* out, int B, int L, int Tv, int repeats, void* ws,
                                  size_t ws_bytes, void* stream, void* video_ready, int rnn_kind, int split);
int rtfs_separator_forward_f32(const float* wav, const float* video_vp, const float* pack_enc, const float* pack_bn,
                               const float* pack_block, const float* pack_caf, const float* pack_s3,
                               const float* pack_dec, float* out, int B, int L, int Tv, int repeats, void* ws,
                               size_t ws_bytes, void* stream, void* video_ready, int rnn_kind /* 0 SRU, 1 LSTM block pack */);

/* Operator-level seam: sru.SRU(input_size=512, hidden_size=32, num_layers=4, bidirectional=True).forward
 * (call site src/models/layers/rnn_layers.py:150; third-party asappresearch `sru`, v2 recurrence).
 * x (L,N,512) -> h (L,N,64).  pack = the DUALPATH pack (only its SRU part is read). */
size_t rtfs_sru_workspace_bytes(int L, int N);
int rtfs_sru_f32(const float* x, const float* pack, float* h, int L, int N, void* ws, size_t ws_bytes, void* stream);

/* Training side of the same operator (upstream sru's forward/backward pair behind
 * rnn_layers.py:150 when the module is used from train.py).  The forward keeps U = x.W, the cell states and the
 * inter-layer activations of all four layers in `saved` (rtfs_sru_saved_floats(L,N) floats, caller-owned) for the backward.
 * tpack (rtfs_sru_train_pack_floats() floats, rtfs-net_amd/packing.py:pack_sru_train):
 *   Wt0 (256,512) | Wt1..3 (192,64) | Wp0 (512,256) | Wp1..3 (64,192) | weight_c (4,128) | bias (4,128)
 *   with Wt = Wp^T and projection columns re-ordered to m*64 + dir*32 + j.
 * backward: dh (L,N,64) -> dx (L,N,512) and dparams (rtfs_sru_grad_floats() floats, overwritten):
 *   dWp0 (512,256) | dWp1..3 (64,192) | d weight_c (4,128) | d bias (4,128). */
size_t rtfs_sru_train_pack_floats(void);
size_t rtfs_sru_grad_floats(void);
size_t rtfs_sru_saved_floats(int L, int N);
size_t rtfs_sru_backward_workspace_bytes(int L, int N);
int rtfs_sru_forward_train_f32(const float* x, const float* tpack, float* h, float* saved, int L, int N, void* stream);
int rtfs_sru_backward_f32(const float* x, const float* tpack, const float* saved, const float* dh, float* dx, float* dparams,
                          int L, int N, void* ws, size_t ws_bytes, void* stream);
/* DualPathRNN.forward / backward for training (src/models/layers/rnn_layers.py:136-162 with rnn_type SRU).
 * x, out, dout, dx (B,64,T,F); dim as in the reference (4: sweep along F, 3: along T); dim 14 / 13: the same sweeps with x, out, dout, dx
 * as rows (B,T,F,64) (the layout the training kernels of a block hand each other; size queries take the plain 4 / 3).  `saved` (rtfs_dualpath_saved_floats)
 * is written by the forward and read by the backward; the same workspace size serves both.
 * tpack (rtfs_dualpath_train_pack_floats(), packing.py:pack_dualpath_train):
 *   LN gamma (64) | LN beta (64) | SRU training pack with layer-0 rows in k*64 + c order | ConvTranspose1d weight as
 *   (co, (7-k)*64 + ci) | as (ci, k*64 + co) | bias (64).
 * dparams (rtfs_dualpath_grad_floats(), overwritten): dgamma | dbeta | SRU gradients (rtfs_sru_backward_f32 layout, layer-0
 *   rows k*64 + c) | d ConvTranspose1d weight as ((7-k)*64 + ci, co) | d bias. */
size_t rtfs_dualpath_train_pack_floats(void);
size_t rtfs_dualpath_grad_floats(void);
size_t rtfs_dualpath_saved_floats(int B, int T, int F, int dim);
size_t rtfs_dualpath_train_workspace_bytes(int B, int T, int F, int dim);
int rtfs_dualpath_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim,
                                    void* ws, size_t ws_bytes, void* stream);
int rtfs_dualpath_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx,
                               float* dparams, int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream);
/* The same module with rnn_type LSTM (rnn_layers.py:116-122: nn.LSTM(512, 32, 4 layers, bidirectional)) in training.
 * tpack (rtfs_dualpath_lstm_train_pack_floats(), packing.py:pack_dualpath_lstm_train): LN gamma | beta | per layer [W_ih both directions
 *   (256, Din), rows dir*128 + gate*32 + j, layer-0 columns in k*64 + c order | its transpose | b_ih + b_hh (256) | W_hh (2,128,32)] |
 *   ConvTranspose1d weight as (co, (7-k)*64 + ci) | as (ci, k*64 + co) | bias.
 * dparams (rtfs_dualpath_lstm_grad_floats(), overwritten): dgamma | dbeta | per layer [dW_ih (256, Din) | d bias (256; the gradient of
 *   b_ih and of b_hh alike) | dW_hh (2,128,32)] | d ConvTranspose1d weight ((7-k)*64 + ci, co) | d bias. */
size_t rtfs_dualpath_lstm_train_pack_floats(void);
size_t rtfs_dualpath_lstm_grad_floats(void);
size_t rtfs_dualpath_lstm_saved_floats(int B, int T, int F, int dim);
size_t rtfs_dualpath_lstm_train_workspace_bytes(int B, int T, int F, int dim);
int rtfs_dualpath_lstm_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim,
                                         void* ws, size_t ws_bytes, void* stream);
int rtfs_dualpath_lstm_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx,
                                    float* dparams, int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream);
/* The same module with rnn_type GRU (rnn_layers.py:116-122: nn.GRU(512, 32, 4 layers, bidirectional); gates r, z, n).  There is no fused
 * inference kernel for this cell (no reference yaml uses it in a DualPathRNN): the forward below also serves inference.
 * tpack (rtfs_dualpath_gru_train_pack_floats(), packing.py:pack_dualpath_gru_train): LN gamma | beta | per layer [W_ih both directions
 *   (192, Din), rows dir*96 + gate*32 + j, layer-0 columns in k*64 + c order | its transpose | b_ih (192) | W_hh (2,96,32) | b_hh (192)] |
 *   ConvTranspose1d weight as (co, (7-k)*64 + ci) | as (ci, k*64 + co) | bias.
 * dparams (rtfs_dualpath_gru_grad_floats(), overwritten): dgamma | dbeta | per layer [dW_ih | db_ih | dW_hh | db_hh] | d ConvTranspose1d
 *   weight ((7-k)*64 + ci, co) | d bias. */
size_t rtfs_dualpath_gru_train_pack_floats(void);
size_t rtfs_dualpath_gru_grad_floats(void);
size_t rtfs_dualpath_gru_saved_floats(int B, int T, int F, int dim);
size_t rtfs_dualpath_gru_train_workspace_bytes(int B, int T, int F, int dim);
int rtfs_dualpath_gru_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int F, int dim,
                                        void* ws, size_t ws_bytes, void* stream);
int rtfs_dualpath_gru_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams,
                                   int B, int T, int F, int dim, void* ws, size_t ws_bytes, void* stream);
/* ConvNormAct.forward / backward for training (src/models/layers/conv_layers.py:65-129: pre_norm -> pre_act -> conv -> norm -> act),
 * 1x1 dense (channels up to 1024) or depthwise k x k (taps up to 4 x 5, stride 1 "same" or stride 2 symmetric), norms: none | gLN |
 * BatchNorm (post-norm only; frozen running statistics, or train mode = statistics of the batch), acts: none | ReLU | PReLU | Sigmoid.
 * cfg (HOST int[15]): Cin, Cout, k, stride, depthwise, pre_norm (0/1), pre_act (0 none, 1 ReLU, 2 PReLU, 3 Sigmoid), norm (0 none, 1 gLN,
 *   2 frozen BatchNorm, 3 train-mode BatchNorm), act, has_bias, is2d, phase, world, in_rows, out_rows.
 *   in_rows / out_rows: the input / output (and their gradients) are (B, H, W, C) rows instead of (B, C, H, W): modules chained inside a
 *   training step hand rows to each other and skip the layout changes; a rows input is not copied into `saved`, the backward takes it
 *   again as `x` (NULL otherwise).
 *   phase / world serve SyncBatchNorm (norm 3 only): phase 1 runs the forward up to the batch statistics (rtfs_cna_saved_stats_offset:
 *   2*Cout doubles inside `saved`, which the caller all-reduces), phase 2 resumes with the normalisation over rows * world samples; the
 *   backward likewise stops after the dgamma / dbeta sums (rtfs_cna_grad_norm_offsets) and resumes with the input gradient; phase 0 =
 *   everything in one call, world = 1.  x (B,Cin,H,W) -> out (B,Cout,Ho,Wo) (rtfs_cna_out_shape).
 * params (rtfs_cna_param_floats, packing.py:pack_cna_train; every slot padded to 64 floats, unused slots ignored):
 *   pre gamma | pre beta | pre slope | W (Cout,Cin) or (C,kh*kw) | W^T (dense only) | bias | gamma | beta | slope | running mean | running var.
 * dparams (rtfs_cna_grad_floats, overwritten): the slots pre gamma ... slope without W^T. */
size_t rtfs_cna_param_floats(const int* cfg);
size_t rtfs_cna_grad_floats(const int* cfg);
size_t rtfs_cna_saved_floats(const int* cfg, int B, int H, int W);
size_t rtfs_cna_workspace_bytes(const int* cfg, int B, int H, int W);
void rtfs_cna_out_shape(const int* cfg, int H, int W, int* Ho, int* Wo);
size_t rtfs_cna_saved_stats_offset(const int* cfg, int B, int H, int W);
void rtfs_cna_grad_norm_offsets(const int* cfg, size_t* dgamma, size_t* dbeta);
int rtfs_cna_forward_train_f32(const float* x, const float* params, float* out, float* saved, const int* cfg, int B, int H, int W,
                               void* ws, size_t ws_bytes, void* stream);
int rtfs_cna_backward_f32(const float* x, const float* params, const float* saved, const float* dout, float* dx, float* dparams,
                          const int* cfg, int B, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* after a forward with norm = 3: nn.BatchNorm's running_mean / running_var update (momentum, unbiased variance) from the batch
 * statistics kept in `saved`; the two pointers are the module's buffers on the device. */
int rtfs_cna_bn_update_f32(const float* saved, const int* cfg, int B, int H, int W, float* running_mean, float* running_var,
                           float momentum, void* stream);
/* MultiHeadSelfAttention2D.forward / backward for training (src/models/layers/attention.py:149-189; 4 heads, hid_chan 4, n_freqs 64).
 * x, out, dout, dx (B,64,T,64).  tpack (rtfs_tf_attention_train_pack_floats(), packing.py:pack_attention_train):
 *   W_qkv (128,64) rows [Q h0..3 (4 each) | K h0..3 | V h0..3 (16 each) | 32 zero rows] | its transpose | bias (128) | PReLU slope per
 *   row (128) | LN gamma (128,64) | LN beta (128,64) | W_proj (64,64) | its transpose | bias (64) | slope per row (64) | gamma (64,64) | beta.
 * dparams (rtfs_tf_attention_grad_floats(), overwritten): dW_qkv | dbias (128) | dslope per module (64 slots, 12 used: Q h0..3, K h0..3,
 *   V h0..3) | dgamma (128,64) | dbeta | dW_proj | dbias (64) | dslope (64 slots, 1 used) | dgamma (64,64) | dbeta. */
size_t rtfs_tf_attention_train_pack_floats(void);
size_t rtfs_tf_attention_grad_floats(void);
size_t rtfs_tf_attention_saved_floats(int B, int T);
size_t rtfs_tf_attention_train_workspace_bytes(int B, int T);
/* rows != 0: x, out, dout, dx are rows (B, T, 64 f, 64 c) instead of (B, 64, T, 64); a rows input is read in place and handed to the
 * backward again as `x` (NULL otherwise). */
int rtfs_tf_attention_forward_train_f32(const float* x, const float* tpack, float* out, float* saved, int B, int T, int rows, void* ws,
                                        size_t ws_bytes, void* stream);
int rtfs_tf_attention_backward_f32(const float* x, const float* tpack, const float* saved, const float* dout, float* dx, float* dparams,
                                   int B, int T, int rows, void* ws, size_t ws_bytes, void* stream);
/* Layout change between the reference's (B, C, P) and the rows (B, P, C) the training kernels hand each other (to_rows != 0: the
 * former to the latter); each direction is the other's adjoint. */
int rtfs_layout_f32(const float* x, float* y, int B, int C, int P, int to_rows, void* stream);

/* Glue of the RTFS block with its adjoints (training side).  inner = 1: channel-first planes (N = B*C, H, W); inner = C > 1: rows
 * (N = B, H, W, C), channels fastest (what the training kernels hand each other inside a block).
 * F.adaptive_avg_pool2d as called at separators/tdanet.py:116 and its adjoint;
 * the last line of InjectionMultiSum.forward (layers/fusion.py:54-69): out = local * up(gate) + up(glob), up = nearest
 * interpolation (Hg, Wg) -> (H, W) (identity when equal), and its adjoint (dlocal like local; dgate, dglob like gate). */
int rtfs_adaptive_avg_pool2d_f32(const float* x, float* y, int N, int H, int W, int Ho, int Wo, int inner, void* stream);
int rtfs_adaptive_avg_pool2d_backward_f32(const float* dy, float* dx, int N, int H, int W, int Ho, int Wo, int inner, void* stream);
int rtfs_tfar_combine_f32(const float* local, const float* gate, const float* glob, float* out, int N, int H, int W, int Hg, int Wg,
                          int inner, void* stream);
int rtfs_tfar_combine_backward_f32(const float* dout, const float* local, const float* gate, float* dlocal, float* dgate, float* dglob,
                                   int N, int H, int W, int Hg, int Wg, int inner, void* stream);
/* Training side of STFTEncoder / STFTDecoder / the S^3 multiply.
 * Encoder (encoder.py:161-175): the waveform is data, so only the Conv2d(2->256, 3x3) weight has a gradient: dw (256,2,3,3) from
 *   wav (B,L) and da0 (B,256,T,129).
 * Decoder (decoder.py:110-132): dwav (B,L) -> dx (B,256,T,129) and dw (256,2,3,3) (ConvTranspose2d weight as stored): the adjoint of
 *   torch.istft (window, overlap-add envelope, crop) followed by the adjoint of the transposed convolution.
 * S^3 (mask_generator.py:71-82): complex multiply of [re 128 | im 128]-split maps (B,256,P); conj_first selects conj(a) (x) b, which is
 *   the adjoint with respect to either factor. */
size_t rtfs_stft_encoder_backward_workspace_bytes(int B, int L);
int rtfs_stft_encoder_backward_f32(const float* wav, const float* da0, float* dw, int B, int L, void* ws, size_t ws_bytes, void* stream);
size_t rtfs_istft_decoder_backward_workspace_bytes(int B, int T);
int rtfs_istft_decoder_backward_f32(const float* x, const float* w, const float* dwav, float* dx, float* dw, int B, int T, int L, void* ws,
                                    size_t ws_bytes, void* stream);
int rtfs_s3_cmul_f32(const float* a, const float* b, float* out, int B, int P, int conj_first, void* stream);

/* Glue of ATTNFusionCell.forward (layers/fusion.py:252-274) with its adjoints (training side).
 * attention: att_embed (B, 4C, Tv) -> reshape (B, C, 4, Tv) -> mean over the 4 -> softmax over Tv -> att (B, C, Tv).
 * combine: fused = key * up(resized) + up(att) * value with key/value/fused (N = B*C, T, F), resized/att (N, Tv), up = nearest
 * interpolation over time broadcast over F; the backward returns all four gradients. */
int rtfs_caf_attention_f32(const float* att_embed, float* att, int B, int C, int Tv, void* stream);
int rtfs_caf_attention_backward_f32(const float* att, const float* datt, float* datt_embed, int B, int C, int Tv, void* stream);
int rtfs_caf_combine_f32(const float* key, const float* value, const float* resized, const float* att, float* out, int N, int T, int F,
                         int Tv, void* stream);
int rtfs_caf_combine_backward_f32(const float* dout, const float* key, const float* value, const float* resized, const float* att,
                                  float* dkey, float* dvalue, float* dresized, float* datt, int N, int T, int F, int Tv, void* stream);
/* rtfs_caf_combine[_backward]_f32 on rows: key, value, out and their gradients (B, T, F, C) with C fastest; resized, att and their
 * gradients stay (B, C, Tv). */
int rtfs_caf_combine_rows_f32(const float* key, const float* value, const float* resized, const float* att, float* out, int B, int T, int F,
                              int C, int Tv, void* stream);
int rtfs_caf_combine_rows_backward_f32(const float* dout, const float* key, const float* value, const float* resized, const float* att,
                                       float* dkey, float* dvalue, float* dresized, float* datt, int B, int T, int F, int C, int Tv,
                                       void* stream);
/* The RTFS block's gateway on rows (B, T, F, C), C fastest (reference separators/tdanet.py:30-38 `gateway = ConvNormAct(in_chan, in_chan, 1,
 * groups=in_chan, act_type)` applied at :106-108 to `x + x_res`): out = PReLU(w_c * (x + x_res) + b_c) in one pass (x_res may be NULL);
 * backward in one pass: dx (the gradient of x and of x_res alike) and dparams = [dw C | db C | dslope 1] (each slot rounded up to 64
 * floats; rtfs_gateway_grad_floats).  w, b: the depthwise Conv2d's weight (C,1,1,1) and bias; slope: nn.PReLU's single weight. */
size_t rtfs_gateway_grad_floats(int C);
size_t rtfs_gateway_workspace_bytes(int C);
int rtfs_gateway_forward_train_f32(const float* x, const float* x_res, const float* w, const float* b, const float* slope, float* out,
                                   size_t rows, int C, void* stream);
int rtfs_gateway_backward_f32(const float* x, const float* x_res, const float* w, const float* b, const float* slope, const float* dout,
                              float* dx, float* dparams, size_t rows, int C, void* ws, size_t ws_bytes, void* stream);
/* Gradient of PITLossWrapper(PairwiseNegSDR) (src/losses/pit_wrapper.py:84-110 around matrix.py:22-53) with respect to the estimates,
 * for the permutation the forward chose: dmin_loss (B) = upstream gradient of min_loss, perm (B, n_src) as returned by
 * rtfs_pit_pairwise_sdr_f32 -> dests (B, n_src, L).  (The targets are data.) */
int rtfs_pit_sdr_backward_f32(const float* ests, const float* targets, const int* perm, const float* dmin_loss, float* dests, int B,
                              int n_src, int L, int sdr_type, int zero_mean, int take_log, void* stream);
/* Pieces of the video-side MultiHeadSelfAttention (src/models/layers/attention.py:28-73) with their adjoints, on rows (b, t) x C:
 * nn.LayerNorm over C; nn.Linear (in_proj / out_proj of nn.MultiheadAttention; N, K multiples of 64); the attention core
 * softmax(q k^T / sqrt(head_dim)) v on the packed projections [q | k | v] (T <= 256, head_dim <= 16), with an optional keep-mask
 * (B*n_head, T, T), already scaled by 1/(1-p), for the dropout nn.MultiheadAttention applies to the attention weights in train mode. */
int rtfs_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float* y, int N, int C, void* stream);
int rtfs_layernorm_rows_backward_f32(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int N,
                                     int C, void* stream);
int rtfs_linear_rows_f32(const float* x, const float* W, const float* bias, float* y, int M, int N, int K, void* stream);
int rtfs_linear_rows_backward_f32(const float* x, const float* W, const float* dy, float* dx, float* dW, float* dbias, int M, int N, int K,
                                  void* ws, size_t ws_bytes, void* stream);
int rtfs_mha_core_f32(const float* qkv, const float* pmask, float* o, int B, int T, int n_head, int head_dim, void* stream);
int rtfs_mha_core_backward_f32(const float* qkv, const float* pmask, const float* dout, float* dqkv, int B, int T, int n_head, int head_dim,
                               void* stream);
/* The two GEMM forms of the training path (bf16x3 split on the matrix cores), exposed for tests:
 * kind 0: C (M,N) = A (M,K) . B (N,K)^T (accumulate != 0: C += ...), N % 64 == 0, K % 16 == 0;
 * kind 1: C (M,N) += A (K,M)^T . B (K,N), M % 64 == 0, N % 64 == 0. */
int rtfs_debug_gemm_f32(int kind, const float* A, const float* B, float* C, int M, int N, int K, int accumulate, void* stream);

/* Diagnostic build of the dual-path sweep with s_memtime stamps at phase boundaries (profiling aid only).
 * x, out: (B, 64, R, Ls) with sequences along the last axis; stamps: DEVICE u64 [ceil(B*R/seqs_per_wg)][16]. */
int rtfs_debug_sweep_stamps(const float* x, const float* pack, float* out, int B, int R, int Ls, unsigned long long* stamps,
                            void* stream);

/* Self test of the f16 MFMA fragment layout the split-precision GEMM kernels assume: D (32x32) = A (32x16) . B (16x32),
 * all DEVICE pointers, row-major; exact for small integer data. */
int rtfs_selftest_mfma_f16(const float* A, const float* B, float* D, void* stream);

/* Measurement hook (bench.py roofline leg; no reference counterpart).  While enabled (on = n > 0), every n-th launch of the fused
 * dual-path sweep kernel is bracketed by HIP events recorded on the stream it is launched on (on = 0: off).  collect() waits for
 * the recorded launches (host-side, call it outside any timed region / graph capture), writes per-launch
 * milliseconds + sequence length + sequence count (HOST pointers, up to cap entries), clears the log and returns the
 * number of entries (or <0 on error). */
int rtfs_sweep_timing_enable(int on);
/* Diagnostics: kernel launches issued by this library in this process so far (every launcher counts; memsets and event records do not).
 * tests/test_hip_parity.py pins the launches of one small-batch forward with it. */
unsigned long long rtfs_debug_launch_count(void);
int rtfs_sweep_timing_collect(float* ms, int* seq_len, int* n_seq, int cap);

/* Evaluation-side loss (the step after the path): PairwiseNegSDR.forward
 * (src/losses/matrix.py:22-53; sdr_type 0 = "snr", 1 = "sisdr", 2 = "sdsdr") followed by PITLossWrapper's factorial search
 * over source permutations (src/losses/pit_wrapper.py:84-110, pit_from = "pw_mtx", perm_reduce = None), n_src <= 4.
 * ests, targets (B, n_src, L) -> pw_loss (B, n_src[est], n_src[target]); min_loss (B) = loss of the best permutation;
 * perm (B, n_src) int32 with perm[b][i] = estimate assigned to target i (the reference's batch_indices, so
 * reordered[b][i] = ests[b][perm[b][i]]).  All device pointers; the batch mean of min_loss is the wrapper's return value. */
int rtfs_pit_pairwise_sdr_f32(const float* ests, const float* targets, int B, int n_src, int L, int sdr_type, int zero_mean,
                              int take_log, float* pw_loss, float* min_loss, int* perm, void* stream);

/* Video front-end (the step before the path): FRCNNVideoModel.forward with backbone_type "resnet",
 * relu_type "prelu", eval mode (src/models/videomodels/frcnn_videomodel.py:61-72, resnet.py:23-118).
 * lips (B, 1, T, 88, 88) grey-scale mouth crops -> out (B, 512, T), the lip embedding AVNet.forward takes.
 * pack: rtfs-net_amd/packing.py:pack_video (eval BatchNorm folded into f16x3 weight images + bias, PReLU slopes). */
size_t rtfs_video_pack_floats(void);
size_t rtfs_video_workspace_bytes(int B, int T);
int rtfs_video_frontend_f32(const float* lips, const float* pack, float* out, int B, int T, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
