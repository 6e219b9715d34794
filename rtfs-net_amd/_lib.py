"""ctypes binding of librtfs_amd.so (the C ABI declared in include/rtfs_amd.h).

There is no fallback: if the shared library is missing the import of any hot-path module raises,
and every call checks that its tensors live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtfs_amd.so")

PACK_ENCODER, PACK_AUDIO_BN, PACK_BLOCK, PACK_DUALPATH, PACK_ATTENTION, PACK_TFAR, PACK_CAF, PACK_S3, PACK_DECODER, PACK_BLOCK_LSTM, PACK_DUALPATH_LSTM = range(11)

_ERR = {-1: "bad shape", -2: "workspace too small", -3: "kernel launch failure", -4: "bad argument"}

_p, _i, _z = C.c_void_p, C.c_int, C.c_size_t
# name -> (restype, argtypes); must list every symbol of include/rtfs_amd.h
SIGNATURES = {
    "rtfs_version": (C.c_char_p, []),
    "rtfs_pack_floats": (_z, [_i]),
    "rtfs_num_frames": (_i, [_i]),
    "rtfs_stft_encoder_workspace_bytes": (_z, [_i, _i]),
    "rtfs_stft_encoder_f32": (_i, [_p, _p, _p, _p, _i, _i, _p, _z, _p]),
    "rtfs_audio_bottleneck_workspace_bytes": (_z, [_i]),
    "rtfs_audio_bottleneck_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_block_workspace_bytes": (_z, [_i, _i, _i]),
    "rtfs_block_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p, _i]),
    "rtfs_dualpath_workspace_bytes": (_z, [_i, _i, _i]),
    "rtfs_dualpath_sru_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_lstm_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_tf_attention_workspace_bytes": (_z, [_i, _i]),
    "rtfs_tf_attention_f32": (_i, [_p, _p, _p, _i, _i, _p, _z, _p]),
    "rtfs_tfar_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "rtfs_tfar_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_caf_workspace_bytes": (_z, [_i, _i]),
    "rtfs_caf_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_vp_pack_floats": (_z, []),
    "rtfs_vp_block_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "rtfs_s3_mask_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "rtfs_istft_decoder_workspace_bytes": (_z, [_i, _i]),
    "rtfs_istft_decoder_f32": (_i, [_p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_separator_workspace_bytes": (_z, [_i, _i, _i]),
    "rtfs_set_batch_split": (_i, [_i]),
    "rtfs_debug_launch_count": (C.c_ulonglong, []),
    "rtfs_separator_workspace_bytes_ex": (_z, [_i, _i, _i, _i]),
    "rtfs_separator_forward_ex_f32": (_i, [_p] * 9 + [_i, _i, _i, _i, _p, _z, _p, _p, _i, _i]),
    "rtfs_separator_forward_f32": (_i, [_p] * 9 + [_i, _i, _i, _i, _p, _z, _p, _p, _i]),
    "rtfs_sru_workspace_bytes": (_z, [_i, _i]),
    "rtfs_sru_f32": (_i, [_p, _p, _p, _i, _i, _p, _z, _p]),
    "rtfs_sru_train_pack_floats": (_z, []),
    "rtfs_sru_grad_floats": (_z, []),
    "rtfs_sru_saved_floats": (_z, [_i, _i]),
    "rtfs_sru_backward_workspace_bytes": (_z, [_i, _i]),
    "rtfs_sru_forward_train_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "rtfs_sru_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_train_pack_floats": (_z, []),
    "rtfs_dualpath_grad_floats": (_z, []),
    "rtfs_dualpath_saved_floats": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_train_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_forward_train_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_lstm_train_pack_floats": (_z, []),
    "rtfs_dualpath_lstm_grad_floats": (_z, []),
    "rtfs_dualpath_lstm_saved_floats": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_lstm_train_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_lstm_forward_train_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_lstm_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_gru_train_pack_floats": (_z, []),
    "rtfs_dualpath_gru_grad_floats": (_z, []),
    "rtfs_dualpath_gru_saved_floats": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_gru_train_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "rtfs_dualpath_gru_forward_train_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_dualpath_gru_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "rtfs_cna_param_floats": (_z, [_p]),
    "rtfs_cna_grad_floats": (_z, [_p]),
    "rtfs_cna_saved_floats": (_z, [_p, _i, _i, _i]),
    "rtfs_cna_workspace_bytes": (_z, [_p, _i, _i, _i]),
    "rtfs_cna_saved_stats_offset": (_z, [_p, _i, _i, _i]),
    "rtfs_cna_grad_norm_offsets": (None, [_p, _p, _p]),
    "rtfs_cna_out_shape": (None, [_p, _i, _i, _p, _p]),
    "rtfs_cna_forward_train_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_cna_bn_update_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, C.c_float, _p]),
    "rtfs_cna_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_tf_attention_train_pack_floats": (_z, []),
    "rtfs_tf_attention_grad_floats": (_z, []),
    "rtfs_tf_attention_saved_floats": (_z, [_i, _i]),
    "rtfs_tf_attention_train_workspace_bytes": (_z, [_i, _i]),
    "rtfs_tf_attention_forward_train_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_tf_attention_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_layout_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "rtfs_adaptive_avg_pool2d_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rtfs_adaptive_avg_pool2d_backward_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rtfs_tfar_combine_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rtfs_tfar_combine_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rtfs_stft_encoder_backward_workspace_bytes": (_z, [_i, _i]),
    "rtfs_stft_encoder_backward_f32": (_i, [_p, _p, _p, _i, _i, _p, _z, _p]),
    "rtfs_istft_decoder_backward_workspace_bytes": (_z, [_i, _i]),
    "rtfs_istft_decoder_backward_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_s3_cmul_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "rtfs_caf_attention_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "rtfs_caf_attention_backward_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "rtfs_caf_combine_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rtfs_caf_combine_backward_f32": (_i, [_p] * 9 + [_i, _i, _i, _i, _p]),
    "rtfs_caf_combine_rows_f32": (_i, [_p] * 5 + [_i] * 5 + [_p]),
    "rtfs_caf_combine_rows_backward_f32": (_i, [_p] * 9 + [_i] * 5 + [_p]),
    "rtfs_gateway_grad_floats": (_z, [_i]),
    "rtfs_gateway_workspace_bytes": (_z, [_i]),
    "rtfs_gateway_forward_train_f32": (_i, [_p] * 6 + [_z, _i, _p]),
    "rtfs_gateway_backward_f32": (_i, [_p] * 8 + [_z, _i, _p, _z, _p]),
    "rtfs_pit_sdr_backward_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "rtfs_layernorm_rows_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "rtfs_layernorm_rows_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "rtfs_linear_rows_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "rtfs_linear_rows_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "rtfs_mha_core_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "rtfs_mha_core_backward_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rtfs_debug_gemm_f32": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "rtfs_debug_sweep_stamps": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "rtfs_selftest_mfma_f16": (_i, [_p, _p, _p, _p]),
    "rtfs_sweep_timing_enable": (_i, [_i]),
    "rtfs_sweep_timing_collect": (_i, [_p, _p, _p, _i]),
    "rtfs_pit_pairwise_sdr_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "rtfs_video_pack_floats": (_z, []),
    "rtfs_video_workspace_bytes": (_z, [_i, _i]),
    "rtfs_video_frontend_f32": (_i, [_p, _p, _p, _i, _i, _p, _z, _p]),
}

_lib = None


def load():
    """Load librtfs_amd.so once; raise (never fall back) when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C rtfs-net_amd/csrc` (or __graft_entry__.build()); "
            "the RTFS-Net MI355X path has no non-HIP fallback"
        )
    lib = C.CDLL(LIB_PATH)
    import functools
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
        # size queries are pure functions of a few integers and a training step asks ~1000 of them: memoise (an instance attribute
        # shadows the CDLL's own lookup)
        if res is _z and args and all(a in (_i, _z) for a in args):
            setattr(lib, name, functools.lru_cache(maxsize=None)(fn))
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(code, code)}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t: torch.Tensor):
    """torch's current stream on the tensor's device as the void* the C ABI takes (the raw-handle query when this torch has it: building a
    torch.cuda.Stream object per call costs microseconds, ~800 times a step)."""
    if _raw_stream is not None:
        idx = t.device.index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("rtfs_net_amd kernels run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"rtfs_net_amd kernels are float32; got {t.dtype}")


_POISON = bool(os.environ.get("RTFS_POISON_WS"))


def workspace(nbytes: int, device):
    """Caller-owned scratch for one C call.  The kernels must never read a workspace byte they have not written in the same call;
    RTFS_POISON_WS=1 (tests) fills every workspace with 0xFF bytes (NaN as f32 / f64) so such a read shows up in the output."""
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    if _POISON:
        ws.fill_(0xFF)
    return ws
