"""Model-level mirror of the reference's ``src/models`` API for the RTFS-Net path.

north_star names -> reference classes:  RTFSNet = AVNet (tdavnet.py), RTFSBlock = TDANetBlock
(separators/tdanet.py), CAFBlock = ATTNFusion (TDAVNet/fusion.py), S3Block = MaskGenerator
(TDAVNet/mask_generator.py).  Constructor keywords, ``forward`` contracts and ``state_dict`` keys follow the
reference so ``AVNet(**yaml["audionet"])``, ``from_pretrain`` and Lightning checkpoints keep working; the
arithmetic runs in ``librtfs_amd.so``.
"""
from __future__ import annotations

import os

import ctypes
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, layers, packing
from .layers import ConvNormAct, InjectionMultiSum, PackedModule, _config_of


# ----------------------------------------------------------------------------- encoder / decoder
def L_recording(*objs):
    from .layers import _recording
    return _recording(*[o for o in objs if o is not None])


class _EncoderTrainFn(torch.autograd.Function):
    """STFTEncoder inside a training step: the inference kernels forward, the Conv2d weight gradient backward (the waveform is data)."""

    @staticmethod
    def forward(ctx, wav, weight, pack):
        lib = _lib.load()
        B, L = wav.shape
        T = lib.rtfs_num_frames(L)
        a0 = torch.empty(B, 256, T, 129, device=wav.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_stft_encoder_workspace_bytes(B, L), wav.device)
        _lib.check(lib.rtfs_stft_encoder_f32(_lib.ptr(wav), _lib.ptr(pack), _lib.ptr(a0), None, B, L, _lib.ptr(ws), ws.numel(),
                                             _lib.stream_of(wav)), "rtfs_stft_encoder_f32")
        ctx.save_for_backward(wav)
        ctx.wshape = weight.shape
        return a0

    @staticmethod
    def backward(ctx, da0):
        lib = _lib.load()
        (wav,) = ctx.saved_tensors
        B, L = wav.shape
        da0 = da0.contiguous()
        dw = torch.empty(256 * 18, device=wav.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_stft_encoder_backward_workspace_bytes(B, L), wav.device)
        _lib.check(lib.rtfs_stft_encoder_backward_f32(_lib.ptr(wav), _lib.ptr(da0), _lib.ptr(dw), B, L, _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_of(wav)), "rtfs_stft_encoder_backward_f32")
        return None, dw.reshape(ctx.wshape), None


class _DecoderTrainFn(torch.autograd.Function):
    """STFTDecoder inside a training step: inference kernels forward; iSTFT adjoint + ConvTranspose2d adjoints backward."""

    @staticmethod
    def forward(ctx, x, weight, pack, length):
        lib = _lib.load()
        B, _, T, _ = x.shape
        wav = torch.empty(B, 1, length, device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_istft_decoder_workspace_bytes(B, T), x.device)
        _lib.check(lib.rtfs_istft_decoder_f32(_lib.ptr(x), _lib.ptr(pack), _lib.ptr(wav), B, T, length, _lib.ptr(ws), ws.numel(),
                                              _lib.stream_of(x)), "rtfs_istft_decoder_f32")
        ctx.save_for_backward(x, weight)
        ctx.length = length
        return wav

    @staticmethod
    def backward(ctx, dwav):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        B, _, T, _ = x.shape
        dwav = dwav.contiguous().to(torch.float32)
        dx = torch.empty_like(x)
        dw = torch.empty(256 * 18, device=x.device, dtype=torch.float32)
        w = weight.detach().contiguous()
        ws = _lib.workspace(lib.rtfs_istft_decoder_backward_workspace_bytes(B, T), x.device)
        _lib.check(lib.rtfs_istft_decoder_backward_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dwav), _lib.ptr(dx), _lib.ptr(dw), B, T, ctx.length,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_istft_decoder_backward_f32")
        return dx, dw.reshape(weight.shape), None, None


class _S3MulFn(torch.autograd.Function):
    """masks (B,256,T,F) x encoder output (B,256,T,F), both [re 128 | im 128]: complex product and its two adjoints."""

    @staticmethod
    def _cmul(a, b, conj):
        lib = _lib.load()
        out = torch.empty_like(a)
        B, P = a.shape[0], a.shape[2] * a.shape[3]
        _lib.check(lib.rtfs_s3_cmul_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(out), B, P, conj, _lib.stream_of(a)), "rtfs_s3_cmul_f32")
        return out

    @staticmethod
    def forward(ctx, masks, emb):
        masks, emb = masks.contiguous(), emb.contiguous()
        ctx.save_for_backward(masks, emb)
        return _S3MulFn._cmul(emb, masks, 0)

    @staticmethod
    def backward(ctx, dout):
        masks, emb = ctx.saved_tensors
        dout = dout.contiguous()
        return _S3MulFn._cmul(emb, dout, 1), _S3MulFn._cmul(masks, dout, 1)


class STFTEncoder(PackedModule):
    """reference TDAVNet/encoder.py:122-175."""

    _pack_fn = staticmethod(packing.pack_encoder)

    def __init__(self, win, hop_length, out_chan=2, kernel_size=-1, stride=1, act_type="ReLU", norm_type="gLN", bias=False, *args, **kwargs):
        super().__init__()
        if not (win == 256 and hop_length == 128 and out_chan == 256 and kernel_size == 3 and stride == 1 and not act_type
                and not norm_type and not bias):
            raise ValueError("MI355X STFTEncoder supports the RTFS-Net yaml: win 256, hop 128, 3x3 conv to 256 ch, no bias/norm/act")
        self.win, self.hop_length, self.out_chan, self.kernel_size, self.stride = win, hop_length, out_chan, kernel_size, stride
        self.act_type, self.norm_type, self.bias = act_type, norm_type, bias
        self.conv = ConvNormAct(in_chan=2, out_chan=out_chan, kernel_size=kernel_size, stride=stride, act_type=act_type,
                                norm_type=norm_type, xavier_init=True, bias=bias, is2d=True)
        self.register_buffer("window", torch.hann_window(win), False)

    @staticmethod
    def unsqueeze_to_2D(x):
        if x.ndim == 1:
            return x.reshape(1, -1)
        if x.ndim == 3:
            assert x.shape[1] == 1
            return x.reshape(x.shape[0], -1)
        return x

    def get_out_chan(self):
        return self.out_chan

    def forward(self, x, return_stats=False):
        x = self.unsqueeze_to_2D(x)
        if x.is_cuda and L_recording(self):
            _lib.need_gpu(x)
            a0 = _EncoderTrainFn.apply(x.contiguous(), self.conv.full_layer[2].weight, self.pack())
            return (a0, None) if return_stats else a0
        self._guard(x)
        lib = _lib.load()
        x = x.contiguous()
        B, L = x.shape
        T = lib.rtfs_num_frames(L)
        a0 = torch.empty(B, 256, T, 129, device=x.device, dtype=torch.float32)
        stats = torch.empty(B, 2, device=x.device, dtype=torch.float64) if return_stats else None
        ws = _lib.workspace(lib.rtfs_stft_encoder_workspace_bytes(B, L), x.device)
        _lib.check(lib.rtfs_stft_encoder_f32(_lib.ptr(x), _lib.ptr(self.pack()), _lib.ptr(a0), _lib.ptr(stats), B, L, _lib.ptr(ws),
                                             ws.numel(), _lib.stream_of(x)), "rtfs_stft_encoder_f32")
        return (a0, stats) if return_stats else a0

    def get_config(self):
        return _config_of(self)


class STFTDecoder(PackedModule):
    """reference TDAVNet/decoder.py:72-132."""

    _pack_fn = staticmethod(packing.pack_decoder)

    def __init__(self, win, hop_length, in_chan, n_src, kernel_size=-1, stride=1, bias=False, *args, **kwargs):
        super().__init__()
        if not (win == 256 and hop_length == 128 and in_chan == 256 and n_src == 1 and kernel_size == 3 and stride == 1 and not bias):
            raise ValueError("MI355X STFTDecoder supports the RTFS-Net yaml: win 256, hop 128, 256 ch, n_src 1, 3x3, no bias")
        self.win, self.hop_length, self.in_chan, self.n_src, self.kernel_size = win, hop_length, in_chan, n_src, kernel_size
        self.padding, self.stride, self.bias = (kernel_size - 1) // 2, stride, bias
        self.decoder = nn.ConvTranspose2d(in_chan, 2, kernel_size, stride=stride, padding=self.padding, bias=bias)
        nn.init.xavier_uniform_(self.decoder.weight)
        self.register_buffer("window", torch.hann_window(win), False)

    def forward(self, x, input_shape):
        batch, length = int(input_shape[0]), int(input_shape[-1])
        T = x.shape[-2]
        if x.shape[-1] != 129:
            raise ValueError("expected 129 frequency bins")
        if x.is_cuda and L_recording(x, self):
            _lib.need_gpu(x)
            return _DecoderTrainFn.apply(x.contiguous().view(batch * self.n_src, self.in_chan, T, 129), self.decoder.weight, self.pack(), length)
        self._guard(x)
        lib = _lib.load()
        x = x.contiguous().view(batch * self.n_src, self.in_chan, T, x.shape[-1])
        wav = torch.empty(batch, self.n_src, length, device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_istft_decoder_workspace_bytes(batch, T), x.device)
        _lib.check(lib.rtfs_istft_decoder_f32(_lib.ptr(x), _lib.ptr(self.pack()), _lib.ptr(wav), batch, T, length, _lib.ptr(ws), ws.numel(),
                                              _lib.stream_of(x)), "rtfs_istft_decoder_f32")
        return wav

    def get_config(self):
        return _config_of(self)


class AudioBottleneck(ConvNormAct, PackedModule):
    """AVNet.audio_bottleneck = ConvNormAct(gLN -> ReLU -> 1x1 256->256) on the HIP path (tdavnet.py:59,89)."""

    _pack_fn = staticmethod(packing.pack_audio_bn)

    def forward(self, x, stats=None, rows_out=False):
        if x.is_cuda and L_recording(x, self):
            return self._forward_train(x, (False, rows_out))  # ConvNormAct's training kernels; rows_out: (B, T, F, C) for the RTFS blocks
        self._guard(x)
        if not (self.in_chan == 256 and self.out_chan == 256 and self.kernel_size == 1 and self.pre_norm_type == "gLN"
                and self.pre_act_type == "ReLU" and not self.norm_type and not self.act_type and self.bias):
            raise ValueError("MI355X audio bottleneck supports the RTFS-Net yaml configuration only")
        lib = _lib.load()
        x = x.contiguous()
        B, _, T, Fq = x.shape
        out = torch.empty_like(x)
        ws = _lib.workspace(lib.rtfs_audio_bottleneck_workspace_bytes(B), x.device)
        _lib.check(lib.rtfs_audio_bottleneck_f32(_lib.ptr(x), _lib.ptr(stats), _lib.ptr(self.pack()), _lib.ptr(out), B, T, Fq, _lib.ptr(ws),
                                                 ws.numel(), _lib.stream_of(x)), "rtfs_audio_bottleneck_f32")
        return out


# ----------------------------------------------------------------------------- RTFS block
class TDANetBlock(PackedModule):
    """RTFS block (2-D) / VP block (1-D): reference separators/tdanet.py:8-131."""

    _pack_fn = staticmethod(packing.pack_block)

    def __init__(self, in_chan, hid_chan, kernel_size=5, stride=2, norm_type="gLN", act_type="PReLU", upsampling_depth=4,
                 layers=dict(), is2d=False):
        super().__init__()
        self.in_chan, self.hid_chan, self.kernel_size, self.stride = in_chan, hid_chan, kernel_size, stride
        self.norm_type, self.act_type, self.upsampling_depth, self.layers, self.is2d = norm_type, act_type, upsampling_depth, layers, is2d
        cna = lambda **kw: ConvNormAct(is2d=is2d, **kw)
        self.gateway = cna(in_chan=in_chan, out_chan=in_chan, kernel_size=1, groups=in_chan, act_type=act_type)
        self.projection = cna(in_chan=in_chan, out_chan=hid_chan, kernel_size=1)
        self.downsample_layers = nn.ModuleList(
            cna(in_chan=hid_chan, out_chan=hid_chan, kernel_size=kernel_size, stride=1 if i == 0 else stride, groups=hid_chan, norm_type=norm_type)
            for i in range(upsampling_depth))
        from . import layers as L
        self.globalatt = nn.Sequential(*[L.get(layer["layer_type"])(in_chan=hid_chan, **layer) for _, layer in self.layers.items()])
        ims = lambda: InjectionMultiSum(in_chan=hid_chan, kernel_size=kernel_size, norm_type=norm_type, is2d=is2d)
        self.fusion_layers = nn.ModuleList(ims() for _ in range(upsampling_depth))
        self.concat_layers = nn.ModuleList(ims() for _ in range(upsampling_depth - 1))
        self.residual_conv = cna(in_chan=hid_chan, out_chan=in_chan, kernel_size=1)
        self._hip = bool(is2d)
        if self._hip:
            kinds = [type(m).__name__ for m in self.globalatt]
            dims = [getattr(m, "dim", None) for m in self.globalatt]
            cells = [getattr(m, "rnn_type", None) for m in self.globalatt][:2]
            if not (in_chan == 256 and hid_chan == 64 and kernel_size == 4 and stride == 2 and norm_type == "gLN" and act_type == "PReLU"
                    and upsampling_depth == 2 and kinds == ["DualPathRNN", "DualPathRNN", "MultiHeadSelfAttention2D"] and dims == [4, 3, 3]
                    and cells in (["SRU", "SRU"], ["LSTM", "LSTM"], ["GRU", "GRU"])):
                raise ValueError("MI355X RTFS block supports the RTFS-Net yaml audio_params only (both sweeps SRU, both LSTM or both GRU)")
            self.rnn_kind = {"SRU": 0, "LSTM": 1, "GRU": 2}[cells[0]]  # 2: no fused block kernel, the unfused HIP kernels serve inference too

    def _vp_supported(self):
        kinds = [type(m).__name__ for m in self.globalatt]
        return (not self.is2d and self.in_chan == 512 and self.hid_chan == 64 and self.kernel_size == 3 and self.stride == 2
                and self.norm_type == "BatchNorm1d" and self.act_type == "PReLU" and self.upsampling_depth == 4 and kinds == ["GlobalAttention"]
                and self.globalatt[0].n_head == 8 and self.globalatt[0].kernel_size == 3 and self.globalatt[0].hid_chan == 128)

    def pack_vp(self):
        sd = self._state_tensors()
        key = (packing.pack_epoch(),) + tuple((v.data_ptr(), v._version) for v in sd.values())
        if getattr(self, "_vp_key", None) != key:
            with torch.no_grad():
                pk = packing.pack_vp(sd)
            object.__setattr__(self, "_vp_buf", pk)
            object.__setattr__(self, "_vp_key", key)
        return self._vp_buf

    def forward_vp(self, x):
        """VP block on the fused HIP kernel (one workgroup per sample)."""
        self._guard(x)
        lib = _lib.load()
        x = x.contiguous()
        B, _, Tv = x.shape
        out = torch.empty_like(x)
        pk = self.pack_vp()
        assert pk.numel() == lib.rtfs_vp_pack_floats()
        _lib.check(lib.rtfs_vp_block_f32(_lib.ptr(x), _lib.ptr(pk), _lib.ptr(out), B, Tv, _lib.stream_of(x)), "rtfs_vp_block_f32")
        return out

    def _forward_train_rows(self, x, x_res=None, rows_in=False, rows_out=False):
        """The audio block inside a training step with rows (B, T, F, C) between its modules: the same composition as _forward_train,
        but only the block's input and output change layout (every module converting at its own boundary costs 10 % of a step) - and
        not even those when the caller keeps rows between blocks (``rows_in`` / ``rows_out``; AVNet.forward_train)."""
        from . import layers as L
        rr = (True, True)
        residual = None
        if x.is_cuda:  # gateway fused with the residual input: one pass each way over the block's largest tensor
            xr = x if rows_in else L._LayoutFn.apply(x, True)
            rr_res = x_res if (rows_in or x_res is None) else L._LayoutFn.apply(x_res, True)
            residual = L.gateway_train(self.gateway, xr, rr_res)
        if residual is None:
            if x_res is not None:
                x = x + x_res
            residual = self.gateway._forward_train(x, (rows_in, True))
        x_enc = self.projection._forward_train(residual, rr)
        down = [self.downsample_layers[0]._forward_train(x_enc, rr)]
        for i in range(1, self.upsampling_depth):
            down.append(self.downsample_layers[i]._forward_train(down[-1], rr))
        size = down[-1].shape[1:3]
        g = down[-1] + sum(L.adaptive_avg_pool(f, size, True) for f in down[:-1])
        for m in self.globalatt:
            if isinstance(m, L.DualPathRNN):
                sru = [p for cell in m.rnn.rnn_lst for p in (cell.weight, cell.weight_c, cell.bias)]
                g = L.dualpath_train(g, m.dim + 10, m.norm.gamma, m.norm.beta, sru, m.linear.weight, m.linear.bias)
            else:
                names, params = zip(*m.named_parameters())
                g = L.attention_train(g, names, True, params)
        fused = [self.fusion_layers[i]._forward_train(down[i], g, True) for i in range(self.upsampling_depth)]
        expanded = self.concat_layers[-1]._forward_train(fused[-2], fused[-1], True) + down[-2]
        for i in range(self.upsampling_depth - 3, -1, -1):
            expanded = self.concat_layers[i]._forward_train(fused[i], expanded, True) + down[i]
        out = self.residual_conv._forward_train(expanded, rr) + residual
        return out if rows_out else L._LayoutFn.apply(out, False)

    def _forward_train(self, x, x_res=None):
        """The block inside a training step (reference separators/tdanet.py:104-131, line by line): every module runs its HIP
        forward-with-saved-state and hands autograd its HIP backward; tensor additions are the only torch ops."""
        from . import layers as L
        if self._hip and self.rnn_kind == 0 and not os.environ.get("RTFS_TRAIN_CF"):
            return self._forward_train_rows(x, x_res)
        if x_res is not None:
            x = x + x_res
        residual = self.gateway(x)
        x_enc = self.projection(residual)
        down = [self.downsample_layers[0](x_enc)]
        for i in range(1, self.upsampling_depth):
            down.append(self.downsample_layers[i](down[-1]))
        size = down[-1].shape[2:]
        g = down[-1] + sum(L.adaptive_avg_pool(f, size) for f in down[:-1])  # pooling the last level onto itself is the identity
        g = self.globalatt(g)
        fused = [self.fusion_layers[i](down[i], g) for i in range(self.upsampling_depth)]
        expanded = self.concat_layers[-1](fused[-2], fused[-1]) + down[-2]
        for i in range(self.upsampling_depth - 3, -1, -1):
            expanded = self.concat_layers[i](fused[i], expanded) + down[i]
        return self.residual_conv(expanded) + residual

    def forward(self, x, x_res=None):
        if x.is_cuda and L_recording(x, x_res, self):  # audio (2-D) and video (1-D) blocks alike
            return self._forward_train(x, x_res)
        if self._hip and (self.rnn_kind == 2 or x.shape[2] // 2 > layers.FUSED_MAX_SWEEP):
            # GRU cells, or a time axis past the fused kernels' on-chip sweep (> 4 s): the block composed from the unfused HIP kernels
            _lib.need_gpu(x, x_res)
            if self.training:
                raise RuntimeError("TDANetBlock: call .eval() for inference")
            with torch.no_grad(), layers.force_train_kernels():
                return self._forward_train(x, x_res)
        if not self._hip:
            x = x if x_res is None else x + x_res
            if x.is_cuda and not self.training and self._vp_supported() and x.shape[-1] <= layers.FUSED_MAX_VIDEO_FRAMES:
                return self.forward_vp(x)
            if x.is_cuda and not self.training:  # more frames than the one-workgroup-per-clip kernel holds in LDS: the per-layer HIP kernels
                with torch.no_grad(), layers.force_train_kernels():
                    return self._forward_train(x, None)
            return self._forward_1d(x)
        self._guard(x, x_res)
        lib = _lib.load()
        x = x.contiguous()
        x_res = None if x_res is None else x_res.contiguous()
        B, _, T, Fq = x.shape
        out = torch.empty_like(x)
        ws = _lib.workspace(lib.rtfs_block_workspace_bytes(B, T, Fq), x.device)
        _lib.check(lib.rtfs_block_f32(_lib.ptr(x), _lib.ptr(x_res), _lib.ptr(self.pack()), _lib.ptr(out), B, T, Fq, _lib.ptr(ws), ws.numel(),
                                      _lib.stream_of(x), self.rnn_kind), "rtfs_block_f32")
        return out

    def _forward_1d(self, x):
        """VP block on stock torch ops (0.004 GMAC)."""
        residual = self.gateway(x)
        downs = [self.downsample_layers[0](self.projection(residual))]
        for i in range(1, self.upsampling_depth):
            downs.append(self.downsample_layers[i](downs[-1]))
        size = downs[-1].shape[-1]
        g = self.globalatt(sum(F.adaptive_avg_pool1d(d, size) for d in downs))
        fused = [self.fusion_layers[i](downs[i], g) for i in range(self.upsampling_depth)]
        expanded = self.concat_layers[-1](fused[-2], fused[-1]) + downs[-2]
        for i in range(self.upsampling_depth - 3, -1, -1):
            expanded = self.concat_layers[i](fused[i], expanded) + downs[i]
        return self.residual_conv(expanded) + residual


class TDANet(nn.Module):
    """reference separators/tdanet.py:134-209: ``repeats`` applications, one shared block when ``shared``."""

    def __init__(self, in_chan=-1, hid_chan=-1, kernel_size=5, stride=2, norm_type="gLN", act_type="PReLU", upsampling_depth=4,
                 layers=[], repeats=4, shared=False, is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan, self.hid_chan, self.kernel_size, self.stride, self.norm_type = in_chan, hid_chan, kernel_size, stride, norm_type
        self.act_type, self.upsampling_depth, self.layers, self.repeats, self.shared, self.is2d = act_type, upsampling_depth, layers, repeats, shared, is2d
        mk = (lambda: TDANetBlock(in_chan, hid_chan, kernel_size, stride, norm_type, act_type, upsampling_depth, layers, is2d)) \
            if (in_chan > 0 and hid_chan > 0) else nn.Identity
        self.blocks = mk() if shared else nn.ModuleList(mk() for _ in range(repeats))

    def get_block(self, i):
        return self.blocks if self.shared else self.blocks[i]

    def forward(self, x):
        residual = x
        for i in range(self.repeats):
            blk = self.get_block(i)
            x = blk(x) if i == 0 else (blk(x, residual) if getattr(blk, "_hip", False) else blk(x + residual))
        return x


_SEPARATORS = {"TDANet": TDANet}


# ----------------------------------------------------------------------------- CAF
class ATTNFusion(nn.Module):
    """CAF block: reference TDAVNet/fusion.py:187-212."""

    def __init__(self, ain_chan, vin_chan, kernel_size, video_fusion=True, is2d=True, *args, **kwargs):
        super().__init__()
        self.ain_chan, self.vin_chan, self.kernel_size, self.video_fusion, self.is2d = ain_chan, vin_chan, kernel_size, video_fusion, is2d
        if video_fusion:
            raise ValueError("MI355X CAF: video_fusion (fusion_repeats > 1) is not used by any RTFS-Net yaml and is not built")
        self.audio_lstm = layers.ATTNFusionCell(ain_chan, vin_chan, kernel_size, is2d)

    def forward(self, audio, video):
        return self.audio_lstm(audio, video), video


class MultiModalFusion(nn.Module):
    """reference TDAVNet/fusion.py:215-281."""

    def __init__(self, audio_bn_chan, video_bn_chan, kernel_size=1, fusion_repeats=3, fusion_type="ConcatFusion", fusion_shared=False,
                 is2d=False, **kwargs):
        super().__init__()
        self.audio_bn_chan, self.video_bn_chan, self.kernel_size, self.fusion_repeats = audio_bn_chan, video_bn_chan, kernel_size, fusion_repeats
        self.fusion_type, self.fusion_shared, self.is2d = fusion_type, fusion_shared, is2d
        if fusion_repeats > 0 and fusion_type != "ATTNFusion":
            raise ValueError(f"fusion_type {fusion_type}: only ATTNFusion (CAF) is on the RTFS-Net path")
        mk = lambda vf: ATTNFusion(ain_chan=audio_bn_chan, vin_chan=video_bn_chan, kernel_size=kernel_size, video_fusion=vf, is2d=is2d, **kwargs)
        if fusion_repeats <= 0:
            self.fusion_module = nn.Identity()
        elif fusion_shared:
            self.fusion_module = mk(fusion_repeats > 1)
        else:
            self.fusion_module = nn.ModuleList(mk(i != fusion_repeats - 1) for i in range(fusion_repeats))

    def get_fusion_block(self, i):
        return self.fusion_module if self.fusion_shared else self.fusion_module[i]

    def forward(self, audio, video):
        a_res, v_res = audio, video
        for i in range(self.fusion_repeats):
            if i == 0:
                a, v = self.get_fusion_block(i)(audio, video)
            else:
                a, v = self.get_fusion_block(i)(a + a_res, v + v_res)
        return a


# ----------------------------------------------------------------------------- refinement module
class RefinementModule(nn.Module):
    """reference TDAVNet/refinement_module.py:10-62."""

    def __init__(self, audio_params, video_params, audio_bn_chan, video_bn_chan, fusion_params):
        super().__init__()
        self.audio_params, self.video_params, self.audio_bn_chan, self.video_bn_chan, self.fusion_params = \
            audio_params, video_params, audio_bn_chan, video_bn_chan, fusion_params
        self.fusion_repeats = self.video_params.get("repeats", 0)
        self.audio_repeats = self.audio_params["repeats"] - self.fusion_repeats
        self.audio_net = _separator(self.audio_params.get("audio_net", None))(**self.audio_params, in_chan=audio_bn_chan)
        self.video_net = _separator(self.video_params.get("video_net", None))(**self.video_params, in_chan=video_bn_chan)
        self.crossmodal_fusion = MultiModalFusion(**self.fusion_params, audio_bn_chan=audio_bn_chan, video_bn_chan=video_bn_chan,
                                                  fusion_repeats=self.fusion_repeats)

    def forward(self, audio, video):
        a_res, v_res = audio, video
        for i in range(self.fusion_repeats):
            audio = self.audio_net.get_block(i)(audio) if i == 0 else self.audio_net.get_block(i)(audio, a_res)
            video = self.video_net.get_block(i)(video if i == 0 else video + v_res)
            audio, video = self.crossmodal_fusion.get_fusion_block(i)(audio, video)
        for j in range(self.audio_repeats):
            i = j + self.fusion_repeats
            audio = self.audio_net.get_block(i)(audio) if i == 0 else self.audio_net.get_block(i)(audio, a_res)
        return audio

    def get_config(self):
        return _config_of(self)


def _separator(identifier):
    if identifier is None:
        return nn.Identity
    if callable(identifier):
        return identifier
    if identifier in _SEPARATORS:
        return _SEPARATORS[identifier]
    raise ValueError("Could not interpret normalization identifier: " + str(identifier))


# ----------------------------------------------------------------------------- S^3 mask head
class MaskGenerator(PackedModule):
    """S^3 block: reference TDAVNet/mask_generator.py:20-99 with RI_split."""

    _pack_fn = staticmethod(packing.pack_s3)

    def __init__(self, n_src, audio_emb_dim, bottleneck_chan, kernel_size=1, mask_act="ReLU", RI_split=False, output_gate=False,
                 dw_gate=False, direct=False, is2d=False, *args, **kwargs):
        super().__init__()
        if not (n_src == 1 and audio_emb_dim == 256 and bottleneck_chan == 256 and kernel_size == 1 and mask_act == "ReLU" and RI_split
                and not output_gate and not direct and is2d):
            raise ValueError("MI355X S^3 block supports the RTFS-Net yaml mask_generation_params only")
        self.n_src, self.in_chan, self.bottleneck_chan, self.kernel_size, self.mask_act = n_src, audio_emb_dim, bottleneck_chan, kernel_size, mask_act
        self.output_gate, self.dw_gate, self.RI_split, self.direct, self.is2d = output_gate, dw_gate, RI_split, direct, is2d
        self.mask_generator = nn.Sequential(nn.PReLU(), ConvNormAct(bottleneck_chan, n_src * audio_emb_dim, kernel_size, act_type=mask_act, is2d=is2d))

    def forward(self, refined_features, audio_mixture_embedding, rows_in=False):
        if refined_features.is_cuda and L_recording(refined_features, audio_mixture_embedding, self):
            from .layers import _CNATrainFn
            prelu, cna = self.mask_generator
            conv = cna.full_layer[2]
            # PReLU -> 1x1 conv -> ReLU as one ConvNormAct whose pre-activation is the stand-alone nn.PReLU (mask_generator.py:52-58);
            # rows_in: the refined features arrive as (B, T, F, C) rows from the last RTFS block
            masks = _CNATrainFn.apply(refined_features, (256, 256, 1, 1, 0, 0, 2, 0, 1, 1, 1, 0, int(rows_in), 0), None, None, prelu.weight,
                                      conv.weight, conv.bias, None, None, None)
            return _S3MulFn.apply(masks, audio_mixture_embedding).unsqueeze(1)
        self._guard(refined_features, audio_mixture_embedding)
        lib = _lib.load()
        r, a0 = refined_features.contiguous(), audio_mixture_embedding.contiguous()
        B, _, T, Fq = r.shape
        out = torch.empty(B, 1, 256, T, Fq, device=r.device, dtype=torch.float32)
        _lib.check(lib.rtfs_s3_mask_f32(_lib.ptr(r), _lib.ptr(a0), _lib.ptr(self.pack()), _lib.ptr(out), B, T, Fq, _lib.stream_of(r)), "rtfs_s3_mask_f32")
        return out

    def get_config(self):
        return _config_of(self)


_ENCODERS = {"STFTEncoder": STFTEncoder}
_DECODERS = {"STFTDecoder": STFTDecoder}
_MASKGENS = {"MaskGenerator": MaskGenerator}


def _lookup(table, identifier):
    if isinstance(identifier, str) and identifier in table:
        return table[identifier]
    raise ValueError("Could not interpret normalization identifier: " + str(identifier))


# ----------------------------------------------------------------------------- top-level model
class BaseAVModel(nn.Module):
    """reference TDAVNet/base_av_model.py."""

    @staticmethod
    def load_state_dict_in(model, pretrained_dict):
        model_dict = model.state_dict()
        model_dict.update({k[12:]: v for k, v in pretrained_dict.items() if "audio_model" in k})
        model.load_state_dict(model_dict)
        return model

    @staticmethod
    def from_pretrain(pretrained_model_conf_or_path, *args, **kwargs):
        if isinstance(pretrained_model_conf_or_path, str):
            # non-executing loader; the reference's serialize() stores torch.__version__ (a str subclass) under infos
            with torch.serialization.safe_globals([torch.torch_version.TorchVersion]):
                conf = torch.load(pretrained_model_conf_or_path, map_location="cpu", weights_only=True)
        else:
            conf = pretrained_model_conf_or_path
        model = get(conf["model_name"])(print_macs=False, *args, **kwargs)
        model.load_state_dict(conf["state_dict"])
        return model

    def serialize(self):
        infos = dict(software_versions=dict(torch_version=str(torch.__version__), python_version=sys.version))
        return dict(model_name=self.__class__.__name__, state_dict=self.get_state_dict(), model_args=self.get_config(), infos=infos)

    def get_state_dict(self):
        return self.state_dict()


class AVNet(BaseAVModel):
    """RTFS-Net: reference tdavnet.py:14-108.  forward(audio_mixture (B,L)|(L)|(B,1,L), mouth_embedding (B,512,Tv)) -> (B,n_src,L)."""

    def __init__(self, n_src, enc_dec_params, audio_bn_params, audio_params, mask_generation_params, pretrained_vout_chan=-1,
                 video_bn_params=dict(), video_params=dict(), fusion_params=dict(), print_macs=True, *args, **kwargs):
        super().__init__()
        self.n_src, self.pretrained_vout_chan = n_src, pretrained_vout_chan
        self.audio_bn_params, self.video_bn_params, self.enc_dec_params = audio_bn_params, video_bn_params, enc_dec_params
        self.audio_params, self.video_params, self.fusion_params = audio_params, video_params, fusion_params
        self.mask_generation_params, self.print_macs = mask_generation_params, print_macs
        self.encoder = _lookup(_ENCODERS, enc_dec_params["encoder_type"])(**enc_dec_params, in_chan=1, upsampling_depth=audio_params.get("upsampling_depth", 1))
        self.enc_out_chan = self.encoder.get_out_chan()
        self.mask_generation_params["mask_generator_type"] = self.mask_generation_params.get("mask_generator_type", "MaskGenerator")
        self.audio_bn_chan = self.audio_bn_params.get("out_chan", self.enc_out_chan)
        self.audio_bn_params["out_chan"] = self.audio_bn_chan
        self.video_bn_chan = self.video_bn_params.get("out_chan", self.pretrained_vout_chan)
        self.audio_bottleneck = AudioBottleneck(**self.audio_bn_params, in_chan=self.enc_out_chan)
        self.video_bottleneck = ConvNormAct(**self.video_bn_params, in_chan=self.pretrained_vout_chan)
        if self.video_bn_params.get("kernel_size", -1) > 0:
            raise ValueError("MI355X AVNet: a non-identity video bottleneck is not on the RTFS-Net path")
        self.refinement_module = RefinementModule(fusion_params=fusion_params, audio_params=audio_params, video_params=video_params,
                                                  audio_bn_chan=self.audio_bn_chan, video_bn_chan=self.video_bn_chan)
        if self.refinement_module.fusion_repeats != 1 or not audio_params.get("shared", False):
            raise ValueError("MI355X AVNet supports video repeats = 1 and a shared audio block (all RTFS-Net yamls)")
        self.mask_generator = _lookup(_MASKGENS, self.mask_generation_params["mask_generator_type"])(
            **self.mask_generation_params, n_src=n_src, audio_emb_dim=self.enc_out_chan, bottleneck_chan=self.audio_bn_chan)
        self.decoder = _lookup(_DECODERS, enc_dec_params["decoder_type"])(**enc_dec_params, in_chan=self.enc_out_chan * n_src, n_src=n_src)
        self.fused = True  # one C call for the whole separator; False composes the per-module entry points
        if print_macs:
            self.get_MACs()

    # -- forward
    def forward(self, audio_mixture, mouth_embedding=None):
        if not self.fused:
            return self.forward_modular(audio_mixture, mouth_embedding)
        wav = STFTEncoder.unsqueeze_to_2D(audio_mixture)
        _lib.need_gpu(wav, mouth_embedding)
        if L_recording(self):
            return self.forward_train(audio_mixture, mouth_embedding)
        frames = int(_lib.load().rtfs_num_frames(int(wav.shape[-1])))
        too_long = frames // 2 > layers.FUSED_MAX_SWEEP or (mouth_embedding is not None and mouth_embedding.shape[-1] > layers.FUSED_MAX_VIDEO_FRAMES)
        if self.refinement_module.audio_net.get_block(0).rnn_kind == 2 or too_long:
            # GRU cells, or an utterance past the fused kernels' on-chip limits (> 4 s of audio / > 120 video frames): the separator
            # composed from the unfused HIP kernels - any length, like the reference (infer_any_video.py:86 feeds whole files)
            if self.training:
                raise RuntimeError("AVNet: call .eval() for inference")
            with torch.no_grad(), layers.force_train_kernels():
                return self.forward_train(audio_mixture, mouth_embedding)
        if self.training:
            raise RuntimeError("AVNet: in .train() mode only the gradient-recording forward exists (see forward_train); "
                               "call .eval() for inference")
        lib = _lib.load()
        wav = wav.contiguous()
        B, L = wav.shape
        rm = self.refinement_module
        # VP block on a side stream: it only feeds the CAF block, so it overlaps the encoder and the first RTFS block
        main = torch.cuda.current_stream(wav.device)
        side = self._side_stream(wav.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            vp = rm.video_net.get_block(0)(self.video_bottleneck(mouth_embedding)).contiguous()
            ready = torch.cuda.Event()
            ready.record(side)
        vp.record_stream(main)
        Tv = vp.shape[-1]
        out = torch.empty(B, self.n_src, L, device=wav.device, dtype=torch.float32)
        # batch split of THIS model's calls (0 = the library's process-wide default, rtfs_set_batch_split / RTFS_SPLIT): a per-call argument
        # of the C ABI, so two models / threads can run different schedules
        split = int(getattr(self, "batch_split", 0) or 0)
        ws = _lib.workspace(lib.rtfs_separator_workspace_bytes_ex(B, L, Tv, split), wav.device)
        packs = [self.encoder.pack(), self.audio_bottleneck.pack(), rm.audio_net.get_block(0).pack(),
                 rm.crossmodal_fusion.get_fusion_block(0).audio_lstm.pack(), self.mask_generator.pack(), self.decoder.pack()]
        _lib.check(lib.rtfs_separator_forward_ex_f32(_lib.ptr(wav), _lib.ptr(vp), *[_lib.ptr(p) for p in packs], _lib.ptr(out), B, L, Tv,
                                                     int(self.audio_params["repeats"]), _lib.ptr(ws), ws.numel(), _lib.stream_of(wav),
                                                     ctypes.c_void_p(ready.cuda_event), rm.audio_net.get_block(0).rnn_kind, split),
                   "rtfs_separator_forward_ex_f32")
        return out

    def _side_stream(self, device):
        streams = self.__dict__.setdefault("_side_streams", {})
        key = (device.type, device.index)
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

    def forward_modular(self, audio_mixture, mouth_embedding=None):
        """Same result through the per-module entry points (the reference's own call sequence, tdavnet.py:86-97)."""
        emb, stats = self.encoder(audio_mixture, return_stats=True)
        audio = self.audio_bottleneck(emb, stats)
        video = self.video_bottleneck(mouth_embedding)
        refined = self.refinement_module(audio, video)
        sep = self.mask_generator(refined, emb)
        return self.decoder(sep, STFTEncoder.unsqueeze_to_2D(audio_mixture).shape)

    def forward_train(self, audio_mixture, mouth_embedding):
        """The separator inside a training step (reference tdavnet.py:86-97 called from src/system/core.py:94-123): every module runs
        its HIP forward-with-saved-state and hands autograd its HIP backward.  BatchNorm layers follow their own mode (train: statistics
        of this rank's batch; eval: running statistics), the video-side VP block is differentiated when it is in train mode and has
        trainable parameters, and evaluated without a graph on its inference kernel otherwise (``freeze_for_finetune``)."""
        rm = self.refinement_module
        vp_block = rm.video_net.get_block(0)
        # The VP block is ~1500 tiny launches that only feed the CAF block: it runs on the side stream under the encoder and the first RTFS
        # block, and - autograd replays every node on the stream its forward ran on - its backward runs under theirs as well.
        main = torch.cuda.current_stream(mouth_embedding.device)
        side = self._side_stream(mouth_embedding.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            video = self.video_bottleneck(mouth_embedding)
            if vp_block.training and any(p.requires_grad for p in vp_block.parameters()):
                video = vp_block(video)
            else:
                with torch.no_grad():
                    video = vp_block(video)
        emb = self.encoder(audio_mixture)
        blk = rm.audio_net.get_block(0)
        # rows (B, T, F, C) from the bottleneck to the mask generator wherever the modules take them (the SRU block's rows pipeline); only
        # the CAF block and the S^3 product work channel-first (each layout change of a 256-channel tensor is a 50 us transpose, twice
        # per step with the backward)
        rows = (isinstance(self.audio_bottleneck, AudioBottleneck) and isinstance(self.mask_generator, MaskGenerator)
                and getattr(blk, "_hip", False) and blk.rnn_kind == 0 and not os.environ.get("RTFS_TRAIN_CF")
                and L_recording(emb, self.audio_bottleneck) and blk.training and self.mask_generator.training)
        fb = rm.crossmodal_fusion.get_fusion_block(0)
        caf_rows = (rows and isinstance(fb, ATTNFusion) and fb.audio_lstm.kernel_size == 4 and fb.audio_lstm.is2d
                    and fb.audio_lstm.in_chan_a in (64, 128, 256, 512, 1024) and L_recording(emb, fb.audio_lstm))
        if rows:
            a_res = self.audio_bottleneck(emb, rows_out=True)
            audio = blk._forward_train_rows(a_res, None, rows_in=True, rows_out=caf_rows)
        else:
            a_res = audio = self.audio_bottleneck(emb)
            audio = blk(audio)
        main.wait_stream(side)
        video.record_stream(main)
        if caf_rows:
            audio = fb.audio_lstm._forward_train(audio, video, rows=True)
        else:
            audio, _ = fb(audio, video)
        if rows:
            if not caf_rows:
                audio = layers._LayoutFn.apply(audio, True)
            for _ in range(rm.audio_repeats):
                audio = blk._forward_train_rows(audio, a_res, rows_in=True, rows_out=True)
            sep = self.mask_generator(audio, emb, rows_in=True)
        else:
            for _ in range(rm.audio_repeats):
                audio = blk(audio, a_res)
            sep = self.mask_generator(audio, emb)
        return self.decoder(sep, STFTEncoder.unsqueeze_to_2D(audio_mixture).shape)

    def freeze_for_finetune(self):
        """Fine-tuning configuration: everything in train mode except BatchNorm layers (frozen running statistics) and the video-side VP
        block (eval mode, requires_grad False; it then runs on its fused inference kernel).  Plain ``.train()`` trains everything, with
        BatchNorm on the batch statistics of the local rank and the VP block's dropout / DropPath drawn from torch's RNG.  Returns self."""
        self.train()
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)):
                m.eval()
        vp = self.refinement_module.video_net
        vp.eval()
        for p in vp.parameters():
            p.requires_grad_(False)
        self.video_bottleneck.eval()
        return self

    def get_config(self):
        return dict(encoder=self.encoder.get_config(), audio_bottleneck=self.audio_bottleneck.get_config(),
                    video_bottleneck=self.video_bottleneck.get_config(), refinement_module=self.refinement_module.get_config(),
                    mask_generator=self.mask_generator.get_config(), decoder=self.decoder.get_config())

    def get_MACs(self):
        """Analytic MAC / parameter report in the reference's table format (base_av_model.py:61-118; the reference
        uses thop on a 2 s input).  Sets ``self.macs_parms`` (read by test.py:91,96)."""
        from .macs import macs_report
        self.macs_parms = macs_report(self)
        print(self.macs_parms)


# north_star vocabulary
RTFSNet = AVNet
RTFSBlock = TDANetBlock
CAFBlock = ATTNFusion
S3Block = MaskGenerator

_MODELS = {"avnet": AVNet, "rtfsnet": AVNet}


def register_model(custom_model):
    """reference src/models/__init__.py:15-24."""
    name = custom_model.__name__.lower()
    if name in _MODELS:
        raise ValueError(f"Model {custom_model.__name__} already exists. Choose another name.")
    _MODELS[name] = custom_model


def get(identifier):
    """reference src/models/__init__.py:27-42 (case-insensitive)."""
    if isinstance(identifier, str) and identifier.lower() in _MODELS:
        return _MODELS[identifier.lower()]
    raise ValueError(f"Could not interpret model name : {str(identifier)}")
