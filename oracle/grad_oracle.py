"""Gradient oracle for the training side (TEST INFRASTRUCTURE ONLY - the product never imports this).

Differentiable float64 restatements, in torch ops, of the functions in ``rtfs_oracle.py`` that the backward kernels
are checked against; gradients come from torch.autograd, so nothing here restates a backward formula.  Pinning:
forward values are compared with the numpy restatement (``tests/test_oracle_golden.py``), which in turn is pinned by the
reference's vectors where the reference arithmetic exists; the SRU cell is third-party (``sru``, absent offline), so its
recurrence - forward and therefore backward - stays "parity unpinned" exactly as in ``rtfs_oracle.sru_forward``.
"""
from __future__ import annotations

import torch


def sru_forward_torch(x, layers):
    """rtfs_oracle.sru_forward in torch (call site reference rnn_layers.py:99-105,150).
    x (L,N,Din); layers = [(weight (Din, 2d*k), weight_c (4d), bias (4d))]; returns h (L,N,2d)."""
    L, N, _ = x.shape
    h_in = x
    for (W, wc, bias) in layers:
        d2 = wc.shape[0] // 2
        d = d2 // 2
        k = W.shape[1] // d2
        U = (h_in.reshape(L * N, -1) @ W).reshape(L, N, 2, d, k)
        vf, vr = wc.reshape(2, 2, d)
        bf, br = bias.reshape(2, 2, d)
        xprime = U[..., 3] if k == 4 else h_in.reshape(L, N, 2, d)
        outs = []
        for di in range(2):
            c = torch.zeros(N, d, dtype=x.dtype)
            hs = [None] * L
            for t in (range(L) if di == 0 else range(L - 1, -1, -1)):
                u0 = U[t, :, di, :, 0]
                f = torch.sigmoid(U[t, :, di, :, 1] + c * vf[di] + bf[di])
                r = torch.sigmoid(U[t, :, di, :, 2] + c * vr[di] + br[di])
                c = u0 + (c - u0) * f
                xp = xprime[t, :, di, :]
                hs[t] = xp + (c - xp) * r
            outs.append(torch.stack(hs))
        h_in = torch.stack(outs, 2).reshape(L, N, d2)
    return h_in


def sru_grads(x, layers, dh):
    """numpy in / numpy out: (h, dx, [(dW, dwc, dbias)]) of sum(h * dh) by autograd in float64."""
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lt = [tuple(torch.tensor(p, dtype=torch.float64, requires_grad=True) for p in lay) for lay in layers]
    h = sru_forward_torch(xt, lt)
    flat = [p for lay in lt for p in lay]
    grads = torch.autograd.grad(h, [xt] + flat, torch.tensor(dh, dtype=torch.float64))
    gl = [tuple(g.numpy() for g in grads[1 + 3 * i: 4 + 3 * i]) for i in range(len(lt))]
    return h.detach().numpy(), grads[0].numpy(), gl


def dualpath_rnn_torch(x, p, dim, kernel_size=8):
    """rtfs_oracle.dualpath_rnn in torch (reference rnn_layers.py:136-162, SRU cell).  x (B,C,T,F) float64 tensor;
    p = dict of tensors with the module's state_dict names."""
    if dim == 4:
        x = x.permute(0, 1, 3, 2)
    B, C, nT, nF = x.shape
    res = x
    mu = x.mean(1, keepdim=True)
    var = ((x - mu) ** 2).mean(1, keepdim=True)
    xn = (x - mu) / torch.sqrt(var + 1e-5) * p["norm.gamma"].reshape(1, C, 1, 1) + p["norm.beta"].reshape(1, C, 1, 1)
    seq = xn.permute(0, 3, 1, 2).reshape(B * nF, C, nT)
    Lr = nT - kernel_size + 1
    unf = torch.stack([seq[:, :, kk:kk + Lr] for kk in range(kernel_size)], 2).reshape(B * nF, C * kernel_size, Lr).permute(2, 0, 1)
    layers = []
    i = 0
    while f"rnn.rnn_lst.{i}.weight" in p:
        layers.append((p[f"rnn.rnn_lst.{i}.weight"], p[f"rnn.rnn_lst.{i}.weight_c"], p[f"rnn.rnn_lst.{i}.bias"]))
        i += 1
    h = sru_forward_torch(unf, layers).permute(1, 2, 0)  # N, 2d, L
    y = torch.nn.functional.conv_transpose1d(h, p["linear.weight"], p["linear.bias"])
    y = y.reshape(B, nF, C, nT).permute(0, 2, 3, 1) + res
    if dim == 4:
        y = y.permute(0, 1, 3, 2)
    return y


def dualpath_grads(x, p, dim, dout):
    """numpy in / numpy out: (out, dx, {name: dparam}) of sum(out * dout) by autograd in float64."""
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    out = dualpath_rnn_torch(xt, pt, dim)
    names = sorted(pt)
    grads = torch.autograd.grad(out, [xt] + [pt[k] for k in names], torch.tensor(dout, dtype=torch.float64))
    return out.detach().numpy(), grads[0].numpy(), {k: g.numpy() for k, g in zip(names, grads[1:])}


def conv_norm_act_torch(x, p, cfg):
    """ConvNormAct (reference conv_layers.py:65-129: pre_norm -> pre_act -> conv -> norm -> act) in float64 torch ops.
    cfg = (Cin, Cout, k, stride, depthwise, pre_norm, pre_act, norm, act, has_bias, is2d); p = dict with the state_dict names
    full_layer.{0.norm.weight/bias, 1.weight, 2.weight/bias, 3.norm.weight/bias, 4.weight}."""
    import torch.nn.functional as F
    cin, cout, k, stride, depthwise, pre_norm, pre_act, norm, act, has_bias, is2d = cfg

    def act_fn(v, kind, slope):
        if kind == 1:
            return torch.relu(v)
        if kind == 2:
            return F.prelu(v, slope)
        if kind == 3:
            return torch.sigmoid(v)
        return v
    if pre_norm:
        x = F.group_norm(x, 1, p["full_layer.0.norm.weight"], p["full_layer.0.norm.bias"], 1e-5)
    x = act_fn(x, pre_act, p.get("full_layer.1.weight"))
    conv = F.conv2d if is2d else F.conv1d
    pad = (k - 1) // 2 if stride > 1 else "same"
    x = conv(x, p["full_layer.2.weight"], p.get("full_layer.2.bias") if has_bias else None, stride=stride, padding=pad,
             groups=cin if depthwise else 1)
    if norm:
        x = F.group_norm(x, 1, p["full_layer.3.norm.weight"], p["full_layer.3.norm.bias"], 1e-5)
    return act_fn(x, act, p.get("full_layer.4.weight"))


def cna_grads(x, p, cfg, dout):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    out = conv_norm_act_torch(xt, pt, cfg)
    names = sorted(pt)
    grads = torch.autograd.grad(out, [xt] + [pt[k] for k in names], torch.tensor(dout, dtype=torch.float64))
    return out.detach().numpy(), grads[0].numpy(), {k: g.numpy() for k, g in zip(names, grads[1:])}


def mhsa2d_torch(x, p, n_head=4, masks=None):
    """rtfs_oracle.mhsa2d in torch (reference attention.py:149-189, dim 3).  x (B,C,T,F) float64; p = state_dict-named tensors.
    masks (optional): module name -> bool tensor "pre-activation >= 0".  PReLU's derivative jumps at 0, so a pre-activation within
    rounding error of 0 makes the gradient of the float64 oracle and of an fp32 implementation differ by O(1) on that element;
    with the implementation's own sign pattern given, prelu(z) = where(mask, z, slope*z) is evaluated on the same linear piece
    (the value changes by < 1e-5 * slope for the handful of elements concerned)."""
    import math
    import torch.nn.functional as F

    def can(v, pre):  # ConvActNorm: 1x1 conv -> PReLU -> LayerNormalization4D((C, F))
        y = F.conv2d(v, p[pre + ".conv.weight"], p[pre + ".conv.bias"])
        y = F.prelu(y, p[pre + ".act.weight"]) if masks is None else torch.where(masks[pre], y, p[pre + ".act.weight"] * y)
        mu = y.mean((1, 3), keepdim=True)
        var = ((y - mu) ** 2).mean((1, 3), keepdim=True)
        return (y - mu) / torch.sqrt(var + 1e-5) * p[pre + ".norm.gamma"] + p[pre + ".norm.beta"]
    B, C, T, Fq = x.shape
    outs = []
    for h in range(n_head):
        Q = can(x, f"Queries.{h}").permute(0, 2, 1, 3).reshape(B, T, -1)
        K = can(x, f"Keys.{h}").permute(0, 2, 1, 3).reshape(B, T, -1)
        V = can(x, f"Values.{h}").permute(0, 2, 1, 3)
        vs = V.shape
        a = torch.softmax(Q @ K.transpose(1, 2) / math.sqrt(Q.shape[-1]), 2)
        outs.append((a @ V.reshape(B, T, -1)).reshape(vs).permute(0, 2, 1, 3))
    return can(torch.cat(outs, 1), "attn_concat_proj") + x


def module_grads(fn, x, p, dout):
    """Generic: (out, dx, {name: dparam}) of sum(fn(x, p) * dout) in float64."""
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    out = fn(xt, pt)
    names = sorted(pt)
    grads = torch.autograd.grad(out, [xt] + [pt[k] for k in names], torch.tensor(dout, dtype=torch.float64), allow_unused=True)
    return out.detach().numpy(), grads[0].numpy(), {k: (None if g is None else g.numpy()) for k, g in zip(names, grads[1:])}


def _sub(p, prefix):
    q = prefix + "."
    return {k[len(q):]: v for k, v in p.items() if k.startswith(q)}


def _cna(x, p, *, k=1, stride=1, depthwise=False, norm=False, act=0, is2d=True):
    """ConvNormAct from its state_dict sub-dict (keys full_layer.N.*)."""
    cfg = (x.shape[1], p["full_layer.2.weight"].shape[0], k, stride, int(depthwise), 0, 0, int(norm), act, int("full_layer.2.bias" in p), int(is2d))
    return conv_norm_act_torch(x, p, cfg)


def injection_multi_sum_torch(local, glob, p):
    """rtfs_oracle.injection_multi_sum in torch (reference layers/fusion.py:54-69), 2-D, kernel 4, gLN."""
    import torch.nn.functional as F
    kw = dict(k=4, depthwise=True, norm=True)
    le = _cna(local, _sub(p, "local_embedding"), **kw)
    if local.shape[-2] * local.shape[-1] > glob.shape[-2] * glob.shape[-1]:
        ge = F.interpolate(_cna(glob, _sub(p, "global_embedding"), **kw), size=local.shape[-2:], mode="nearest")
        gate = F.interpolate(_cna(glob, _sub(p, "global_gate"), act=3, **kw), size=local.shape[-2:], mode="nearest")
    else:
        gi = F.interpolate(glob, size=local.shape[-2:], mode="nearest")
        ge, gate = _cna(gi, _sub(p, "global_embedding"), **kw), _cna(gi, _sub(p, "global_gate"), act=3, **kw)
    return le * gate + ge


def rtfs_block_torch(x, p):
    """rtfs_oracle.rtfs_block in torch (reference separators/tdanet.py:104-131; is2d, upsampling_depth 2, globalatt =
    DualPathRNN(dim 4), DualPathRNN(dim 3), MultiHeadSelfAttention2D)."""
    import torch.nn.functional as F
    residual = _cna(x, _sub(p, "gateway"), depthwise=True, act=2)
    x_enc = _cna(residual, _sub(p, "projection"))
    d0 = _cna(x_enc, _sub(p, "downsample_layers.0"), k=4, depthwise=True, norm=True)
    d1 = _cna(d0, _sub(p, "downsample_layers.1"), k=4, stride=2, depthwise=True, norm=True)
    g = F.adaptive_avg_pool2d(d0, d1.shape[-2:]) + d1
    dp = dualpath_lstm_torch if "globalatt.0.rnn.weight_ih_l0" in p else dualpath_rnn_torch  # yaml rnn_type LSTM / SRU
    g = dp(g, _sub(p, "globalatt.0"), 4)
    g = dp(g, _sub(p, "globalatt.1"), 3)
    g = mhsa2d_torch(g, _sub(p, "globalatt.2"))
    xf0 = injection_multi_sum_torch(d0, g, _sub(p, "fusion_layers.0"))
    xf1 = injection_multi_sum_torch(d1, g, _sub(p, "fusion_layers.1"))
    expanded = injection_multi_sum_torch(xf0, xf1, _sub(p, "concat_layers.0")) + d0
    return _cna(expanded, _sub(p, "residual_conv")) + residual


def stft_encoder_torch(wav, w):
    """STFTEncoder.forward (reference TDAVNet/encoder.py:161-175): torch.stft -> (B,2,T,F) -> Conv2d 3x3 'same', no bias."""
    import torch.nn.functional as F
    win = torch.hann_window(256, dtype=torch.float32).to(wav.dtype)  # the reference's buffer is float32 (encoder.py:159)
    spec = torch.stft(wav, n_fft=256, hop_length=128, window=win, return_complex=True)  # (B, F, T)
    x = torch.stack([spec.real, spec.imag], 1).transpose(2, 3).contiguous()
    return F.conv2d(x, w, padding=1)


def stft_decoder_torch(x, w, length):
    """STFTDecoder.forward (reference TDAVNet/decoder.py:110-132): ConvTranspose2d 256->2 3x3 pad 1 -> complex -> istft."""
    import torch.nn.functional as F
    y = F.conv_transpose2d(x, w, padding=1)  # (B, 2, T, F)
    spec = torch.complex(y[:, 0], y[:, 1]).transpose(1, 2)
    win = torch.hann_window(256, dtype=torch.float32).to(x.dtype)  # float32 buffer in the reference (decoder.py:108)
    return torch.istft(spec, n_fft=256, hop_length=128, window=win, length=length).unsqueeze(1)


def s3_torch(refined, a0, p):
    """MaskGenerator.forward + apply_masks (reference TDAVNet/mask_generator.py:67-99), RI_split, n_src 1."""
    import torch.nn.functional as F
    m = F.relu(F.conv2d(F.prelu(refined, p["mask_generator.0.weight"]), p["mask_generator.1.full_layer.2.weight"],
                        p["mask_generator.1.full_layer.2.bias"]))
    mr, mi, er, ei = m[:, :128], m[:, 128:], a0[:, :128], a0[:, 128:]
    return torch.cat([er * mr - ei * mi, er * mi + ei * mr], 1).unsqueeze(1)


def audio_chain_torch(wav, p):
    """encoder -> audio bottleneck -> S^3 (on the bottleneck output) -> decoder: the separator without its refinement module."""
    a0 = stft_encoder_torch(wav, p["encoder.conv.full_layer.2.weight"])
    a1 = conv_norm_act_torch(a0, _sub(p, "audio_bottleneck"), (256, 256, 1, 1, 0, 1, 1, 0, 0, 1, 1))
    s = s3_torch(a1, a0, _sub(p, "mask_generator"))
    return stft_decoder_torch(s[:, 0], p["decoder.decoder.weight"], wav.shape[-1])


def caf_torch(a, v, p, bn_train=False):
    """ATTNFusionCell.forward (reference layers/fusion.py:252-274), is2d, kernel_size 4, BatchNorm in eval mode (running stats).
    a (B,256,T,F), v (B,512,Tv); p = the cell's state_dict (running statistics included, treated as constants)."""
    import torch.nn.functional as F
    B, C, T, _ = a.shape

    def video_conv(pre):
        w = p[pre + ".full_layer.2.weight"]
        y = F.conv1d(v, w, p[pre + ".full_layer.2.bias"], groups=C)
        return F.group_norm(y, 1, p[pre + ".full_layer.3.norm.weight"], p[pre + ".full_layer.3.norm.bias"], 1e-5)

    def audio_conv(pre, relu):
        y = F.conv2d(a, p[pre + ".full_layer.2.weight"], None, groups=C)
        # bn_train: statistics of the batch (and torch updates the two running tensors in place, momentum 0.1, as nn.BatchNorm2d does)
        y = F.batch_norm(y, p[pre + ".full_layer.3.running_mean"].detach(), p[pre + ".full_layer.3.running_var"].detach(),
                         p[pre + ".full_layer.3.weight"], p[pre + ".full_layer.3.bias"], bn_train, 0.1, 1e-5)
        return torch.relu(y) if relu else y
    b_t = F.interpolate(video_conv("resize"), size=T, mode="nearest").unsqueeze(-1)
    k1 = audio_conv("key_embed", True) * b_t
    val = audio_conv("value_embed", False)
    att = video_conv("attention_embed").reshape(B, C, 4, -1).mean(2)
    att = F.interpolate(torch.softmax(att, -1), size=T, mode="nearest").unsqueeze(-1)
    return k1 + att * val


def avnet_torch(wav, video_vp, p, repeats, vp_trainable=False, bn_train=False):
    """AVNet.forward (reference tdavnet.py:86-97 + refinement_module.py:45-62) in float64 torch ops: encoder -> bottleneck -> block ->
    CAF -> (repeats-1) x block(x + a1) -> S^3 -> decoder.  video_vp: the VP block's output (a constant), or with vp_trainable the lip
    embedding the VP block (dropout off) is applied to; bn_train: BatchNorm layers on batch statistics."""
    if vp_trainable:
        video_vp = vp_block_torch(video_vp, _sub(p, "refinement_module.video_net.blocks"), bn_train=bn_train)
    a0 = stft_encoder_torch(wav, p["encoder.conv.full_layer.2.weight"])
    a1 = conv_norm_act_torch(a0, _sub(p, "audio_bottleneck"), (256, 256, 1, 1, 0, 1, 1, 0, 0, 1, 1))
    blk = _sub(p, "refinement_module.audio_net.blocks")
    x = rtfs_block_torch(a1, blk)
    x = caf_torch(x, video_vp, _sub(p, "refinement_module.crossmodal_fusion.fusion_module.audio_lstm"), bn_train=bn_train)
    for _ in range(repeats - 1):
        x = rtfs_block_torch(x + a1, blk)
    s = s3_torch(x, a0, _sub(p, "mask_generator"))
    return stft_decoder_torch(s[:, 0], p["decoder.decoder.weight"], wav.shape[-1])


def pit_loss_torch(est, tgt, kind="snr", zero_mean=True, take_log=True, eps=1e-8):
    """PITLossWrapper(PairwiseNegSDR(kind)) for n_src = 1 (src/losses/matrix.py:22-53): mean over the batch."""
    if zero_mean:
        est, tgt = est - est.mean(2, keepdim=True), tgt - tgt.mean(2, keepdim=True)
    if kind in ("sisdr", "sdsdr"):
        proj = (est * tgt).sum(2, keepdim=True) * tgt / ((tgt ** 2).sum(2, keepdim=True) + eps)
    else:
        proj = tgt
    noise = est - tgt if kind in ("sdsdr", "snr") else est - proj
    sdr = (proj ** 2).sum(2) / ((noise ** 2).sum(2) + eps)
    if take_log:
        sdr = 10 * torch.log10(sdr + eps)
    return (-sdr).mean()


def _cna1d(x, p, *, k=1, stride=1, depthwise=False, norm=None, act=0, bn_train=False):
    """1-D ConvNormAct from its state_dict sub-dict; norm in {None, "gLN", "BN"} (BN: running statistics, or the batch's if bn_train)."""
    import torch.nn.functional as F
    y = F.conv1d(x, p["full_layer.2.weight"], p.get("full_layer.2.bias"), stride=stride, padding=(k - 1) // 2,
                 groups=x.shape[1] if depthwise else 1)
    if norm == "gLN":
        y = F.group_norm(y, 1, p["full_layer.3.norm.weight"], p["full_layer.3.norm.bias"], 1e-5)
    elif norm == "BN":
        y = F.batch_norm(y, p["full_layer.3.running_mean"].detach(), p["full_layer.3.running_var"].detach(), p["full_layer.3.weight"],
                         p["full_layer.3.bias"], bn_train, 0.1, 1e-5)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = F.prelu(y, p["full_layer.4.weight"])
    elif act == 3:
        y = torch.sigmoid(y)
    return y


def mhsa_1d_torch(x, p, n_head=8, pmask=None):
    """rtfs_oracle.mhsa_1d in torch (reference attention.py:57-73), dropout / DropPath off; pmask: optional keep-mask (B*n_head, T, T)
    multiplying the attention weights (what nn.MultiheadAttention's dropout does in train mode)."""
    import math
    import torch.nn.functional as F
    res = x
    y = x.transpose(1, 2)
    C = y.shape[-1]
    y = F.layer_norm(y, (C,), p["norm1.weight"], p["norm1.bias"], 1e-5) + p["pos_enc.pe"][0, : y.shape[1]].detach()
    residual = y
    B, T, _ = y.shape
    hd = C // n_head
    qkv = y @ p["attention.in_proj_weight"].t() + p["attention.in_proj_bias"]
    q, k, v = [t.reshape(B, T, n_head, hd).transpose(1, 2) for t in qkv.split(C, -1)]
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    if pmask is not None:
        a = a * pmask.reshape(B, n_head, T, T)
    o = (a @ v).transpose(1, 2).reshape(B, T, C) @ p["attention.out_proj.weight"].t() + p["attention.out_proj.bias"]
    y = F.layer_norm(o + residual, (C,), p["norm2.weight"], p["norm2.bias"], 1e-5)
    return y.transpose(1, 2) + res


def vp_block_torch(v, p, depth=4, bn_train=False):
    """rtfs_oracle.vp_block in torch (reference separators/tdanet.py:104-131 with yaml video_params), dropout / DropPath off."""
    import torch.nn.functional as F
    residual = _cna1d(v, _sub(p, "gateway"), depthwise=True, act=2)
    x_enc = _cna1d(residual, _sub(p, "projection"))
    kw = dict(k=3, depthwise=True, norm="BN", bn_train=bn_train)
    downs = [_cna1d(x_enc, _sub(p, "downsample_layers.0"), **kw)]
    for i in range(1, depth):
        downs.append(_cna1d(downs[-1], _sub(p, f"downsample_layers.{i}"), stride=2, **kw))
    tgt = downs[-1].shape[-1]
    g = sum(F.adaptive_avg_pool1d(d, tgt) for d in downs)
    g = mhsa_1d_torch(g, _sub(p, "globalatt.0.MHSA"))
    f = _sub(p, "globalatt.0.FFN")
    y = _cna1d(g, _sub(f, "encoder"), norm="gLN")
    y = _cna1d(y, _sub(f, "refiner"), k=f["refiner.full_layer.2.weight"].shape[-1], depthwise=True, act=1)
    g = _cna1d(y, _sub(f, "decoder"), norm="gLN") + g

    def ims(loc, glo, name):
        q = _sub(p, name)
        le = _cna1d(loc, _sub(q, "local_embedding"), **kw)
        if loc.shape[-1] > glo.shape[-1]:
            ge = F.interpolate(_cna1d(glo, _sub(q, "global_embedding"), **kw), size=loc.shape[-1], mode="nearest")
            gate = F.interpolate(_cna1d(glo, _sub(q, "global_gate"), act=3, **kw), size=loc.shape[-1], mode="nearest")
        else:
            gi = F.interpolate(glo, size=loc.shape[-1], mode="nearest")
            ge, gate = _cna1d(gi, _sub(q, "global_embedding"), **kw), _cna1d(gi, _sub(q, "global_gate"), act=3, **kw)
        return le * gate + ge
    xf = [ims(downs[i], g, f"fusion_layers.{i}") for i in range(depth)]
    expanded = ims(xf[-2], xf[-1], f"concat_layers.{depth - 2}") + downs[-2]
    for i in range(depth - 3, -1, -1):
        expanded = ims(xf[i], expanded, f"concat_layers.{i}") + downs[i]
    return _cna1d(expanded, _sub(p, "residual_conv")) + residual


def dualpath_lstm_torch(x, p, dim, kernel_size=8):
    """DualPathRNN.forward with rnn_type LSTM or GRU (reference rnn_layers.py:116-122,136-162): the cell is stock
    torch.nn.LSTM / nn.GRU (512, 32, 4 layers, bidirectional; told apart by the 128 / 96 rows of weight_hh), so this function is the
    reference's own arithmetic - forward and backward."""
    if dim == 4:
        x = x.permute(0, 1, 3, 2)
    B, C, nT, nF = x.shape
    res = x
    mu = x.mean(1, keepdim=True)
    var = ((x - mu) ** 2).mean(1, keepdim=True)
    xn = (x - mu) / torch.sqrt(var + 1e-5) * p["norm.gamma"].reshape(1, C, 1, 1) + p["norm.beta"].reshape(1, C, 1, 1)
    seq = xn.permute(0, 3, 1, 2).reshape(B * nF, C, nT)
    Lr = nT - kernel_size + 1
    unf = torch.stack([seq[:, :, kk:kk + Lr] for kk in range(kernel_size)], 2).reshape(B * nF, C * kernel_size, Lr).permute(2, 0, 1)
    names = [k for k in p if k.startswith("rnn.")]
    flat = [p[f"rnn.{n}_l{l}{suf}"] for l in range(4) for suf in ("", "_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    assert len(flat) == len(names)
    if p["rnn.weight_hh_l0"].shape[0] == 128:
        h = torch._VF.lstm(unf, (unf.new_zeros(8, B * nF, 32), unf.new_zeros(8, B * nF, 32)), flat, True, 4, 0.0, False, True, False)[0]
    else:
        h = torch._VF.gru(unf, unf.new_zeros(8, B * nF, 32), flat, True, 4, 0.0, False, True, False)[0]
    y = torch.nn.functional.conv_transpose1d(h.permute(1, 2, 0), p["linear.weight"], p["linear.bias"])
    y = y.reshape(B, nF, C, nT).permute(0, 2, 3, 1) + res
    if dim == 4:
        y = y.permute(0, 1, 3, 2)
    return y
