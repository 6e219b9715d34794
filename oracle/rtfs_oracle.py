"""CPU oracle for the RTFS-Net separator forward pass.

TEST INFRASTRUCTURE ONLY.  This file is a numpy restatement of the reference's
``AVNet.forward`` (reference ``src/models/tdavnet.py:86-97``) and of every
module under it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``rtfs-net_amd/``) never does and fails loudly when its HIP library is absent.

Parity status
-------------
* Every function except :func:`sru_forward` is pinned against outputs of the
  reference itself, captured in this container by ``oracle/make_golden.py``
  and committed under ``tests/golden/`` (see ``tests/test_oracle_golden.py``).
* :func:`sru_forward` restates the third-party ``sru`` package (asappresearch /
  taolei87 ``sru``; requested un-pinned at ``setup/requirements.yaml:33``,
  alternative pin ``sru==2.6.0`` at ``:18``).  That package is not in
  ``/root/reference`` and not installable here, so the SRU cell is
  **parity unpinned**: it follows upstream's published v2 recurrence
  (``elementwise_recurrence_naive``), anchored on the reference call site
  ``src/models/layers/rnn_layers.py:99-105,150``.

All tensors are float32 numpy arrays in the reference's layouts
((B, C, T, F) with F fastest for spectrogram-shaped data).  Parameters come in
as a flat ``dict`` with the reference's ``state_dict`` key names.
"""
from __future__ import annotations

import math
import numpy as np

EPS = 1e-5  # reference src/models/layers/normalizations.py:5

f32 = np.float32


# --------------------------------------------------------------------------- helpers
def _sub(sd: dict, prefix: str) -> dict:
    """Sub-dictionary of ``sd`` under ``prefix.`` with the prefix stripped."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def prelu(x, a):
    """nn.PReLU() with one shared slope (reference activations.py:10-11)."""
    a = f32(np.asarray(a).reshape(-1)[0])
    return np.where(x >= 0, x, a * x).astype(f32)


def relu(x):
    return np.maximum(x, 0).astype(f32)


def gln(x, w, b, eps=EPS):
    """GroupNorm(1, C): per-sample stats over every non-batch axis, per-channel
    affine (reference normalizations.py:8-17)."""
    B = x.shape[0]
    xf = x.reshape(B, -1).astype(np.float64)
    mu = xf.mean(1)
    var = xf.var(1)
    shp = (B,) + (1,) * (x.ndim - 1)
    cshp = (1, -1) + (1,) * (x.ndim - 2)
    y = (x.astype(np.float64) - mu.reshape(shp)) / np.sqrt(var.reshape(shp) + eps)
    return (y * w.reshape(cshp) + b.reshape(cshp)).astype(f32)


def batchnorm_eval(x, w, b, rm, rv, eps=EPS):
    """nn.BatchNorm{1,2}d in eval mode (running statistics)."""
    cshp = (1, -1) + (1,) * (x.ndim - 2)
    scale = (w.astype(np.float64) / np.sqrt(rv.astype(np.float64) + eps))
    shift = b.astype(np.float64) - rm.astype(np.float64) * scale
    return (x * scale.reshape(cshp) + shift.reshape(cshp)).astype(f32)


def ln4d(x, gamma, beta, eps=EPS):
    """LayerNormalization4D (reference normalizations.py:20-41): stats over
    dim 1 when gamma is (1,C,1,1), over dims (1,3) when gamma is (1,C,1,F)."""
    axes = (1, 3) if gamma.shape[-1] > 1 else (1,)
    xd = x.astype(np.float64)
    mu = xd.mean(axis=axes, keepdims=True)
    var = xd.var(axis=axes, keepdims=True)
    return (((xd - mu) / np.sqrt(var + eps)) * gamma + beta).astype(f32)


def layernorm_last(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last axis."""
    xd = x.astype(np.float64)
    mu = xd.mean(-1, keepdims=True)
    var = xd.var(-1, keepdims=True)
    return ((xd - mu) / np.sqrt(var + eps) * w + b).astype(f32)


def pointwise(x, w, b=None):
    """1x1 convolution, any number of trailing spatial axes.
    x (B,Cin,...), w (Cout,Cin,1[,1])."""
    w2 = w.reshape(w.shape[0], w.shape[1])
    B, C = x.shape[:2]
    sp = x.shape[2:]
    y = np.matmul(w2[None], x.reshape(B, C, -1))
    if b is not None:
        y = y + b.reshape(1, -1, 1)
    return y.reshape((B, w2.shape[0]) + sp).astype(f32)


def grouped_pointwise_1d(x, w, b, groups):
    """Conv1d kernel 1 with groups: x (B,Cin,T), w (Cout,Cin/groups,1)."""
    B, Cin, T = x.shape
    Cout = w.shape[0]
    ipg, opg = Cin // groups, Cout // groups
    xg = x.reshape(B, groups, ipg, T)
    wg = w.reshape(groups, opg, ipg)
    y = np.einsum("goi,bgit->bgot", wg, xg).reshape(B, Cout, T)
    if b is not None:
        y = y + b.reshape(1, -1, 1)
    return y.astype(f32)


def _same_pads(k):
    """torch ``padding='same'`` (stride 1): total k-1, the extra one on the high
    side (k=4 -> low 1, high 2)."""
    tot = k - 1
    lo = tot // 2
    return lo, tot - lo


def dwconv2d(x, w, b=None, stride=1, pad=None):
    """Depthwise Conv2d (cross-correlation).  x (B,C,H,W), w (C,1,kh,kw).
    ``pad=None`` means 'same' (stride 1)."""
    B, C, H, W = x.shape
    kh, kw = w.shape[-2:]
    if pad is None:
        (pt, pb), (pl, pr) = _same_pads(kh), _same_pads(kw)
    else:
        pt = pb = pl = pr = pad
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr)))
    Ho = (H + pt + pb - kh) // stride + 1
    Wo = (W + pl + pr - kw) // stride + 1
    y = np.zeros((B, C, Ho, Wo), np.float64)
    for i in range(kh):
        for j in range(kw):
            y += xp[:, :, i:i + stride * (Ho - 1) + 1:stride, j:j + stride * (Wo - 1) + 1:stride] * \
                w[:, 0, i, j].reshape(1, C, 1, 1).astype(np.float64)
    if b is not None:
        y += b.reshape(1, C, 1, 1)
    return y.astype(f32)


def dwconv1d(x, w, b=None, stride=1, pad=None):
    """Depthwise Conv1d.  x (B,C,T), w (C,1,k)."""
    B, C, T = x.shape
    k = w.shape[-1]
    if pad is None:
        pl, pr = _same_pads(k)
    else:
        pl = pr = pad
    xp = np.pad(x, ((0, 0), (0, 0), (pl, pr)))
    To = (T + pl + pr - k) // stride + 1
    y = np.zeros((B, C, To), np.float64)
    for j in range(k):
        y += xp[:, :, j:j + stride * (To - 1) + 1:stride] * w[:, 0, j].reshape(1, C, 1).astype(np.float64)
    if b is not None:
        y += b.reshape(1, C, 1)
    return y.astype(f32)


def conv2d_dense_same(x, w):
    """Dense Conv2d, stride 1, odd kernel, 'same', no bias.  x (B,Ci,H,W), w (Co,Ci,kh,kw)."""
    B, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    (pt, pb), (pl, pr) = _same_pads(kh), _same_pads(kw)
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr))).astype(np.float64)
    y = np.zeros((B, Co, H, W), np.float64)
    for i in range(kh):
        for j in range(kw):
            y += np.einsum("oc,bchw->bohw", w[:, :, i, j].astype(np.float64), xp[:, :, i:i + H, j:j + W])
    return y.astype(f32)


def conv_transpose2d_s1(x, w, pad):
    """ConvTranspose2d stride 1, no bias.  x (B,Ci,H,W), w (Ci,Co,kh,kw)."""
    B, Ci, H, W = x.shape
    _, Co, kh, kw = w.shape
    full = np.zeros((B, Co, H + kh - 1, W + kw - 1), np.float64)
    for i in range(kh):
        for j in range(kw):
            full[:, :, i:i + H, j:j + W] += np.einsum("co,bchw->bohw", w[:, :, i, j].astype(np.float64), x.astype(np.float64))
    return full[:, :, pad:pad + H + kh - 1 - 2 * pad, pad:pad + W + kw - 1 - 2 * pad].astype(f32)


def adaptive_avg_pool_axis(x, out, axis):
    """F.adaptive_avg_pool along one axis: window i = [floor(i*n/out), ceil((i+1)*n/out))."""
    n = x.shape[axis]
    if n == out:
        return x
    xs = np.moveaxis(x, axis, -1).astype(np.float64)
    y = np.empty(xs.shape[:-1] + (out,), np.float64)
    for i in range(out):
        s = (i * n) // out
        e = -((-(i + 1) * n) // out)
        y[..., i] = xs[..., s:e].mean(-1)
    return np.moveaxis(y, -1, axis).astype(f32)


def nearest_index(n_in, n_out):
    """Legacy ``mode='nearest'`` source index: floor(dst * in / out)."""
    return np.minimum((np.arange(n_out) * n_in) // n_out, n_in - 1)


def nearest_up(x, size):
    """F.interpolate(mode='nearest') over the trailing len(size) axes."""
    for ax, n_out in zip(range(x.ndim - len(size), x.ndim), size):
        n_in = x.shape[ax]
        if n_in != n_out:
            x = np.take(x, nearest_index(n_in, n_out), axis=ax)
    return x


# --------------------------------------------------------------------------- STFT / iSTFT
def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(f32)


def stft_ri(x, win=256, hop=128):
    """torch.stft(n_fft=win, hop, hann, center=True, reflect, onesided) then
    stack(re, im).transpose(2,3): (B,L) -> (B,2,T,F) (reference encoder.py:161-172)."""
    B, L = x.shape
    w = hann_periodic(win).astype(np.float64)
    xp = np.pad(x.astype(np.float64), ((0, 0), (win // 2, win // 2)), mode="reflect")
    T = 1 + L // hop
    idx = np.arange(T)[:, None] * hop + np.arange(win)[None, :]
    frames = xp[:, idx] * w  # B,T,win
    spec = np.fft.rfft(frames, axis=-1)  # B,T,F
    return np.stack([spec.real, spec.imag], 1).astype(f32)


def istft_ri(re, im, length, win=256, hop=128):
    """torch.istft(n_fft=win, hop, hann, center=True, length): re/im (B,T,F) -> (B,length)
    (reference decoder.py:119-128)."""
    B, T, F = re.shape
    w = hann_periodic(win).astype(np.float64)
    spec = re.astype(np.float64) + 1j * im.astype(np.float64)
    frames = np.fft.irfft(spec, n=win, axis=-1) * w  # B,T,win
    n = win + hop * (T - 1)
    y = np.zeros((B, n), np.float64)
    env = np.zeros(n, np.float64)
    for t in range(T):
        y[:, t * hop:t * hop + win] += frames[:, t]
        env[t * hop:t * hop + win] += w * w
    y = y[:, win // 2:]
    env = env[win // 2:]
    out = np.zeros((B, length), np.float64)
    m = min(length, y.shape[1])
    out[:, :m] = y[:, :m] / env[:m]
    return out.astype(f32)


# --------------------------------------------------------------------------- ConvNormAct interpreter
def conv_norm_act(x, p: dict, *, groups=1, stride=1, is2d=True, pre_act=None, act=None, norm=None, pre_norm=None, pad=None):
    """Interprets one reference ``ConvNormAct`` (conv_layers.py:65-129).  ``p`` holds the
    module's parameters under ``full_layer.N...`` names: 0 pre_norm, 2 conv, 3 norm, 4 act."""
    if pre_norm == "gLN":
        x = gln(x, p["full_layer.0.norm.weight"], p["full_layer.0.norm.bias"])
    if pre_act == "ReLU":
        x = relu(x)
    w = p["full_layer.2.weight"]
    b = p.get("full_layer.2.bias")
    k = w.shape[-1]
    if k == 1 and groups == 1:
        x = pointwise(x, w, b)
    elif k == 1 and is2d and groups == x.shape[1]:
        x = (x * w.reshape(1, -1, 1, 1) + (b.reshape(1, -1, 1, 1) if b is not None else 0)).astype(f32)
    elif k == 1 and (not is2d) and groups == x.shape[1] and w.shape[0] == x.shape[1]:
        x = (x * w.reshape(1, -1, 1) + (b.reshape(1, -1, 1) if b is not None else 0)).astype(f32)
    elif k == 1 and not is2d:
        x = grouped_pointwise_1d(x, w, b, groups)
    elif is2d:
        assert groups == x.shape[1]
        x = dwconv2d(x, w, b, stride=stride, pad=pad if stride > 1 else None)
    else:
        assert groups == x.shape[1]
        x = dwconv1d(x, w, b, stride=stride, pad=pad if stride > 1 else None)
    if norm == "gLN":
        x = gln(x, p["full_layer.3.norm.weight"], p["full_layer.3.norm.bias"])
    elif norm in ("BatchNorm1d", "BatchNorm2d"):
        x = batchnorm_eval(x, p["full_layer.3.weight"], p["full_layer.3.bias"],
                           p["full_layer.3.running_mean"], p["full_layer.3.running_var"])
    if act == "PReLU":
        x = prelu(x, p["full_layer.4.weight"])
    elif act == "ReLU":
        x = relu(x)
    elif act == "Sigmoid":
        x = sigmoid(x)
    return x


# --------------------------------------------------------------------------- SRU (third party; parity unpinned)
def sru_forward(x, layers):
    """Stacked bidirectional SRU, upstream v2 semantics.  x (L,N,Din) -> h (L,N,2d).

    ``layers`` = list of (weight (Din,2d*k), weight_c (4d,), bias (4d,)).  k = 4 when
    Din != 2d (projected highway term) else 3.  Per (n, dir, j):
        f = sigmoid(u1 + v_f*c + b_f);  r = sigmoid(u2 + v_r*c + b_r)
        c' = u0 + (c - u0)*f;           h = x' + (c' - x')*r
    with both gates using the *previous* c, c_0 = 0, direction 1 walking t = L-1..0,
    identity activation, no rescale (call site rnn_layers.py:99-105 passes only
    input_size/hidden_size/num_layers/bidirectional)."""
    L, N, _ = x.shape
    h_in = x.astype(f32)
    for (W, wc, bias) in layers:
        d2 = wc.shape[0] // 2  # dirs*d
        d = d2 // 2
        k = W.shape[1] // d2
        U = np.matmul(h_in.reshape(L * N, -1), W).reshape(L, N, 2, d, k)
        vf, vr = wc.reshape(2, 2, d)
        bf, br = bias.reshape(2, 2, d)
        xprime = U[..., 3] if k == 4 else h_in.reshape(L, N, 2, d)
        h = np.empty((L, N, 2, d), f32)
        for di in range(2):
            c = np.zeros((N, d), f32)
            order = range(L) if di == 0 else range(L - 1, -1, -1)
            for t in order:
                u0 = U[t, :, di, :, 0]
                f = sigmoid(U[t, :, di, :, 1] + c * vf[di] + bf[di])
                r = sigmoid(U[t, :, di, :, 2] + c * vr[di] + br[di])
                c = (u0 + (c - u0) * f).astype(f32)
                xp = xprime[t, :, di, :]
                h[t, :, di, :] = xp + (c - xp) * r
        h_in = h.reshape(L, N, d2)
    return h_in


def lstm_forward(x, p: dict, num_layers=4, prefix="rnn."):
    """Stacked bidirectional nn.LSTM (the reference's other DualPathRNN cell: rnn_layers.py:116-122 with
    rnn_type LSTM; stock torch, so this function IS pinned by reference vectors).  x (L,N,Din) -> (L,N,2d).
    Gate order in the weights: i, f, g, o;  c' = f*c + i*g;  h' = o*tanh(c')."""
    L, N, _ = x.shape
    h_in = x.astype(np.float64)
    for layer in range(num_layers):
        outs = []
        for di, suf in enumerate(("", "_reverse")):
            Wih = p[f"{prefix}weight_ih_l{layer}{suf}"].astype(np.float64)
            Whh = p[f"{prefix}weight_hh_l{layer}{suf}"].astype(np.float64)
            b = (p[f"{prefix}bias_ih_l{layer}{suf}"] + p[f"{prefix}bias_hh_l{layer}{suf}"]).astype(np.float64)
            d = Whh.shape[1]
            U = np.matmul(h_in.reshape(L * N, -1), Wih.T).reshape(L, N, 4 * d) + b
            h = np.zeros((N, d))
            c = np.zeros((N, d))
            out = np.empty((L, N, d))
            for t in (range(L) if di == 0 else range(L - 1, -1, -1)):
                g = U[t] + h @ Whh.T
                i_, f_, g_, o_ = g[:, :d], g[:, d:2 * d], g[:, 2 * d:3 * d], g[:, 3 * d:]
                c = 1.0 / (1.0 + np.exp(-f_)) * c + 1.0 / (1.0 + np.exp(-i_)) * np.tanh(g_)
                h = 1.0 / (1.0 + np.exp(-o_)) * np.tanh(c)
                out[t] = h
            outs.append(out)
        h_in = np.concatenate(outs, -1)
    return h_in.astype(f32)


def _sru_layers(p: dict, prefix="rnn.rnn_lst"):
    out = []
    i = 0
    while f"{prefix}.{i}.weight" in p:
        out.append((p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.weight_c"], p[f"{prefix}.{i}.bias"]))
        i += 1
    return out


# --------------------------------------------------------------------------- dual-path RNN
def dualpath_rnn(x, p: dict, dim: int, kernel_size=8):
    """DualPathRNN.forward (reference rnn_layers.py:136-162), stride 1, SRU cell.
    x (B,C,T,F); dim=4 sweeps along F, dim=3 along T."""
    if dim == 4:
        x = x.transpose(0, 1, 3, 2)
    B, C, oT, oF = x.shape
    # rnn_layers.py:141-143: new_T = ceil((T-k)/stride)*stride + k == T at stride 1, so the
    # zero-pad is a no-op; a sweep axis shorter than k makes nn.Unfold raise in the reference.
    if oT < kernel_size:
        raise ValueError(f"sweep axis {oT} shorter than kernel_size {kernel_size}")
    nT, nF = oT, oF
    res = x
    xn = ln4d(x, p["norm.gamma"], p["norm.beta"])
    seq = xn.transpose(0, 3, 1, 2).reshape(B * nF, C, nT)  # (N, C, T)
    Lr = nT - kernel_size + 1
    # nn.Unfold((k,1)): feature index c*k + kk, value seq[:, c, l+kk]
    unf = np.stack([seq[:, :, kk:kk + Lr] for kk in range(kernel_size)], 2)  # N,C,k,L
    unf = unf.reshape(B * nF, C * kernel_size, Lr).transpose(2, 0, 1)  # L,N,C*k
    if "rnn.weight_ih_l0" in p:  # rnn_type LSTM
        h = lstm_forward(np.ascontiguousarray(unf), p)
    else:
        h = sru_forward(np.ascontiguousarray(unf), _sru_layers(p))  # L,N,2d
    hh = h.transpose(1, 2, 0)  # N,2d,L
    Wt = p["linear.weight"]  # (Cin=2d, Cout=C, k)
    y = np.zeros((B * nF, C, nT), np.float64)
    for kk in range(kernel_size):
        y[:, :, kk:kk + Lr] += np.einsum("io,nil->nol", Wt[:, :, kk].astype(np.float64), hh.astype(np.float64))
    y += p["linear.bias"].reshape(1, -1, 1)
    y = y.reshape(B, nF, C, nT).transpose(0, 2, 3, 1).astype(f32)
    y = (y + res)[:, :, :oT, :oF]
    if dim == 4:
        y = y.transpose(0, 1, 3, 2)
    return np.ascontiguousarray(y)


# --------------------------------------------------------------------------- TF self-attention
def _conv_act_norm(x, p):
    """ConvActNorm (reference conv_layers.py:142-215): 1x1 conv -> PReLU -> LN4D((C,F))."""
    y = pointwise(x, p["conv.weight"], p["conv.bias"])
    y = prelu(y, p["act.weight"])
    return ln4d(y, p["norm.gamma"], p["norm.beta"])


def mhsa2d(x, p: dict, n_head=4):
    """MultiHeadSelfAttention2D.forward (reference attention.py:149-189), dim=3."""
    B, C, T, Fq = x.shape
    outs = []
    for h in range(n_head):
        Q = _conv_act_norm(x, _sub(p, f"Queries.{h}")).transpose(0, 2, 1, 3).reshape(B, T, -1)
        K = _conv_act_norm(x, _sub(p, f"Keys.{h}")).transpose(0, 2, 1, 3).reshape(B, T, -1)
        V = _conv_act_norm(x, _sub(p, f"Values.{h}")).transpose(0, 2, 1, 3)  # B,T,Cv,F
        vshape = V.shape
        V = V.reshape(B, T, -1)
        s = np.matmul(Q.astype(np.float64), K.astype(np.float64).transpose(0, 2, 1)) / math.sqrt(Q.shape[-1])
        s = s - s.max(-1, keepdims=True)
        a = np.exp(s)
        a /= a.sum(-1, keepdims=True)
        o = np.matmul(a, V.astype(np.float64)).reshape(vshape).transpose(0, 2, 1, 3)  # B,Cv,T,F
        outs.append(o.astype(f32))
    y = np.concatenate(outs, 1)  # channel = head*Cv + c
    y = _conv_act_norm(y, _sub(p, "attn_concat_proj"))
    return (y + x).astype(f32)


# --------------------------------------------------------------------------- TFAR (InjectionMultiSum)
def injection_multi_sum(local, glob, p: dict, *, is2d=True, norm="gLN"):
    """InjectionMultiSum.forward (reference layers/fusion.py:54-69)."""
    nsp = 2 if is2d else 1
    new_shape = local.shape[-nsp:]
    old_shape = glob.shape[-nsp:]
    C = local.shape[1]
    kw = dict(groups=C, is2d=is2d, norm=norm)
    le = conv_norm_act(local, _sub(p, "local_embedding"), **kw)
    if np.prod(new_shape) > np.prod(old_shape):
        ge = nearest_up(conv_norm_act(glob, _sub(p, "global_embedding"), **kw), new_shape)
        gate = nearest_up(conv_norm_act(glob, _sub(p, "global_gate"), act="Sigmoid", **kw), new_shape)
    else:
        gi = nearest_up(glob, new_shape)
        ge = conv_norm_act(gi, _sub(p, "global_embedding"), **kw)
        gate = conv_norm_act(gi, _sub(p, "global_gate"), act="Sigmoid", **kw)
    return (le * gate + ge).astype(f32)


# --------------------------------------------------------------------------- RTFS block (audio, 2-D)
def rtfs_block(x, p: dict, *, return_internals=False):
    """TDANetBlock.forward with is2d=True, upsampling_depth=2 (reference
    separators/tdanet.py:104-131) and globalatt = [DualPathRNN(dim 4), DualPathRNN(dim 3),
    MultiHeadSelfAttention2D] (yaml audio_params.layers)."""
    C = x.shape[1]
    residual = conv_norm_act(x, _sub(p, "gateway"), groups=C, act="PReLU")
    x_enc = conv_norm_act(residual, _sub(p, "projection"))
    H = x_enc.shape[1]
    d0 = conv_norm_act(x_enc, _sub(p, "downsample_layers.0"), groups=H, norm="gLN")
    d1 = conv_norm_act(d0, _sub(p, "downsample_layers.1"), groups=H, stride=2, pad=1, norm="gLN")
    tgt = d1.shape[-2:]
    g = adaptive_avg_pool_axis(adaptive_avg_pool_axis(d0, tgt[0], 2), tgt[1], 3) + d1
    g_in = g
    g = dualpath_rnn(g, _sub(p, "globalatt.0"), dim=4)
    g_f = g
    g = dualpath_rnn(g, _sub(p, "globalatt.1"), dim=3)
    g_t = g
    g = mhsa2d(g, _sub(p, "globalatt.2"))
    xf0 = injection_multi_sum(d0, g, _sub(p, "fusion_layers.0"))
    xf1 = injection_multi_sum(d1, g, _sub(p, "fusion_layers.1"))
    expanded = injection_multi_sum(xf0, xf1, _sub(p, "concat_layers.0")) + d0
    out = conv_norm_act(expanded, _sub(p, "residual_conv")) + residual
    if return_internals:
        return out.astype(f32), dict(residual=residual, x_enc=x_enc, d0=d0, d1=d1, g_in=g_in, g_f=g_f, g_t=g_t,
                                     g_att=g, xf0=xf0, xf1=xf1, expanded=expanded)
    return out.astype(f32)


# --------------------------------------------------------------------------- VP block (video, 1-D)
def positional_encoding(n_pos, channels, max_len=10000):
    """PositionalEncoding buffer (reference attention.py:9-25), float32 arithmetic."""
    pos = np.arange(n_pos, dtype=f32)[:, None]
    div = np.exp(np.arange(0, channels, 2, dtype=f32) * f32(-(math.log(float(max_len)) / channels))).astype(f32)
    pe = np.zeros((n_pos, channels), f32)
    pe[:, 0::2] = np.sin((pos * div).astype(f32))
    pe[:, 1::2] = np.cos((pos * div).astype(f32))
    return pe


def mhsa_1d(x, p: dict, n_head=8):
    """MultiHeadSelfAttention.forward, batch_first (reference attention.py:57-73); eval
    mode so dropout / DropPath are identity.  x (B,C,T)."""
    res = x
    y = x.transpose(0, 2, 1)  # B,T,C
    y = layernorm_last(y, p["norm1.weight"], p["norm1.bias"])
    pe = p["pos_enc.pe"][0, : y.shape[1]] if "pos_enc.pe" in p else positional_encoding(y.shape[1], y.shape[2])
    y = (y + pe).astype(f32)
    residual = y
    B, T, C = y.shape
    hd = C // n_head
    Wi, bi = p["attention.in_proj_weight"], p["attention.in_proj_bias"]
    qkv = np.matmul(y.astype(np.float64), Wi.T.astype(np.float64)) + bi
    q, k, v = [t.reshape(B, T, n_head, hd).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, -1)]
    s = np.matmul(q, k.transpose(0, 1, 3, 2)) / math.sqrt(hd)
    s = s - s.max(-1, keepdims=True)
    a = np.exp(s)
    a /= a.sum(-1, keepdims=True)
    o = np.matmul(a, v).transpose(0, 2, 1, 3).reshape(B, T, C)
    o = np.matmul(o, p["attention.out_proj.weight"].T.astype(np.float64)) + p["attention.out_proj.bias"]
    y = (o + residual).astype(f32)
    y = layernorm_last(y, p["norm2.weight"], p["norm2.bias"])
    return (y.transpose(0, 2, 1) + res).astype(f32)


def ffn_1d(x, p: dict):
    """FeedForwardNetwork.forward (reference conv_layers.py:252-259), 1-D, eval."""
    y = conv_norm_act(x, _sub(p, "encoder"), is2d=False, norm="gLN")
    y = conv_norm_act(y, _sub(p, "refiner"), groups=y.shape[1], is2d=False, act="ReLU")
    y = conv_norm_act(y, _sub(p, "decoder"), is2d=False, norm="gLN")
    return (y + x).astype(f32)


def vp_block(v, p: dict, depth=4):
    """Video TDANetBlock: 1-D, upsampling_depth 4, k=3, BatchNorm1d, GlobalAttention
    (reference separators/tdanet.py:104-131 with yaml video_params)."""
    C = v.shape[1]
    residual = conv_norm_act(v, _sub(p, "gateway"), groups=C, is2d=False, act="PReLU")
    x_enc = conv_norm_act(residual, _sub(p, "projection"), is2d=False)
    H = x_enc.shape[1]
    downs = [conv_norm_act(x_enc, _sub(p, "downsample_layers.0"), groups=H, is2d=False, norm="BatchNorm1d")]
    for i in range(1, depth):
        downs.append(conv_norm_act(downs[-1], _sub(p, f"downsample_layers.{i}"), groups=H, is2d=False,
                                   stride=2, pad=1, norm="BatchNorm1d"))
    tgt = downs[-1].shape[-1]
    g = sum(adaptive_avg_pool_axis(d, tgt, 2) for d in downs)
    g = mhsa_1d(g, _sub(p, "globalatt.0.MHSA"))
    g = ffn_1d(g, _sub(p, "globalatt.0.FFN"))
    ims = lambda a, b, name: injection_multi_sum(a, b, _sub(p, name), is2d=False, norm="BatchNorm1d")
    xf = [ims(downs[i], g, f"fusion_layers.{i}") for i in range(depth)]
    expanded = ims(xf[-2], xf[-1], f"concat_layers.{depth - 2}") + downs[-2]
    for i in range(depth - 3, -1, -1):
        expanded = ims(xf[i], expanded, f"concat_layers.{i}") + downs[i]
    out = conv_norm_act(expanded, _sub(p, "residual_conv"), is2d=False) + residual
    return out.astype(f32)


# --------------------------------------------------------------------------- CAF
def caf_video_terms(video, p: dict, in_chan_a=256, kernel_size=4):
    """Video-side half of ATTNFusionCell.forward (reference layers/fusion.py:255,261-264):
    returns (resize(video) (B,Ca,Tv), softmax_Tv(mean_k attention_embed(video)) (B,Ca,Tv))."""
    B = video.shape[0]
    r = conv_norm_act(video, _sub(p, "resize"), groups=in_chan_a, is2d=False, norm="gLN")
    att = conv_norm_act(video, _sub(p, "attention_embed"), groups=in_chan_a, is2d=False, norm="gLN")
    att = att.reshape(B, in_chan_a, kernel_size, -1).astype(np.float64).mean(2)
    att = att - att.max(-1, keepdims=True)
    e = np.exp(att)
    att = (e / e.sum(-1, keepdims=True)).astype(f32)
    return r, att


def caf(audio, video, p: dict, kernel_size=4):
    """ATTNFusion.forward with video_fusion=False -> ATTNFusionCell (reference
    TDAVNet/fusion.py:204-212, layers/fusion.py:252-274).  audio (B,Ca,T,F), video (B,Cv,Tv)."""
    Ca, T = audio.shape[1], audio.shape[2]
    r, att = caf_video_terms(video, p, Ca, kernel_size)
    r_up = nearest_up(r, (T,))[..., None]
    att_up = nearest_up(att, (T,))[..., None]
    key = conv_norm_act(audio, _sub(p, "key_embed"), groups=Ca, norm="BatchNorm2d", act="ReLU")
    val = conv_norm_act(audio, _sub(p, "value_embed"), groups=Ca, norm="BatchNorm2d")
    return (key * r_up + att_up * val).astype(f32)


# --------------------------------------------------------------------------- S3 mask head
def s3_mask(refined, a0, p: dict):
    """MaskGenerator.forward with RI_split, n_src=1 (reference mask_generator.py:67-99).
    Returns (B, 1, C, T, F)."""
    m = prelu(refined, p["mask_generator.0.weight"])
    m = conv_norm_act(m, _sub(p, "mask_generator.1"), act="ReLU")
    C = a0.shape[1]
    h = C // 2
    mr, mi = m[:, :h], m[:, h:]
    er, ei = a0[:, :h], a0[:, h:]
    out = np.concatenate([er * mr - ei * mi, er * mi + ei * mr], 1)
    return out[:, None].astype(f32)


# --------------------------------------------------------------------------- encoder / decoder
def stft_encoder(wav, p: dict, win=256, hop=128):
    """STFTEncoder.forward (reference encoder.py:161-175): returns (a0, spec)."""
    if wav.ndim == 1:
        wav = wav[None]
    elif wav.ndim == 3:
        wav = wav.reshape(wav.shape[0], -1)
    spec = stft_ri(wav, win, hop)
    return conv2d_dense_same(spec, p["conv.full_layer.2.weight"]), spec


def stft_decoder(x, p: dict, length, win=256, hop=128):
    """STFTDecoder.forward (reference decoder.py:110-132).  x (B,n_src,C,T,F) -> (B,n_src,L)."""
    B, S = x.shape[:2]
    y = conv_transpose2d_s1(x.reshape((B * S,) + x.shape[2:]), p["decoder.weight"], pad=1)
    out = istft_ri(y[:, 0], y[:, 1], length, win, hop)
    return out.reshape(B, S, length)


# --------------------------------------------------------------------------- whole model
def refinement(a1, video, sd: dict, repeats: int):
    """RefinementModule.forward (reference refinement_module.py:45-62) with
    fusion_repeats = 1 and shared block weights."""
    pa = _sub(sd, "refinement_module.audio_net.blocks")
    pv = _sub(sd, "refinement_module.video_net.blocks")
    pf = _sub(sd, "refinement_module.crossmodal_fusion.fusion_module.audio_lstm")
    audio = rtfs_block(a1, pa)
    v = vp_block(video, pv)
    audio = caf(audio, v, pf)
    for _ in range(repeats - 1):
        audio = rtfs_block(audio + a1, pa)
    return audio


def avnet_forward(wav, mouth_emb, sd: dict, repeats=4, return_internals=False):
    """AVNet.forward (reference tdavnet.py:86-97).  wav (B,L), mouth_emb (B,512,Tv) -> (B,1,L)."""
    wav = np.asarray(wav, f32)
    if wav.ndim == 1:
        wav = wav[None]
    L = wav.shape[-1]
    a0, _ = stft_encoder(wav, _sub(sd, "encoder"))
    a1 = conv_norm_act(a0, _sub(sd, "audio_bottleneck"), pre_norm="gLN", pre_act="ReLU")
    refined = refinement(a1, np.asarray(mouth_emb, f32), sd, repeats)
    sep = s3_mask(refined, a0, _sub(sd, "mask_generator"))
    out = stft_decoder(sep, _sub(sd, "decoder"), L)
    if return_internals:
        return out, dict(a0=a0, a1=a1, refined=refined, sep=sep)
    return out
