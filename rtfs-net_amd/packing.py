"""Parameter packs: the flat float32 device buffers the C ABI reads (layout contract with
``csrc/api.hip``; every tensor padded to a multiple of 64 floats, 1x1 weights stored [cin][cout]).

Each ``pack_*`` takes a mapping ``name -> tensor`` with the reference's ``state_dict`` key names
*relative to the module it packs* (e.g. ``full_layer.2.weight``), so the same functions serve live
``nn.Module`` parameters and loaded checkpoints.
"""
from __future__ import annotations

import torch

ALIGN = 64


def _sub(sd, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


_ZEROS = {}


def zeros_view(n, device):
    """n float32 zeros on ``device`` as a slice of one shared constant (read-only by contract: it only ever feeds torch.cat) - a fresh
    torch.zeros per absent parameter / padding run cost ~400 one-workgroup fill kernels per training step."""
    z = _ZEROS.get(device)
    if z is None or z.numel() < n:
        z = _ZEROS[device] = torch.zeros(max(n, 4096), device=device, dtype=torch.float32)
    return z[:n]


def _cat(tensors):
    out = []
    for t in tensors:
        t = t.detach().to(torch.float32).reshape(-1)
        out.append(t)
        pad = (-t.numel()) % ALIGN
        if pad:
            out.append(zeros_view(pad, t.device))
    return torch.cat(out).contiguous()


LOG2E = 1.4426950408889634
W16_SCALE = 256.0  # weights are stored pre-scaled by 2^8 so their f16 low parts stay normal; kernels undo it


def split16_image(w, kc=32):
    """(cout, cin) f32 weight -> the f16x3 kernels' staged image [cin/kc][hi|lo][cout][kc] halfs, returned bit-cast
    to float32 (same element count as w).  w*256 = hi + lo with hi, lo exactly representable in f16."""
    cout, cin = w.shape
    ws = w.detach().to(torch.float32) * W16_SCALE
    hi = ws.to(torch.float16)
    lo = (ws - hi.to(torch.float32)).to(torch.float16)
    img = torch.stack([hi.reshape(cout, cin // kc, kc).permute(1, 0, 2), lo.reshape(cout, cin // kc, kc).permute(1, 0, 2)], 1)
    return img.contiguous().view(torch.float32).reshape(-1)


def frag_image_gate(img):
    """split16_image of a (256 cols, K) gate weight ([K/32][hi|lo][256][32] halfs, as float32) -> the same halfs in MFMA fragment order for
    k_dualpath16s.hip: [K step 16][dir 2][gate tile m 4][hi|lo][lane = h*32 + r][8] with element j of lane (h, r) =
    W[k = 16 step + 8 h + j][col = dir*128 + m*32 + r]: a wave's B fragments of one K step are eight contiguous 1 KiB pieces."""
    h16 = img.view(torch.float16).reshape(-1, 2, 2, 4, 32, 2, 2, 8)  # chunk32, part, dir, m, r, half-chunk, h, j
    return h16.permute(0, 5, 2, 3, 1, 6, 4, 7).contiguous().view(torch.float32).reshape(-1)


def frag_image_gate_rot(img):
    """frag_image_gate of a layer 1-3 gate image (K = 64: four K steps) with the BACKWARD direction's K steps rotated by two: stream step s
    holds input channels 16 ((s + 2 dir) mod 4) .. + 15 of direction dir.  Gate tile 3 of these layers is the identity block of the highway
    input (rows 32 dir + j), so for BOTH directions it is non-zero in stream steps 0-1 only: the sweep kernel runs one code path that
    multiplies tile 3 in the first two steps and skips it in the last two."""
    h16 = frag_image_gate(img).view(torch.float16).reshape(4, 2, -1)
    h16 = torch.stack([h16[:, 0], torch.roll(h16[:, 1], -2, 0)], 1)
    return h16.contiguous().view(torch.float32).reshape(-1)


def frag_image_ct(img):
    """split16_image(., 64) of the (64 co, 512 k') conv-transpose weight ([8 taps][hi|lo][64 co][64] halfs) -> A fragments in order
    [tap 8][co tile 2][ks 4][hi|lo][lane = h*32 + r][8]: element j of lane (h, r) = W[co = 32 tile + r][k' = 64 tap + 16 ks + 8 h + j]."""
    h16 = img.view(torch.float16).reshape(8, 2, 2, 32, 4, 2, 8)  # tap, part, co tile, r, ks, h, j
    return h16.permute(0, 2, 4, 1, 5, 3, 6).contiguous().view(torch.float32).reshape(-1)


def pack_encoder(sd):
    """STFTEncoder: conv.full_layer.2.weight (256,2,3,3) -> (256,18)."""
    return _cat([sd["conv.full_layer.2.weight"].reshape(256, 18)])


def pack_audio_bn(sd):
    w = sd["full_layer.2.weight"].reshape(256, 256)
    return _cat([sd["full_layer.0.norm.weight"], sd["full_layer.0.norm.bias"], w.t(), sd["full_layer.2.bias"], split16_image(w)])


def _dualpath_parts(sd):
    parts = [sd["norm.gamma"].reshape(64), sd["norm.beta"].reshape(64), sd["rnn.rnn_lst.0.weight"]]
    wl = []
    for i in (1, 2, 3):
        w = sd[f"rnn.rnn_lst.{i}.weight"].reshape(64, 64, 3)  # (in, dir*32+j, m)
        wl.append(torch.cat([w, w.new_zeros(64, 64, 1)], 2).reshape(64, 256))
    parts.append(torch.stack(wl))
    parts.append(torch.stack([sd[f"rnn.rnn_lst.{i}.weight_c"] for i in range(4)]))
    parts.append(torch.stack([sd[f"rnn.rnn_lst.{i}.bias"] for i in range(4)]))
    parts.append(sd["linear.weight"].permute(2, 0, 1).reshape(512, 64))  # (ci,co,k) -> (k*64+ci, co)
    parts.append(sd["linear.bias"])
    # ---- f16x3 images for k_dualpath16.hip.  Gate columns (m = 1 forget, 2 reset), v_f/v_r and b_f/b_r carry the
    # factor -log2(e) so the kernel's sigmoid is rcp(1 + exp2(z)).
    gate_scale = torch.tensor([1.0, -LOG2E, -LOG2E, 1.0], dtype=torch.float32, device=sd["norm.gamma"].device)
    w0 = sd["rnn.rnn_lst.0.weight"].detach().to(torch.float32).reshape(64, 8, 2, 32, 4) * gate_scale  # (c, kk, dir, j, m)
    w0 = w0.permute(1, 0, 2, 4, 3).reshape(512, 256)  # rows k' = kk*64 + c, cols dir*128 + m*32 + j
    img0 = split16_image(w0.t())
    parts.append(img0)
    imgs = []
    eye = torch.eye(64, dtype=torch.float32, device=w0.device).reshape(64, 2, 32, 1)  # highway input via an identity gate
    for i in (1, 2, 3):
        w = sd[f"rnn.rnn_lst.{i}.weight"].detach().to(torch.float32).reshape(64, 2, 32, 3)
        w = torch.cat([w, eye], 3) * gate_scale  # (k, dir, j, m)
        imgs.append(split16_image(w.permute(0, 1, 3, 2).reshape(64, 256).t()))
    parts.append(torch.stack(imgs))
    imgct = split16_image(sd["linear.weight"].detach().to(torch.float32).permute(1, 2, 0).reshape(64, 512), 64)  # rows co, k' = kk*64+ci
    parts.append(imgct)
    parts.append(torch.stack([sd[f"rnn.rnn_lst.{i}.weight_c"] for i in range(4)]) * (-LOG2E))
    parts.append(torch.stack([sd[f"rnn.rnn_lst.{i}.bias"] for i in range(4)]) * (-LOG2E))
    # the same three images in fragment order (generation-3 sweep kernel: B fragments straight from L2, no LDS staging)
    f0, fl, fct = frag_image_gate(img0), [frag_image_gate_rot(i) for i in imgs], frag_image_ct(imgct)
    parts.append(f0)
    parts.append(torch.stack(fl))
    parts.append(fct)
    return parts


_PARAM_STORES = {}


def param_store(anchor, name):
    """A private dict tied to the lifetime of the tensor ``anchor`` (normally a module's first nn.Parameter) without touching the tensor:
    entries live in a module-level table keyed by id(anchor) and a weakref finalizer drops them when the tensor dies, so a new tensor that
    reuses the address never sees them - and nothing rides along when a module is pickled or deep-copied."""
    import weakref
    key = (id(anchor), name)
    st = _PARAM_STORES.get(key)
    if st is None:
        st = _PARAM_STORES[key] = {}
        weakref.finalize(anchor, _PARAM_STORES.pop, key, None)
    return st


_PACK_EPOCH = [0]


def invalidate_packs():
    """Drop every cached parameter pack (inference packs, training packs, gradient bundles) at its next use.

    The caches are keyed on (data_ptr, Tensor._version) of the live parameters, which every in-place update THROUGH THE PARAMETER bumps
    (optimizer steps, load_state_dict, p.copy_ under no_grad).  An update through ``p.data`` (``p.data.copy_(ema)``, ``p.data.clamp_()``
    - EMA swaps, weight clipping in older code) runs on a detached alias with its OWN version counter and is invisible to that key:
    call this after such an update."""
    _PACK_EPOCH[0] += 1


def pack_epoch():
    return _PACK_EPOCH[0]


def cached_train_pack(kind, tensors, build):
    """The training kernels take re-ordered copies of the live parameters.  Within one step the shared RTFS block is applied R times
    with unchanged parameters, so the copy is cached - in a store tied to the first parameter object (param_store: it dies with the
    module, so a new module that happens to reuse the addresses cannot see it), keyed by (storage address, version counter) of every parameter: an optimizer step
    bumps the versions.  Only genuine nn.Parameters are cached (a temporary may reuse an address with different contents)."""
    if not all(t is None or isinstance(t, torch.nn.Parameter) for t in tensors):
        return build()
    anchor = next((t for t in tensors if t is not None), None)
    if anchor is None:
        return build()
    store = param_store(anchor, "train_pack")
    key = (kind, _PACK_EPOCH[0]) + tuple(None if t is None else (id(t), t.data_ptr(), t._version) for t in tensors)
    pk = store.get(key)
    if pk is None:
        store.clear()  # one live entry per (anchor): older versions are dead weight
        pk = store[key] = build()
    return pk


def pack_sru_train(weights, weight_cs, biases):
    """Training-side pack of sru.SRU(512, 32, 4 layers, bidirectional) (layout contract: include/rtfs_amd.h,
    rtfs_sru_forward_train_f32): projections re-ordered to column m*64 + dir*32 + j, once K-major (Wt) and once as is (Wp)."""
    wp = []
    for w in weights:
        din = w.shape[0]
        k = w.shape[1] // 64
        wp.append(w.detach().to(torch.float32).reshape(din, 64, k).permute(0, 2, 1).reshape(din, 64 * k))
    parts = [wp[0].t()] + [w.t() for w in wp[1:]] + wp + [torch.stack(list(weight_cs)), torch.stack(list(biases))]
    return _cat([t.contiguous() for t in parts])


def unpack_sru_grads(flat):
    """Inverse of the gradient layout of rtfs_sru_backward_f32: flat -> ([dweight], [dweight_c], [dbias]) in the module's shapes."""
    sizes = [(512, 4), (64, 3), (64, 3), (64, 3)]
    dws, off = [], 0
    for din, k in sizes:
        n = din * 64 * k
        dws.append(flat[off:off + n].reshape(din, k, 64).permute(0, 2, 1).reshape(din, 64 * k))
        off += n
    dwc = flat[off:off + 512].reshape(4, 128)
    db = flat[off + 512:off + 1024].reshape(4, 128)
    return dws, list(dwc), list(db)


def pack_dualpath_train(gamma, beta, weights, weight_cs, biases, lin_w, lin_b):
    """Training-side pack of DualPathRNN with the SRU cell (layout contract: include/rtfs_amd.h,
    rtfs_dualpath_forward_train_f32).  The Unfold feature order c*8 + k becomes k*64 + c (a window of the channel-last
    sequence), the ConvTranspose1d weight (ci, co, k) is laid out once for the forward windows and once for the adjoint."""
    w0 = weights[0].detach().to(torch.float32).reshape(64, 8, 256).permute(1, 0, 2).reshape(512, 256)  # rows k*64 + c
    sru = pack_sru_train([w0] + list(weights[1:]), weight_cs, biases)
    lw = lin_w.detach().to(torch.float32)
    wcf = lw.flip(2).permute(1, 2, 0).reshape(64, 512)  # (co, (7-k)*64 + ci)
    wcb = lw.permute(0, 2, 1).reshape(64, 512)          # (ci, k*64 + co)
    return _cat([gamma.reshape(64), beta.reshape(64), sru, wcf.contiguous(), wcb.contiguous(), lin_b])


def unpack_dualpath_grads(flat):
    """rtfs_dualpath_backward_f32's gradient buffer -> (dgamma (64), dbeta (64), [dweight], [dweight_c], [dbias], dlin_w, dlin_b)."""
    n_sru = 512 * 256 + 3 * 64 * 192 + 1024
    dws, dwcs, dbs = unpack_sru_grads(flat[128:128 + n_sru])
    dws[0] = dws[0].reshape(8, 64, 256).permute(1, 0, 2).reshape(512, 256)  # rows back to c*8 + k
    off = 128 + n_sru
    dlw = flat[off:off + 512 * 64].reshape(8, 64, 64).flip(0).permute(1, 2, 0)  # ((7-k), ci, co) -> (ci, co, k)
    return flat[0:64], flat[64:128], dws, dwcs, dbs, dlw, flat[off + 512 * 64:off + 512 * 64 + 64]


def pack_cna_train(cfg, pre_g, pre_b, pre_s, w, bias, g, b, s, rmean=None, rvar=None):
    """Training-side parameter buffer of one ConvNormAct (layout contract: include/rtfs_amd.h, rtfs_cna_forward_train_f32).
    cfg = (Cin, Cout, k, stride, depthwise, pre_norm, pre_act, norm, act, has_bias, is2d); absent parameters are zero-filled."""
    cin, cout, depthwise = cfg[0], cfg[1], cfg[4]
    dev = w.device
    z = lambda n: zeros_view(n, dev)
    w2 = w.detach().to(torch.float32).reshape(cout, -1)
    parts = [pre_g if pre_g is not None else z(cin), pre_b if pre_b is not None else z(cin), pre_s if pre_s is not None else z(1), w2]
    if not depthwise:
        parts.append(w2.t().contiguous())
    parts += [bias if bias is not None else z(cout), g if g is not None else z(cout), b if b is not None else z(cout),
              s if s is not None else z(1), rmean if rmean is not None else z(cout), rvar if rvar is not None else z(cout)]
    return _cat(parts)


def unpack_cna_grads(cfg, flat, w_shape):
    """rtfs_cna_backward_f32's gradient buffer -> (dpre_g, dpre_b, dpre_s, dw, dbias, dg, db, ds) (flat views; caller reshapes)."""
    cin, cout = cfg[0], cfg[1]
    pad = lambda n: (n + ALIGN - 1) // ALIGN * ALIGN
    wn = 1
    for d in w_shape:
        wn *= d
    sizes = [cin, cin, 1, wn, cout, cout, cout, 1]
    out, off = [], 0
    for n in sizes:
        out.append(flat[off:off + n])
        off += pad(n)
    out[3] = out[3].reshape(w_shape)
    return out


_ATT_MODULES = [f"Queries.{h}" for h in range(4)] + [f"Keys.{h}" for h in range(4)] + [f"Values.{h}" for h in range(4)]


def pack_attention_train(sd):
    """Training-side pack of MultiHeadSelfAttention2D (layout contract: include/rtfs_amd.h, rtfs_tf_attention_forward_train_f32).
    sd: name -> tensor with the module's state_dict names (live parameters are fine)."""
    f32 = lambda t: t.detach().to(torch.float32)
    w = torch.cat([f32(sd[m + ".conv.weight"]).reshape(-1, 64) for m in _ATT_MODULES])
    dev = w.device
    w = torch.cat([w, torch.zeros(32, 64, device=dev)])
    b = torch.cat([f32(sd[m + ".conv.bias"]) for m in _ATT_MODULES] + [torch.zeros(32, device=dev)])
    sl = torch.cat([f32(sd[m + ".act.weight"]).expand(sd[m + ".conv.bias"].numel()) for m in _ATT_MODULES] + [torch.zeros(32, device=dev)])
    ga = torch.cat([f32(sd[m + ".norm.gamma"]).reshape(-1, 64) for m in _ATT_MODULES] + [torch.zeros(32, 64, device=dev)])
    be = torch.cat([f32(sd[m + ".norm.beta"]).reshape(-1, 64) for m in _ATT_MODULES] + [torch.zeros(32, 64, device=dev)])
    p = "attn_concat_proj"
    wp = f32(sd[p + ".conv.weight"]).reshape(64, 64)
    return _cat([w, w.t().contiguous(), b, sl, ga, be, wp, wp.t().contiguous(), f32(sd[p + ".conv.bias"]), f32(sd[p + ".act.weight"]).expand(64),
                 f32(sd[p + ".norm.gamma"]).reshape(64, 64), f32(sd[p + ".norm.beta"]).reshape(64, 64)])


def unpack_attention_grads(flat):
    """rtfs_tf_attention_backward_f32's gradient buffer -> dict name -> gradient in the module's parameter shapes."""
    o = 0
    take = lambda n: flat[o:o + n]
    dw = flat[0:128 * 64].reshape(128, 64); o = 128 * 64
    db = flat[o:o + 128]; o += 128
    dsl = flat[o:o + 64]; o += 64
    dg = flat[o:o + 128 * 64].reshape(128, 64); o += 128 * 64
    dbe = flat[o:o + 128 * 64].reshape(128, 64); o += 128 * 64
    out, c0 = {}, 0
    for i, m in enumerate(_ATT_MODULES):
        c = 4 if i < 8 else 16
        out[m + ".conv.weight"] = dw[c0:c0 + c].reshape(c, 64, 1, 1)
        out[m + ".conv.bias"] = db[c0:c0 + c]
        out[m + ".act.weight"] = dsl[i:i + 1]
        out[m + ".norm.gamma"] = dg[c0:c0 + c].reshape(1, c, 1, 64)
        out[m + ".norm.beta"] = dbe[c0:c0 + c].reshape(1, c, 1, 64)
        c0 += c
    p = "attn_concat_proj"
    out[p + ".conv.weight"] = flat[o:o + 4096].reshape(64, 64, 1, 1); o += 4096
    out[p + ".conv.bias"] = flat[o:o + 64]; o += 64
    out[p + ".act.weight"] = flat[o:o + 1]; o += 64
    out[p + ".norm.gamma"] = flat[o:o + 4096].reshape(1, 64, 1, 64); o += 4096
    out[p + ".norm.beta"] = flat[o:o + 4096].reshape(1, 64, 1, 64)
    return out


_LSTM_SUFFIX = ("", "_reverse")


def lstm_param_names(num_layers=4):
    """nn.LSTM's parameter names in the order the training pack / gradient buffer uses: per layer, per direction, (w_ih, w_hh, b_ih, b_hh)."""
    return [f"{n}_l{l}{suf}" for l in range(num_layers) for suf in _LSTM_SUFFIX for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


def pack_dualpath_lstm_train(gamma, beta, lstm, lin_w, lin_b):
    """Training-side pack of DualPathRNN with nn.LSTM(512, 32, 4 layers, bidirectional) (layout contract: include/rtfs_amd.h,
    rtfs_dualpath_lstm_forward_train_f32).  lstm: name -> tensor with nn.LSTM's names."""
    f32 = lambda t: t.detach().to(torch.float32)
    parts = [gamma.reshape(64), beta.reshape(64)]
    for l in range(4):
        wih = torch.cat([f32(lstm[f"weight_ih_l{l}{suf}"]) for suf in _LSTM_SUFFIX])  # (256, Din), rows dir*128 + gate*32 + j
        if l == 0:
            wih = wih.reshape(256, 64, 8).permute(0, 2, 1).reshape(256, 512)  # Unfold feature c*8 + k -> window order k*64 + c
        bias = torch.cat([f32(lstm[f"bias_ih_l{l}{suf}"]) + f32(lstm[f"bias_hh_l{l}{suf}"]) for suf in _LSTM_SUFFIX])
        whh = torch.stack([f32(lstm[f"weight_hh_l{l}{suf}"]) for suf in _LSTM_SUFFIX])
        parts += [wih.contiguous(), wih.t().contiguous(), bias, whh.contiguous()]
    lw = f32(lin_w)
    parts += [lw.flip(2).permute(1, 2, 0).reshape(64, 512).contiguous(), lw.permute(0, 2, 1).reshape(64, 512).contiguous(), lin_b]
    return _cat(parts)


def unpack_dualpath_lstm_grads(flat):
    """rtfs_dualpath_lstm_backward_f32's gradient buffer -> (dgamma, dbeta, {lstm name: grad}, dlin_w, dlin_b)."""
    out, o = {}, 128
    for l in range(4):
        din = 512 if l == 0 else 64
        dwih = flat[o:o + 256 * din].reshape(256, din); o += 256 * din
        if l == 0:
            dwih = dwih.reshape(256, 8, 64).permute(0, 2, 1).reshape(256, 512)
        db = flat[o:o + 256]; o += 256
        dwhh = flat[o:o + 2 * 128 * 32].reshape(2, 128, 32); o += 2 * 128 * 32
        for d, suf in enumerate(_LSTM_SUFFIX):
            out[f"weight_ih_l{l}{suf}"] = dwih[d * 128:(d + 1) * 128]
            out[f"weight_hh_l{l}{suf}"] = dwhh[d]
            out[f"bias_ih_l{l}{suf}"] = db[d * 128:(d + 1) * 128]
            out[f"bias_hh_l{l}{suf}"] = db[d * 128:(d + 1) * 128]
    dlw = flat[o:o + 512 * 64].reshape(8, 64, 64).flip(0).permute(1, 2, 0)
    return flat[0:64], flat[64:128], out, dlw, flat[o + 512 * 64:o + 512 * 64 + 64]


def pack_dualpath_gru_train(gamma, beta, gru, lin_w, lin_b):
    """Pack of DualPathRNN with nn.GRU(512, 32, 4 layers, bidirectional) (layout contract: include/rtfs_amd.h,
    rtfs_dualpath_gru_forward_train_f32).  gru: name -> tensor with nn.GRU's names (the same as nn.LSTM's)."""
    f32 = lambda t: t.detach().to(torch.float32)
    parts = [gamma.reshape(64), beta.reshape(64)]
    for l in range(4):
        wih = torch.cat([f32(gru[f"weight_ih_l{l}{suf}"]) for suf in _LSTM_SUFFIX])  # (192, Din), rows dir*96 + gate*32 + j
        if l == 0:
            wih = wih.reshape(192, 64, 8).permute(0, 2, 1).reshape(192, 512)
        bih = torch.cat([f32(gru[f"bias_ih_l{l}{suf}"]) for suf in _LSTM_SUFFIX])
        bhh = torch.cat([f32(gru[f"bias_hh_l{l}{suf}"]) for suf in _LSTM_SUFFIX])
        whh = torch.stack([f32(gru[f"weight_hh_l{l}{suf}"]) for suf in _LSTM_SUFFIX])
        parts += [wih.contiguous(), wih.t().contiguous(), bih, whh.contiguous(), bhh]
    lw = f32(lin_w)
    parts += [lw.flip(2).permute(1, 2, 0).reshape(64, 512).contiguous(), lw.permute(0, 2, 1).reshape(64, 512).contiguous(), lin_b]
    return _cat(parts)


def unpack_dualpath_gru_grads(flat):
    """rtfs_dualpath_gru_backward_f32's gradient buffer -> (dgamma, dbeta, {gru name: grad}, dlin_w, dlin_b)."""
    out, o = {}, 128
    for l in range(4):
        din = 512 if l == 0 else 64
        dwih = flat[o:o + 192 * din].reshape(192, din); o += 192 * din
        if l == 0:
            dwih = dwih.reshape(192, 8, 64).permute(0, 2, 1).reshape(192, 512)
        dbih = flat[o:o + 192]; o += 192
        dwhh = flat[o:o + 2 * 96 * 32].reshape(2, 96, 32); o += 2 * 96 * 32
        dbhh = flat[o:o + 192]; o += 192
        for d, suf in enumerate(_LSTM_SUFFIX):
            out[f"weight_ih_l{l}{suf}"] = dwih[d * 96:(d + 1) * 96]
            out[f"weight_hh_l{l}{suf}"] = dwhh[d]
            out[f"bias_ih_l{l}{suf}"] = dbih[d * 96:(d + 1) * 96]
            out[f"bias_hh_l{l}{suf}"] = dbhh[d * 96:(d + 1) * 96]
    dlw = flat[o:o + 512 * 64].reshape(8, 64, 64).flip(0).permute(1, 2, 0)
    return flat[0:64], flat[64:128], out, dlw, flat[o + 512 * 64:o + 512 * 64 + 64]


def _dualpath_lstm_parts(sd):
    """DualPathRNN with rnn_type LSTM (nn.LSTM(512, 32, 4 layers, bidirectional)).  Columns of the input projections:
    dir*128 + gate*32 + j (gates i,f,g,o); bias = b_ih + b_hh; recurrent weights as [layer][dir][k][gate*32 + j]."""
    def wih(layer):  # (K, 256)
        return torch.cat([sd[f"rnn.weight_ih_l{layer}{suf}"].t() for suf in ("", "_reverse")], 1)
    parts = [sd["norm.gamma"].reshape(64), sd["norm.beta"].reshape(64), wih(0), torch.stack([wih(i) for i in (1, 2, 3)])]
    parts.append(torch.stack([torch.cat([sd[f"rnn.bias_ih_l{i}{suf}"] + sd[f"rnn.bias_hh_l{i}{suf}"] for suf in ("", "_reverse")]) for i in range(4)]))
    parts.append(torch.stack([torch.stack([sd[f"rnn.weight_hh_l{i}{suf}"].t() for suf in ("", "_reverse")]) for i in range(4)]))  # (4,2,32,128)
    parts.append(sd["linear.weight"].permute(2, 0, 1).reshape(512, 64))
    parts.append(sd["linear.bias"])
    return parts


def _is_lstm(sd):
    return "rnn.weight_ih_l0" in sd


def pack_dualpath(sd):
    return _cat(_dualpath_lstm_parts(sd) if _is_lstm(sd) else _dualpath_parts(sd))


def _attention_parts(sd):
    names = [f"Queries.{h}" for h in range(4)] + [f"Keys.{h}" for h in range(4)] + [f"Values.{h}" for h in range(4)]
    w = torch.cat([sd[n + ".conv.weight"].reshape(-1, 64) for n in names])  # (96,64)
    b = torch.cat([sd[n + ".conv.bias"] for n in names])
    slope = torch.cat([sd[n + ".act.weight"].reshape(1) for n in names])
    gamma = torch.cat([sd[n + ".norm.gamma"].reshape(-1, 64) for n in names])
    beta = torch.cat([sd[n + ".norm.beta"].reshape(-1, 64) for n in names])
    p = "attn_concat_proj"
    return [w.t(), b, slope, gamma, beta, sd[p + ".conv.weight"].reshape(64, 64).t(), sd[p + ".conv.bias"],
            sd[p + ".act.weight"].reshape(1), sd[p + ".norm.gamma"].reshape(64, 64), sd[p + ".norm.beta"].reshape(64, 64)]


def pack_attention(sd):
    return _cat(_attention_parts(sd))


def _tfar_parts(sd):
    parts = []
    for n in ("local_embedding", "global_embedding", "global_gate"):
        parts += [sd[n + ".full_layer.2.weight"].reshape(64, 16), sd[n + ".full_layer.3.norm.weight"], sd[n + ".full_layer.3.norm.bias"]]
    return parts


def pack_tfar(sd):
    return _cat(_tfar_parts(sd))


def proj_perm():
    """K order in which the back-to-back kernel feeds GEMM-1 accumulators to the projection GEMM: position
    (2m+s)*16 + 8h + j holds channel 32m + 16s + (j&3) + 8(j>>2) + 4h  (m co-tile, s register octet, h lane half)."""
    idx = []
    for m in range(8):
        for s_ in range(2):
            for h in range(2):
                for j in range(8):
                    idx.append(32 * m + 16 * s_ + (j & 3) + 8 * (j >> 2) + 4 * h)
    return idx


def pack_block(sd):
    """TDANetBlock (2-D, depth 2) with globalatt = [DualPathRNN, DualPathRNN, MultiHeadSelfAttention2D]."""
    parts = [sd["gateway.full_layer.2.weight"].reshape(256), sd["gateway.full_layer.2.bias"], sd["gateway.full_layer.4.weight"].reshape(1),
             sd["projection.full_layer.2.weight"].reshape(64, 256).t(), sd["projection.full_layer.2.bias"]]
    for i in (0, 1):
        p = f"downsample_layers.{i}.full_layer."
        parts += [sd[p + "2.weight"].reshape(64, 16), sd[p + "2.bias"], sd[p + "3.norm.weight"], sd[p + "3.norm.bias"]]
    for g in ("globalatt.0", "globalatt.1"):
        sub = _sub(sd, g)
        parts += _dualpath_lstm_parts(sub) if _is_lstm(sub) else _dualpath_parts(sub)
    parts += _attention_parts(_sub(sd, "globalatt.2"))
    parts += _tfar_parts(_sub(sd, "fusion_layers.0"))
    parts += _tfar_parts(_sub(sd, "fusion_layers.1"))
    parts += _tfar_parts(_sub(sd, "concat_layers.0"))
    parts += [sd["residual_conv.full_layer.2.weight"].reshape(256, 64).t(), sd["residual_conv.full_layer.2.bias"]]
    wp = sd["projection.full_layer.2.weight"].reshape(64, 256)
    parts += [split16_image(wp), split16_image(sd["residual_conv.full_layer.2.weight"].reshape(256, 64))]
    parts += [split16_image(wp[:, torch.tensor(proj_perm(), device=wp.device)])]
    return _cat(parts)


def pack_caf(sd):
    """ATTNFusionCell (audio_lstm)."""
    def bn(n):
        p = n + ".full_layer.3."
        return torch.stack([sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"]])
    return _cat([
        sd["key_embed.full_layer.2.weight"].reshape(256), bn("key_embed"),
        sd["value_embed.full_layer.2.weight"].reshape(256), bn("value_embed"),
        sd["attention_embed.full_layer.2.weight"].reshape(1024, 2), sd["attention_embed.full_layer.2.bias"],
        sd["attention_embed.full_layer.3.norm.weight"], sd["attention_embed.full_layer.3.norm.bias"],
        sd["resize.full_layer.2.weight"].reshape(256, 2), sd["resize.full_layer.2.bias"],
        sd["resize.full_layer.3.norm.weight"], sd["resize.full_layer.3.norm.bias"],
    ])


def pack_s3(sd):
    """MaskGenerator: mask_generator = Sequential(PReLU, ConvNormAct(1x1, ReLU))."""
    w = sd["mask_generator.1.full_layer.2.weight"].reshape(256, 256)
    return _cat([sd["mask_generator.0.weight"].reshape(1), w.t(), sd["mask_generator.1.full_layer.2.bias"], split16_image(w)])


def pack_decoder(sd):
    """STFTDecoder: ConvTranspose2d weight (256,2,3,3) -> 18 per-tap 1x1 maps, zero padded to 32."""
    w = sd["decoder.weight"].reshape(256, 18)
    w = torch.cat([w, w.new_zeros(256, 14)], 1)  # (cin 256, 32 taps)
    return _cat([w, split16_image(w.t()), taps_perm_image(w.t())])


def taps_perm_image(wt):
    """(32 taps, 256 ch) decoder tap maps -> A fragments of the S3 + taps kernel (k_pwr.hip, PWR_S3T): the separated
    spectrum reaches the matrix cores as accumulator registers, so K runs in register order.  Layout
    [m 4][part 2][s 2][hi|lo][tap 32][16] halfs with k = 8h + j <-> channel part*128 + m*32 + 16s + 4h + (j&3) + 8(j>>2)."""
    ws = wt.detach().to(torch.float32) * W16_SCALE
    idx = torch.empty(4, 2, 2, 16, dtype=torch.long)
    for m in range(4):
        for part in range(2):
            for s_ in range(2):
                for hh in range(2):
                    for j in range(8):
                        idx[m, part, s_, 8 * hh + j] = part * 128 + m * 32 + 16 * s_ + 4 * hh + (j & 3) + 8 * (j >> 2)
    g = ws[:, idx.reshape(-1)].reshape(32, 4, 2, 2, 16).permute(1, 2, 3, 0, 4)  # [m][part][s][tap][16]
    hi = g.to(torch.float16)
    lo = (g - hi.to(torch.float32)).to(torch.float16)
    img = torch.stack([hi, lo], 3)  # [m][part][s][hl][tap][16]
    return img.contiguous().view(torch.float32).reshape(-1)


def _bn_fold(sd, prefix, conv_bias=None, eps=1e-5):
    """Eval BatchNorm1d (and the conv bias in front of it) -> y = scale * conv + shift."""
    scale = sd[prefix + ".weight"] / torch.sqrt(sd[prefix + ".running_var"] + eps)
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    if conv_bias is not None:
        shift = shift + scale * conv_bias
    return scale, shift


def pack_vp(sd):
    """Video TDANetBlock (1-D, depth 4, k 3, BatchNorm1d, GlobalAttention): order = csrc/k_vp.hip."""
    parts = [sd["gateway.full_layer.2.weight"].reshape(512), sd["gateway.full_layer.2.bias"], sd["gateway.full_layer.4.weight"].reshape(1),
             sd["projection.full_layer.2.weight"].reshape(64, 512).t(), sd["projection.full_layer.2.bias"]]
    for i in range(4):
        p = f"downsample_layers.{i}.full_layer."
        sc, sh = _bn_fold(sd, p + "3", sd[p + "2.bias"])
        parts += [sd[p + "2.weight"].reshape(64, 3), sc, sh]
    m = "globalatt.0.MHSA."
    parts += [sd[m + "norm1.weight"], sd[m + "norm1.bias"], sd[m + "pos_enc.pe"][0, :16], sd[m + "attention.in_proj_weight"],
              sd[m + "attention.in_proj_bias"], sd[m + "attention.out_proj.weight"], sd[m + "attention.out_proj.bias"],
              sd[m + "norm2.weight"], sd[m + "norm2.bias"]]
    f = "globalatt.0.FFN."
    parts += [sd[f + "encoder.full_layer.2.weight"].reshape(128, 64), sd[f + "encoder.full_layer.3.norm.weight"], sd[f + "encoder.full_layer.3.norm.bias"],
              sd[f + "refiner.full_layer.2.weight"].reshape(128, 3), sd[f + "refiner.full_layer.2.bias"],
              sd[f + "decoder.full_layer.2.weight"].reshape(64, 128), sd[f + "decoder.full_layer.3.norm.weight"], sd[f + "decoder.full_layer.3.norm.bias"]]
    for grp, n in (("fusion_layers", 4), ("concat_layers", 3)):
        for i in range(n):
            for name in ("local_embedding", "global_embedding", "global_gate"):
                p = f"{grp}.{i}.{name}.full_layer."
                sc, sh = _bn_fold(sd, p + "3")
                parts += [sd[p + "2.weight"].reshape(64, 3), sc, sh]
    parts += [sd["residual_conv.full_layer.2.weight"].reshape(512, 64).t(), sd["residual_conv.full_layer.2.bias"]]
    return _cat(parts)


# ----------------------------------------------------------------------------- video front-end (k_video.hip)
def _video_conv_image(w2d, bn_scale):
    """(cout, K) f32 conv weight with the eval-BatchNorm scale folded in -> [cout/64][K/32][hi|lo][64][32] halfs (bit-cast
    to float32), K zero-padded to a multiple of 32."""
    cout, K = w2d.shape
    w = (w2d.detach().to(torch.float32) * bn_scale.reshape(-1, 1)) * W16_SCALE
    Kp = (K + 31) // 32 * 32
    if Kp != K:
        w = torch.cat([w, w.new_zeros(cout, Kp - K)], 1)
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    img = torch.stack([hi.reshape(cout // 64, 64, Kp // 32, 32).permute(0, 2, 1, 3), lo.reshape(cout // 64, 64, Kp // 32, 32).permute(0, 2, 1, 3)], 2)
    return img.contiguous().view(torch.float32).reshape(-1)  # [cb][chunk][hi|lo][64][32]


def _video_bn_fold(sd, prefix, eps=1e-5):
    scale = sd[prefix + ".weight"].float() / torch.sqrt(sd[prefix + ".running_var"].float() + eps)
    return scale, sd[prefix + ".bias"].float() - sd[prefix + ".running_mean"].float() * scale


def pack_video(sd):
    """FRCNNVideoModel (backbone 'resnet', relu_type 'prelu') state_dict -> the pack of rtfs_video_frontend_f32: for the stem
    and then every trunk convolution in forward order (conv1, [downsample], conv2 per block): weight image, bias, slopes."""
    parts = []

    def add(w2d, bn_prefix, slope):
        scale, shift = _video_bn_fold(sd, bn_prefix)
        parts.extend([_video_conv_image(w2d, scale), shift, slope if slope is not None else torch.zeros_like(shift)])

    add(sd["frontend3D.0.weight"].reshape(64, 245), "frontend3D.1", sd["frontend3D.2.weight"])
    for li in (1, 2, 3, 4):
        for bi in (0, 1):
            pre = f"trunk.layer{li}.{bi}"
            w1 = sd[pre + ".conv1.weight"]  # (planes, cin, 3, 3) -> k = (dy*3+dx)*cin + ci
            add(w1.permute(0, 2, 3, 1).reshape(w1.shape[0], -1), pre + ".bn1", sd[pre + ".relu1.weight"])
            if pre + ".downsample.0.weight" in sd:
                wd = sd[pre + ".downsample.0.weight"]
                add(wd.reshape(wd.shape[0], -1), pre + ".downsample.1", None)
            w2 = sd[pre + ".conv2.weight"]
            add(w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1), pre + ".bn2", sd[pre + ".relu2.weight"])
    return _cat(parts)
