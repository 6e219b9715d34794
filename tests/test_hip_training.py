"""GPU parity of the training side: every HIP forward-with-saved-state / backward pair, the composed blocks and
the whole training step against the float64 autograd oracle (oracle/grad_oracle.py, itself pinned by gradients captured from the
reference's own autograd, tests/golden/grad_R2_L4096_B2.npz).  Tolerances are written per test; see DESIGN.md "parity / kinks" for why
piecewise-linear activations need sign masks or l2 bounds at block level."""
import os

import numpy as np
import pytest
import torch

from oracle import rtfs_oracle as O
from oracle.params import make_inputs, make_state_dict
from tests.test_hip_parity import BLK, CELL, SD, _conf, close, dev, host, lstm_model, model
from tests.util import l2_rel, rand, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,M,N,K,acc", [(0, 300, 192, 64, 0), (0, 1000, 512, 256, 1), (0, 37, 64, 512, 0), (1, 512, 256, 1237, 0),
                                             (1, 64, 192, 5, 0), (1, 64, 64, 40000, 0)])
def test_training_gemms(kind, M, N, K, acc):
    """The two bf16x3 GEMM forms of the training path against float64 numpy (error budget ~2^-17 per product)."""
    from rtfs_net_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    if kind == 0:
        A, B = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
    else:
        A, B = rng.standard_normal((K, M)).astype(np.float32), rng.standard_normal((K, N)).astype(np.float32)
        A[:, 0] *= 1e-12  # a column far below f16's range: bf16 keeps the exponent
        ref = A.astype(np.float64).T @ B.astype(np.float64)
    C0 = rng.standard_normal((M, N)).astype(np.float32) if (acc or kind == 1) else np.zeros((M, N), np.float32)
    if kind == 1:
        C0[0] = 0
    C = dev(C0)
    a, b = dev(A), dev(B)
    _lib.check(lib.rtfs_debug_gemm_f32(kind, _lib.ptr(a), _lib.ptr(b), _lib.ptr(C), M, N, K, acc, _lib.stream_of(a)), "gemm")
    want = ref + (C0 if (acc or kind == 1) else 0)
    close(f"gemm kind {kind} {M}x{N}x{K}", host(C), want, tol=2e-5)
    if kind == 1:  # the tiny column by itself
        close("gemm tiny column", host(C)[0], ref[0], tol=2e-5)


@pytest.mark.parametrize("L,N,seed", [(19, 5, 21), (57, 37, 22), (250, 3, 23)])
def test_sru_training_forward_backward(L, N, seed):
    """sru.SRU used from a training step: forward (with saved state) and backward kernels against the autograd oracle."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    p = O._sub(BLK, "globalatt.0")
    layers = O._sru_layers(p)
    sru = R.layers.SRU(512, 32, num_layers=4, bidirectional=True)
    sru.load_state_dict({k[len("rnn."):]: torch.from_numpy(v) for k, v in p.items() if k.startswith("rnn.")})
    sru = sru.cuda().train()
    x = rand((L, N, 512), seed)
    dh = rand((L, N, 64), seed + 100)
    xt = dev(x).requires_grad_(True)
    h, _ = sru(xt)
    h.backward(dev(dh))
    h_ref, dx_ref, g_ref = G.sru_grads(x, layers, dh)
    close("sru train forward", host(h), h_ref)
    close("sru dx", host(xt.grad), dx_ref, tol=2e-4)
    for i, cell in enumerate(sru.rnn_lst):
        close(f"sru layer {i} dW", host(cell.weight.grad), g_ref[i][0], tol=2e-4)
        close(f"sru layer {i} dweight_c", host(cell.weight_c.grad), g_ref[i][1], tol=2e-4)
        close(f"sru layer {i} dbias", host(cell.bias.grad), g_ref[i][2], tol=2e-4)
    # inference kernel and training forward agree
    with torch.no_grad():
        if L <= 243:
            close("sru eval vs train forward", host(sru(dev(x))[0]), host(h))


@pytest.mark.parametrize("idx,shape,seed", [(0, (2, 64, 11, 13), 31), (1, (2, 64, 11, 13), 32), (0, (1, 64, 5, 64), 33), (1, (1, 64, 125, 9), 34),
                                            (1, (1, 64, 250, 3), 35)])
def test_dualpath_training_forward_backward(idx, shape, seed):
    """DualPathRNN (SRU cell) used from a training step: forward + backward kernels against the autograd oracle
    (LayerNorm, Unfold windows, SRU, ConvTranspose1d, residual; both sweep directions)."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    p = O._sub(BLK, f"globalatt.{idx}")
    dim = 4 if idx == 0 else 3
    mod = R.layers.DualPathRNN(64, 32, dim, kernel_size=8, stride=1, rnn_type="SRU", num_layers=4, bidirectional=True)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    mod = mod.cuda().train()
    x = rand(shape, seed)
    dout = rand(shape, seed + 100)
    xt = dev(x).requires_grad_(True)
    out = mod(xt)
    out.backward(dev(dout))
    o_ref, dx_ref, g_ref = G.dualpath_grads(x, p, dim, dout)
    close("dualpath train forward", host(out), o_ref)
    close("dualpath dx", host(xt.grad), dx_ref, tol=2e-4)
    got = {k: v.grad for k, v in mod.named_parameters()}
    assert set(got) == set(g_ref)
    for k in sorted(g_ref):
        close(f"dualpath d {k}", host(got[k]).reshape(g_ref[k].shape), g_ref[k], tol=2e-4)
    with torch.no_grad():
        close("dualpath eval vs train forward", host(mod(dev(x))), host(out))


CNA_CASES = {
    # name: (ctor kwargs, input shape)  -- the ConvNormAct configurations on the path (yaml + tdanet.py / fusion.py / tdavnet.py ctors)
    "audio_bn": (dict(in_chan=256, out_chan=256, kernel_size=1, pre_norm_type="gLN", pre_act_type="ReLU", is2d=True), (2, 256, 9, 7)),
    "projection": (dict(in_chan=256, out_chan=64, kernel_size=1, norm_type="gLN", act_type="PReLU", is2d=True), (2, 256, 9, 7)),
    "gateway": (dict(in_chan=256, out_chan=256, kernel_size=1, groups=256, act_type="PReLU", is2d=True), (2, 256, 6, 5)),
    "downsample": (dict(in_chan=64, out_chan=64, kernel_size=4, stride=2, groups=64, norm_type="gLN", is2d=True), (2, 64, 11, 9)),
    "tfar_gate": (dict(in_chan=64, out_chan=64, kernel_size=4, groups=64, norm_type="gLN", act_type="Sigmoid", bias=False, is2d=True), (2, 64, 10, 7)),
    "tfar_plain": (dict(in_chan=64, out_chan=64, kernel_size=4, groups=64, norm_type="gLN", bias=False, is2d=True), (1, 64, 5, 12)),
    "residual_conv": (dict(in_chan=64, out_chan=256, kernel_size=1, is2d=True), (2, 64, 9, 7)),
    "ffn_refiner_1d": (dict(in_chan=128, out_chan=128, kernel_size=5, groups=128, act_type="ReLU", is2d=False), (2, 128, 13)),
    "ffn_encoder_1d": (dict(in_chan=64, out_chan=128, kernel_size=1, norm_type="gLN", bias=False, is2d=False), (3, 64, 50)),
}


@pytest.mark.parametrize("name", sorted(CNA_CASES))
def test_conv_norm_act_training_forward_backward(name):
    """ConvNormAct used from a training step (every configuration the path instantiates): forward + all gradients vs the autograd oracle."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    kw, shape = CNA_CASES[name]
    torch.manual_seed(sum(map(ord, name)))
    mod = R.layers.ConvNormAct(**kw)
    with torch.no_grad():
        for k, v in mod.named_parameters():  # away from the init values (gamma 1, beta 0, slope 0.25)
            if "norm" in k or k.endswith("1.weight") or k.endswith("4.weight") or k.endswith("bias"):
                v.add_(0.3 * torch.randn_like(v))
    p = {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}
    mod = mod.cuda().train()
    conv = mod.full_layer[2]
    depthwise = conv.groups > 1
    code = R.layers._ACT_CODE
    cfg = (conv.in_channels, conv.out_channels, kw["kernel_size"], kw.get("stride", 1), int(depthwise), int(kw.get("pre_norm_type") == "gLN"),
           code[type(mod.full_layer[1])], int(kw.get("norm_type") == "gLN"), code[type(mod.full_layer[4])], int(conv.bias is not None),
           int(kw["is2d"]))
    x = rand(shape, 7)
    xt = dev(x).requires_grad_(True)
    out = mod(xt)
    dout = rand(tuple(out.shape), 8)
    out.backward(dev(dout))
    o_ref, dx_ref, g_ref = G.cna_grads(x, p, cfg, dout)
    close(f"{name} forward", host(out), o_ref)
    close(f"{name} dx", host(xt.grad), dx_ref, tol=2e-4)
    got = {k: v.grad for k, v in mod.named_parameters()}
    assert set(got) == set(g_ref)
    for k in sorted(g_ref):
        assert got[k] is not None, k
        close(f"{name} d {k}", host(got[k]), g_ref[k], tol=2e-4)


@pytest.mark.parametrize("shape,seed", [((2, 64, 12, 64), 41), ((1, 64, 125, 64), 42), ((3, 64, 1, 64), 43), ((1, 64, 70, 64), 44)])
def test_mhsa2d_training_forward_backward(shape, seed):
    """MultiHeadSelfAttention2D used from a training step: forward + every gradient (12 Q/K/V ConvActNorms, softmax attention,
    concat projection, residual) against the autograd oracle."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    p = {k: v.copy() for k, v in O._sub(BLK, "globalatt.2").items()}
    rng = np.random.default_rng(seed)
    for k in p:  # away from the init values so every gradient path is exercised
        if "norm" in k or "act" in k or "bias" in k:
            p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
    mod = R.layers.MultiHeadSelfAttention2D(64, 64, n_head=4, hid_chan=4)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    mod = mod.cuda().train()
    x = rand(shape, seed)
    dout = rand(shape, seed + 100)
    xt = dev(x).requires_grad_(True)
    out = mod(xt)
    sv = out.grad_fn.saved_tensors[1].clone()
    out.backward(dev(dout))
    # the implementation's PReLU sign pattern (see grad_oracle.mhsa2d_torch): pre-activations sit in the saved-state buffer as
    # r0 (R,64) | Z (R,128: Q h0-3, K h0-3 (4 ch each), V h0-3 (16 each), 32 pad) | stats | Qp Kp Vp | P | r_att (R,64) | Z2 (R,64)
    B, _, T, _ = shape
    Rr, Tp = B * T * 64, (T + 63) // 64 * 64
    Z = host(sv[Rr * 64:Rr * 192]).reshape(B, T, 64, 128).transpose(0, 3, 1, 2)  # (B, ch, T, F)
    off2 = Rr * 192 + B * T * 32 + 4 * B * Tp * (256 + 256 + 1024) + 4 * B * Tp * Tp + Rr * 64
    Z2 = host(sv[off2:off2 + Rr * 64]).reshape(B, T, 64, 64).transpose(0, 3, 1, 2)
    masks, c0 = {}, 0
    for i, m in enumerate([f"Queries.{h}" for h in range(4)] + [f"Keys.{h}" for h in range(4)] + [f"Values.{h}" for h in range(4)]):
        c = 4 if i < 8 else 16
        masks[m] = torch.from_numpy(Z[:, c0:c0 + c] >= 0)
        c0 += c
    masks["attn_concat_proj"] = torch.from_numpy(Z2 >= 0)
    o_ref, dx_ref, g_ref = G.module_grads(lambda a, b: G.mhsa2d_torch(a, b, masks=masks), x, p, dout)
    close("mhsa2d train forward", host(out), o_ref)
    got = {k: v.grad for k, v in mod.named_parameters()}
    assert set(got) == set(g_ref)
    # a key bias common to all keys shifts every score of a query row equally: softmax cancels it, the true gradient is 0
    gscale = max(float(np.abs(v).max()) for v in g_ref.values())
    # the PReLU slopes' gradients are scalars (sums with cancellation): judged against the largest of them, not their own size
    ascale = max(float(np.abs(v).max()) for k, v in g_ref.items() if k.endswith("act.weight"))

    def err_of(k):
        if ".norm.beta" in k and k.startswith("Keys"):
            return float(np.abs(host(got[k])).max()) / gscale
        if k.endswith("act.weight"):
            return float(np.abs(host(got[k]) - g_ref[k]).max()) / ascale
        return rel_err(host(got[k]), g_ref[k])
    errs = {k: err_of(k) for k in sorted(g_ref)}
    bad = {k: e for k, e in errs.items() if not e <= 2e-4}
    print(f"[parity] mhsa2d {len(g_ref)} parameter gradients: worst max-rel {max(errs.values()):.3e}")
    assert not bad, f"mhsa2d parameter gradients off: {bad}"
    close("mhsa2d dx", host(xt.grad), dx_ref, tol=2e-4)
    with torch.no_grad():
        close("mhsa2d eval vs train forward", host(mod(dev(x))), host(out))


def test_pool_and_tfar_combine_adjoints():
    """adaptive_avg_pool2d and the TFAR combine with their adjoints vs torch autograd (odd sizes: overlapping pooling windows,
    uneven nearest-neighbour fan-out)."""
    import rtfs_net_amd as R
    import torch.nn.functional as F
    L = R.layers
    for (H, W, Ho, Wo) in [(17, 129, 8, 64), (251, 9, 125, 4), (8, 64, 8, 64), (5, 7, 2, 3)]:
        x = rand((2, 3, H, W), H + W)
        dy = rand((2, 3, Ho, Wo), H)
        xt = dev(x).requires_grad_(True)
        y = L.adaptive_avg_pool(xt, (Ho, Wo))
        y.backward(dev(dy))
        xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        yr = F.adaptive_avg_pool2d(xr, (Ho, Wo))
        yr.backward(torch.tensor(dy, dtype=torch.float64))
        close(f"pool {H}x{W}->{Ho}x{Wo}", host(y), yr.detach().numpy(), tol=1e-6)
        close("pool adjoint", host(xt.grad), xr.grad.numpy(), tol=1e-6)
        le, ga, ge = rand((2, 3, H, W), 1), rand((2, 3, Ho, Wo), 2), rand((2, 3, Ho, Wo), 3)
        ts = [dev(a).requires_grad_(True) for a in (le, ga, ge)]
        o = L._TfarCombineFn.apply(*ts)
        do = rand((2, 3, H, W), 4)
        o.backward(dev(do))
        rs = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (le, ga, ge)]
        orf = rs[0] * F.interpolate(rs[1], size=(H, W), mode="nearest") + F.interpolate(rs[2], size=(H, W), mode="nearest")
        orf.backward(torch.tensor(do, dtype=torch.float64))
        close("tfar combine", host(o), orf.detach().numpy(), tol=1e-6)
        for t, r, nm in zip(ts, rs, ("dlocal", "dgate", "dglobal")):
            close(f"tfar combine {nm}", host(t.grad), r.grad.numpy(), tol=2e-6)


@pytest.mark.parametrize("shape,seed,with_res,cell", [((1, 256, 17, 129), 51, False, "SRU"), ((2, 256, 21, 129), 52, True, "SRU"),
                                                      ((1, 256, 17, 129), 53, True, "LSTM")])
def test_block_training_forward_backward(shape, seed, with_res, cell):
    """The whole RTFS block inside a training step (gateway, projection, 2-level pyramid, pooling, both sweeps, TF attention,
    three TFAR fusions, residual convolution; 139 parameter tensors) against the autograd oracle.  Tolerances are loose by design:
    PReLU's derivative jumps at 0 and a handful of the ~10^5 pre-activations land within fp32 rounding of 0, which moves single
    gradient elements by O(1) of their size (the per-module tests pin the kernels to ~1e-5 with the kinks taken out)."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    import copy
    if cell == "LSTM":  # the block with rnn_type LSTM in both sweeps (legacy yamls): stock nn.LSTM, fully reference arithmetic
        lm, lsd = lstm_model()
        p = {k: v.copy() for k, v in O._sub(lsd, "refinement_module.audio_net.blocks").items()}
        blk = copy.deepcopy(lm.refinement_module.audio_net.get_block(0)).train()
    else:
        p = {k: v.copy() for k, v in BLK.items()}
        blk = copy.deepcopy(model().refinement_module.audio_net.get_block(0)).train()
    x = rand(shape, seed)
    res = rand(shape, seed + 1) if with_res else None
    dout = rand(shape, seed + 100)
    xt = dev(x).requires_grad_(True)
    rt = dev(res).requires_grad_(True) if with_res else None
    out = blk(xt, rt)
    out.backward(dev(dout))
    xin = x + res if with_res else x
    o_ref, dx_ref, g_ref = G.module_grads(G.rtfs_block_torch, xin, p, dout)
    close("block train forward", host(out), o_ref)
    print(f"[parity] block dx: max-rel {rel_err(host(xt.grad), dx_ref):.3e} l2-rel {l2_rel(host(xt.grad), dx_ref):.3e}")
    assert l2_rel(host(xt.grad), dx_ref) <= 2e-3
    if with_res:
        assert torch.equal(rt.grad, xt.grad)
    got = {k: v.grad for k, v in blk.named_parameters()}
    assert set(got) == set(g_ref)
    gscale = {k: float(np.abs(v).max()) for k, v in g_ref.items()}
    l2 = {k: l2_rel(host(got[k]).reshape(g_ref[k].shape), g_ref[k]) for k in g_ref if gscale[k] > 1e-9 * max(gscale.values())}
    worst = sorted(l2.items(), key=lambda kv: -kv[1])[:3]
    print(f"[parity] block {len(g_ref)} parameter gradients: median l2-rel {np.median(list(l2.values())):.3e}, worst {worst}")
    assert np.median(list(l2.values())) <= 1e-4
    assert worst[0][1] <= 5e-2, worst
    with torch.no_grad():
        blk.eval()
        close("block eval vs train forward", host(blk(dev(x), dev(res) if with_res else None)), host(out))


@pytest.mark.parametrize("B,L,seed,smooth", [(2, 4096, 61, True), (2, 4096, 61, False), (1, 5000, 62, False)])
def test_encoder_bottleneck_s3_decoder_training(B, L, seed, smooth):
    """The separator without its refinement module inside a training step: STFT encoder (weight gradient), audio bottleneck, S^3 mask +
    complex multiply, decoder (iSTFT adjoint, ConvTranspose2d adjoints) against torch autograd over torch.stft / torch.istft."""
    from oracle import grad_oracle as G
    import copy
    m = copy.deepcopy(model()).train()
    names = ("encoder.", "audio_bottleneck.", "mask_generator.", "decoder.")
    p = {k: v.copy() for k, v in SD.items() if k.startswith(names)}
    if smooth:  # no activation kink within reach: PReLU slope 1 (derivative continuous), every mask pre-activation far above 0
        p["mask_generator.mask_generator.0.weight"][:] = 1.0
        p["mask_generator.mask_generator.1.full_layer.2.bias"] += 5.0
        with torch.no_grad():
            for k, v in m.named_parameters():
                if k in p:
                    v.copy_(torch.from_numpy(p[k]))
    wav = rand((B, L), seed) * 0.1
    dwav = rand((B, 1, L), seed + 1)
    wt = dev(wav)
    a0 = m.encoder(wt)
    a1 = m.audio_bottleneck(a0)
    sep = m.mask_generator(a1, a0)
    out = m.decoder(sep, wt.shape)
    out.backward(dev(dwav))
    o_ref, _, g_ref = G.module_grads(G.audio_chain_torch, wav, p, dwav)
    close("audio chain forward", host(out), o_ref)
    got = {k: v.grad for k, v in m.named_parameters() if k.startswith(names)}
    assert set(got) == set(g_ref)
    errs = {k: (rel_err(host(got[k]), g_ref[k]), l2_rel(host(got[k]), g_ref[k])) for k in sorted(g_ref)}
    for k, (e, l2) in errs.items():
        print(f"[parity] audio chain d {k}: max-rel {e:.3e} l2-rel {l2:.3e}")
    # the decoder's own gradient sees no activation kink; everything upstream of the mask's ReLU / PReLU does (2 x 10^6 pre-activations,
    # a few tens of them within fp32 rounding of 0: see test_block_training_forward_backward), hence the looser bound there
    assert errs["decoder.decoder.weight"][0] <= 2e-4
    assert max(l2 for _, l2 in errs.values()) <= 1e-2
    if smooth:
        assert max(e for e, _ in errs.values()) <= 2e-4, errs


@pytest.mark.parametrize("B,T,F,Tv,seed,bn_train", [(2, 9, 5, 4, 71, False), (1, 33, 129, 7, 72, False), (2, 20, 16, 20, 73, False),
                                                     (2, 9, 5, 4, 74, True), (3, 33, 129, 7, 75, True)])
def test_caf_training_forward_backward(B, T, F, Tv, seed, bn_train):
    """CAF cell inside a training step with its BatchNorm layers frozen (eval-mode statistics): grouped video-side convolutions +
    gLN, depthwise audio-side convolutions + BatchNorm (+ReLU), attention softmax, nearest up-sampling; all gradients (audio input,
    video input, 14 parameter tensors) against the autograd oracle."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    rng = np.random.default_rng(seed)
    p = {k: v.copy() for k, v in CELL.items()}
    for k in p:  # BatchNorm statistics and affines away from their init values
        if k.endswith("running_mean") or k.endswith(".bias"):
            p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
        if k.endswith("running_var") or k.endswith("3.weight") or k.endswith("norm.weight"):
            p[k] = (p[k] * (1 + 0.5 * rng.random(p[k].shape))).astype(np.float32)
    cell = R.layers.ATTNFusionCell(256, 512, kernel_size=4, is2d=True)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    cell = cell.cuda().train()
    for mod_ in cell.modules():  # bn_train False: frozen BatchNorm statistics (eval), everything else in train mode
        if isinstance(mod_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) and not bn_train:
            mod_.eval()
    a, v = rand((B, 256, T, F), seed), rand((B, 512, Tv), seed + 1)
    dout = rand((B, 256, T, F), seed + 2)
    at, vt = dev(a).requires_grad_(True), dev(v).requires_grad_(True)
    out = cell(at, vt)
    out.backward(dev(dout))
    pt = {k: torch.tensor(val, dtype=torch.float64, requires_grad=("running" not in k)) for k, val in p.items() if "num_batches" not in k}
    ar, vr = torch.tensor(a, dtype=torch.float64, requires_grad=True), torch.tensor(v, dtype=torch.float64, requires_grad=True)
    o_ref = G.caf_torch(ar, vr, pt, bn_train=bn_train)
    o_ref.backward(torch.tensor(dout, dtype=torch.float64))
    if bn_train:  # nn.BatchNorm2d's side effects: running statistics (momentum 0.1, unbiased variance) and the batch counter
        for pre in ("key_embed", "value_embed"):
            bnm = getattr(cell, pre).full_layer[3]
            close(f"caf {pre} running_mean", host(bnm.running_mean), pt[pre + ".full_layer.3.running_mean"].numpy(), tol=1e-5)
            close(f"caf {pre} running_var", host(bnm.running_var), pt[pre + ".full_layer.3.running_var"].numpy(), tol=1e-5)
            assert int(bnm.num_batches_tracked) == 1
    close("caf train forward", host(out), o_ref.detach().numpy())
    close("caf d audio", host(at.grad), ar.grad.numpy(), tol=2e-4)
    close("caf d video", host(vt.grad), vr.grad.numpy(), tol=2e-4)
    got = {k: v.grad for k, v in cell.named_parameters()}
    gscale = max(float(pt[k].grad.abs().max()) for k in got)
    for k, g in got.items():
        assert g is not None, k
        if k == "attention_embed.full_layer.3.norm.bias":  # constant over time: the softmax cancels it, the true gradient is 0
            assert float(np.abs(host(g)).max()) <= 1e-5 * gscale
            continue
        close(f"caf d {k}", host(g), pt[k].grad.numpy(), tol=2e-4)
    if not bn_train:
        close("caf inference vs training forward", host(cell.eval()(dev(a), dev(v))), host(out))  # eval mode: inference kernels, no graph


@pytest.mark.parametrize("kind,zero_mean,take_log,n", [("snr", True, True, 1), ("sisdr", True, True, 2), ("sdsdr", False, True, 3),
                                                       ("sisdr", True, False, 1)])
def test_pit_loss_gradient(kind, zero_mean, take_log, n):
    """Gradient of PITLossWrapper(PairwiseNegSDR) w.r.t. the estimates (HIP kernel) vs torch autograd over the reference's formula
    evaluated for the permutation the forward picked."""
    import rtfs_net_amd as R
    B, L = 3, 4000
    est, tgt = rand((B, n, L), 80 + n), rand((B, n, L), 90 + n)
    est = (0.7 * tgt[:, ::-1] + 0.5 * est).astype(np.float32) + 0.1  # correlated with a permuted target, non-zero mean
    loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR(kind, zero_mean=zero_mean, take_log=take_log), pit_from="pw_mtx")
    et = dev(est).requires_grad_(True)
    loss, reordered = loss_mod(et, dev(tgt), return_ests=True)
    (2.5 * loss).backward()
    _, _, perm = R.losses._pairwise(dev(est), dev(tgt), kind, zero_mean, take_log)
    perm = host(perm).astype(np.int64)
    er = torch.tensor(est, dtype=torch.float64, requires_grad=True)
    tr = torch.tensor(tgt, dtype=torch.float64)
    from oracle import grad_oracle as G
    # per target i the estimate perm[b][i]; mean over sources, then over the batch
    picked = torch.gather(er, 1, torch.from_numpy(perm)[:, :, None].expand(-1, -1, L))
    tot = sum(G.pit_loss_torch(picked[:, i:i + 1], tr[:, i:i + 1], kind, zero_mean, take_log) for i in range(n)) / n
    (2.5 * tot).backward()
    close(f"pit loss {kind}", np.array([float(loss)]), np.array([float(tot)]), tol=1e-5)
    close(f"pit loss {kind} d est", host(et.grad), er.grad.numpy(), tol=2e-5)


@pytest.mark.parametrize("smooth,full", [(True, False), (False, False), (True, True)])
def test_avnet_training_step_end_to_end(smooth, full):
    """AVNet.forward_train + PIT loss + backward through every audio-side module (encoder, bottleneck, shared RTFS block x R, CAF,
    S^3, decoder), frozen BatchNorm / VP block, against the float64 autograd oracle of the whole separator.  R = 2, 0.26 s input.
    Tolerances as in test_block_training_forward_backward (activation kinks); then one optimizer step through System."""
    import copy
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    m = copy.deepcopy(model(2))
    if full:  # everything trains: VP block differentiated (dropout 0 so the oracle can follow), BatchNorm on batch statistics
        m.train()
        ga = m.refinement_module.video_net.get_block(0).globalatt[0]
        ga.MHSA.dropout, ga.MHSA.dropout_layer.p, ga.FFN.dropout = 0.0, 0.0, 0.0
    else:
        m.freeze_for_finetune()
    if smooth:  # take the activation kinks out of reach (see test_encoder_bottleneck_s3_decoder_training): every PReLU slope 1, mask ReLU inactive
        with torch.no_grad():
            for k, v in m.named_parameters():
                if k.endswith("act.weight") or k.endswith("full_layer.4.weight") or k == "mask_generator.mask_generator.0.weight":
                    v.fill_(1.0)
            m.mask_generator.mask_generator[1].full_layer[2].bias.add_(5.0)
    B, L, Tv = 2, 4096, 7
    wav, emb = make_inputs(B, L, Tv, seed=5)
    tgt = rand((B, 1, L), 6) * 0.05
    wt, vt = dev(wav), dev(emb)
    p0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items() if "num_batches" not in k}  # before BatchNorm's running update
    out = m(wt, vt)
    loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
    loss = loss_mod(out, dev(tgt))
    loss.backward()
    if full:
        vp = emb
    else:
        with torch.no_grad():
            vp = host(m.refinement_module.video_net.get_block(0)(vt))
    skip = () if full else ("refinement_module.video_net.",)
    p = {k: v for k, v in p0.items() if not (skip and k.startswith(skip))}
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=("running" not in k and not k.endswith("pos_enc.pe"))) for k, v in p.items()}
    o_ref = G.avnet_torch(torch.tensor(wav, dtype=torch.float64), torch.tensor(vp, dtype=torch.float64), pt, 2, vp_trainable=full, bn_train=full)
    l_ref = G.pit_loss_torch(o_ref, torch.tensor(tgt, dtype=torch.float64), "snr")
    l_ref.backward()
    close("avnet train forward", host(out), o_ref.detach().numpy(), tol_l2=3e-5)  # bf16x3 GEMMs (2^-17 per product): measured 1.1e-5
    close("avnet loss", np.array([float(loss)]), np.array([float(l_ref)]), tol=1e-5)
    got = {k: v.grad for k, v in m.named_parameters() if v.requires_grad}
    assert set(got) == {k for k, v in pt.items() if v.requires_grad}
    gsc = max(float(v.grad.abs().max()) for v in pt.values() if v.requires_grad)
    l2 = {k: l2_rel(host(g), pt[k].grad.numpy()) for k, g in got.items() if float(pt[k].grad.abs().max()) > 1e-7 * gsc}
    worst = sorted(l2.items(), key=lambda kv: -kv[1])[:3]
    print(f"[parity] avnet {len(got)} parameter gradients: median l2-rel {np.median(list(l2.values())):.3e}, worst {worst}")
    if smooth:  # what is left is the accumulation of the bf16x3 GEMM error (~5e-6 per GEMM) over the two block applications and,
        # for the scalar PReLU slopes, cancellation in their sums
        assert np.median(list(l2.values())) <= 5e-4
        assert np.mean([v <= 2e-3 for v in l2.values()]) >= 0.95 and worst[0][1] <= 2e-2, worst
    else:  # ~10^2 of the 2 x 10^6 mask pre-activations flip side within fp32 rounding: every upstream gradient moves by ~1e-3
        assert np.median(list(l2.values())) <= 5e-3
        assert np.mean([v <= 2e-2 for v in l2.values()]) >= 0.9, worst
    # one optimizer step through System (core.py:119-123 + what Lightning does around it)
    opt = torch.optim.AdamW([q for q in m.parameters() if q.requires_grad], lr=1e-3)
    system = R.System(audio_model=m, loss_func={"train": loss_mod, "val": loss_mod}, optimizer=opt)
    before = float(loss)
    for _ in range(3):
        system.optimization_step((wt, dev(tgt), vt, None))
    m.eval()  # what Lightning does around validation_step
    after = float(system.validation_step((wt, dev(tgt), vt, None), 0)["val_loss"])
    print(f"[train] loss {before:.4f} -> {after:.4f} after 3 AdamW steps")
    assert after < before


@pytest.mark.parametrize("idx,shape,seed", [(0, (2, 64, 11, 13), 131), (1, (2, 64, 11, 13), 132), (0, (1, 64, 5, 64), 133), (1, (1, 64, 125, 9), 134)])
def test_dualpath_lstm_training_forward_backward(idx, shape, seed):
    """DualPathRNN with the LSTM cell inside a training step.  Its oracle is stock torch.nn.LSTM in float64, i.e. the reference's own
    arithmetic forward AND backward (rnn_layers.py:116-122): the fully pinned training parity for the dual-path module."""
    import json
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    from tests.util import ROOT
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "state_spec_R4_lstm.json")))
    p = O._sub(O._sub(make_state_dict(spec, 0), "refinement_module.audio_net.blocks"), f"globalatt.{idx}")
    dim = 4 if idx == 0 else 3
    mod = R.layers.DualPathRNN(64, 32, dim, kernel_size=8, stride=1, rnn_type="LSTM", num_layers=4, bidirectional=True)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    mod = mod.cuda().train()
    x, dout = rand(shape, seed), rand(shape, seed + 100)
    xt = dev(x).requires_grad_(True)
    out = mod(xt)
    out.backward(dev(dout))
    o_ref, dx_ref, g_ref = G.module_grads(lambda a, b: G.dualpath_lstm_torch(a, b, dim), x, p, dout)
    close("dualpath lstm train forward", host(out), o_ref)
    close("dualpath lstm dx", host(xt.grad), dx_ref, tol=2e-4)
    got = {k: v.grad for k, v in mod.named_parameters()}
    assert set(got) == set(g_ref)
    worst = 0.0
    for k in sorted(g_ref):
        e = rel_err(host(got[k]).reshape(g_ref[k].shape), g_ref[k])
        worst = max(worst, e)
        assert e <= 2e-4, (k, e)
    print(f"[parity] dualpath lstm {len(g_ref)} parameter gradients: worst max-rel {worst:.3e}")
    close("dualpath lstm eval vs train forward", host(mod.eval()(dev(x))), host(out))


@pytest.mark.parametrize("idx,shape,seed", [(0, (2, 64, 11, 13), 141), (1, (2, 64, 11, 13), 142), (1, (1, 64, 125, 9), 143)])
def test_dualpath_gru_forward_and_backward(idx, shape, seed):
    """DualPathRNN with the GRU cell (SURVEY 8 row a8': nn.LSTM/GRU): inference forward and training forward + backward on the GEMM +
    scan kernels, against stock torch.nn.GRU in float64 = the reference's own arithmetic."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    dim = 4 if idx == 0 else 3
    torch.manual_seed(seed)
    mod = R.layers.DualPathRNN(64, 32, dim, kernel_size=8, stride=1, rnn_type="GRU", num_layers=4, bidirectional=True)
    with torch.no_grad():
        mod.norm.gamma.add_(0.2 * torch.randn_like(mod.norm.gamma))
        mod.norm.beta.add_(0.2 * torch.randn_like(mod.norm.beta))
    p = {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}
    mod = mod.cuda().train()
    x, dout = rand(shape, seed), rand(shape, seed + 100)
    xt = dev(x).requires_grad_(True)
    out = mod(xt)
    out.backward(dev(dout))
    o_ref, dx_ref, g_ref = G.module_grads(lambda a, b: G.dualpath_lstm_torch(a, b, dim), x, p, dout)
    close("dualpath gru train forward", host(out), o_ref)
    close("dualpath gru dx", host(xt.grad), dx_ref, tol=2e-4)
    got = {k: v.grad for k, v in mod.named_parameters()}
    assert set(got) == set(g_ref)
    worst = max(rel_err(host(got[k]).reshape(g_ref[k].shape), g_ref[k]) for k in g_ref)
    print(f"[parity] dualpath gru {len(g_ref)} parameter gradients: worst max-rel {worst:.3e}")
    assert worst <= 2e-4
    with torch.no_grad():
        close("dualpath gru inference forward", host(mod.eval()(dev(x))), o_ref)


def test_avnet_gru_cells_inference_and_training_step():
    """The whole separator with rnn_type GRU in both sweeps: there is no fused kernel for it, eval() composes the unfused HIP kernels;
    checked against the float64 oracle (stock nn.GRU), then one training step through System."""
    import copy
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    conf = _conf(2)
    for k in ("layer_1", "layer_2"):
        conf["audio_params"]["layers"][k]["rnn_type"] = "GRU"
    torch.manual_seed(7)
    m = R.AVNet(print_macs=False, **conf).cuda().eval()
    B, L, Tv = 2, 4096, 7
    wav, emb = make_inputs(B, L, Tv, seed=5)
    with torch.no_grad():
        out = m(dev(wav), dev(emb))
        vp = host(m.refinement_module.video_net.get_block(0)(dev(emb)))
    p = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if "num_batches" not in k}
    pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in p.items()}
    o_ref = G.avnet_torch(torch.tensor(wav, dtype=torch.float64), torch.tensor(vp, dtype=torch.float64), pt, 2)
    close("avnet (GRU cells) inference", host(out), o_ref.numpy())
    m.train()
    loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
    system = R.System(audio_model=m, loss_func={"train": loss_mod, "val": loss_mod}, optimizer=torch.optim.AdamW(m.parameters(), lr=1e-3))
    tgt = dev(rand((B, 1, L), 6) * 0.05)
    l0 = float(system.optimization_step((dev(wav), tgt, dev(emb), None)))
    for _ in range(2):
        l1 = float(system.optimization_step((dev(wav), tgt, dev(emb), None)))
    print(f"[train] GRU-cell model: loss {l0:.4f} -> {l1:.4f}")
    assert l1 < l0


def test_sync_batchnorm_two_emulated_ranks():
    """SyncBatchNorm (train.py:145 sync_batchnorm=True): two ranks each hold half a batch; outputs, input gradients and running
    statistics must equal plain BatchNorm over the whole batch on one rank, and the two ranks' local parameter gradients must add up to
    the whole-batch ones.  The ranks are emulated in one process by running them one after the other three times with an all-reduce hook
    that first records each rank's contribution and then hands out the recorded sum (round 1 settles the forward statistics, round 2
    the backward sums under the right forward, round 3 is the synchronised step that is checked)."""
    import copy
    import rtfs_net_amd as R
    L = R.layers
    torch.manual_seed(3)
    ref = L.ConvNormAct(in_chan=256, out_chan=256, kernel_size=1, groups=256, norm_type="BatchNorm2d", act_type="ReLU", bias=False, is2d=True)
    with torch.no_grad():
        for k, v in ref.named_parameters():
            v.add_(0.3 * torch.randn_like(v))
    ref = ref.cuda().train()
    init = copy.deepcopy(ref.state_dict())
    x, dout = rand((4, 256, 9, 7), 1), rand((4, 256, 9, 7), 2)
    ranks = [torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(ref)).train() for _ in range(2)]
    assert isinstance(ranks[0].full_layer[3], torch.nn.SyncBatchNorm)
    xt = dev(x).requires_grad_(True)
    out = ref(xt)
    out.backward(dev(dout))
    state = {"rank": 0, "rec": {"fwd": [None, None], "bwd": [None, None]}, "use": {"fwd": False, "bwd": False}}

    def all_reduce(t):
        site = "fwd" if t.dtype == torch.float64 else "bwd"
        if state["use"][site]:
            t.copy_(state["rec"][site][0] + state["rec"][site][1])
        else:
            state["rec"][site][state["rank"]] = t.clone()
    old = (L._bn_world, L._bn_all_reduce)
    L._bn_world, L._bn_all_reduce = (lambda: 2), all_reduce
    res = [None, None]
    try:
        for rnd in range(3):
            state["use"] = {"fwd": rnd >= 1, "bwd": rnd >= 2}
            for r in range(2):
                state["rank"] = r
                ranks[r].load_state_dict(init)
                for q in ranks[r].parameters():
                    q.grad = None
                xr = dev(x[2 * r:2 * r + 2]).requires_grad_(True)
                o = ranks[r](xr)
                o.backward(dev(dout[2 * r:2 * r + 2]))
                res[r] = (host(o), host(xr.grad))
    finally:
        L._bn_world, L._bn_all_reduce = old
    close("syncbn output", np.concatenate([res[0][0], res[1][0]]), host(out), tol=1e-5)
    close("syncbn dx", np.concatenate([res[0][1], res[1][1]]), host(xt.grad), tol=1e-5)
    for (k, pr), (_, p0), (_, p1) in zip(ref.named_parameters(), ranks[0].named_parameters(), ranks[1].named_parameters()):
        close(f"syncbn d {k} (sum of the local gradients)", host(p0.grad) + host(p1.grad), host(pr.grad), tol=1e-5)
    for r in ranks:
        close("syncbn running_mean", host(r.full_layer[3].running_mean), host(ref.full_layer[3].running_mean), tol=1e-6)
        close("syncbn running_var", host(r.full_layer[3].running_var), host(ref.full_layer[3].running_var), tol=1e-6)


VP = O._sub(SD, "refinement_module.video_net.blocks")


@pytest.mark.parametrize("B,T,masked", [(2, 7, False), (3, 50, False), (2, 13, True)])
def test_video_mhsa_training_forward_backward(B, T, masked):
    """Video-side MultiHeadSelfAttention (LayerNorm, PE, nn.MultiheadAttention 8 heads, LayerNorm) inside a training step: with
    dropout 0 the module against the autograd oracle; with an explicit keep-mask the attention core alone against the same formula."""
    import rtfs_net_amd as R
    from oracle import grad_oracle as G
    p = {k: v.copy() for k, v in O._sub(VP, "globalatt.0.MHSA").items()}
    rng = np.random.default_rng(T)
    for k in p:
        if "norm" in k or "bias" in k:
            p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
    x, dout = rand((B, 64, T), 5), rand((B, 64, T), 6)
    if not masked:
        mod = R.layers.MultiHeadSelfAttention(64, n_head=8, dropout=0.0)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        mod = mod.cuda().train()
        xt = dev(x).requires_grad_(True)
        out = mod(xt)
        out.backward(dev(dout))
        pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=(k != "pos_enc.pe")) for k, v in p.items()}
        xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        o_ref = G.mhsa_1d_torch(xr, pt)
        o_ref.backward(torch.tensor(dout, dtype=torch.float64))
        close("video mhsa train forward", host(out), o_ref.detach().numpy())
        close("video mhsa dx", host(xt.grad), xr.grad.numpy(), tol=2e-4)
        for k, v in mod.named_parameters():
            close(f"video mhsa d {k}", host(v.grad), pt[k].grad.numpy(), tol=2e-4)
        close("video mhsa eval vs train forward", host(mod.eval()(dev(x))), host(out))
    else:
        qkv = rand((B, T, 192), 7)
        mask = (rng.random((B * 8, T, T)) >= 0.3).astype(np.float32) / 0.7
        do = rand((B, T, 64), 8)
        qt = dev(qkv).requires_grad_(True)
        o = R.layers._MhaCoreFn.apply(qt, 8, dev(mask))
        o.backward(dev(do))
        qr = torch.tensor(qkv, dtype=torch.float64, requires_grad=True)
        q, k_, v_ = [t.reshape(B, T, 8, 8).transpose(1, 2) for t in qr.split(64, -1)]
        a = torch.softmax(q @ k_.transpose(-1, -2) / np.sqrt(8.0), -1) * torch.tensor(mask, dtype=torch.float64).reshape(B, 8, T, T)
        o_ref = (a @ v_).transpose(1, 2).reshape(B, T, 64)
        o_ref.backward(torch.tensor(do, dtype=torch.float64))
        close("mha core (masked) forward", host(o), o_ref.detach().numpy(), tol=1e-5)
        close("mha core (masked) d qkv", host(qt.grad), qr.grad.numpy(), tol=1e-5)


@pytest.mark.parametrize("B,Tv,bn_train,seed", [(2, 50, False, 91), (3, 50, True, 92), (2, 17, True, 93)])
def test_vp_block_training_forward_backward(B, Tv, bn_train, seed):
    """The video-side VP block (1-D TDANetBlock depth 4, BatchNorm1d frozen or in train mode, GlobalAttention = MHSA + FFN; dropout 0)
    inside a training step against the autograd oracle: output, input gradient, 150 parameter tensors, BatchNorm running statistics."""
    import copy
    from oracle import grad_oracle as G
    blk = copy.deepcopy(model().refinement_module.video_net.get_block(0))
    blk.globalatt[0].MHSA.dropout = 0.0
    blk.globalatt[0].MHSA.dropout_layer.p = 0.0
    blk.globalatt[0].FFN.dropout = 0.0
    rng = np.random.default_rng(seed)
    with torch.no_grad():  # BatchNorm statistics / affines away from their init values
        for k, v in blk.state_dict().items():
            if k.endswith("running_mean") or (k.endswith(".bias") and "full_layer.3" in k):
                v.add_(torch.from_numpy(0.3 * rng.standard_normal(tuple(v.shape))).to(v))
            if k.endswith("running_var") or (k.endswith("full_layer.3.weight")):
                v.mul_(torch.from_numpy(1 + 0.5 * rng.random(tuple(v.shape))).to(v))
    p = {k: v.detach().cpu().numpy().copy() for k, v in blk.state_dict().items() if "num_batches" not in k}
    blk.train()
    if not bn_train:
        for m_ in blk.modules():
            if isinstance(m_, torch.nn.BatchNorm1d):
                m_.eval()
    x, dout = rand((B, 512, Tv), seed), rand((B, 512, Tv), seed + 1)
    xt = dev(x).requires_grad_(True)
    out = blk(xt)
    out.backward(dev(dout))
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=("running" not in k and k != "globalatt.0.MHSA.pos_enc.pe")) for k, v in p.items()}
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    o_ref = G.vp_block_torch(xr, pt, bn_train=bn_train)
    o_ref.backward(torch.tensor(dout, dtype=torch.float64))
    close("vp block train forward", host(out), o_ref.detach().numpy())
    print(f"[parity] vp block dx: max-rel {rel_err(host(xt.grad), xr.grad.numpy()):.3e} l2-rel {l2_rel(host(xt.grad), xr.grad.numpy()):.3e}")
    assert l2_rel(host(xt.grad), xr.grad.numpy()) <= 2e-3
    got = {k: v.grad for k, v in blk.named_parameters()}
    gsc = max(float(pt[k].grad.abs().max()) for k in got)
    l2 = {k: l2_rel(host(g), pt[k].grad.numpy()) for k, g in got.items() if float(pt[k].grad.abs().max()) > 1e-7 * gsc}
    worst = sorted(l2.items(), key=lambda kv: -kv[1])[:3]
    print(f"[parity] vp block {len(got)} parameter gradients: median l2-rel {np.median(list(l2.values())):.3e}, worst {worst}")
    assert np.median(list(l2.values())) <= 1e-4 and worst[0][1] <= 5e-2, worst  # PReLU kinks: see test_block_training_forward_backward
    if bn_train:
        sd = blk.state_dict()
        for k in p:
            if "running" in k:
                close(f"vp {k}", host(sd[k]), pt[k].numpy(), tol=1e-5)
    else:
        close("vp block inference kernel vs training forward", host(blk.eval()(dev(x))), host(out))


@pytest.mark.parametrize("case", ["eval", "train"])
def test_training_gradients_vs_reference_golden(case):
    """HIP training step directly against gradients the REFERENCE produced under torch autograd (tests/golden/grad_R2_L4096_B2.npz,
    oracle/make_golden_grad.py): loss, separated waveform, and all 264 parameter gradients.  case "eval": BatchNorm frozen (here: BatchNorm
    layers in eval mode inside a train()-mode model), "train": BatchNorm on batch statistics.  Bounds as in
    test_avnet_training_step_end_to_end (activation kinks move single fp32 gradient elements; the tight bounds are the per-module tests)."""
    import copy
    import zlib
    import rtfs_net_amd as R
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "grad_R2_L4096_B2.npz"))
    m = copy.deepcopy(model(2)).train()
    ga = m.refinement_module.video_net.get_block(0).globalatt[0]
    ga.MHSA.dropout, ga.MHSA.dropout_layer.p, ga.FFN.dropout = 0.0, 0.0, 0.0
    if case == "eval":
        for mod_ in m.modules():
            if isinstance(mod_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod_.eval()
    B, L, Tv = 2, 4096, 7
    wav, emb = make_inputs(B, L, Tv, seed=5)
    tgt = (0.05 * np.random.default_rng(6).standard_normal((B, 1, L))).astype(np.float32)
    out = m(dev(wav), dev(emb))
    loss = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")(out, dev(tgt))
    loss.backward()
    close(f"hip vs reference ({case}) separated waveform", host(out), gold[f"{case}/est"], tol_l2=3e-5)  # training-side bf16x3 GEMMs
    assert abs(float(loss) - float(gold[f"{case}/loss"])) <= 1e-5 * abs(float(gold[f"{case}/loss"]))
    errs = {}
    gscale = max(np.abs(gold[f"{case}/{k}"]).max() for k, _ in m.named_parameters())
    for k, p_ in m.named_parameters():
        g = host(p_.grad).reshape(-1).astype(np.float64)
        ref = gold[f"{case}/{k}"]
        if g.size > 4096:
            rs = np.random.RandomState(zlib.crc32(k.encode()) & 0x7FFFFFFF)
            g = g[rs.choice(g.size, 4096, replace=False).astype(np.int64)]
        if np.abs(ref).max() > 1e-6 * gscale:
            errs[k] = l2_rel(g, ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f"[parity] hip vs reference gradients ({case}), {len(errs)} tensors: median l2-rel {np.median(list(errs.values())):.3e}, worst {worst}")
    assert np.median(list(errs.values())) <= 5e-3
    assert np.mean([v <= 2e-2 for v in errs.values()]) >= 0.9, worst


def test_training_edge_shapes_and_gradient_accumulation():
    """Smallest shapes the sweeps allow (B = 1, T' = 9 > kernel 8, R = 1) through a training step, and autograd's accumulation contract:
    two forward/backward passes without zero_grad leave exactly twice the gradient of one."""
    import copy
    import rtfs_net_amd as R
    conf = _conf(1)
    mdl = R.AVNet(print_macs=False, **conf)
    mdl.load_state_dict({k: torch.from_numpy(v) for k, v in SD.items()})
    mdl = mdl.cuda().train()
    ga = mdl.refinement_module.video_net.get_block(0).globalatt[0]
    ga.MHSA.dropout, ga.MHSA.dropout_layer.p, ga.FFN.dropout = 0.0, 0.0, 0.0
    for mod_ in mdl.modules():  # frozen statistics: a second pass must not see different BatchNorm buffers
        if isinstance(mod_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod_.eval()
    wav, emb = make_inputs(1, 2176, 4, seed=9)  # T = 18 frames -> T' = 9
    tgt = dev(rand((1, 1, 2176), 10) * 0.05)
    loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("sisdr"), pit_from="pw_mtx")

    def step():
        out = mdl(dev(wav), dev(emb))
        assert out.shape == (1, 1, 2176) and torch.isfinite(out).all()
        loss_mod(out, tgt).backward()
    step()
    g1 = {k: v.grad.clone() for k, v in mdl.named_parameters()}
    assert all(torch.isfinite(g).all() for g in g1.values()) and sum(float(g.abs().sum()) for g in g1.values()) > 0
    step()
    gmax = max(float(g.abs().max()) for g in g1.values())
    dev_ = {k: float((v.grad - 2 * g1[k]).abs().max() / g1[k].abs().max()) for k, v in mdl.named_parameters()
            if float(g1[k].abs().max()) > 1e-6 * gmax}  # exactly-zero gradients (bias before BatchNorm, ...) are rounding noise
    worst = max(dev_.values())
    print(f"[parity] accumulated gradient vs 2x single: worst relative deviation {worst:.2e} ({max(dev_, key=dev_.get)})")
    assert worst <= 1e-3  # f32 atomics make single gradients reproducible only to rounding


DDP_WORKER = r"""
import json, os, sys, copy
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RTFS_ROOT"])
import rtfs_net_amd as R
from rtfs_net_amd.configs import RTFS4_AUDIONET
from oracle.params import make_inputs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)   # two ranks share the one GPU of the box: gloo moves the CUDA tensors
conf = copy.deepcopy(RTFS4_AUDIONET); conf["audio_params"]["repeats"] = 2
torch.manual_seed(0)
m = R.AVNet(print_macs=False, **conf).cuda().train()
ga = m.refinement_module.video_net.get_block(0).globalatt[0]
ga.MHSA.dropout, ga.MHSA.dropout_layer.p, ga.FFN.dropout = 0.0, 0.0, 0.0           # no RNG in the comparison
loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
system = R.System(audio_model=m, loss_func={"train": loss_mod, "val": loss_mod})
if world > 1:
    system.convert_sync_batchnorm()
    m = system.audio_model
B = 2
wav, emb = make_inputs(B, 4096, 7, seed=5)
tgt = 0.05 * np.random.default_rng(6).standard_normal((B, 1, 4096)).astype(np.float32)
sl = slice(rank * B // world, (rank + 1) * B // world)
batch = tuple(torch.from_numpy(a[sl]).cuda() for a in (wav, tgt, emb)) + (None,)
batch = (batch[0], batch[1], batch[2], None)
loss = system.training_step(batch, 0)["loss"]
loss.backward()
n = system.allreduce_gradients()
torch.cuda.synchronize()
if rank == 0:
    g = {k: p.grad.detach().double().cpu().numpy() for k, p in m.named_parameters()}
    np.savez(os.environ["RTFS_OUT"], loss=float(loss), n=n, **{k.replace(".", "/"): v for k, v in g.items()})
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def test_two_process_training_step_matches_single_process(tmp_path):
    """Data-parallel training step for real: two processes (gloo, sharing the box's GPU), each with half the batch, SyncBatchNorm and the
    flattened gradient all-reduce, against one process with the whole batch: every averaged parameter gradient must agree."""
    import socket
    import subprocess
    import sys
    from tests.util import ROOT

    def run(world, out):
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       RTFS_ROOT=ROOT, RTFS_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, "-c", DDP_WORKER], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for p_ in procs:
            o_, e_ = p_.communicate(timeout=600)
            assert p_.returncode == 0, e_[-3000:]
        return dict(np.load(out))
    one = run(1, tmp_path / "one.npz")
    two = run(2, tmp_path / "two.npz")
    assert int(two["n"]) == sum(v.size for k, v in two.items() if k not in ("loss", "n")) > 700000  # one flattened buffer, every parameter
    gmax = max(np.abs(v).max() for k, v in one.items() if k not in ("loss", "n"))
    # a bias in front of a BatchNorm and an additive constant in front of a softmax have an exactly zero gradient: noise, not compared
    errs = {k: l2_rel(two[k], one[k]) for k in one if k not in ("loss", "n") and np.abs(one[k]).max() > 1e-6 * gmax}
    assert abs(float(two["loss"]) - float(one["loss"])) < 10.0  # rank 0's loss is its own half batch's: not comparable beyond sanity
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f"[parity] 2-process vs 1-process gradients over {len(errs)} tensors: median l2-rel {np.median(list(errs.values())):.3e}, worst {worst}")
    assert np.median(list(errs.values())) <= 1e-4 and worst[0][1] <= 2e-2, worst


@pytest.mark.parametrize("shape,with_res,seed", [((2, 5, 7, 256), True, 151), ((1, 33, 129, 256), False, 152), ((3, 1, 3, 64), True, 153)])
def test_gateway_one_pass_forward_backward(shape, with_res, seed):
    """rtfs_gateway_forward_train_f32 / rtfs_gateway_backward_f32 (PReLU(depthwise 1x1(x + x_res)) on rows, tdanet.py:30-38,106-108) against
    torch autograd in float64 on the same values; the kink is kept out of reach (|z| > 1e-3) so the bound can be tight (1e-5)."""
    import rtfs_net_amd as R
    from rtfs_net_amd import layers as L
    C = shape[-1]
    cna = R.layers.ConvNormAct(C, C, 1, groups=C, act_type="PReLU", is2d=True).cuda().train()
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        cna.full_layer[2].weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, (C, 1, 1, 1)).astype(np.float32)))
        cna.full_layer[2].bias.copy_(torch.from_numpy(rng.uniform(-0.2, 0.2, C).astype(np.float32)))
        cna.full_layer[4].weight.fill_(0.3)
    x = rand(shape, seed)
    res = rand(shape, seed + 1) if with_res else None
    w64 = cna.full_layer[2].weight.detach().double().cpu().reshape(C)
    b64 = cna.full_layer[2].bias.detach().double().cpu()
    z0 = w64 * (torch.from_numpy(x).double() + (torch.from_numpy(res).double() if with_res else 0)) + b64
    x = np.where(np.abs(z0.numpy()) < 1e-3, x + 0.01, x).astype(np.float32)  # move the few pre-activations near 0 away from the kink
    dout = rand(shape, seed + 2)
    xt = dev(x).requires_grad_(True)
    rt = dev(res).requires_grad_(True) if with_res else None
    out = L.gateway_train(cna, xt, rt)
    assert out is not None
    out.backward(dev(dout))
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    r64 = torch.tensor(res, dtype=torch.float64, requires_grad=True) if with_res else None
    w = w64.clone().requires_grad_(True)
    b = b64.clone().requires_grad_(True)
    sl = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    z = w * (x64 + r64 if with_res else x64) + b
    ref = torch.where(z >= 0, z, sl * z)
    ref.backward(torch.tensor(dout, dtype=torch.float64))
    close("gateway forward", host(out), ref.detach().numpy(), 1e-6)
    close("gateway dx", host(xt.grad), x64.grad.numpy(), 1e-5)
    if with_res:
        close("gateway dx_res", host(rt.grad), r64.grad.numpy(), 1e-5)
    close("gateway dweight", host(cna.full_layer[2].weight.grad).reshape(C), w.grad.numpy(), 1e-5)
    close("gateway dbias", host(cna.full_layer[2].bias.grad), b.grad.numpy(), 1e-5)
    close("gateway dslope", host(cna.full_layer[4].weight.grad), sl.grad.numpy(), 1e-5)
