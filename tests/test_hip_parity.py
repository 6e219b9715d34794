"""GPU parity: the HIP path (through the C ABI via the nn.Module mirror) vs the CPU oracle on the same
seeded inputs, and vs the golden vectors captured from the reference.  Tolerance: the north_star's
1e-4 relative (fp32), read as max|got-ref| / max|ref| per tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import rtfs_oracle as O
from oracle.params import make_inputs, make_state_dict
from tests.util import check_probe, l2_rel, load_golden, rand, rel_err, spec_R4

pytestmark = pytest.mark.gpu

TOL = 1e-4
SD = make_state_dict(spec_R4(), 0)
BLK = O._sub(SD, "refinement_module.audio_net.blocks")
CELL = O._sub(SD, "refinement_module.crossmodal_fusion.fusion_module.audio_lstm")


def _conf(repeats=4):
    import copy
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    c = copy.deepcopy(RTFS4_AUDIONET)
    c["audio_params"]["repeats"] = repeats
    return c


_MODELS = {}


def model(repeats=4):
    import rtfs_net_amd as R
    if repeats not in _MODELS:
        m = R.AVNet(print_macs=False, **_conf(repeats))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in SD.items()})
        _MODELS[repeats] = m.cuda().eval()
    return _MODELS[repeats]


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def close(name, got, ref, tol=TOL, tol_l2=None):
    """Two readings of the north_star's "1e-4 rel", both asserted: max|d| / max|ref| <= tol, and the stricter aggregate
    ||d||_2 / ||ref||_2 <= tol / 10 (measured 1e-7 ... 1e-6 on every case)."""
    e, l2 = rel_err(got, ref), l2_rel(got, ref)
    if tol_l2 is None:  # the inference path's bar; tests that pass their own (training-side, bf16x3) tolerance get it for both readings
        tol_l2 = tol / 10 if tol == TOL else tol
    print(f"[parity] {name}: max-rel {e:.3e}  l2-rel {l2:.3e}  shape {tuple(np.shape(ref))}")
    assert np.isfinite(np.asarray(got)).all(), f"{name}: non-finite output"
    assert e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"
    assert l2 <= tol_l2, f"{name}: l2-rel err {l2:.3e} > {tol_l2:.1e}"


def test_library_loaded_is_hip():
    from rtfs_net_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.rtfs_version()


def test_mfma_f16_fragment_layout():
    """Exact integer data, asymmetric operands: catches any row/col or k-order mistake in the assumed lane maps."""
    import ctypes
    from rtfs_net_amd import _lib
    lib = _lib.load()
    i, k, j = np.arange(32)[:, None], np.arange(16), np.arange(32)[None, :]
    A = ((i * 3 + k[None, :] * 5) % 7 - 3).astype(np.float32)
    Bm = ((k[:, None] * 2 + j * 11) % 5 - 2).astype(np.float32)
    dA, dB = dev(A), dev(Bm)
    dD = torch.zeros(32 * 32 + 32 * 16, device="cuda")
    _lib.check(lib.rtfs_selftest_mfma_f16(_lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dD), _lib.stream_of(dD)), "selftest")
    assert np.array_equal(host(dD)[:1024].reshape(32, 32), A @ Bm)
    # f16 subnormal operands must not be flushed (the split-precision low parts of small weights are subnormal)
    A2 = np.zeros((32, 16), np.float32); B2 = np.zeros((16, 32), np.float32)
    A2[3, 5] = 2.0 ** -20; B2[5, 7] = 1024.0
    A2[9, 2] = 3.0; B2[2, 11] = 2.0 ** -22
    _lib.check(lib.rtfs_selftest_mfma_f16(_lib.ptr(dev(A2)), _lib.ptr(dev(B2)), _lib.ptr(dD), _lib.stream_of(dD)), "selftest")
    D2 = host(dD)[:1024].reshape(32, 32)
    rt = host(dD)[1024:].reshape(32, 16)
    print("[selftest] f16 subnormal: cvt round trip", rt[3, 5], "(want", 2.0 ** -20, ") products", D2[3, 7], D2[9, 11])
    # On this toolchain the f32->f16 conversion flushes f16 subnormals (round trip gives 0): that is why every weight
    # image is pre-scaled by 2^8 -- the low parts of the split weights would otherwise vanish.
    assert rt[3, 5] in (0.0, 2.0 ** -20)


def test_encoder():
    m = model()
    x = rand((2, 2048), 101, 0.07)
    a0 = host(m.encoder(dev(x)))
    ref, _ = O.stft_encoder(x, O._sub(SD, "encoder"))
    close("encoder", a0, ref)
    check_probe(load_golden("mod_encoder"), "out", a0, TOL)


def test_encoder_stats_and_ragged_length():
    m = model()
    x = rand((3, 5000), 7, 0.07)
    a0, st = m.encoder(dev(x), return_stats=True)
    a0, st = host(a0), host(st)
    ref, _ = O.stft_encoder(x, O._sub(SD, "encoder"))
    close("encoder L=5000", a0, ref)
    want = np.stack([ref.reshape(3, -1).astype(np.float64).sum(1), (ref.reshape(3, -1).astype(np.float64) ** 2).sum(1)], 1)
    assert np.allclose(st[:, 1], want[:, 1], rtol=1e-5)
    assert np.allclose(st[:, 0], want[:, 0], rtol=1e-3, atol=1e-3 * np.abs(ref).sum() / 3 * 1e-3)


def test_audio_bn():
    m = model()
    x = rand((2, 256, 9, 129), 102)
    y = host(m.audio_bottleneck(dev(x)))
    close("audio_bn", y, O.conv_norm_act(x, O._sub(SD, "audio_bottleneck"), pre_norm="gLN", pre_act="ReLU"))
    check_probe(load_golden("mod_audio_bn"), "out", y, TOL)


@pytest.mark.parametrize("name,idx,dim,shape,seed", [
    ("mod_dualpath_f", 0, 4, (2, 64, 12, 64), 103),
    ("mod_dualpath_t", 1, 3, (2, 64, 12, 64), 103),
    ("mod_dualpath_t_min", 1, 3, (1, 64, 8, 64), 113),
])
def test_dualpath(name, idx, dim, shape, seed):
    m = model()
    x = rand(shape, seed)
    y = host(m.refinement_module.audio_net.blocks.globalatt[idx](dev(x)))
    close(name, y, O.dualpath_rnn(x, O._sub(BLK, f"globalatt.{idx}"), dim))
    check_probe(load_golden(name), "out", y, TOL)


@pytest.mark.parametrize("dim,shape", [(4, (1, 64, 125, 64)), (3, (1, 64, 125, 64)), (3, (1, 64, 250, 64)), (4, (2, 64, 9, 64)), (3, (3, 64, 37, 64)),
                                       (3, (1, 64, 251, 64)), (3, (2, 64, 400, 64)), (3, (1, 64, 700, 64))])
def test_dualpath_full_size_rows(dim, shape):
    """2 s (T'=125) and 4 s (T'=250) sweep lengths, odd tile remainders; 250 = the longest sweep of the fused kernel, 251 / 400 / 700 = past it
    (6.4 s, 11 s: the unfused GEMM + scan + GEMM kernels; the reference has no length limit, rnn_layers.py:136-162)."""
    m = model()
    x = rand(shape, 11 + dim)
    idx = 0 if dim == 4 else 1
    y = host(m.refinement_module.audio_net.blocks.globalatt[idx](dev(x)))
    close(f"dualpath dim{dim} {shape}", y, O.dualpath_rnn(x, O._sub(BLK, f"globalatt.{idx}"), dim))


@pytest.mark.parametrize("Ls", [15, 23, 24, 39, 40, 47, 70, 71, 72, 87, 103, 104, 119, 134, 135, 136, 160, 198, 199, 200, 231, 249, 250])
def test_dualpath_sweep_lengths(Ls):
    """Sweep lengths L = Ls - 7 around every tile / time-part / kernel boundary of the fused sweep (8, 16, 17, 32, 33, 40, 63, 64 | 65, 80, 96, 97,
    112, 127, 128 | 129, 153, 191-193, 224, 242, 243 in the four-part variant of the 4 s shapes): the write-back stores whole groups of steps and lets the ones behind the sequence end land in scratch rows, so every
    remainder is its own case.  Both directions and both kernels (sequence pair, L <= 64; single sequence, L <= 128; generation 2 above)."""
    m = model()
    x = rand((1, 64, Ls, 10), 500 + Ls)
    y = host(m.refinement_module.audio_net.blocks.globalatt[1](dev(x)))
    close(f"dualpath sweep length {Ls - 7}", y, O.dualpath_rnn(x, O._sub(BLK, "globalatt.1"), 3))


def test_generation2_sweep_kernels_still_pass():
    """The generation-2 sweep kernels stay in the library as the fall-back for tensors spanning >= 4 GB (generation 3 addresses with 32-bit
    offsets) and behind RTFS_SWEEP_GEN2=1; the switch is read once per process, so the length sweep is re-run in a child process."""
    import subprocess, sys
    if os.environ.get("RTFS_SWEEP_GEN2"):
        pytest.skip("already inside the generation-2 run")
    env = dict(os.environ, RTFS_SWEEP_GEN2="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_parity.py", "-q", "-x", "-m", "gpu", "-k", "test_dualpath_sweep_lengths or test_dualpath_full_size_rows"],
                        cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stdout[-3000:]


def test_exact_f32_gemm_switch_still_passes():
    """RTFS_GEMM_F32=1 (read once per process) routes the pointwise / dual-path GEMMs to the exact v_mfma_f32_32x32x2_f32 kernels and the
    separator to its unfused, contiguous-row sequence - the A/B path for the f16x3 arithmetic and for the padded-row kernels.  Re-run the
    end-to-end and block cases in a child process."""
    import subprocess, sys
    if os.environ.get("RTFS_GEMM_F32"):
        pytest.skip("already inside the exact-f32 run")
    env = dict(os.environ, RTFS_GEMM_F32="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, "-m", "pytest", "tests/test_hip_parity.py", "-q", "-x", "-m", "gpu", "-k",
                         "test_end_to_end_vs_golden or test_rtfs_block or test_audio_bn or test_s3 or test_decoder"],
                        cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stdout[-3000:]


def test_dualpath_short_axis_raises():
    m = model()
    with pytest.raises(ValueError):
        m.refinement_module.audio_net.blocks.globalatt[1](dev(rand((1, 64, 5, 64), 1)))


def test_sru_operator():
    """Operator-level seam: sru.SRU.forward(x (L,N,512)) -> (h (L,N,64), c)."""
    m = model()
    sru = m.refinement_module.audio_net.blocks.globalatt[0].rnn
    x = rand((19, 5, 512), 21)
    h, _ = sru(dev(x))
    p = O._sub(BLK, "globalatt.0")
    close("sru operator", host(h), O.sru_forward(x, O._sru_layers(p)))


@pytest.mark.parametrize("shape,seed", [((2, 64, 12, 64), 103), ((1, 64, 125, 64), 5), ((1, 64, 250, 64), 6), ((2, 64, 33, 64), 8),
                                        ((1, 64, 256, 64), 9), ((1, 64, 257, 64), 10), ((2, 64, 400, 64), 11)])
def test_mhsa2d(shape, seed):
    m = model()
    x = rand(shape, seed)
    y = host(m.refinement_module.audio_net.blocks.globalatt[2](dev(x)))
    close(f"mhsa2d {shape}", y, O.mhsa2d(x, O._sub(BLK, "globalatt.2")))
    if seed == 103:
        check_probe(load_golden("mod_mhsa2d"), "out", y, TOL)


def test_tfar():
    m = model()
    blk = m.refinement_module.audio_net.blocks
    glo = rand((2, 64, 8, 64), 105)
    loc = rand((2, 64, 17, 129), 104)
    y = host(blk.fusion_layers[0](dev(loc), dev(glo)))
    close("tfar up", y, O.injection_multi_sum(loc, glo, O._sub(BLK, "fusion_layers.0")))
    check_probe(load_golden("mod_tfar_up"), "out", y, TOL)
    loc = rand((2, 64, 8, 64), 106)
    y = host(blk.fusion_layers[1](dev(loc), dev(glo)))
    close("tfar same", y, O.injection_multi_sum(loc, glo, O._sub(BLK, "fusion_layers.1")))
    check_probe(load_golden("mod_tfar_same"), "out", y, TOL)


def test_caf():
    m = model()
    a, v = rand((2, 256, 17, 129), 107), rand((2, 512, 7), 108)
    y, v2 = m.refinement_module.crossmodal_fusion.fusion_module(dev(a), dev(v))
    y = host(y)
    close("caf", y, O.caf(a, v, CELL))
    check_probe(load_golden("mod_caf"), "out", y, TOL)


@pytest.mark.parametrize("name,tv,seed", [("mod_vp50", 50, 109), ("mod_vp7", 7, 110)])
def test_vp_block(name, tv, seed):
    m = model()
    v = rand((2, 512, tv), seed)
    y = host(m.refinement_module.video_net.blocks(dev(v)))
    check_probe(load_golden(name), "out", y, TOL)


def test_s3():
    m = model()
    r, a0 = rand((2, 256, 9, 129), 111), rand((2, 256, 9, 129), 112)
    y = host(m.mask_generator(dev(r), dev(a0)))
    close("s3", y, O.s3_mask(r, a0, O._sub(SD, "mask_generator")))
    check_probe(load_golden("mod_s3"), "out", y, TOL)


def test_decoder():
    m = model()
    x = rand((2, 1, 256, 17, 129), 114, 0.3)
    y = host(m.decoder(dev(x), torch.Size([2, 2048])))
    close("decoder", y, O.stft_decoder(x, O._sub(SD, "decoder"), 2048))
    check_probe(load_golden("mod_decoder"), "out", y, TOL)


def test_rtfs_block():
    m = model()
    x = rand((1, 256, 17, 129), 115)
    y = host(m.refinement_module.audio_net.blocks(dev(x)))
    close("rtfs block", y, O.rtfs_block(x, BLK))
    check_probe(load_golden("mod_rtfs_block"), "out", y, TOL)


def test_rtfs_block_with_residual_input_and_batch():
    m = model()
    x, r = rand((3, 256, 20, 129), 31), rand((3, 256, 20, 129), 32)
    y = host(m.refinement_module.audio_net.blocks(dev(x), dev(r)))
    close("rtfs block(x + res)", y, O.rtfs_block(x + r, BLK))


@pytest.mark.parametrize("name,R,B,L,Tv,seed", [
    ("e2e_R4_L4096_B2", 4, 2, 4096, 7, 1),
    ("e2e_R4_L5000_B3", 4, 3, 5000, 8, 3),
    ("e2e_R6_L8000_B2", 6, 2, 8000, 13, 5),
    ("e2e_R12_L8000_B1", 12, 1, 8000, 13, 4),
    ("e2e_R4_L32000_B1", 4, 1, 32000, 50, 2),
])
@pytest.mark.parametrize("fused", [True, False])
def test_end_to_end_vs_golden(name, R, B, L, Tv, seed, fused):
    g = load_golden(name)
    m = model(R)
    m.fused = fused
    wav, emb = make_inputs(B, L, Tv, seed)
    try:
        out = host(m(dev(wav), dev(emb)))
    finally:
        m.fused = True
    close(f"{name} fused={fused}", out, g["out"])


def test_end_to_end_4s_vs_oracle():
    """BASELINE config 5 shape (4 s @16 kHz: T = 501, T' = 250, Tv = 100): exercises the unpaired 4 s sweep kernel,
    250-key attention and the 100-frame VP block inside the fused separator.  No reference vector at this size
    (the oracle is the checker), R = 2 keeps the CPU side short."""
    m = model(4)
    import copy, rtfs_net_amd as R
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    c = copy.deepcopy(RTFS4_AUDIONET); c["audio_params"]["repeats"] = 2
    m2 = R.AVNet(print_macs=False, **c)
    m2.load_state_dict(m.state_dict())
    m2 = m2.cuda().eval()
    wav, emb = make_inputs(1, 64000, 100, 21)
    out = host(m2(dev(wav), dev(emb)))
    ref = O.avnet_forward(wav, emb, SD, repeats=2)
    close("e2e 4 s R=2", out, ref)


@pytest.mark.parametrize("L,Tv", [(16700, 27), (33100, 52), (20000, 31)])
def test_end_to_end_odd_lengths_vs_oracle(L, Tv):
    """Utterance lengths whose down-sampled time axis (T' = 65, 129, 78 frames) falls between the shapes of the recipes - 65 and 129 sit in the
    windows (65-71, 129-135) where the sweep kernels used to be picked by L instead of Ls and left the last positions unwritten.  No reference
    vector at these sizes (the oracle is the checker); R = 2 keeps the CPU side short."""
    m = model(4)
    import copy, rtfs_net_amd as R
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    c = copy.deepcopy(RTFS4_AUDIONET); c["audio_params"]["repeats"] = 2
    m2 = R.AVNet(print_macs=False, **c)
    m2.load_state_dict(m.state_dict())
    m2 = m2.cuda().eval()
    wav, emb = make_inputs(1, L, Tv, 77 + Tv)
    out = host(m2(dev(wav), dev(emb)))
    ref = O.avnet_forward(wav, emb, SD, repeats=2)
    close(f"e2e {L} samples R=2", out, ref)


@pytest.mark.parametrize("T", [63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193])
def test_end_to_end_frame_counts_around_tile_boundaries(T):
    """Utterances of T STFT frames (T' = T // 2 low-resolution frames) around every row-band / tile boundary the kernels use: 64-row bands
    of the depthwise family at full resolution (T = 64, 128, 192) and at low resolution (T' = 32, 64, 96), 64 x 64 transpose tiles (T' = 63,
    64, 65: small path / exact / overlapping last tile), the sweep kernels' 32-row tiles and 64-position limit (T' = 64, 65), attention
    frame groups of 8.  The oracle is the checker (no reference vector at these sizes); R = 2 keeps the CPU side short."""
    m = model(4)
    import copy, rtfs_net_amd as R
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    c = copy.deepcopy(RTFS4_AUDIONET); c["audio_params"]["repeats"] = 2
    m2 = R.AVNet(print_macs=False, **c)
    m2.load_state_dict(m.state_dict())
    m2 = m2.cuda().eval()
    L = (T - 1) * 128 + 17
    Tv = max(2, round(L / 16000 * 25))
    wav, emb = make_inputs(1, L, Tv, 900 + T)
    out = host(m2(dev(wav), dev(emb)))
    ref = O.avnet_forward(wav, emb, SD, repeats=2)
    close(f"e2e T = {T} frames", out, ref)


# ---------------- rnn_type LSTM: every number in these vectors is the reference's own arithmetic (stock nn.LSTM)
_LSTM = {}


def lstm_model():
    import copy, json, os
    import rtfs_net_amd as R
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    from tests.util import GOLDEN
    if "m" not in _LSTM:
        c = copy.deepcopy(RTFS4_AUDIONET)
        for k in ("layer_1", "layer_2"):
            c["audio_params"]["layers"][k]["rnn_type"] = "LSTM"
        sd = make_state_dict(json.load(open(os.path.join(GOLDEN, "state_spec_R4_lstm.json"))), 0)
        m = R.AVNet(print_macs=False, **c)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        _LSTM["m"], _LSTM["sd"] = m.cuda().eval(), sd
    return _LSTM["m"], _LSTM["sd"]


@pytest.mark.parametrize("name,idx,dim", [("mod_dualpath_f_lstm", 0, 4), ("mod_dualpath_t_lstm", 1, 3)])
def test_dualpath_lstm(name, idx, dim):
    m, sd = lstm_model()
    x = rand((2, 64, 12, 64), 103)
    y = host(m.refinement_module.audio_net.blocks.globalatt[idx](dev(x)))
    close(name, y, O.dualpath_rnn(x, O._sub(O._sub(sd, "refinement_module.audio_net.blocks"), f"globalatt.{idx}"), dim))
    check_probe(load_golden(name), "out", y, TOL)


def test_dualpath_lstm_full_rows():
    m, sd = lstm_model()
    for dim, shape in [(4, (1, 64, 125, 64)), (3, (1, 64, 125, 64))]:
        x = rand(shape, 40 + dim)
        idx = 0 if dim == 4 else 1
        y = host(m.refinement_module.audio_net.blocks.globalatt[idx](dev(x)))
        close(f"dualpath lstm dim{dim}", y, O.dualpath_rnn(x, O._sub(O._sub(sd, "refinement_module.audio_net.blocks"), f"globalatt.{idx}"), dim))


@pytest.mark.parametrize("name,B,L,Tv,seed", [("e2e_lstm_R4_L4096_B2", 2, 4096, 7, 1), ("e2e_lstm_R4_L32000_B1", 1, 32000, 50, 2)])
def test_end_to_end_lstm_vs_reference(name, B, L, Tv, seed):
    m, _ = lstm_model()
    g = load_golden(name)
    wav, emb = make_inputs(B, L, Tv, seed)
    out = host(m(dev(wav), dev(emb)))
    close(name, out, g["out"])


def test_input_rank_variants_match():
    m = model()
    wav, emb = make_inputs(1, 4096, 7, 9)
    a = host(m(dev(wav), dev(emb)))
    b = host(m(dev(wav[0]), dev(emb)))
    c = host(m(dev(wav[:, None, :]), dev(emb)))
    # not bitwise: the gLN statistics are accumulated with f64 atomics whose arrival order varies run to run
    assert a.shape == (1, 1, 4096) and b.shape == a.shape and c.shape == a.shape
    assert rel_err(b, a) <= 2e-6 and rel_err(c, a) <= 2e-6


def test_batch_split_option_changes_nothing_but_the_schedule():
    """rtfs_set_batch_split(2): two half batches as independent chains on forked streams - every mixture as in the single chain (not
    bitwise: the gLN statistics are f64 atomics), odd batch sizes split unevenly, parts below 8 mixtures are not split."""
    import rtfs_net_amd as R
    m = model()
    try:
        for B in (17, 16, 9):
            wav, emb = make_inputs(B, 4096, 7, 300 + B)
            R.set_batch_split(1)
            a = host(m(dev(wav), dev(emb)))
            R.set_batch_split(2)
            b = host(m(dev(wav), dev(emb)))
            R.set_batch_split(3)
            c = host(m(dev(wav), dev(emb)))
            for i in range(B):
                assert rel_err(b[i], a[i]) <= 2e-6 and rel_err(c[i], a[i]) <= 2e-6, (B, i)
    finally:
        R.set_batch_split(0)


def test_batch_split_per_call_argument():
    """rtfs_separator_forward_ex_f32(..., split): the schedule as an argument of the call (AVNet.batch_split) instead of the process-wide
    setter - consecutive calls of one model with different settings, each mixture as in the single chain."""
    import rtfs_net_amd as R
    R.set_batch_split(0)
    m = model()
    wav, emb = make_inputs(17, 4096, 7, 411)
    try:
        m.batch_split = 1
        a = host(m(dev(wav), dev(emb)))
        m.batch_split = 2
        b = host(m(dev(wav), dev(emb)))
    finally:
        m.batch_split = 0
    for i in range(17):
        assert rel_err(b[i], a[i]) <= 2e-6, i
    lib = R._lib.load()
    assert lib.rtfs_separator_workspace_bytes_ex(32, 4096, 7, 2) > lib.rtfs_separator_workspace_bytes_ex(32, 4096, 7, 1)
    assert lib.rtfs_separator_workspace_bytes_ex(32, 4096, 7, 9) == 0  # out of range


def test_launches_per_small_batch_forward():
    """A batch-1 forward is a chain of dependent launches at dispatch latency (test.py:51-62, inference.py:55 and infer_any_video.py:86 all run
    one utterance per step): the library counts its launches, and the count of one RTFS-Net-4 forward must not creep up - 70 at the end of
    round 3: 4 blocks x 15 (steps 3 and 14 share a launch) + STFT + encoder statistics + bottleneck / head + 3 boundaries + tail + iSTFT + CAF
    video + VP block."""
    import rtfs_net_amd as R
    lib = R._lib.load()
    m = model()
    wav, emb = make_inputs(1, 4096, 7, 77)
    w, e = dev(wav), dev(emb)
    m(w, e)
    torch.cuda.synchronize()
    n0 = lib.rtfs_debug_launch_count()
    m(w, e)
    torch.cuda.synchronize()
    n = lib.rtfs_debug_launch_count() - n0
    print(f"[launches] {n} per batch-1 forward")
    assert 60 <= n <= 72, n


def test_forward_can_be_captured_in_a_hip_graph():
    """torch.cuda.graph capture of AVNet.forward (library-internal side streams are forked from / joined into the capturing stream by events,
    the workspace comes from torch's graph pool) and replay on new input values written into the captured tensors."""
    m = model()
    wav, emb = make_inputs(2, 8000, 12, 61)
    w, e = dev(wav), dev(emb)
    ref = host(m(w, e))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m(w, e)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(w, e)
    g.replay()
    assert rel_err(host(out), ref) <= 2e-6
    wav2, emb2 = make_inputs(2, 8000, 12, 62)
    w.copy_(dev(wav2)); e.copy_(dev(emb2))
    g.replay()
    assert rel_err(host(out), host(m(w, e))) <= 2e-6


def test_two_host_threads_on_two_streams():
    """The re-entrancy note of include/rtfs_amd.h: two host threads drive the separator concurrently, each on its own stream (each gets its
    own internal side streams); every result equals the one the same input gives alone."""
    import threading
    m = model()
    inputs = [make_inputs(3, 6000 + 512 * k, 9 + k, 400 + k) for k in range(2)]
    alone = [host(m(dev(w), dev(e))) for w, e in inputs]   # also warms the parameter packs (the pack caches are filled once, then read)
    results, errors = [None, None], []

    def run(k):
        try:
            s = torch.cuda.Stream()
            w, e = dev(inputs[k][0]), dev(inputs[k][1])
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(6):
                    out = m(w, e)
            s.synchronize()
            results[k] = host(out)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(2):
        assert rel_err(results[k], alone[k]) <= 2e-6, k


def test_batch_independence_property():
    """Size-independent property at bench batch size: every mixture of a batch separates exactly as it does alone
    (all norms are per-sample in eval mode)."""
    m = model()
    wav, emb = make_inputs(4, 8000, 13, 12)
    full = host(m(dev(wav), dev(emb)))
    for i in (0, 3):
        one = host(m(dev(wav[i:i + 1]), dev(emb[i:i + 1])))
        assert rel_err(full[i:i + 1], one) <= 2e-6


@pytest.mark.parametrize("B", [3, 5, 7, 13, 17, 31, 33])
def test_every_mixture_of_an_odd_batch_equals_its_own_run(B):
    """Odd batch sizes (grids that do not divide evenly over 8 XCDs / 256 CUs, persistent workgroups with a ragged last tile, pixel tiles that
    straddle samples): every mixture of the batch against its batch-1 run.  Short utterances keep it cheap; no oracle needed."""
    m = model()
    wav, emb = make_inputs(B, 3000 + 128 * (B % 5), 5, 700 + B)
    full = host(m(dev(wav), dev(emb)))
    for i in range(B):
        one = host(m(dev(wav[i:i + 1]), dev(emb[i:i + 1])))
        assert rel_err(full[i:i + 1], one) <= 2e-6, (B, i)


def test_relocated_package_runs_the_forward(tmp_path):
    """train.py:95 copies the models package into the experiment directory, test.py:33-36 imports the copy as `<exp>.models` and calls
    AVNet.from_pretrain(...)(mix, mouth_emb): the copied package (with its librtfs_amd.so) must produce the reference's output."""
    import importlib
    import shutil
    import sys
    from tests.util import ROOT
    exp = tmp_path / "exp_gpu"
    shutil.copytree(os.path.join(ROOT, "rtfs-net_amd"), exp / "models", ignore=shutil.ignore_patterns("csrc", "__pycache__"))
    (exp / "__init__.py").write_text("")
    sys.path.append(str(tmp_path))
    try:
        models = importlib.import_module("exp_gpu.models")
        ck = tmp_path / "best_model.pth"
        torch.save(model().serialize(), ck)
        m = models.AVNet.from_pretrain(str(ck), **_conf(4)).cuda().eval()
        wav, emb = make_inputs(2, 4096, 7, 1)
        with torch.no_grad():
            out = host(m(dev(wav), dev(emb)))
        close("relocated package e2e_R4_L4096_B2", out, load_golden("e2e_R4_L4096_B2")["out"])
    finally:
        sys.path.remove(str(tmp_path))
        for k in [k for k in sys.modules if k.startswith("exp_gpu")]:
            del sys.modules[k]


def test_cpu_tensor_rejected():
    m = model()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4096), torch.zeros(1, 512, 7))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_pit_loss_vs_oracle_and_reference(k):
    """rtfs_pit_pairwise_sdr_f32 through the reference-named classes: pairwise matrix vs the oracle (1e-4 dB) and vs the
    reference's own outputs (golden), best permutation + reordered estimates bit-exact."""
    from oracle import loss_oracle as LO
    import rtfs_net_amd as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_cases.npz"))
    est, tgt = LO.make_loss_case(k)
    e, t = torch.from_numpy(est).cuda(), torch.from_numpy(tgt).cuda()
    for kind in ("snr", "sisdr", "sdsdr"):
        pw = R.losses.PairwiseNegSDR(kind)(e, t).cpu().numpy()
        ref = LO.pairwise_neg_sdr(est, tgt, kind)
        assert np.abs(pw - ref).max() < 1e-4
        assert np.abs(pw - g[f"c{k}_{kind}_pw"]).max() < 1e-4
        mean, reo = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR(kind), pit_from="pw_mtx")(e, t, return_ests=True)
        assert abs(float(mean) - float(g[f"c{k}_{kind}_mean"])) < 1e-4
        perm = g[f"c{k}_{kind}_perm"]
        want = np.stack([est[b][perm[b]] for b in range(est.shape[0])])
        assert np.array_equal(reo.cpu().numpy(), want)
    with pytest.raises(TypeError):
        R.losses.PairwiseNegSDR("snr")(e[:, :1], t)


@pytest.mark.gpu
@pytest.mark.parametrize("k,B,T", [(0, 1, 3), (1, 2, 5)])
def test_video_frontend_vs_oracle_and_reference(k, B, T):
    """rtfs_video_frontend_f32 through the reference-named FRCNNVideoModel: vs the numpy oracle and vs the reference
    module's own output (golden); tolerance 1e-4 relative (f16x3 split arithmetic, f32 accumulate)."""
    from oracle import video_oracle as V
    import rtfs_net_amd as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "video_cases.npz"))
    sd = V.make_video_state_dict(0)
    m = R.FRCNNVideoModel(print_macs=False)
    assert [(n, tuple(p.shape)) for n, p in m.state_dict().items()] == [(n, tuple(s)) for n, s in V.video_state_spec()]
    m.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()})
    m = m.cuda().eval()
    x = V.make_video_input(B, T, k)
    from rtfs_net_amd import _lib
    lib = _lib.load()
    _lib.workspace(lib.rtfs_video_workspace_bytes(B, T), torch.device("cuda", 0)).fill_(0xFF)  # poison: scratch is NaN on entry
    with torch.no_grad():
        y = m(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = V.video_frontend(x, sd)
    assert y.shape == (B, 512, T)
    assert rel_err(y, ref) < 1e-4
    assert rel_err(y, g[f"c{k}_out"]) < 1e-4
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 2, 64, 64, device="cuda"))


@pytest.mark.gpu
def test_system_forward_and_validation_step():
    """System.forward (core.py:78-92): raw lips -> video front-end -> separator, and validation_step -> PIT loss, all on
    the HIP path; checked against the composition of the three oracles."""
    import copy
    from oracle import loss_oracle as LO, video_oracle as V
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    import rtfs_net_amd as R
    B, L, Tv = 1, 4096, 3
    sd = make_state_dict(spec_R4(), 0)
    vsd = V.make_video_state_dict(0)
    audio = R.AVNet(print_macs=False, **copy.deepcopy(RTFS4_AUDIONET))
    audio.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    video = R.FRCNNVideoModel(print_macs=False)
    video.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vsd.items()})
    loss = {"val": R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")}
    s = R.System(audio_model=audio, video_model=video, loss_func=loss).cuda().eval()
    wav, _ = make_inputs(B, L, Tv, 5)
    lips = V.make_video_input(B, Tv, 9)
    tgt = np.random.RandomState(3).randn(B, 1, L).astype(np.float32) * 0.05  # the test config has n_src = 1
    with torch.no_grad():
        est = s(torch.from_numpy(wav).cuda(), torch.from_numpy(lips).cuda())
        out = s.validation_step((torch.from_numpy(wav).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(lips).cuda(), None), 0)
    emb = V.video_frontend(lips, vsd)
    ref = O.avnet_forward(wav, emb, sd, repeats=4)
    assert rel_err(est.cpu().numpy(), ref) < 1e-4
    ref_loss = LO.pit_from_pw_mtx(LO.pairwise_neg_sdr(ref, tgt, "snr"))[0]
    assert abs(float(out["val_loss"]) - float(ref_loss)) < 1e-3


@pytest.mark.parametrize("scale", [1e-5, 1e-3, 30.0, 1e4])
def test_input_amplitude_range(scale):
    """Range safety of the f16x3 split-precision GEMMs: the same mixture at very quiet (x 1e-5, x 1e-3) and very hot (x 30, x 1e4)
    amplitudes.  Everything behind the first gLN is amplitude-free, but the encoder output a0 (and with it the S3 product and the
    decoder's taps GEMM, whose B operand is the separated spectrum) scales with the waveform: un-scaled, its f16 low parts are flushed below
    ~1e-3 (measured 8e-6 at x 1e-3, fp16-grade below) and the products overflow f16 above ~1e5; the fused S3 + taps kernel therefore
    normalises the encoder rows by the power of two nearest 1 / rms(a0) of the mixture.  Oracle on the scaled input, same 1e-4 / 1e-5 bar."""
    m = model()
    wav, emb = make_inputs(2, 4096, 7, 1)
    wav = (wav * scale).astype(np.float32)
    out = host(m(dev(wav), dev(emb)))
    ref = O.avnet_forward(wav, emb, SD, repeats=4)
    close(f"input x {scale:g}", out, ref)


@pytest.mark.parametrize("log2_scale", [-10, 10])
def test_block_with_rescaled_intermediate(log2_scale):
    """Range safety inside the RTFS block: the gateway's depthwise weight and bias scaled by 2^-10 / 2^+10 put the block's residual stream and
    the projection GEMM's operand at ~1e-3 / ~1e3 instead of the O(1) of every other fixture (what a trained checkpoint may do); the
    projection output is re-normalised by the gLN behind it, so a precision loss there is amplified back to O(1).  Oracle with the same
    parameters, same 1e-4 / 1e-5 bar."""
    import copy
    import rtfs_net_amd as R
    sc = 2.0 ** log2_scale
    pre = "refinement_module.audio_net.blocks.gateway.full_layer.2."
    sd = dict(SD)
    sd[pre + "weight"] = SD[pre + "weight"] * np.float32(sc)
    sd[pre + "bias"] = SD[pre + "bias"] * np.float32(sc)
    m = R.AVNet(print_macs=False, **_conf(4))
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    x = rand((2, 256, 17, 129), 215)
    y = host(m.refinement_module.audio_net.blocks(dev(x)))
    close(f"rtfs block, gateway x 2^{log2_scale}", y, O.rtfs_block(x, O._sub(sd, "refinement_module.audio_net.blocks")))
