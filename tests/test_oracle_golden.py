"""CPU: pins oracle/rtfs_oracle.py against vectors captured from the reference itself
(oracle/make_golden.py).  Everything but the SRU cell is the reference's own code in those
vectors; the SRU arithmetic is the oracle's (third-party package absent -> parity unpinned)."""
import os

import numpy as np
import pytest

from oracle import rtfs_oracle as O
from oracle.params import make_inputs, make_state_dict
from tests.util import check_probe, load_golden, rand, rel_err, spec_R4

TOL = 2e-5  # fp32 reference vs float64-accumulating oracle

SD = make_state_dict(spec_R4(), 0)
BLK = O._sub(SD, "refinement_module.audio_net.blocks")
VBLK = O._sub(SD, "refinement_module.video_net.blocks")
CELL = O._sub(SD, "refinement_module.crossmodal_fusion.fusion_module.audio_lstm")


def test_encoder():
    a0, _ = O.stft_encoder(rand((2, 2048), 101, 0.07), O._sub(SD, "encoder"))
    check_probe(load_golden("mod_encoder"), "out", a0, TOL)


def test_audio_bn():
    y = O.conv_norm_act(rand((2, 256, 9, 129), 102), O._sub(SD, "audio_bottleneck"), pre_norm="gLN", pre_act="ReLU")
    check_probe(load_golden("mod_audio_bn"), "out", y, TOL)


@pytest.mark.parametrize("name,idx,dim", [("mod_dualpath_f", 0, 4), ("mod_dualpath_t", 1, 3)])
def test_dualpath(name, idx, dim):
    y = O.dualpath_rnn(rand((2, 64, 12, 64), 103), O._sub(BLK, f"globalatt.{idx}"), dim)
    check_probe(load_golden(name), "out", y, TOL)


def test_dualpath_min_and_short():
    y = O.dualpath_rnn(rand((1, 64, 8, 64), 113), O._sub(BLK, "globalatt.1"), 3)
    check_probe(load_golden("mod_dualpath_t_min"), "out", y, TOL)
    with pytest.raises(ValueError):  # reference: nn.Unfold raises for a sweep axis < kernel_size
        O.dualpath_rnn(rand((1, 64, 5, 64), 1), O._sub(BLK, "globalatt.1"), 3)


def test_mhsa2d():
    y = O.mhsa2d(rand((2, 64, 12, 64), 103), O._sub(BLK, "globalatt.2"))
    check_probe(load_golden("mod_mhsa2d"), "out", y, TOL)


def test_tfar():
    glo = rand((2, 64, 8, 64), 105)
    y = O.injection_multi_sum(rand((2, 64, 17, 129), 104), glo, O._sub(BLK, "fusion_layers.0"))
    check_probe(load_golden("mod_tfar_up"), "out", y, TOL)
    y = O.injection_multi_sum(rand((2, 64, 8, 64), 106), glo, O._sub(BLK, "fusion_layers.1"))
    check_probe(load_golden("mod_tfar_same"), "out", y, TOL)


def test_caf():
    y = O.caf(rand((2, 256, 17, 129), 107), rand((2, 512, 7), 108), CELL)
    check_probe(load_golden("mod_caf"), "out", y, TOL)


@pytest.mark.parametrize("name,tv,seed", [("mod_vp50", 50, 109), ("mod_vp7", 7, 110)])
def test_vp_block(name, tv, seed):
    y = O.vp_block(rand((2, 512, tv), seed), VBLK)
    check_probe(load_golden(name), "out", y, TOL)


def test_s3():
    y = O.s3_mask(rand((2, 256, 9, 129), 111), rand((2, 256, 9, 129), 112), O._sub(SD, "mask_generator"))
    check_probe(load_golden("mod_s3"), "out", y, TOL)


def test_decoder():
    y = O.stft_decoder(rand((2, 1, 256, 17, 129), 114, 0.3), O._sub(SD, "decoder"), 2048)
    check_probe(load_golden("mod_decoder"), "out", y, TOL)


def test_rtfs_block_with_internals():
    g = load_golden("mod_rtfs_block")
    out, ints = O.rtfs_block(rand((1, 256, 17, 129), 115), BLK, return_internals=True)
    for k in ["residual", "x_enc", "d0", "d1", "g_f", "g_t", "g_att", "xf0", "xf1"]:
        check_probe(g, k, ints[k], TOL, "block.")
    check_probe(g, "out", out, TOL)


@pytest.mark.parametrize("name,R,B,L,Tv,seed", [
    ("e2e_R4_L4096_B2", 4, 2, 4096, 7, 1),
    ("e2e_R4_L5000_B3", 4, 3, 5000, 8, 3),
    ("e2e_R6_L8000_B2", 6, 2, 8000, 13, 5),
    ("e2e_R12_L8000_B1", 12, 1, 8000, 13, 4),
])
def test_end_to_end(name, R, B, L, Tv, seed):
    g = load_golden(name)
    wav, emb = make_inputs(B, L, Tv, seed)
    out, ints = O.avnet_forward(wav, emb, SD, repeats=R, return_internals=True)
    for k in ["a0", "a1", "refined", "sep"]:
        check_probe(g, k, ints[k], 5 * TOL if R <= 6 else 2e-4, name + ".")
    e = rel_err(out, g["out"])
    assert e <= (5 * TOL if R <= 6 else 2e-4), f"{name}: out rel err {e:.3e}"


@pytest.mark.slow
def test_end_to_end_full_size_2s():
    g = load_golden("e2e_R4_L32000_B1")
    wav, emb = make_inputs(1, 32000, 50, 2)
    out = O.avnet_forward(wav, emb, SD, repeats=4)
    e = rel_err(out, g["out"])
    assert e <= 1e-4, f"2 s end-to-end rel err {e:.3e}"


# ---------------- rnn_type LSTM: the one configuration whose vectors are 100 % reference arithmetic (stock nn.LSTM)
def _lstm_sd():
    import json, os
    from tests.util import GOLDEN
    return make_state_dict(json.load(open(os.path.join(GOLDEN, "state_spec_R4_lstm.json"))), 0)


@pytest.mark.parametrize("name,idx,dim", [("mod_dualpath_f_lstm", 0, 4), ("mod_dualpath_t_lstm", 1, 3)])
def test_dualpath_lstm(name, idx, dim):
    blk = O._sub(_lstm_sd(), "refinement_module.audio_net.blocks")
    y = O.dualpath_rnn(rand((2, 64, 12, 64), 103), O._sub(blk, f"globalatt.{idx}"), dim)
    check_probe(load_golden(name), "out", y, TOL)


def test_end_to_end_lstm():
    g = load_golden("e2e_lstm_R4_L4096_B2")
    wav, emb = make_inputs(2, 4096, 7, 1)
    out, ints = O.avnet_forward(wav, emb, _lstm_sd(), repeats=4, return_internals=True)
    for k in ["a0", "a1", "refined", "sep"]:
        check_probe(g, k, ints[k], 5 * TOL, "lstm.")
    assert rel_err(out, g["out"]) <= 5 * TOL


def test_torch_ops_composition_matches_numpy_oracle():
    """oracle/torch_cpu.py (stock torch CPU ops under the oracle's call graph, SURVEY 8d's second CPU baseline) agrees with
    the numpy restatement."""
    from oracle import rtfs_oracle as O, torch_cpu as TC
    from oracle.params import load_spec, make_inputs, make_state_dict
    sd = make_state_dict(load_spec("state_spec_R4.json"), 0)
    wav, emb = make_inputs(2, 4096, 7, 3)
    a = O.avnet_forward(wav, emb, sd, repeats=2)
    b = TC.avnet_forward(wav, emb, sd, repeats=2)
    assert np.abs(a - b).max() / np.abs(a).max() < 2e-5
    assert O.pointwise.__module__ == "oracle.rtfs_oracle"  # primitives restored


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_loss_oracle_matches_reference(k):
    """oracle/loss_oracle.py vs the reference's PairwiseNegSDR + PITLossWrapper outputs (tests/golden/loss_cases.npz,
    generated by oracle/make_golden_loss.py); tolerance 1e-4 dB (the reference computes in float32)."""
    from oracle import loss_oracle as LO
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_cases.npz"))
    est, tgt = LO.make_loss_case(k)
    for kind in ("snr", "sisdr", "sdsdr"):
        pw = LO.pairwise_neg_sdr(est, tgt, kind)
        assert np.abs(pw - g[f"c{k}_{kind}_pw"]).max() < 1e-4
        mean, _, perm, _ = LO.pit_from_pw_mtx(pw, est)
        assert abs(mean - g[f"c{k}_{kind}_mean"]) < 1e-4
        assert np.array_equal(perm, g[f"c{k}_{kind}_perm"])


@pytest.mark.parametrize("k,B,T", [(0, 1, 3), (1, 2, 5)])
def test_video_oracle_matches_reference(k, B, T):
    """oracle/video_oracle.py vs the reference's FRCNNVideoModel (ResNet-18 trunk, PReLU, eval) outputs and internal
    tensors (tests/golden/video_cases.npz, generated by oracle/make_golden_video.py); tolerance 2e-5 relative."""
    import os
    from oracle import video_oracle as V
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "video_cases.npz"))
    y, it = V.video_frontend(V.make_video_input(B, T, k), V.make_video_state_dict(0), return_internals=True)
    assert np.abs(y - g[f"c{k}_out"]).max() / np.abs(g[f"c{k}_out"]).max() < 2e-5
    assert np.abs(it["stem"][:, ::8, ::3, ::3] - g[f"c{k}_stem"]).max() < 2e-5
    for li in (1, 2, 3, 4):
        ref = g[f"c{k}_layer{li}"]
        assert np.abs(it[f"layer{li}"][:, ::16] - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("case", ["eval", "train"])
def test_gradient_oracle_against_reference_autograd(case):
    """oracle/grad_oracle.py (the float64 torch restatement the HIP backward kernels are tested against) vs golden gradients produced by
    the REFERENCE's own modules and loss under torch autograd (oracle/make_golden_grad.py -> tests/golden/grad_R2_L4096_B2.npz):
    loss, output and every one of the 264 parameter gradients, with BatchNorm in eval mode and on batch statistics."""
    import zlib
    import torch
    from oracle import grad_oracle as G
    from oracle.params import make_inputs, make_state_dict
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "grad_R2_L4096_B2.npz"))
    sd = make_state_dict(spec_R4(), 0)
    B, Ls, Tv = 2, 4096, 7
    wav, emb = make_inputs(B, Ls, Tv, seed=5)
    tgt = (0.05 * np.random.default_rng(6).standard_normal((B, 1, Ls))).astype(np.float32)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=("running" not in k and not k.endswith("pos_enc.pe")))
          for k, v in sd.items() if "num_batches" not in k}
    out = G.avnet_torch(torch.tensor(wav, dtype=torch.float64), torch.tensor(emb, dtype=torch.float64), pt, 2, vp_trainable=True,
                        bn_train=(case == "train"))
    loss = G.pit_loss_torch(out, torch.tensor(tgt, dtype=torch.float64), "snr")
    loss.backward()
    # (the reference's float32 Hann / positional-encoding buffers cast to float64 vs float64 ones here: ~1e-8)
    assert abs(float(loss) - float(gold[f"{case}/loss"])) <= 1e-7 * abs(float(gold[f"{case}/loss"]))
    assert rel_err(out.detach().numpy(), gold[f"{case}/est"]) <= 1e-6
    names = [k for k, v in pt.items() if v.requires_grad]
    assert len(names) == 264
    gscale = max(np.abs(gold[f"{case}/{k}"]).max() for k in names)
    worst = 0.0
    for k in names:
        g = pt[k].grad.numpy().reshape(-1)
        ref = gold[f"{case}/{k}"]
        if g.size > 4096:
            rs = np.random.RandomState(zlib.crc32(k.encode()) & 0x7FFFFFFF)
            idx = rs.choice(g.size, 4096, replace=False).astype(np.int64)
            assert abs(np.sqrt((g ** 2).sum()) - gold[f"{case}/{k}#l2"]) <= 1e-6 * max(gold[f"{case}/{k}#l2"], 1e-12 * gscale), k
            g = g[idx]
        err = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-9 * gscale)
        worst = max(worst, err)
        assert err <= 1e-6, (k, err)
    print(f"gradient oracle vs reference autograd ({case}): worst relative error {worst:.2e} over {len(names)} tensors")


@pytest.mark.slow
def test_end_to_end_lstm_ragged_5s():
    """Past BASELINE config 5's length, 100 % reference arithmetic (stock nn.LSTM): L = 85000 (ragged), T' = 332, Tv = 133
    (oracle/make_golden_sizes.py)."""
    g = load_golden("e2e_lstm_R4_L85000_B1")
    wav, emb = make_inputs(1, 85000, 133, 35)
    out = O.avnet_forward(wav, emb, _lstm_sd(), repeats=4)
    e = rel_err(out, g["out"])
    assert e <= 1e-4, f"5.3 s LSTM end-to-end rel err {e:.3e}"
