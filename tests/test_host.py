"""CPU tests: host logic, state_dict / pack contracts, and that the C-ABI library loads and exports every
symbol include/rtfs_amd.h declares (no compute calls without a GPU)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from tests.util import ROOT, spec_R4

# config/lrs2_RTFSNet_4_layer.yaml `audionet` section, restated as data so the tests do not read /root/reference
RTFS4_AUDIONET = {
    "n_src": 1,
    "pretrained_vout_chan": 512,
    "video_bn_params": {"kernel_size": -1},
    "audio_bn_params": {"pre_norm_type": "gLN", "pre_act_type": "ReLU", "out_chan": 256, "kernel_size": 1, "is2d": True},
    "enc_dec_params": {"encoder_type": "STFTEncoder", "decoder_type": "STFTDecoder", "win": 256, "hop_length": 128, "out_chan": 256,
                       "kernel_size": 3, "stride": 1, "bias": False, "act_type": None, "norm_type": None},
    "audio_params": {"audio_net": "TDANet", "hid_chan": 64, "kernel_size": 4, "stride": 2, "norm_type": "gLN", "act_type": "PReLU",
                     "upsampling_depth": 2, "repeats": 4, "shared": True, "is2d": True,
                     "layers": {
                         "layer_1": {"layer_type": "DualPathRNN", "hid_chan": 32, "dim": 4, "kernel_size": 8, "stride": 1, "rnn_type": "SRU",
                                     "num_layers": 4, "bidirectional": True},
                         "layer_2": {"layer_type": "DualPathRNN", "hid_chan": 32, "dim": 3, "kernel_size": 8, "stride": 1, "rnn_type": "SRU",
                                     "num_layers": 4, "bidirectional": True},
                         "layer_3": {"layer_type": "MultiHeadSelfAttention2D", "dim": 3, "n_freqs": 64, "n_head": 4, "hid_chan": 4,
                                     "act_type": "PReLU", "norm_type": "LayerNormalization4D"}}},
    "video_params": {"video_net": "TDANet", "hid_chan": 64, "kernel_size": 3, "stride": 2, "norm_type": "BatchNorm1d", "act_type": "PReLU",
                     "upsampling_depth": 4, "repeats": 1, "shared": True, "is2d": False,
                     "layers": {"layer_1": {"layer_type": "GlobalAttention", "ffn_name": "FeedForwardNetwork", "kernel_size": 3, "n_head": 8,
                                            "dropout": 0.1}}},
    "fusion_params": {"fusion_type": "ATTNFusion", "fusion_shared": True, "kernel_size": 4, "is2d": True},
    "mask_generation_params": {"mask_generator_type": "MaskGenerator", "mask_act": "ReLU", "RI_split": True, "is2d": True},
}


def build(repeats=4):
    import copy
    import rtfs_net_amd as R
    c = copy.deepcopy(RTFS4_AUDIONET)
    c["audio_params"]["repeats"] = repeats
    return R.AVNet(print_macs=False, **c).eval()


def test_config_matches_reference_yaml_if_present():
    path = "/root/reference/config/lrs2_RTFSNet_4_layer.yaml"
    if not os.path.exists(path):
        pytest.skip("reference not mounted")
    import yaml
    assert yaml.safe_load(open(path))["audionet"] == RTFS4_AUDIONET


def test_header_symbols_exported_and_bound():
    """Every function declared in include/rtfs_amd.h is exported by the .so and has a ctypes signature."""
    from rtfs_net_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rtfs_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtfs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.load().rtfs_version()


def test_state_dict_keys_shapes_order_match_reference():
    m = build()
    spec = spec_R4()
    sd = m.state_dict()
    assert [k for k, _, _ in spec] == list(sd.keys())
    for k, shape, dtype in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
        assert str(sd[k].dtype) == "torch." + dtype, k
    assert sum(p.numel() for p in m.parameters()) == 739952


def test_same_parameter_set_for_every_repeat_count():
    assert list(build(12).state_dict().keys()) == list(build(4).state_dict().keys())


def test_pack_sizes_match_library():
    from rtfs_net_amd import _lib
    lib = _lib.load()
    m = build()
    rm = m.refinement_module
    blk = rm.audio_net.blocks
    mods = {_lib.PACK_ENCODER: m.encoder, _lib.PACK_AUDIO_BN: m.audio_bottleneck, _lib.PACK_BLOCK: blk, _lib.PACK_DUALPATH: blk.globalatt[0],
            _lib.PACK_ATTENTION: blk.globalatt[2], _lib.PACK_TFAR: blk.fusion_layers[0],
            _lib.PACK_CAF: rm.crossmodal_fusion.fusion_module.audio_lstm, _lib.PACK_S3: m.mask_generator, _lib.PACK_DECODER: m.decoder}
    for kind, mod in mods.items():
        assert lib.rtfs_pack_floats(kind) == mod.pack().numel(), kind


def test_pack_layout_and_cache_invalidation():
    m = build()
    dp = m.refinement_module.audio_net.blocks.globalatt[0]
    p = dp.pack()
    assert p is dp.pack()  # cached
    # layer-1 projection: (64,192) columns (dir*32+j)*3+m are re-laid to (dir*32+j)*4+m with a zero 4th gate column
    off = 64 + 64 + 512 * 256
    w1 = dp.rnn.rnn_lst[1].weight.detach()
    got = p[off:off + 64 * 256].view(64, 64, 4)
    assert torch.equal(got[:, :, :3], w1.view(64, 64, 3)) and float(got[:, :, 3].abs().max()) == 0.0
    # conv-transpose weight (ci,co,k) -> (k*64+ci, co)
    off += 3 * 64 * 256 + 2 * 4 * 128
    wt = p[off:off + 512 * 64].view(8, 64, 64)
    assert torch.equal(wt[3, 5], dp.linear.weight.detach()[5, :, 3])
    with torch.no_grad():
        dp.linear.bias.add_(1.0)
    off += 512 * 64
    assert dp.pack() is not p and torch.equal(dp.pack()[off:off + 64], dp.linear.bias.detach())
    # f16x3 image of the layer-0 projection: [chunk][hi|lo][col = dir*128 + gate*32 + j][32 k'], k' = kk*64 + c, scaled by 256
    img = dp.pack()[off + 64:off + 64 + 512 * 256].view(torch.float16).view(16, 2, 256, 32).float()
    w0 = dp.rnn.rnn_lst[0].weight.detach()
    c, kk, d, j = 37, 5, 1, 9
    kp = kk * 64 + c
    rec = (img[kp // 32, 0, d * 128 + 0 * 32 + j, kp % 32] + img[kp // 32, 1, d * 128 + j, kp % 32]) / 256.0
    assert abs(float(rec) - float(w0[c * 8 + kk, (d * 32 + j) * 4 + 0])) < 1e-6 * abs(float(w0[c * 8 + kk, (d * 32 + j) * 4]))+1e-9


def test_upstream_sru_checkpoint_keys_accepted():
    """Upstream `sru` state dicts carry rnn_lst.{i}.weight|weight_c|bias (+ a scale_x buffer)."""
    m = build()
    sd = dict(m.state_dict())
    sd["refinement_module.audio_net.blocks.globalatt.0.rnn.rnn_lst.0.scale_x"] = torch.ones(1)
    m.load_state_dict(sd)


def test_lightning_checkpoint_prefix_loader_and_serialize(tmp_path):
    import rtfs_net_amd as R
    m = build()
    ck = {"audio_model." + k: v + 1 if v.dtype.is_floating_point else v for k, v in m.state_dict().items()}
    R.BaseAVModel.load_state_dict_in(m, ck)
    k0 = "encoder.conv.full_layer.2.weight"
    assert torch.equal(m.state_dict()[k0], ck["audio_model." + k0])
    conf = m.serialize()
    assert conf["model_name"] == "AVNet" and set(conf) == {"model_name", "state_dict", "model_args", "infos"}
    path = str(tmp_path / "best_model.pth")
    torch.save({"model_name": conf["model_name"], "state_dict": conf["state_dict"]}, path)
    import copy
    m2 = R.AVNet.from_pretrain(path, **copy.deepcopy(RTFS4_AUDIONET))
    assert torch.equal(m2.state_dict()[k0], m.state_dict()[k0])


def test_model_registry():
    import rtfs_net_amd as R
    assert R.get("AVNet") is R.AVNet and R.get("avnet") is R.AVNet and R.RTFSNet is R.AVNet
    with pytest.raises(ValueError):
        R.get("nope")
    with pytest.raises(ValueError):
        R.register_model(R.AVNet)


def test_macs_report_matches_published_total():
    m = build()
    m.get_MACs()
    total = [l for l in m.macs_parms.splitlines() if l.startswith("Total")][0]
    assert "21,9" in total and "739 K" in total  # published: 21.9 G MACs, 0.7 M params


def test_unsupported_configs_fail_loudly():
    import copy
    import rtfs_net_amd as R
    c = copy.deepcopy(RTFS4_AUDIONET)
    c["audio_params"]["layers"]["layer_1"]["rnn_type"] = "GRU"
    with pytest.raises(ValueError):
        R.AVNet(print_macs=False, **c)
    c["audio_params"]["layers"]["layer_1"]["rnn_type"] = "LSTM"  # mixed cells are not on any yaml
    with pytest.raises(ValueError):
        R.AVNet(print_macs=False, **c)


def test_lstm_variant_state_dict_matches_reference():
    import copy
    import rtfs_net_amd as R
    from tests.util import GOLDEN
    c = copy.deepcopy(RTFS4_AUDIONET)
    for k in ("layer_1", "layer_2"):
        c["audio_params"]["layers"][k]["rnn_type"] = "LSTM"
    m = R.AVNet(print_macs=False, **c)
    spec = json.load(open(os.path.join(GOLDEN, "state_spec_R4_lstm.json")))
    assert [k for k, _, _ in spec] == list(m.state_dict().keys())
    assert sum(p.numel() for p in m.parameters()) == 832112  # RTFS-Net-4 with rnn_type LSTM


def test_no_cpu_fallback_and_no_training_mode():
    m = build()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4096), torch.zeros(1, 512, 7))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.refinement_module.audio_net.blocks(torch.zeros(1, 256, 17, 129))


def test_missing_library_raises(monkeypatch):
    from rtfs_net_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librtfs_amd.so")
    with pytest.raises(RuntimeError, match="no non-HIP fallback"):
        _lib.load()


def test_vp_block_torch_ops_match_golden_on_cpu():
    """The video-side VP block runs on stock torch ops; pin it against the reference's vectors."""
    from oracle.params import make_state_dict
    from tests.util import check_probe, load_golden, rand
    m = build()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(spec_R4(), 0).items()})
    with torch.no_grad():
        for name, tv, seed in [("mod_vp50", 50, 109), ("mod_vp7", 7, 110)]:
            y = m.refinement_module.video_net.blocks(torch.from_numpy(rand((2, 512, tv), seed))).numpy()
            check_probe(load_golden(name), "out", y, 2e-5)


def test_system_checkpoint_formats(tmp_path):
    """Lightning .ckpt (audio_model./video_model. prefixes, core.py:178-181) and best_model.pth (train.py:156-160) round
    trips through the non-executing loaders."""
    import copy
    import torch
    import rtfs_net_amd as R
    cfg = copy.deepcopy(RTFS4_AUDIONET)
    torch.manual_seed(1)
    a0, v0 = R.AVNet(print_macs=False, **copy.deepcopy(cfg)), R.FRCNNVideoModel(print_macs=False)
    sd = {**{"audio_model." + k: v for k, v in a0.state_dict().items()}, **{"video_model." + k: v for k, v in v0.state_dict().items()}}
    ck = tmp_path / "epoch=1.ckpt"
    torch.save({"state_dict": sd, "training_config": {"exp": {"exp_name": "x"}}}, ck)
    torch.manual_seed(2)
    s = R.System(audio_model=R.AVNet(print_macs=False, **copy.deepcopy(cfg)), video_model=R.FRCNNVideoModel(print_macs=False))
    assert s.load_lightning_checkpoint(str(ck)) == {"exp": {"exp_name": "x"}}
    for k, v in a0.state_dict().items():
        assert torch.equal(v, s.audio_model.state_dict()[k]), k
    for k, v in v0.state_dict().items():
        assert torch.equal(v, s.video_model.state_dict()[k]), k
    best = tmp_path / "best_model.pth"
    ser = a0.serialize()
    ser["infos"]["software_versions"]["torch_version"] = torch.__version__  # as the reference writes it (TorchVersion)
    torch.save(ser, best)
    a1 = R.system.load_best_model(str(best), **copy.deepcopy(cfg))
    for k, v in a0.state_dict().items():
        assert torch.equal(v, a1.state_dict()[k]), k
    with pytest.raises(ValueError):
        R.System(audio_model=a0, optimizer=object())
