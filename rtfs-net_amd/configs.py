"""The `audionet` section of the reference's three RTFS-Net yaml files as package data.

config/lrs2_RTFSNet_{4,6,12}_layer.yaml differ only in `audio_params.repeats` (the block weights are shared, `shared: true`), so one
dictionary + the repeat count covers the BASELINE configurations; bench.py, __graft_entry__.smoke() and the tests build their models
from here (tests/test_host.py checks the dictionary against the reference's yaml when /root/reference is mounted).
"""
import copy

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


def audionet_config(repeats=4, rnn_type="SRU"):
    """`conf["audionet"]` of RTFS-Net-`repeats` (4 / 6 / 12 are the published ones); `rnn_type` selects the DualPathRNN cell
    (rnn_layers.py:99-122: SRU as in the three yamls, or LSTM / GRU)."""
    c = copy.deepcopy(RTFS4_AUDIONET)
    c["audio_params"]["repeats"] = int(repeats)
    for k in ("layer_1", "layer_2"):
        c["audio_params"]["layers"][k]["rnn_type"] = rnn_type
    return c
