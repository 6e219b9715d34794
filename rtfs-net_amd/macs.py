"""Analytic multiply-accumulate / parameter report for the RTFS-Net path, in the table format of the
reference's ``BaseAVModel.get_MACs`` (TDAVNet/base_av_model.py:61-118; the reference measures with thop on a
2 s @16 kHz mixture and a 2 s @25 fps lip embedding, batch 1)."""
from __future__ import annotations


def _params_k(m):
    return int(sum(p.numel() for p in m.parameters() if p.requires_grad) / 1000)


def block_macs(T, F):
    """One application of the 2-D RTFS block on a (256, T, F) input."""
    P, Tp, Fp = T * F, T // 2, F // 2
    Pg = Tp * Fp
    m = P * 256                      # gateway depthwise 1x1
    m += P * 256 * 64 * 2            # projection + residual_conv
    m += P * 64 * 16 + Pg * 64 * 16  # downsample convs
    for n_seq, ls in ((Tp, Fp), (Fp, Tp)):  # dual-path sweeps: along F then along T
        L = ls - 7
        m += n_seq * (L * (512 * 256 + 3 * 64 * 192) + L * 64 * 64 * 8)
    m += Pg * 64 * (96 + 64)         # attention 1x1 convs (Q,K,V for 4 heads; concat projection)
    m += 4 * Tp * Tp * (256 + 1024)  # QK^T and AV
    m += 2 * P * 64 * 16 + 7 * Pg * 64 * 16  # TFAR depthwise convs (2 full-res local, 7 compressed)
    return m


def vp_macs(Tv):
    n = [Tv]
    for _ in range(3):
        n.append((n[-1] + 2 - 3) // 2 + 1)
    m = Tv * 512 + Tv * 512 * 64 * 2 + sum(n) * 64 * 3
    t = n[-1]
    m += t * 64 * 64 * 4 + 2 * 8 * t * t * 8 + t * (64 * 128 * 2 + 128 * 3)
    m += sum(n) * 64 * 3 * 3 + sum(n[:-1]) * 64 * 3 * 3
    return m


def macs_report(model, seconds=2):
    L = seconds * 16000
    T, F, Tv = 1 + L // 128, 129, seconds * 25
    P = T * F
    rm = model.refinement_module
    R = int(model.audio_params["repeats"])
    enc = P * 256 * 18
    bn = P * 256 * 256
    audio = R * block_macs(T, F)
    video = vp_macs(Tv)
    fusion = 2 * P * 256 + Tv * (256 * 2 + 1024 * 2)
    mask = P * 256 * 256
    dec = P * 256 * 18
    total = enc + bn + audio + video + fusion + mask + dec
    M = lambda v: "{:,}".format(int(v / 1e6))
    K = lambda m: "{:,}".format(_params_k(m))
    rows = [
        ("Encoder ------------- ", enc, model.encoder), ("Audio BN ------------ ", bn, model.audio_bottleneck),
        ("Video BN ------------ ", 0, model.video_bottleneck), ("RefinementModule ---- ", audio + video + fusion, rm),
        ("   AudioNet --------- ", audio, rm.audio_net), ("   VideoNet --------- ", video, rm.video_net),
        ("   FusionNet -------- ", fusion, rm.crossmodal_fusion), ("Mask Generator ------ ", mask, model.mask_generator),
        ("Decoder ------------- ", dec, model.decoder), ("Total --------------- ", total, model),
    ]
    return "CTCNet\n" + "".join("{}MACs: {:>8} M    Params: {:>6} K\n".format(n, M(v), K(m)) for n, v, m in rows)
