"""CPU restatement of the video front-end (TEST INFRASTRUCTURE ONLY; SURVEY 8f rank 2, "the step before the path"):
``FRCNNVideoModel`` with the ResNet-18 trunk and PReLU activations, eval mode
(reference src/models/videomodels/frcnn_videomodel.py:16-72, resnet.py:23-118).

  lips (B, 1, T, 88, 88) -> Conv3d(1,64,(5,7,7),s(1,2,2),p(2,3,3)) -> BatchNorm3d -> PReLU(64) -> MaxPool3d((1,3,3),s(1,2,2),p(0,1,1))
       -> frames (B*T, 64, 22, 22) -> ResNet BasicBlock x [2,2,2,2] (64,128,256,512; stride 2 from layer2 on, 1x1-conv
       downsample) -> AdaptiveAvgPool2d(1) -> (B, 512, T)

Pinned by tests/golden/video_*.npz generated from the reference module itself (oracle/make_golden_video.py).
float64 accumulation on float32 data.
"""
from __future__ import annotations

import zlib

import numpy as np

f32 = np.float32
BN_EPS = 1e-5


def bn_eval(x, p, prefix):
    """nn.BatchNorm{2,3}d in eval mode (running statistics); channel axis 1."""
    w, b = p[prefix + ".weight"].astype(np.float64), p[prefix + ".bias"].astype(np.float64)
    rm, rv = p[prefix + ".running_mean"].astype(np.float64), p[prefix + ".running_var"].astype(np.float64)
    scale = w / np.sqrt(rv + BN_EPS)
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return (x * scale.reshape(shp) + (b - rm * scale).reshape(shp)).astype(f32)


def prelu_ch(x, a):
    """nn.PReLU(num_parameters=C): one slope per channel (axis 1)."""
    shp = (1, -1) + (1,) * (x.ndim - 2)
    return np.where(x >= 0, x, a.reshape(shp) * x).astype(f32)


def conv2d(x, w, stride=1, pad=0):
    """Dense Conv2d (cross-correlation), no bias.  x (N,Ci,H,W), w (Co,Ci,kh,kw)."""
    N, Ci, H, W = x.shape
    Co, _, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))).astype(np.float64)
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = np.zeros((N, Co, Ho, Wo), np.float64)
    w64 = w.astype(np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i:i + stride * (Ho - 1) + 1:stride, j:j + stride * (Wo - 1) + 1:stride]
            y += np.einsum("oc,nchw->nohw", w64[:, :, i, j], patch)
    return y.astype(f32)


def stem(x, p):
    """frontend3D (frcnn_videomodel.py:41-53): x (B,1,T,H,W) -> (B,64,T,H/4,W/4)."""
    B, _, T, H, W = x.shape
    w = p["frontend3D.0.weight"].astype(np.float64)  # (64,1,5,7,7)
    xp = np.pad(x[:, 0], ((0, 0), (2, 2), (3, 3), (3, 3))).astype(np.float64)  # (B, T+4, H+6, W+6)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    y = np.zeros((B, 64, T, Ho, Wo), np.float64)
    for dt in range(5):
        for i in range(7):
            for j in range(7):
                patch = xp[:, dt:dt + T, i:i + 2 * (Ho - 1) + 1:2, j:j + 2 * (Wo - 1) + 1:2]  # (B,T,Ho,Wo)
                y += w[:, 0, dt, i, j].reshape(1, 64, 1, 1, 1) * patch[:, None]
    y = prelu_ch(bn_eval(y.astype(f32), p, "frontend3D.1"), p["frontend3D.2.weight"])
    # MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1)): -inf padding
    yp = np.pad(y, ((0, 0), (0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    out = np.full((B, 64, T, Hp, Wp), -np.inf, f32)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, yp[:, :, :, i:i + 2 * (Hp - 1) + 1:2, j:j + 2 * (Wp - 1) + 1:2])
    return out


def basic_block(x, p, prefix, stride):
    """BasicBlock.forward (resnet.py:51-66) with relu_type='prelu'."""
    out = prelu_ch(bn_eval(conv2d(x, p[prefix + ".conv1.weight"], stride, 1), p, prefix + ".bn1"), p[prefix + ".relu1.weight"])
    out = bn_eval(conv2d(out, p[prefix + ".conv2.weight"], 1, 1), p, prefix + ".bn2")
    if prefix + ".downsample.0.weight" in p:
        res = bn_eval(conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0), p, prefix + ".downsample.1")
    else:
        res = x
    return prelu_ch((out + res).astype(f32), p[prefix + ".relu2.weight"])


def video_frontend(x, p, return_internals=False):
    """FRCNNVideoModel.forward (frcnn_videomodel.py:61-72), backbone 'resnet': x (B,1,T,88,88) -> (B,512,T)."""
    B, _, T, _, _ = x.shape
    y = stem(x, p)
    f = y.transpose(0, 2, 1, 3, 4).reshape(B * T, 64, y.shape[3], y.shape[4])  # threeD_to_2D_tensor
    internals = {"stem": f}
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            f = basic_block(f, p, f"trunk.layer{li}.{bi}", stride if bi == 0 else 1)
        internals[f"layer{li}"] = f
    v = f.astype(np.float64).mean((2, 3)).astype(f32)  # AdaptiveAvgPool2d(1) + view
    out = np.ascontiguousarray(v.reshape(B, T, -1).transpose(0, 2, 1))
    return (out, internals) if return_internals else out


# ------------------------------------------------------------------ deterministic synthetic parameters / inputs
def video_state_spec():
    """[name, shape] of the reference FRCNNVideoModel(backbone_type='resnet', relu_type='prelu') state_dict, in order."""
    spec = []  # the reference registers the trunk before the 3-D front end
    inpl = 64
    for li, planes in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for bi in range(2):
            pre = f"trunk.layer{li}.{bi}"
            cin = inpl if bi == 0 else planes
            spec += [(pre + ".conv1.weight", (planes, cin, 3, 3))]
            spec += [(pre + f".bn1.{k}", s) for k, s in (("weight", (planes,)), ("bias", (planes,)), ("running_mean", (planes,)), ("running_var", (planes,)), ("num_batches_tracked", ()))]
            spec += [(pre + ".relu1.weight", (planes,)), (pre + ".relu2.weight", (planes,)), (pre + ".conv2.weight", (planes, planes, 3, 3))]
            spec += [(pre + f".bn2.{k}", s) for k, s in (("weight", (planes,)), ("bias", (planes,)), ("running_mean", (planes,)), ("running_var", (planes,)), ("num_batches_tracked", ()))]
            if bi == 0 and (li > 1):
                spec += [(pre + ".downsample.0.weight", (planes, cin, 1, 1))]
                spec += [(pre + f".downsample.1.{k}", s) for k, s in (("weight", (planes,)), ("bias", (planes,)), ("running_mean", (planes,)), ("running_var", (planes,)), ("num_batches_tracked", ()))]
        inpl = planes
    spec += [("frontend3D.0.weight", (64, 1, 5, 7, 7))]
    spec += [(f"frontend3D.1.{k}", s) for k, s in (("weight", (64,)), ("bias", (64,)), ("running_mean", (64,)), ("running_var", (64,)), ("num_batches_tracked", ()))]
    spec += [("frontend3D.2.weight", (64,))]
    return spec


def make_video_state_dict(seed=0):
    """Per-name seeded values: He-like conv weights, non-trivial BatchNorm statistics / affines, PReLU slopes in (0.1, 0.4)."""
    sd = {}
    for name, shape in video_state_spec():
        rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        leaf = name.split(".")[-1]
        if leaf == "num_batches_tracked":
            sd[name] = np.zeros(shape, np.int64)
        elif leaf == "running_mean":
            sd[name] = (rs.randn(*shape) * 0.1).astype(f32)
        elif leaf == "running_var":
            sd[name] = rs.uniform(0.5, 1.5, shape).astype(f32)
        elif len(shape) == 1 and ("bn" in name or "downsample.1" in name or "frontend3D.1" in name):
            sd[name] = (rs.uniform(0.5, 1.5, shape) if leaf == "weight" else rs.uniform(-0.2, 0.2, shape)).astype(f32)
        elif len(shape) == 1:  # PReLU slopes
            sd[name] = rs.uniform(0.1, 0.4, shape).astype(f32)
        else:
            fan_in = int(np.prod(shape[1:]))
            sd[name] = (rs.randn(*shape) * np.sqrt(2.0 / fan_in)).astype(f32)
    return sd


def make_video_input(B, T, seed=0, size=88):
    rs = np.random.RandomState(4321 + seed)
    return rs.rand(B, 1, T, size, size).astype(f32)
