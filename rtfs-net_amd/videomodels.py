"""Video front-end of the reference on the MI355X path: ``FRCNNVideoModel`` (ResNet-18 trunk, PReLU), reference
``src/models/videomodels/frcnn_videomodel.py:16-72`` / ``resnet.py:23-118`` -- same model class, constructor keywords and
``state_dict`` keys (the trunk is a table-built tree of parameter holders).  ``forward`` (eval mode) marshals one call into ``rtfs_video_frontend_f32`` (implicit-GEMM convolutions
on the f16 matrix cores, ``csrc/k_video.hip``); there is no CPU fallback.  Only the ``resnet`` backbone with
``relu_type="prelu"`` (what the RTFS-Net recipes load) is on this path.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, packing


# The trunk holds no arithmetic here (the fused front-end call does it all): it is a tree of parameter holders whose
# state_dict keys, order, shapes and default initialisation are those a reference checkpoint carries (resnet.py:23-118).
# One row per residual stage: (stage name, input width, width, stride of its first unit); two units per stage.
_TRUNK_STAGES = (("layer1", 64, 64, 1), ("layer2", 64, 128, 2), ("layer3", 128, 256, 2), ("layer4", 256, 512, 2))
_UNITS_PER_STAGE = 2


def _unit_entries(cin: int, width: int, stride: int):
    """(key, kind, dims) of one residual unit, in checkpoint order.  kind: 'c' conv (cout, cin, k, stride), 'n' batch norm
    (channels), 'a' per-channel PReLU slope (channels)."""
    rows = [("conv1", "c", (width, cin, 3, stride)), ("bn1", "n", (width,)), ("relu1", "a", (width,)), ("relu2", "a", (width,)),
            ("conv2", "c", (width, width, 3, 1)), ("bn2", "n", (width,))]
    if stride != 1 or cin != width:  # the 1x1 projection on the skip path where the shape changes
        rows.append(("downsample", "s", (("0", "c", (width, cin, 1, stride)), ("1", "n", (width,)))))
    return rows


def _holder(kind: str, dims) -> nn.Module:
    if kind == "c":
        cout, cin, k, stride = dims
        m = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / (k * k * cout)))  # resnet.py:90-94
        return m
    if kind == "n":
        return nn.BatchNorm2d(*dims)  # weight 1, bias 0 (resnet.py:95-97 == torch's default)
    if kind == "a":
        return nn.PReLU(num_parameters=dims[0])
    seq = nn.Sequential()
    for key, k2, d2 in dims:
        seq.add_module(key, _holder(k2, d2))
    return seq


class _Holders(nn.Module):
    """A module that only owns named children built from a table."""

    def __init__(self, rows):
        super().__init__()
        for key, kind, dims in rows:
            self.add_module(key, _holder(kind, dims))


def _trunk(gamma_zero: bool = False) -> nn.Module:
    trunk = nn.Module()
    for name, cin, width, stride in _TRUNK_STAGES:
        units = [_Holders(_unit_entries(cin if u == 0 else width, width, stride if u == 0 else 1)) for u in range(_UNITS_PER_STAGE)]
        if gamma_zero:
            for u in units:
                nn.init.zeros_(u.bn2.weight)
        trunk.add_module(name, nn.Sequential(*units))
    return trunk


class FRCNNVideoModel(nn.Module):
    """frcnn_videomodel.py:16-115.  forward(x (B,1,T,88,88)) -> (B,512,T) lip embedding."""

    def __init__(self, backbone_type="resnet", relu_type="prelu", width_mult=1.0, pretrain=None, print_macs=True, *args, **kwargs):
        super().__init__()
        if backbone_type != "resnet" or relu_type != "prelu":
            raise ValueError("MI355X FRCNNVideoModel supports backbone_type='resnet' with relu_type='prelu'")
        self.backbone_type, self.frontend_nout, self.backend_out = backbone_type, 64, 512
        self.trunk = _trunk()
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64), nn.PReLU(num_parameters=64),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        self.pretrain = pretrain
        if pretrain:
            self.init_from(pretrain)

    def pack(self) -> torch.Tensor:
        sd = {k: v for k, v in self.state_dict().items() if v.is_floating_point()}
        key = (packing.pack_epoch(),) + tuple((v.data_ptr(), v._version) for v in sd.values())
        if getattr(self, "_pack_key", None) != key:
            with torch.no_grad():
                object.__setattr__(self, "_pack_buf", packing.pack_video(sd))
            object.__setattr__(self, "_pack_key", key)
        return self._pack_buf

    def forward(self, x: torch.Tensor):
        _lib.need_gpu(x)
        if self.training:
            raise RuntimeError("FRCNNVideoModel: only the eval-mode forward is implemented on the MI355X path (the reference "
                               "freezes this model anyway, frcnn_videomodel.py:78-84)")
        if x.ndim != 5 or x.shape[1] != 1 or tuple(x.shape[3:]) != (88, 88):
            raise ValueError("expected lips of shape (B, 1, T, 88, 88)")
        lib = _lib.load()
        x = x.contiguous().float()
        B, _, T, _, _ = x.shape
        out = torch.empty(B, 512, T, device=x.device, dtype=torch.float32)
        pk = self.pack()
        assert pk.numel() == lib.rtfs_video_pack_floats()
        ws = _lib.workspace(lib.rtfs_video_workspace_bytes(B, T), x.device)
        _lib.check(lib.rtfs_video_frontend_f32(_lib.ptr(x), _lib.ptr(pk), _lib.ptr(out), B, T, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
                   "rtfs_video_frontend_f32")
        return out

    def init_from(self, path):
        """frcnn_videomodel.py:74-76 + update_frcnn_parameter (:103-115), with a non-executing loader."""
        pretrained = torch.load(path, map_location="cpu", weights_only=True)["model_state_dict"]
        own = self.state_dict()
        own.update({k: v for k, v in pretrained.items() if "tcn" not in k})
        self.load_state_dict(own)
        for p in self.parameters():
            p.requires_grad = False
        return self
