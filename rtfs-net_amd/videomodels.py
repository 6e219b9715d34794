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
# state_dict keys, order, shapes, construction order (= consumption of the random stream) and final initialisation pass are
# those of the reference's ResNet (resnet.py:5-121), under the reference's public names.
def conv3x3(in_planes: int, out_planes: int, stride: int = 1) -> nn.Conv2d:
    """resnet.py:5-6."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def downsample_basic_block(inplanes: int, outplanes: int, stride: int) -> nn.Sequential:
    """The 1x1 projection + batch norm on a skip path whose shape changes (resnet.py:9-13)."""
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(outplanes))


class BasicBlock(nn.Module):
    """Parameter holder of one residual unit (resnet.py:24-50): conv1, bn1, relu1, relu2, conv2, bn2 in checkpoint order, then
    ``downsample`` (None where the skip path is the identity) and ``stride``.  Only the PReLU variant is on the MI355X path."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="prelu"):
        super().__init__()
        if relu_type != "prelu":
            raise ValueError("MI355X video trunk: relu_type='prelu' only")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = nn.PReLU(num_parameters=planes)
        self.relu2 = nn.PReLU(num_parameters=planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("the MI355X video trunk runs inside rtfs_video_frontend_f32 (FRCNNVideoModel.forward), not unit by unit")


class ResNet(nn.Module):
    """ResNet-18 trunk as the reference builds it (resnet.py:68-121): four stages of two BasicBlocks, widths 64 .. 512, the first
    unit of stages 2-4 strided with a projected skip path; convolutions re-drawn N(0, sqrt(2 / (k*k*cout))) and batch norms reset
    in ONE pass over modules() after construction (so a seed gives the reference's initial weights); ``gamma_zero`` zeroes each
    unit's last batch-norm scale.  Configurations the fused front-end kernel does not implement are rejected here."""

    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), num_classes=1000, relu_type="prelu", gamma_zero=False, avg_pool_downsample=False):
        super().__init__()
        if list(layers) != [2, 2, 2, 2] or block is not BasicBlock:
            raise ValueError("MI355X video trunk: ResNet-18 (BasicBlock, layers [2, 2, 2, 2]) only")
        if avg_pool_downsample:
            raise ValueError("MI355X video trunk: avg_pool_downsample is not implemented (the RTFS-Net recipes do not use it)")
        self.inplanes, self.relu_type, self.gamma_zero = 64, relu_type, gamma_zero
        self.downsample_block = downsample_basic_block
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            setattr(self, f"layer{i + 1}", self._make_layer(block, planes, layers[i], stride))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():  # resnet.py:90-97
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if gamma_zero:  # resnet.py:99-104
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)

    def _make_layer(self, block, planes, blocks, stride=1):
        # the projection is built BEFORE its unit (resnet.py:106-121): the order in which default initialisers draw from the random stream
        skip = self.downsample_block(self.inplanes, planes * block.expansion, stride) if stride != 1 or self.inplanes != planes * block.expansion else None
        units = [block(self.inplanes, planes, stride, skip, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        units += [block(self.inplanes, planes, relu_type=self.relu_type) for _ in range(1, blocks)]
        return nn.Sequential(*units)

    def forward(self, x):
        raise RuntimeError("the MI355X video trunk runs inside rtfs_video_frontend_f32 (FRCNNVideoModel.forward)")


def _trunk(gamma_zero: bool = False) -> nn.Module:
    return ResNet(BasicBlock, [2, 2, 2, 2], relu_type="prelu", gamma_zero=gamma_zero)


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
