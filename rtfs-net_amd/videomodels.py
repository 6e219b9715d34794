"""Video front-end of the reference on the MI355X path: ``FRCNNVideoModel`` (ResNet-18 trunk, PReLU), reference
``src/models/videomodels/frcnn_videomodel.py:16-72`` / ``resnet.py:23-118`` -- same class names, constructor keywords and
``state_dict`` keys.  ``forward`` (eval mode) marshals one call into ``rtfs_video_frontend_f32`` (implicit-GEMM convolutions
on the f16 matrix cores, ``csrc/k_video.hip``); there is no CPU fallback.  Only the ``resnet`` backbone with
``relu_type="prelu"`` (what the RTFS-Net recipes load) is on this path.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, packing


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def downsample_basic_block(inplanes, outplanes, stride):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(outplanes))


class BasicBlock(nn.Module):
    """Parameter container with the reference's keys (resnet.py:23-66); the arithmetic runs in the fused front-end call."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="relu"):
        super().__init__()
        if relu_type != "prelu":
            raise ValueError("MI355X BasicBlock supports relu_type='prelu'")
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = nn.PReLU(num_parameters=planes)
        self.relu2 = nn.PReLU(num_parameters=planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    """resnet.py:69-118 (layers [2,2,2,2], BasicBlock): parameter container."""

    def __init__(self, block, layers, num_classes=1000, relu_type="relu", gamma_zero=False, avg_pool_downsample=False):
        super().__init__()
        if list(layers) != [2, 2, 2, 2] or avg_pool_downsample or block is not BasicBlock:
            raise ValueError("MI355X ResNet supports BasicBlock [2,2,2,2] with the 1x1-conv downsample")
        self.inplanes, self.relu_type, self.gamma_zero = 64, relu_type, gamma_zero
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():  # the reference's default init (resnet.py:90-98)
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if gamma_zero:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    m.bn2.weight.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = downsample_basic_block(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, relu_type=self.relu_type))
        return nn.Sequential(*layers)


class FRCNNVideoModel(nn.Module):
    """frcnn_videomodel.py:16-115.  forward(x (B,1,T,88,88)) -> (B,512,T) lip embedding."""

    def __init__(self, backbone_type="resnet", relu_type="prelu", width_mult=1.0, pretrain=None, print_macs=True, *args, **kwargs):
        super().__init__()
        if backbone_type != "resnet" or relu_type != "prelu":
            raise ValueError("MI355X FRCNNVideoModel supports backbone_type='resnet' with relu_type='prelu'")
        self.backbone_type, self.frontend_nout, self.backend_out = backbone_type, 64, 512
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
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
