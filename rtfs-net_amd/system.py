"""Inference-side mirror of the reference's ``System`` (``src/system/core.py:50-123``) and its on-disk formats
(SURVEY 8f rank 4): ``forward(wav, mouth)`` chains the video front-end, the separator and -- in ``validation_step`` -- the
PIT loss, all on the HIP path; ``load_lightning_checkpoint`` reads a Lightning ``.ckpt`` (``state_dict`` with
``audio_model.`` / ``video_model.`` prefixes, ``core.py:178-181``) and ``load_best_model`` the ``best_model.pth`` that
``train.py:156-160`` writes, both with non-executing loaders.  No training loop (no backward yet).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class System(nn.Module):
    """core.py:53-92: audio_model (AVNet), optional video_model (FRCNNVideoModel), optional loss_func {"val": PITLossWrapper}."""

    default_monitor: str = "val_loss"

    def __init__(self, audio_model=None, video_model=None, optimizer=None, loss_func=None, train_loader=None, val_loader=None,
                 scheduler=None, config=None, train_video_model=False):
        super().__init__()
        if optimizer is not None or scheduler is not None or train_video_model:
            raise ValueError("MI355X System is inference-only (no backward pass yet): no optimizer / scheduler / video training")
        self.audio_model, self.video_model, self.loss_func = audio_model, video_model, loss_func
        self.train_loader, self.val_loader = train_loader, val_loader
        self.config = {} if config is None else config

    def forward(self, wav, mouth=None):
        """core.py:78-92."""
        if self.video_model is None:
            return self.audio_model(wav)
        with torch.no_grad():
            mouth_emb = self.video_model(mouth.type_as(wav))
        return self.audio_model(wav, mouth_emb)

    def common_step(self, batch, batch_nb, is_train=False):
        """core.py:94-118, validation side."""
        if is_train:
            raise RuntimeError("MI355X System: training_step needs the backward pass, which is not built yet")
        if self.video_model is None:
            inputs, targets, _ = batch
            est_targets = self(inputs)
        else:
            inputs, targets, target_mouths, _ = batch
            est_targets = self(inputs, target_mouths)
        if targets.ndim == 2:
            targets = targets.unsqueeze(1)
        return self.loss_func["val"](est_targets, targets)

    def validation_step(self, batch, batch_nb):
        return {"val_loss": self.common_step(batch, batch_nb, is_train=False)}

    # ---------------------------------------------------------------- on-disk formats
    def load_lightning_checkpoint(self, path, strict=True):
        """Lightning ``.ckpt``: {"state_dict": {"audio_model.*", "video_model.*"}, "training_config": ...}."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True) if isinstance(path, str) else path
        sd = ckpt["state_dict"]
        audio = {k[len("audio_model."):]: v for k, v in sd.items() if k.startswith("audio_model.")}
        video = {k[len("video_model."):]: v for k, v in sd.items() if k.startswith("video_model.")}
        self.audio_model.load_state_dict(audio, strict=strict)
        if self.video_model is not None and video:
            self.video_model.load_state_dict(video, strict=strict)
        return ckpt.get("training_config")


def load_best_model(path, **audionet_kwargs):
    """``best_model.pth`` (train.py:156-160 -> BaseAVModel.serialize) -> AVNet, as test.py:38-39 does."""
    from .models import AVNet
    return AVNet.from_pretrain(path, **audionet_kwargs)
