"""Mirror of the reference's ``System`` (``src/system/core.py:50-123``) and its on-disk formats (SURVEY 8f rank 4 + the start of
rank 1): ``forward(wav, mouth)`` chains the video front-end and the separator; ``validation_step`` / ``training_step`` add the PIT
loss, all on the HIP path.  ``optimization_step`` is what Lightning does around ``training_step`` in the reference (``train.py:135-148``:
backward, gradient all-reduce over the data-parallel ranks, ``gradient_clip_val`` 5.0, optimizer step) written out, because Lightning is
not a dependency here; the gradient exchange is ONE all-reduce of one flattened buffer (RCCL when the process group is ``nccl``).
Training covers what ``AVNet.forward_train`` covers (frozen BatchNorm statistics, frozen video-side VP block).
``load_lightning_checkpoint`` reads a Lightning ``.ckpt`` (``state_dict`` with ``audio_model.`` / ``video_model.`` prefixes,
``core.py:178-181``) and ``load_best_model`` the ``best_model.pth`` that ``train.py:156-160`` writes, both with non-executing loaders.
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
        if train_video_model:
            raise ValueError("MI355X System: the video front-end is inference-only (the reference freezes it too: yaml videonet)")
        self.audio_model, self.video_model, self.loss_func = audio_model, video_model, loss_func
        self.optimizer, self.scheduler = optimizer, scheduler
        self.train_loader, self.val_loader = train_loader, val_loader
        self.config = {} if config is None else config

    def forward(self, wav, mouth=None):
        """core.py:78-92."""
        if self.video_model is None:
            return self.audio_model(wav)
        with torch.no_grad():
            mouth_emb = self.video_model(mouth.type_as(wav))
        return self.audio_model(wav, mouth_emb)

    def common_step(self, batch, batch_nb, is_train=True):
        """core.py:94-118."""
        if self.video_model is None and len(batch) == 4:  # extension: pre-computed lip embeddings in the mouth slot
            inputs, targets, mouth_emb, _ = batch
            est_targets = self.audio_model(inputs, mouth_emb)
        elif self.video_model is None:
            inputs, targets, _ = batch
            est_targets = self(inputs)
        else:
            inputs, targets, target_mouths, _ = batch
            est_targets = self(inputs, target_mouths)
        if targets.ndim == 2:
            targets = targets.unsqueeze(1)
        return self.loss_func["train" if is_train else "val"](est_targets, targets)

    def training_step(self, batch, batch_nb):
        """core.py:119-123: the loss tensor carries the HIP backward graph (AVNet.forward_train + the PIT loss gradient kernel)."""
        return {"loss": self.common_step(batch, batch_nb, is_train=True)}

    def validation_step(self, batch, batch_nb):
        with torch.no_grad():
            return {"val_loss": self.common_step(batch, batch_nb, is_train=False)}

    # ---------------------------------------------------------------- what Lightning does around training_step (train.py:135-148)
    def convert_sync_batchnorm(self):
        """What Lightning's ``sync_batchnorm=True`` (train.py:145) does: BatchNorm layers become nn.SyncBatchNorm, which the training
        kernels synchronise with two small all-reduces per layer (batch statistics forward, dgamma / dbeta sums backward)."""
        self.audio_model = nn.SyncBatchNorm.convert_sync_batchnorm(self.audio_model)
        return self

    def trainable_parameters(self):
        return [p for p in self.audio_model.parameters() if p.requires_grad]

    def allreduce_gradients(self):
        """Average the gradients over the data-parallel ranks with ONE collective on one flattened buffer (739,952 floats for
        RTFS-Net; backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests).  No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0
        params = [p for p in self.trainable_parameters()]
        if not params:
            return 0
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
        off = 0
        for p in params:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p).clone()
            off += n
        return flat.numel()

    def broadcast_parameters(self, src=0):
        """What DistributedDataParallel does at construction (train.py:135-146 runs under Lightning's DDP strategy): every rank starts
        from rank `src`'s parameters AND buffers (BatchNorm running statistics), so ranks that were seeded differently cannot drift apart
        silently.  One broadcast of one flattened float buffer (+ one for the integer buffers).  No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0
        tensors = [t for t in list(self.audio_model.parameters()) + list(self.audio_model.buffers())]
        done = 0
        for is_float in (True, False):
            group = [t for t in tensors if t.is_floating_point() == is_float]
            if not group:
                continue
            flat = torch.cat([t.detach().reshape(-1).to(torch.float32 if is_float else torch.int64) for t in group])
            dist.broadcast(flat, src=src)
            off = 0
            with torch.no_grad():
                for t in group:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t).to(t.dtype))  # in place through the tensor itself: bumps _version (pack caches)
                    off += n
            done += flat.numel()
        self._params_broadcast = True
        return done

    def optimization_step(self, batch, batch_nb=0, gradient_clip_val=5.0):
        """zero_grad -> training_step -> backward -> gradient all-reduce -> clip (train.py:142 gradient_clip_val 5.0) -> optimizer step.
        The first step of a multi-rank job broadcasts rank 0's parameters and buffers first (DDP's construction-time broadcast)."""
        if self.optimizer is None:
            raise RuntimeError("System.optimization_step needs an optimizer")
        if not getattr(self, "_params_broadcast", False):
            self.broadcast_parameters()
            self._params_broadcast = True
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.training_step(batch, batch_nb)["loss"]
        loss.backward()
        self.allreduce_gradients()
        if gradient_clip_val:
            torch.nn.utils.clip_grad_norm_(self.trainable_parameters(), gradient_clip_val)
        self.optimizer.step()
        return loss.detach()

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
