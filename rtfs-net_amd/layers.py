"""Layer-level mirror of the reference's ``src/models/layers`` for the RTFS-Net path.

Same class names, constructor keywords and parameter / buffer names as the reference (so
``state_dict`` keys line up), but the audio-path modules do no arithmetic in Python: their
``forward`` hands raw device pointers to ``librtfs_amd.so``.  Only the 0.004-GMAC video-side
(1-D) modules keep a stock-torch-op ``forward``.

reference files: layers/conv_layers.py, normalizations.py, activations.py, rnn_layers.py,
attention.py, fusion.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, packing

EPS = 1e-5


# ----------------------------------------------------------------------------- pack cache
class PackedModule(nn.Module):
    """Caches the flat parameter pack of a module; rebuilt when any tensor is replaced, moved or
    modified in place (tracked through data_ptr + _version)."""

    _pack_fn = None  # staticmethod(packing.pack_*) set by subclasses

    def _state_tensors(self):
        return self.state_dict(keep_vars=True)

    def pack(self) -> torch.Tensor:
        sd = self._state_tensors()
        key = (packing.pack_epoch(),) + tuple((v.data_ptr(), v._version) for v in sd.values())
        if getattr(self, "_pack_key", None) != key:
            with torch.no_grad():
                pk = type(self)._pack_fn(sd)
            object.__setattr__(self, "_pack_buf", pk)
            object.__setattr__(self, "_pack_key", key)
        return self._pack_buf

    def _guard(self, *tensors):
        _lib.need_gpu(*tensors)
        if self.training:
            raise RuntimeError(
                f"{type(self).__name__}: in .train() mode this module runs only with autograd recording (its training kernels); "
                "call .eval() for inference"
            )


_FORCE_TRAIN_KERNELS = [False]

# Fused inference kernels keep a whole sweep / score row / video pyramid on chip: a sweep axis of 250 positions (4 s of audio, BASELINE
# config 5), 256 attention keys, 120 video frames.  The reference's forward has no length limit (rnn_layers.py:136-162, attention.py:149-189;
# infer_any_video.py:86 feeds whole files): longer inputs run on the UNFUSED HIP kernels (GEMM + scan + GEMM, batched-GEMM attention,
# per-layer video block) that also serve training - same arithmetic, tensors through HBM between the steps, any length.
FUSED_MAX_SWEEP = 250
FUSED_MAX_KEYS = 256
FUSED_MAX_VIDEO_FRAMES = 120


class force_train_kernels:
    """Context manager: route every module through its unfused HIP kernels whatever the mode.  Used for configurations that have
    no fused inference kernel (DualPathRNN with the GRU cell); BatchNorm layers still follow their own train / eval flag."""

    def __enter__(self):
        self.prev = _FORCE_TRAIN_KERNELS[0]
        _FORCE_TRAIN_KERNELS[0] = True

    def __exit__(self, *exc):
        _FORCE_TRAIN_KERNELS[0] = self.prev


def _recording(*tensors_and_modules):
    """True when a module call has to take the training kernels: autograd is recording, the module is in train() mode and a
    gradient is wanted (an input or a parameter requires grad).  In eval() mode the inference kernels run and return graph-less
    tensors whatever the grad mode, exactly as before the backward pass existed."""
    if _FORCE_TRAIN_KERNELS[0]:
        return True
    if not torch.is_grad_enabled():
        return False
    wanted = False
    for o in tensors_and_modules:
        if o is None:
            continue
        if isinstance(o, torch.Tensor):
            wanted = wanted or o.requires_grad
        else:
            if not o.training:
                return False
            wanted = wanted or any(p.requires_grad for p in o.parameters())
    return wanted


# ----------------------------------------------------------------------------- normalisations / activations
class GlobalLayerNorm(nn.Module):
    """gLN = GroupNorm(1, C) (reference normalizations.py:8-17); parameter keys ``norm.weight/bias``."""

    def __init__(self, num_channels: int = 1, eps: float = EPS):
        super().__init__()
        self.num_channels, self.eps = num_channels, eps
        self.norm = nn.GroupNorm(1, num_channels, eps=eps)

    def forward(self, x):
        return self.norm(x)


class LayerNormalization4D(nn.Module):
    """Reference normalizations.py:20-41: gamma/beta of shape (1, C, 1, F); statistics over C when F == 1,
    over (C, F) otherwise."""

    def __init__(self, input_dimension, eps: float = EPS):
        super().__init__()
        c, f = input_dimension
        self.dim = (1, 3) if f > 1 else (1,)
        self.gamma = nn.Parameter(torch.ones(1, c, 1, f))
        self.beta = nn.Parameter(torch.zeros(1, c, 1, f))
        self.eps = eps

    def forward(self, x):
        mu = x.mean(dim=self.dim, keepdim=True)
        sd = torch.sqrt(x.var(dim=self.dim, unbiased=False, keepdim=True) + self.eps)
        return (x - mu) / sd * self.gamma + self.beta


gLN = GlobalLayerNorm
LN4d = LayerNormalization4D
_LOCAL_NORMS = {"gLN": GlobalLayerNorm, "GlobalLayerNorm": GlobalLayerNorm, "LayerNormalization4D": LayerNormalization4D, "LN4d": LayerNormalization4D}


def norm_class(identifier):
    if identifier is None:
        return nn.Identity
    if callable(identifier):
        return identifier
    if isinstance(identifier, str):
        cls = getattr(nn, identifier, None) or _LOCAL_NORMS.get(identifier)
        if cls is not None:
            return cls
    raise ValueError("Could not interpret normalization identifier: " + str(identifier))


def act_class(identifier):
    if identifier is None:
        return nn.Identity
    if callable(identifier):
        return identifier
    if isinstance(identifier, str) and hasattr(nn, identifier):
        return getattr(nn, identifier)
    raise ValueError("Could not interpret activation identifier: " + str(identifier))


# ----------------------------------------------------------------------------- conv holders
def _config_of(module):
    return {k: v for k, v in module.__dict__.items() if not k.startswith("_") and k != "training" and not callable(v)}


_ACT_CODE = {nn.Identity: 0, nn.ReLU: 1, nn.PReLU: 2, nn.Sigmoid: 3}


def _bn_world():
    """Number of ranks a SyncBatchNorm layer synchronises over (hook: the tests emulate ranks with threads)."""
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _bn_all_reduce(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM)


class _GradBundleFn(torch.autograd.Function):
    """Stands for "the parameters of one module" in the autograd graph.  forward: the parameters -> an uninitialised tensor the size of
    the module kernels' flat gradient buffer (its values are never read); backward: that buffer -> per-parameter views.  The shared RTFS
    block is applied R times per step: with the parameters as direct inputs autograd summed every parameter's gradient separately
    (~370 one-workgroup add kernels per step); through the bundle it sums one flat buffer per module and application."""

    @staticmethod
    def forward(ctx, n, unpack, *params):
        ctx.unpack = unpack
        return params[0].new_empty(n, dtype=torch.float32)

    @staticmethod
    def backward(ctx, flat):
        return (None, None) + tuple(ctx.unpack(flat.contiguous()))


def _grad_bundle(kind, params, n, unpack):
    """The bundle tensor of ``params`` (non-None tensors): cached on the first parameter like packing.cached_train_pack, so that the R
    applications of a shared module inside one step feed ONE graph node; an optimizer step (version bump) starts a new one."""
    cacheable = torch.is_grad_enabled() and all(isinstance(t, nn.Parameter) for t in params)
    if not cacheable:
        return _GradBundleFn.apply(n, unpack, *params)
    import weakref
    store = packing.param_store(params[0], "grad_bundle")
    key = (kind, n, packing.pack_epoch()) + tuple((id(t), t._version, t.requires_grad) for t in params)
    ref = store.get(key)
    b = ref() if ref is not None else None
    if b is None:
        store.clear()
        b = _GradBundleFn.apply(n, unpack, *params)
        # a WEAK reference: the bundle's graph node holds the parameters, so a strong one in the module-level table would keep every model
        # alive for ever; within a step the graph of the block outputs keeps the bundle alive, which is exactly as long as it must be shared
        store[key] = weakref.ref(b)
    return b


class _BundledFn(torch.autograd.Function):
    """Runs one of the training Functions below (``fn``: inputs x, non-tensor ``head`` arguments, then parameters) with its parameters
    hidden from autograd behind their bundle: the inner backward hands back its flat gradient buffer (``ctx.flat_grads``) as the
    bundle's gradient."""

    @staticmethod
    def forward(ctx, fn, bundle, hidden, x, *head):
        ctx.fn, ctx.nhead, ctx.flat_grads = fn, len(head), True
        ctx.bundle = bundle  # keeps the shared tensor (not just its graph node) alive as long as this application's node: see _grad_bundle
        return fn.forward(ctx, x, *head, *hidden)

    @staticmethod
    def backward(ctx, dout):
        dx, dpar = ctx.fn.backward(ctx, dout)
        return (None, dpar, None, dx) + (None,) * ctx.nhead


def _apply_bundled(fn, kind, x, head, params, n, unpack, tail=()):
    """fn.apply(x, *head, *params, *tail) with the gradient of the (non-None) parameters routed through one bundle; ``unpack(flat)``
    returns the gradients of exactly those parameters, in order.  ``tail``: trailing non-differentiable arguments."""
    live = tuple(p for p in params if p is not None)
    if not any(p.requires_grad for p in live):
        return fn.apply(x, *head, *params, *tail)
    return _BundledFn.apply(fn, _grad_bundle(kind, live, n, unpack), tuple(params) + tuple(tail), x, *head)


class _CNATrainFn(torch.autograd.Function):
    """ConvNormAct forward/backward on the training kernels (csrc/k_train_conv.hip, channel-last rows inside).
    Inputs: x, cfg tuple (11 ints as in include/rtfs_amd.h + an optional 12th: synchronise the BatchNorm statistics across ranks), then
    pre_gamma, pre_beta, pre_slope, weight, bias, gamma, beta, slope (None where the stage is absent) [, running mean, var, momentum]."""

    _CARR, _GEOM = {}, {}

    @staticmethod
    def _carr(cfg, phase, world):
        """The HOST int[15] configuration of the C ABI (cached: a step makes ~200 of these calls and every ctypes object costs microseconds
        the GPU then waits for)."""
        import ctypes
        lay = tuple(int(v) for v in cfg[12:14]) if len(cfg) >= 14 else (0, 0)  # (in_rows, out_rows)
        key = tuple(cfg[:11]) + (phase, world) + lay
        arr = _CNATrainFn._CARR.get(key)
        if arr is None:
            arr = _CNATrainFn._CARR[key] = (ctypes.c_int * 15)(*key)
        return arr

    @staticmethod
    def _geom(cfg, world, B, H, W):
        """(carr, Ho, Wo, saved floats, workspace bytes, gradient floats) of one configuration and input geometry, cached."""
        import ctypes
        carr = _CNATrainFn._carr(cfg, 0, world)
        key = (tuple(carr), B, H, W)
        g = _CNATrainFn._GEOM.get(key)
        if g is None:
            lib = _lib.load()
            ho, wo = ctypes.c_int(), ctypes.c_int()
            lib.rtfs_cna_out_shape(carr, H, W, ctypes.byref(ho), ctypes.byref(wo))
            g = _CNATrainFn._GEOM[key] = (carr, ho.value, wo.value, lib.rtfs_cna_saved_floats(carr, B, H, W),
                                          lib.rtfs_cna_workspace_bytes(carr, B, H, W), lib.rtfs_cna_grad_floats(carr))
        return g

    @staticmethod
    def forward(ctx, x, cfg, *params):
        import ctypes
        lib = _lib.load()
        x = x.contiguous()
        in_rows, out_rows = (bool(cfg[12]), bool(cfg[13])) if len(cfg) >= 14 else (False, False)
        if in_rows:  # (B, H, W, C) / (B, W, C)
            B, H, W = x.shape[0], (x.shape[1] if x.dim() == 4 else 1), x.shape[-2]
        else:
            B, H, W = x.shape[0], (x.shape[2] if x.dim() == 4 else 1), x.shape[-1]
        world = _bn_world() if (len(cfg) > 11 and cfg[11] and cfg[7] == 3) else 1
        carr, ho_v, wo_v, n_saved, ws_bytes, _ = _CNATrainFn._geom(cfg, world, B, H, W)
        params, running = params[:8], params[8:]  # optional: BatchNorm running mean / var (+ momentum in train mode)
        build = lambda: packing.pack_cna_train(cfg, *params, *running[:2])
        # BatchNorm's running statistics are buffers that change without a version bump the cache could see: pack those afresh
        pk = packing.cached_train_pack(("cna", tuple(cfg[:11])), tuple(params), build) if cfg[7] < 2 else build()
        if out_rows:
            oshape = (B, ho_v, wo_v, cfg[1]) if x.dim() == 4 else (B, wo_v, cfg[1])
        else:
            oshape = (B, cfg[1], ho_v, wo_v) if x.dim() == 4 else (B, cfg[1], wo_v)
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
        saved = torch.empty(n_saved, device=x.device, dtype=torch.float32)
        ws = _lib.workspace(ws_bytes, x.device)

        def run(c):
            _lib.check(lib.rtfs_cna_forward_train_f32(_lib.ptr(x), _lib.ptr(pk), _lib.ptr(out), _lib.ptr(saved), c, B, H, W, _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_of(x)), "rtfs_cna_forward_train_f32")
        if world > 1:  # SyncBatchNorm: batch statistics summed over the ranks between the convolution and the normalisation
            run(_CNATrainFn._carr(cfg, 1, world))
            off = lib.rtfs_cna_saved_stats_offset(carr, B, H, W)
            _bn_all_reduce(saved[off:off + 4 * cfg[1]].view(torch.float64))
            run(_CNATrainFn._carr(cfg, 2, world))
        else:
            run(carr)
        if cfg[7] == 3:  # train-mode BatchNorm: nn.BatchNorm's side effect on its buffers (momentum None = cumulative average is not built)
            rm, rv, momentum = running
            _lib.check(lib.rtfs_cna_bn_update_f32(_lib.ptr(saved), carr, B, H, W, _lib.ptr(rm), _lib.ptr(rv), float(momentum), _lib.stream_of(x)),
                       "rtfs_cna_bn_update_f32")
        ctx.save_for_backward(pk, saved, x if in_rows else None)
        ctx.cfg, ctx.geom, ctx.xshape, ctx.world = tuple(cfg[:11]) + (0, int(in_rows), int(out_rows)), (B, H, W), x.shape, world
        ctx.pshapes = [None if p is None else p.shape for p in params]
        ctx.nrunning = len(running)
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        lib = _lib.load()
        pk, saved, xrows = ctx.saved_tensors
        B, H, W = ctx.geom
        world = ctx.world
        carr, _, _, _, ws_bytes, n_grad = _CNATrainFn._geom(ctx.cfg, world, B, H, W)
        dout = dout.contiguous().to(torch.float32)
        dx = torch.empty(ctx.xshape, device=dout.device, dtype=torch.float32)
        dpar = torch.empty(n_grad, device=dout.device, dtype=torch.float32)
        ws = _lib.workspace(ws_bytes, dout.device)

        def run(c):
            _lib.check(lib.rtfs_cna_backward_f32(_lib.ptr(xrows), _lib.ptr(pk), _lib.ptr(saved), _lib.ptr(dout), _lib.ptr(dx), _lib.ptr(dpar), c,
                                                 B, H, W, _lib.ptr(ws), ws.numel(), _lib.stream_of(dout)), "rtfs_cna_backward_f32")
        if world > 1:  # SyncBatchNorm backward: the input gradient needs the sums of dy and dy * xhat over every rank; the
            # parameter gradients stay local (the gradient all-reduce averages them like every other parameter's)
            run(_CNATrainFn._carr(ctx.cfg, 1, world))
            og, ob = ctypes.c_size_t(), ctypes.c_size_t()
            lib.rtfs_cna_grad_norm_offsets(carr, ctypes.byref(og), ctypes.byref(ob))
            C = ctx.cfg[1]
            local = (dpar[og.value:og.value + C].clone(), dpar[ob.value:ob.value + C].clone())
            both = torch.cat(local)
            _bn_all_reduce(both)
            dpar[og.value:og.value + C] = both[:C]
            dpar[ob.value:ob.value + C] = both[C:]
            run(_CNATrainFn._carr(ctx.cfg, 2, world))
            dpar[og.value:og.value + C] = local[0]
            dpar[ob.value:ob.value + C] = local[1]
        else:
            run(carr)
        if getattr(ctx, "flat_grads", False):
            return dx, dpar
        grads = packing.unpack_cna_grads(ctx.cfg, dpar, ctx.pshapes[3])
        out = (dx, None) + tuple(None if shp is None else g.reshape(shp) for g, shp in zip(grads, ctx.pshapes))
        return out + (None,) * ctx.nrunning


class _GatewayFn(torch.autograd.Function):
    """The RTFS block's gateway on rows (tdanet.py:30-38 applied at :106-108): PReLU(depthwise 1x1 conv(x + x_res)) in one pass each way
    (csrc/k_train_conv.hip gateway_kernel).  Inputs: x, x_res (or None) as (B, T, F, C) rows, the gradient bundle of (conv.weight, conv.bias,
    prelu.weight), and those three parameters hidden in a tuple."""

    @staticmethod
    def forward(ctx, x, x_res, bundle, params):
        lib = _lib.load()
        w, b, slope = (p.detach().contiguous() for p in params)
        x = x.contiguous()
        x_res = None if x_res is None else x_res.contiguous()
        C = x.shape[-1]
        out = torch.empty_like(x)
        _lib.check(lib.rtfs_gateway_forward_train_f32(_lib.ptr(x), _lib.ptr(x_res), _lib.ptr(w), _lib.ptr(b), _lib.ptr(slope), _lib.ptr(out),
                                                      x.numel() // C, C, _lib.stream_of(x)), "rtfs_gateway_forward_train_f32")
        ctx.save_for_backward(x, x_res, w, b, slope)
        ctx.bundle = bundle  # see _BundledFn
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, x_res, w, b, slope = ctx.saved_tensors
        C = x.shape[-1]
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dpar = torch.empty(lib.rtfs_gateway_grad_floats(C), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_gateway_workspace_bytes(C), x.device)
        _lib.check(lib.rtfs_gateway_backward_f32(_lib.ptr(x), _lib.ptr(x_res), _lib.ptr(w), _lib.ptr(b), _lib.ptr(slope), _lib.ptr(dout), _lib.ptr(dx),
                                                 _lib.ptr(dpar), x.numel() // C, C, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
                   "rtfs_gateway_backward_f32")
        return dx, (dx if x_res is not None else None), dpar, None


def gateway_train(cna, x, x_res):
    """``cna`` = the block's gateway ConvNormAct (depthwise 1x1 with bias, no norm, PReLU); x, x_res rows.  None when the module is
    configured differently (the caller then composes add + ConvNormAct)."""
    pre_n, pre_a, conv, nrm, act = cna.full_layer
    C = x.shape[-1]
    if not (isinstance(pre_n, nn.Identity) and isinstance(pre_a, nn.Identity) and isinstance(nrm, nn.Identity) and isinstance(act, nn.PReLU)
            and act.weight.numel() == 1 and isinstance(conv, (nn.Conv1d, nn.Conv2d)) and cna.kernel_size == 1 and cna.stride == 1
            and conv.groups == C and conv.in_channels == C and conv.out_channels == C and conv.bias is not None
            and C % 4 == 0 and C <= 1024 and not (C & (C - 1))):
        return None
    params = (conv.weight, conv.bias, act.weight)
    Cp = (C + 63) // 64 * 64
    unpack = lambda flat: [flat[:C].reshape(conv.weight.shape), flat[Cp:Cp + C], flat[2 * Cp:2 * Cp + 1]]
    n = _lib.load().rtfs_gateway_grad_floats(C)
    bundle = _grad_bundle(("gateway", C), params, n, unpack) if any(p.requires_grad for p in params) else None
    return _GatewayFn.apply(x, x_res, bundle, params)


def _cna_apply(x, cfg, params, running=()):
    """_CNATrainFn with the eight (optional) parameters behind a gradient bundle."""
    import ctypes
    shapes = [None if p is None else p.shape for p in params]

    def unpack(flat):
        grads = packing.unpack_cna_grads(cfg, flat, shapes[3])
        return [g.reshape(shp) for g, shp in zip(grads, shapes) if shp is not None]
    n = _CNATrainFn._geom(tuple(cfg[:11]), 1, 1, 8, 8)[5]  # the gradient layout does not depend on the geometry
    # the running statistics (buffers) and the momentum ride along as extra trailing arguments, as before
    return _apply_bundled(_CNATrainFn, ("cna", tuple(cfg[:11])), x, (cfg,), tuple(params), n, unpack, tuple(running))


class ConvNormAct(nn.Module):
    """Parameter holder with the reference's layout ``full_layer = Sequential(pre_norm, pre_act, conv, norm, act)``
    (conv_layers.py:65-129), hence keys ``full_layer.{0,2,3,4}.*``.  kernel_size <= 0 makes every stage an
    Identity and out_chan = in_chan."""

    def __init__(self, in_chan=1, out_chan=1, kernel_size=-1, stride=1, groups=1, dilation=1, padding=None,
                 pre_norm_type=None, pre_act_type=None, norm_type=None, act_type=None, xavier_init=False, bias=True,
                 is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan = in_chan
        self.out_chan = out_chan if kernel_size > 0 else in_chan
        self.kernel_size, self.stride, self.groups, self.dilation = kernel_size, stride, groups, dilation
        self.pre_norm_type, self.pre_act_type, self.norm_type, self.act_type = pre_norm_type, pre_act_type, norm_type, act_type
        self.xavier_init, self.bias = xavier_init, bias
        self.padding = padding
        if self.padding is None:
            self.padding = dilation * (kernel_size - 1) // 2 if stride > 1 else "same"
        stages = [norm_class(pre_norm_type)(in_chan), act_class(pre_act_type)()]
        if kernel_size > 0:
            conv = (nn.Conv2d if is2d else nn.Conv1d)(in_chan, self.out_chan, kernel_size, stride=stride, padding=self.padding,
                                                      dilation=dilation, groups=groups, bias=bias)
            if xavier_init:
                nn.init.xavier_uniform_(conv.weight)
        else:
            conv = nn.Identity()
        stages += [conv, norm_class(norm_type)(self.out_chan), act_class(act_type)()]
        self.full_layer = nn.Sequential(*stages)

    def forward(self, x):
        if x.is_cuda and _recording(x, self):
            return self._forward_train(x)
        return self.full_layer(x)

    def _forward_train(self, x, rows=(False, False)):
        """The module inside a training step: HIP forward-with-saved-state + backward (rtfs_cna_*_f32).  rows = (input, output) are
        (B, H, W, C) rows instead of (B, C, H, W): how the modules of a block hand tensors to each other without layout changes."""
        pre_n, pre_a, conv, nrm, act = self.full_layer
        if not isinstance(conv, (nn.Conv1d, nn.Conv2d)):
            return x
        if not isinstance(pre_n, (nn.Identity, GlobalLayerNorm)):
            raise RuntimeError(f"ConvNormAct: the training kernels implement gLN as pre-norm only, not {type(pre_n).__name__}")
        bn = isinstance(nrm, (nn.BatchNorm1d, nn.BatchNorm2d, nn.SyncBatchNorm))  # SyncBatchNorm: what convert_sync_batchnorm leaves
        if bn and nrm.training and (nrm.momentum is None or not nrm.track_running_stats):
            raise RuntimeError("ConvNormAct: train-mode BatchNorm needs track_running_stats and a momentum")
        world = _bn_world() if isinstance(nrm, nn.SyncBatchNorm) else 1
        if bn and nrm.training and world * (x.numel() // conv.in_channels) <= self.stride:  # one output value per channel (over all ranks
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")  # if synchronised)
        if not bn and not isinstance(nrm, (nn.Identity, GlobalLayerNorm)):
            raise RuntimeError(f"ConvNormAct: norm {type(nrm).__name__} has no training kernel")
        for m in (pre_a, act):
            if type(m) not in _ACT_CODE:
                raise RuntimeError(f"ConvNormAct: activation {type(m).__name__} has no training kernel")
        depthwise = conv.groups == conv.in_channels and conv.groups == conv.out_channels and conv.groups > 1
        if self.dilation != 1:
            raise RuntimeError("ConvNormAct: dilated convolutions have no training kernel")
        weight = conv.weight
        if not depthwise and conv.groups != 1:
            # grouped 1x1 (CAF's video-side convolutions, fusion.py:218-232): run as a dense 1x1 whose weight is the block-diagonal
            # expansion; the scatter is differentiable, so the gradient comes back in the grouped shape
            if self.kernel_size != 1:
                raise RuntimeError("ConvNormAct: grouped convolutions are supported for kernel_size 1 only")
            cout, gin = conv.out_channels, conv.in_channels // conv.groups
            idx = (torch.arange(cout, device=weight.device) // (cout // conv.groups))[:, None] * gin + torch.arange(gin, device=weight.device)[None, :]
            weight = torch.zeros(cout, conv.in_channels, device=weight.device, dtype=weight.dtype).scatter(1, idx, weight.reshape(cout, gin))
            weight = weight.reshape(cout, conv.in_channels, *([1] * (conv.weight.dim() - 2)))
        is2d = isinstance(conv, nn.Conv2d)
        cfg = (conv.in_channels, conv.out_channels, self.kernel_size, self.stride, int(depthwise), int(isinstance(pre_n, GlobalLayerNorm)),
               _ACT_CODE[type(pre_a)], (3 if nrm.training else 2) if bn else int(isinstance(nrm, GlobalLayerNorm)), _ACT_CODE[type(act)],
               int(conv.bias is not None), int(is2d), int(isinstance(nrm, nn.SyncBatchNorm)), int(rows[0]), int(rows[1]))
        gn = lambda m, a: getattr(m.norm, a) if isinstance(m, GlobalLayerNorm) else (getattr(m, a) if bn and m is nrm else None)
        sl = lambda m: m.weight if isinstance(m, nn.PReLU) else None
        running = ((nrm.running_mean, nrm.running_var) + ((nrm.momentum,) if nrm.training else ())) if bn else ()
        out = _cna_apply(x, cfg, (gn(pre_n, "weight"), gn(pre_n, "bias"), sl(pre_a), weight, conv.bias, gn(nrm, "weight"), gn(nrm, "bias"),
                                  sl(act)), running)
        if bn and nrm.training:
            nrm.num_batches_tracked += 1
        return out

    def get_config(self):
        return _config_of(self)


class ConvActNorm(nn.Module):
    """conv -> act -> norm (conv_layers.py:142-215); keys ``conv.*``, ``act.weight``, ``norm.gamma/beta``."""

    def __init__(self, in_chan=1, out_chan=1, kernel_size=-1, stride=1, groups=1, dilation=1, padding=None, norm_type=None,
                 act_type=None, n_freqs=-1, xavier_init=False, bias=True, is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan, self.out_chan, self.kernel_size, self.stride = in_chan, out_chan, kernel_size, stride
        self.groups, self.dilation, self.norm_type, self.act_type, self.n_freqs = groups, dilation, norm_type, act_type, n_freqs
        self.xavier_init, self.bias = xavier_init, bias
        self.padding = (0 if stride > 1 else "same") if padding is None else padding
        if kernel_size > 0:
            self.conv = (nn.Conv2d if is2d else nn.Conv1d)(in_chan, out_chan, kernel_size, stride=stride, padding=self.padding,
                                                           dilation=dilation, groups=groups, bias=bias)
            if xavier_init:
                nn.init.xavier_uniform_(self.conv.weight)
        else:
            self.conv = nn.Identity()
        self.act = act_class(act_type)()
        self.norm = norm_class(norm_type)((out_chan, n_freqs) if norm_type == "LayerNormalization4D" else out_chan)

    def forward(self, x):
        return self.norm(self.act(self.conv(x)))

    def get_config(self):
        return _config_of(self)


class FeedForwardNetwork(nn.Module):
    """Video-side FFN (conv_layers.py:218-259): 1x1 -> dw conv + ReLU -> 1x1, gLN after the 1x1s, + input."""

    def __init__(self, in_chan, hid_chan, kernel_size=5, norm_type="gLN", act_type="ReLU", dropout=0, is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan, self.hid_chan, self.kernel_size = in_chan, hid_chan, kernel_size
        self.norm_type, self.act_type, self.dropout, self.is2d = norm_type, act_type, dropout, is2d
        self.encoder = ConvNormAct(in_chan, hid_chan, 1, norm_type=norm_type, bias=False, is2d=is2d)
        self.refiner = ConvNormAct(hid_chan, hid_chan, kernel_size, groups=hid_chan, act_type=act_type, is2d=is2d)
        self.decoder = ConvNormAct(hid_chan, in_chan, 1, norm_type=norm_type, bias=False, is2d=is2d)
        self.dropout_layer = nn.Identity()  # DropPath: identity in eval, the only supported mode

    def forward(self, x):
        if x.is_cuda and _recording(x, self):  # conv_layers.py:252-259 with its two DropPath applications
            p = float(self.dropout)
            y = _drop_path(self.refiner(self.encoder(x)), p, self.training)
            return _drop_path(self.decoder(y), p, self.training) + x
        return self.decoder(self.refiner(self.encoder(x))) + x


# ----------------------------------------------------------------------------- SRU operator
class SRUCell(nn.Module):
    """Parameters of one upstream ``sru.SRUCell`` (names weight / weight_c / bias)."""

    def __init__(self, input_size, hidden_size, bidirectional=True):
        super().__init__()
        out = hidden_size * (2 if bidirectional else 1)
        k = 4 if input_size != out else 3
        self.input_size, self.hidden_size, self.num_matrices = input_size, hidden_size, k
        self.weight = nn.Parameter(torch.empty(input_size, out * k))
        self.weight_c = nn.Parameter(torch.empty(2 * out))
        self.bias = nn.Parameter(torch.zeros(2 * out))
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        # upstream v2 defaults: weight ~ U(+-sqrt(3/in)), gate columns and weight_c scaled by sqrt(0.5), bias 0
        lim = (3.0 / self.input_size) ** 0.5
        self.weight.uniform_(-lim, lim)
        w = self.weight.view(self.input_size, -1, self.num_matrices)
        w[:, :, 1].mul_(0.5 ** 0.5)
        w[:, :, 2].mul_(0.5 ** 0.5)
        self.weight_c.uniform_(-(3.0 ** 0.5), 3.0 ** 0.5).mul_(0.5 ** 0.5)
        self.bias.zero_()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "scale_x", None)  # upstream buffer; rescale is off at the reference call site
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


def _sru_pack(sd):
    """A DUALPATH pack whose SRU part is filled and the rest zero (what rtfs_sru_f32 reads)."""
    full = {"norm.gamma": sd["rnn_lst.0.weight"].new_zeros(64), "norm.beta": sd["rnn_lst.0.weight"].new_zeros(64),
            "linear.weight": sd["rnn_lst.0.weight"].new_zeros(64, 64, 8), "linear.bias": sd["rnn_lst.0.weight"].new_zeros(64)}
    full.update({"rnn." + k: v for k, v in sd.items()})
    return packing.pack_dualpath(full)


class _SRUTrainFn(torch.autograd.Function):
    """sru.SRU forward/backward on the training kernels (csrc/k_train_gemm.hip, k_train_rnn.hip).  Inputs: x, then (weight, weight_c, bias) x 4."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        x = x.contiguous()
        L, N, _ = x.shape
        tpack = packing.cached_train_pack("sru", params, lambda: packing.pack_sru_train(params[0::3], params[1::3], params[2::3]))
        h = torch.empty(L, N, 64, device=x.device, dtype=torch.float32)
        saved = torch.empty(lib.rtfs_sru_saved_floats(L, N), device=x.device, dtype=torch.float32)
        _lib.check(lib.rtfs_sru_forward_train_f32(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(h), _lib.ptr(saved), L, N, _lib.stream_of(x)),
                   "rtfs_sru_forward_train_f32")
        ctx.save_for_backward(x, tpack, saved)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        x, tpack, saved = ctx.saved_tensors
        L, N, _ = x.shape
        dh = dh.contiguous().to(torch.float32)
        dx = torch.empty_like(x)
        dpar = torch.empty(lib.rtfs_sru_grad_floats(), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_sru_backward_workspace_bytes(L, N), x.device)
        _lib.check(lib.rtfs_sru_backward_f32(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(saved), _lib.ptr(dh), _lib.ptr(dx), _lib.ptr(dpar), L, N,
                                             _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_sru_backward_f32")
        dws, dwcs, dbs = packing.unpack_sru_grads(dpar)
        out = [dx]
        for i in range(4):
            out += [dws[i], dwcs[i], dbs[i]]
        return tuple(out)


class SRU(PackedModule):
    """Drop-in for ``sru.SRU(input_size=512, hidden_size=32, num_layers=4, bidirectional=True)`` as called at
    reference rnn_layers.py:99-105,150: forward(x (L,N,512)) -> (h (L,N,64), None)."""

    _pack_fn = staticmethod(_sru_pack)

    def __init__(self, input_size, hidden_size, num_layers=2, bidirectional=False, **kwargs):
        super().__init__()
        if not (input_size == 512 and hidden_size == 32 and num_layers == 4 and bidirectional):
            raise ValueError("the MI355X SRU kernel is built for input 512, hidden 32, 4 layers, bidirectional")
        self.input_size, self.hidden_size, self.num_layers, self.bidirectional = input_size, hidden_size, num_layers, bidirectional
        self.rnn_lst = nn.ModuleList([SRUCell(input_size if i == 0 else 2 * hidden_size, hidden_size, True) for i in range(num_layers)])

    def forward(self, x):
        _lib.need_gpu(x)  # no train/eval difference in this operator (no dropout at the reference call site)
        lib = _lib.load()
        if _recording(x, self):
            params = [p for cell in self.rnn_lst for p in (cell.weight, cell.weight_c, cell.bias)]
            return _SRUTrainFn.apply(x, *params), None
        x = x.contiguous()
        L, N, _ = x.shape
        h = torch.empty(L, N, 64, device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_sru_workspace_bytes(L, N), x.device)
        _lib.check(lib.rtfs_sru_f32(_lib.ptr(x), _lib.ptr(self.pack()), _lib.ptr(h), L, N, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_sru_f32")
        return h, None


# ----------------------------------------------------------------------------- dual-path RNN
class _DualPathTrainFn(torch.autograd.Function):
    """DualPathRNN (SRU cell) forward/backward on the training kernels.  Inputs: x, dim, gamma, beta, (weight, weight_c, bias) x 4,
    ConvTranspose1d weight, bias."""

    @staticmethod
    def forward(ctx, x, dim, gamma, beta, *rest):
        lib = _lib.load()
        x = x.contiguous()
        if dim >= 10:  # rows layout (B, T, F, 64)
            B, T, Fq, _ = x.shape
        else:
            B, _, T, Fq = x.shape
        sru, lin_w, lin_b = rest[:12], rest[12], rest[13]
        tpack = packing.cached_train_pack("dualpath", (gamma, beta) + tuple(rest),
                                          lambda: packing.pack_dualpath_train(gamma, beta, sru[0::3], sru[1::3], sru[2::3], lin_w, lin_b))
        out = torch.empty_like(x)
        saved = torch.empty(lib.rtfs_dualpath_saved_floats(B, T, Fq, dim % 10), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_dualpath_train_workspace_bytes(B, T, Fq, dim % 10), x.device)
        _lib.check(lib.rtfs_dualpath_forward_train_f32(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(out), _lib.ptr(saved), B, T, Fq, dim,
                                                       _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_dualpath_forward_train_f32")
        ctx.save_for_backward(x, tpack, saved)
        ctx.dim, ctx.geom = dim, (B, T, Fq)
        ctx.shapes = (gamma.shape, beta.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, tpack, saved = ctx.saved_tensors
        B, T, Fq = ctx.geom
        dout = dout.contiguous().to(torch.float32)
        dx = torch.empty_like(x)
        dpar = torch.empty(lib.rtfs_dualpath_grad_floats(), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_dualpath_train_workspace_bytes(B, T, Fq, ctx.dim % 10), x.device)
        _lib.check(lib.rtfs_dualpath_backward_f32(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(saved), _lib.ptr(dout), _lib.ptr(dx), _lib.ptr(dpar),
                                                  B, T, Fq, ctx.dim, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_dualpath_backward_f32")
        if getattr(ctx, "flat_grads", False):
            return dx, dpar
        return (dx, None) + tuple(_dualpath_unpack(ctx.shapes)(dpar))


def _dualpath_unpack(shapes):
    def unpack(flat):
        dg, db, dws, dwcs, dbs, dlw, dlb = packing.unpack_dualpath_grads(flat)
        out = [dg.reshape(shapes[0]), db.reshape(shapes[1])]
        for i in range(4):
            out += [dws[i], dwcs[i], dbs[i]]
        return out + [dlw, dlb]
    return unpack


def dualpath_train(x, dim, gamma, beta, sru, lin_w, lin_b):
    """_DualPathTrainFn (SRU cells) with the 16 parameters behind a gradient bundle."""
    params = (gamma, beta) + tuple(sru) + (lin_w, lin_b)
    return _apply_bundled(_DualPathTrainFn, "dualpath", x, (dim,), params, _lib.load().rtfs_dualpath_grad_floats(),
                          _dualpath_unpack((gamma.shape, beta.shape)))


class _DualPathLstmTrainFn(torch.autograd.Function):
    """DualPathRNN with a stock torch cell (kind "lstm" or "gru") on the GEMM + scan kernels.  Inputs: x, dim, kind, gamma, beta, the 32
    nn.LSTM / nn.GRU parameters in packing.lstm_param_names() order, ConvTranspose1d weight, bias."""

    @staticmethod
    def forward(ctx, x, dim, kind, gamma, beta, *rest):
        lib = _lib.load()
        x = x.contiguous()
        B, _, T, Fq = x.shape
        names = packing.lstm_param_names()
        cell, lin_w, lin_b = dict(zip(names, rest[:len(names)])), rest[len(names)], rest[len(names) + 1]
        pack_fn = packing.pack_dualpath_lstm_train if kind == "lstm" else packing.pack_dualpath_gru_train
        tpack = packing.cached_train_pack("dualpath_" + kind, (gamma, beta) + tuple(rest), lambda: pack_fn(gamma, beta, cell, lin_w, lin_b))
        fn = lambda n: getattr(lib, f"rtfs_dualpath_{kind}_{n}")
        out = torch.empty_like(x)
        saved = torch.empty(fn("saved_floats")(B, T, Fq, dim), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(fn("train_workspace_bytes")(B, T, Fq, dim), x.device)
        _lib.check(fn("forward_train_f32")(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(out), _lib.ptr(saved), B, T, Fq, dim, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_of(x)), f"rtfs_dualpath_{kind}_forward_train_f32")
        ctx.save_for_backward(x, tpack, saved)
        ctx.dim, ctx.shapes, ctx.names, ctx.kind = dim, (gamma.shape, beta.shape), names, kind
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, tpack, saved = ctx.saved_tensors
        B, _, T, Fq = x.shape
        kind = ctx.kind
        fn = lambda n: getattr(lib, f"rtfs_dualpath_{kind}_{n}")
        dout = dout.contiguous().to(torch.float32)
        dx = torch.empty_like(x)
        dpar = torch.empty(fn("grad_floats")(), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(fn("train_workspace_bytes")(B, T, Fq, ctx.dim), x.device)
        _lib.check(fn("backward_f32")(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(saved), _lib.ptr(dout), _lib.ptr(dx), _lib.ptr(dpar), B, T, Fq, ctx.dim,
                                      _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), f"rtfs_dualpath_{kind}_backward_f32")
        unpack = packing.unpack_dualpath_lstm_grads if kind == "lstm" else packing.unpack_dualpath_gru_grads
        dg, db, dl, dlw, dlb = unpack(dpar)
        return (dx, None, None, dg.reshape(ctx.shapes[0]), db.reshape(ctx.shapes[1])) + tuple(dl[n] for n in ctx.names) + (dlw, dlb)


class DualPathRNN(PackedModule):
    """reference rnn_layers.py:62-162 with rnn_type SRU.  x (B,64,T,F) -> same shape."""

    _pack_fn = staticmethod(packing.pack_dualpath)

    def __init__(self, in_chan, hid_chan, dim, kernel_size=8, stride=1, rnn_type="LSTM", num_layers=1,
                 norm_type="LayerNormalization4D", act_type="Tanh", bidirectional=True, apply_ffn=False, *args, **kwargs):
        super().__init__()
        if not (rnn_type in ("SRU", "LSTM", "GRU") and in_chan == 64 and hid_chan == 32 and kernel_size == 8 and stride == 1 and num_layers == 4
                and bidirectional and norm_type == "LayerNormalization4D" and not apply_ffn and dim in (3, 4)):
            raise ValueError("MI355X DualPathRNN supports the RTFS-Net yaml configuration only "
                             "(SRU, LSTM or GRU, in 64, hid 32, kernel 8, stride 1, 4 layers, bidirectional, LN4D)")
        self.in_chan, self.hid_chan, self.dim, self.kernel_size, self.stride = in_chan, hid_chan, dim, kernel_size, stride
        self.rnn_type, self.num_layers, self.norm_type, self.act_type = rnn_type, num_layers, norm_type, act_type
        self.bidirectional, self.apply_ffn = bidirectional, apply_ffn
        self.num_direction = 2
        self.unfolded_chan = in_chan * kernel_size
        self.rnn_out_chan = hid_chan * 2
        self.norm = LayerNormalization4D((in_chan, 1))
        if rnn_type == "SRU":
            self.rnn = SRU(self.unfolded_chan, hid_chan, num_layers=num_layers, bidirectional=True)
        else:  # parameter holder with nn.LSTM's / nn.GRU's names (weight_ih_l0, weight_hh_l0_reverse, ...); the arithmetic is the HIP kernels'
            self.rnn = getattr(nn, rnn_type)(input_size=self.unfolded_chan, hidden_size=hid_chan, num_layers=num_layers, bidirectional=True)
        self.linear = nn.ConvTranspose1d(self.rnn_out_chan, in_chan, kernel_size, stride=stride)

    def forward(self, x):
        _lib.need_gpu(x)  # no train/eval difference in this module (dropout is off at the reference call site)
        lib = _lib.load()
        x = x.contiguous()
        B, C, T, Fq = x.shape
        if (T if self.dim == 3 else Fq) < self.kernel_size:
            raise ValueError(f"sweep axis shorter than kernel_size {self.kernel_size}")  # nn.Unfold raises in the reference
        long_axis = (T if self.dim == 3 else Fq) > FUSED_MAX_SWEEP  # past the fused kernel's on-chip sweep: the unfused kernels, any length
        if _recording(x, self) or self.rnn_type == "GRU" or long_axis:  # GRU: no fused inference kernel, the GEMM + scan kernels serve both
            if self.rnn_type in ("LSTM", "GRU"):
                cell = [getattr(self.rnn, n) for n in packing.lstm_param_names()]
                return _DualPathLstmTrainFn.apply(x, self.dim, self.rnn_type.lower(), self.norm.gamma, self.norm.beta, *cell, self.linear.weight,
                                                  self.linear.bias)
            sru = [p for cell in self.rnn.rnn_lst for p in (cell.weight, cell.weight_c, cell.bias)]
            return dualpath_train(x, self.dim, self.norm.gamma, self.norm.beta, sru, self.linear.weight, self.linear.bias)
        out = torch.empty_like(x)
        ws = _lib.workspace(lib.rtfs_dualpath_workspace_bytes(B, T, Fq), x.device)
        fn = lib.rtfs_dualpath_sru_f32 if self.rnn_type == "SRU" else lib.rtfs_dualpath_lstm_f32
        _lib.check(fn(_lib.ptr(x), _lib.ptr(self.pack()), _lib.ptr(out), B, T, Fq, self.dim, _lib.ptr(ws), ws.numel(), _lib.stream_of(x)),
                   "rtfs_dualpath_%s_f32" % self.rnn_type.lower())
        return out


# ----------------------------------------------------------------------------- TF self-attention
class _AttentionTrainFn(torch.autograd.Function):
    """MultiHeadSelfAttention2D forward/backward on the training kernels.  Inputs: x, names (tuple), then the parameters in that order."""

    @staticmethod
    def forward(ctx, x, names, rows, *params):
        lib = _lib.load()
        x = x.contiguous()
        B, T = (x.shape[0], x.shape[1]) if rows else (x.shape[0], x.shape[2])
        tpack = packing.cached_train_pack("attention", params, lambda: packing.pack_attention_train(dict(zip(names, params))))
        out = torch.empty_like(x)
        saved = torch.empty(lib.rtfs_tf_attention_saved_floats(B, T), device=x.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_tf_attention_train_workspace_bytes(B, T), x.device)
        _lib.check(lib.rtfs_tf_attention_forward_train_f32(_lib.ptr(x), _lib.ptr(tpack), _lib.ptr(out), _lib.ptr(saved), B, T, int(rows),
                                                           _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_tf_attention_forward_train_f32")
        ctx.save_for_backward(tpack, saved, x if rows else None)
        ctx.names, ctx.geom, ctx.rows = names, (B, T), bool(rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        tpack, saved, xrows = ctx.saved_tensors
        B, T = ctx.geom
        dout = dout.contiguous().to(torch.float32)
        dx = torch.empty_like(dout)
        dpar = torch.empty(lib.rtfs_tf_attention_grad_floats(), device=dout.device, dtype=torch.float32)
        ws = _lib.workspace(lib.rtfs_tf_attention_train_workspace_bytes(B, T), dout.device)
        _lib.check(lib.rtfs_tf_attention_backward_f32(_lib.ptr(xrows), _lib.ptr(tpack), _lib.ptr(saved), _lib.ptr(dout), _lib.ptr(dx), _lib.ptr(dpar),
                                                      B, T, int(ctx.rows), _lib.ptr(ws), ws.numel(), _lib.stream_of(dout)),
                   "rtfs_tf_attention_backward_f32")
        if getattr(ctx, "flat_grads", False):
            return dx, dpar
        g = packing.unpack_attention_grads(dpar)
        return (dx, None, None) + tuple(g[n] for n in ctx.names)


def attention_train(x, names, rows, params):
    """_AttentionTrainFn with the parameters behind a gradient bundle."""
    def unpack(flat):
        g = packing.unpack_attention_grads(flat)
        return [g[n] for n in names]
    return _apply_bundled(_AttentionTrainFn, "attention", x, (names, rows), tuple(params), _lib.load().rtfs_tf_attention_grad_floats(), unpack)


class MultiHeadSelfAttention2D(PackedModule):
    """reference attention.py:76-189 (4 heads, hid_chan 4, n_freqs 64, dim 3)."""

    _pack_fn = staticmethod(packing.pack_attention)

    def __init__(self, in_chan, n_freqs, n_head=4, hid_chan=4, act_type="PReLU", norm_type="LayerNormalization4D", dim=3, *args, **kwargs):
        super().__init__()
        if not (in_chan == 64 and n_freqs == 64 and n_head == 4 and hid_chan == 4 and act_type == "PReLU"
                and norm_type == "LayerNormalization4D" and dim == 3):
            raise ValueError("MI355X MultiHeadSelfAttention2D supports in_chan 64, n_freqs 64, 4 heads, hid_chan 4, dim 3")
        self.in_chan, self.n_freqs, self.n_head, self.hid_chan = in_chan, n_freqs, n_head, hid_chan
        self.act_type, self.norm_type, self.dim = act_type, norm_type, dim
        mk = lambda oc: ConvActNorm(in_chan, oc, 1, act_type=act_type, norm_type=norm_type, n_freqs=n_freqs, is2d=True)
        self.Queries = nn.ModuleList([mk(hid_chan) for _ in range(n_head)])
        self.Keys = nn.ModuleList([mk(hid_chan) for _ in range(n_head)])
        self.Values = nn.ModuleList([mk(in_chan // n_head) for _ in range(n_head)])
        self.attn_concat_proj = mk(in_chan)

    def forward(self, x):
        _lib.need_gpu(x)
        lib = _lib.load()
        x = x.contiguous()
        B, C, T, Fq = x.shape
        if C != 64 or Fq != 64:
            raise ValueError("expected (B, 64, T, 64)")
        if _recording(x, self) or T > FUSED_MAX_KEYS:  # more keys than the fused kernel's LDS score tile: batched-GEMM attention, any length
            names, params = zip(*self.named_parameters())
            return attention_train(x, names, False, params)
        out = torch.empty_like(x)
        ws = _lib.workspace(lib.rtfs_tf_attention_workspace_bytes(B, T), x.device)
        _lib.check(lib.rtfs_tf_attention_f32(_lib.ptr(x), _lib.ptr(self.pack()), _lib.ptr(out), B, T, _lib.ptr(ws), ws.numel(),
                                             _lib.stream_of(x)), "rtfs_tf_attention_f32")
        return out


# ----------------------------------------------------------------------------- TFAR
def _geom(t, rows):
    """(N, H, W, inner C) of a contiguous tensor: channel-first (B, C, H, W) / (B, C, W) -> N = B*C planes, inner 1;
    rows (B, H, W, C) / (B, W, C) -> N = B, inner C."""
    if rows:
        return t.shape[0], (t.shape[1] if t.dim() == 4 else 1), t.shape[-2], t.shape[-1]
    return t.shape[0] * t.shape[1], (t.shape[2] if t.dim() == 4 else 1), t.shape[-1], 1


class _AdaptivePoolFn(torch.autograd.Function):
    """F.adaptive_avg_pool2d / 1d with its adjoint on the HIP kernels (reference call site separators/tdanet.py:116)."""

    @staticmethod
    def forward(ctx, x, size, rows=False):
        lib = _lib.load()
        x = x.contiguous()
        N, H, W, C = _geom(x, rows)
        Ho, Wo = (size[0], size[1]) if x.dim() == 4 else (1, size[-1])
        if rows:
            y = torch.empty((x.shape[0], Ho, Wo, C) if x.dim() == 4 else (x.shape[0], Wo, C), device=x.device, dtype=torch.float32)
        else:
            y = torch.empty(x.shape[:2] + ((Ho, Wo) if x.dim() == 4 else (Wo,)), device=x.device, dtype=torch.float32)
        _lib.check(lib.rtfs_adaptive_avg_pool2d_f32(_lib.ptr(x), _lib.ptr(y), N, H, W, Ho, Wo, C, _lib.stream_of(x)), "rtfs_adaptive_avg_pool2d_f32")
        ctx.geom, ctx.xshape = (N, H, W, Ho, Wo, C), x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        N, H, W, Ho, Wo, C = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(ctx.xshape, device=dy.device, dtype=torch.float32)
        _lib.check(lib.rtfs_adaptive_avg_pool2d_backward_f32(_lib.ptr(dy), _lib.ptr(dx), N, H, W, Ho, Wo, C, _lib.stream_of(dy)),
                   "rtfs_adaptive_avg_pool2d_backward_f32")
        return dx, None, None


class _TfarCombineFn(torch.autograd.Function):
    """local * up(gate) + up(glob) with nearest up-sampling, and its adjoint (layers/fusion.py:54-69)."""

    @staticmethod
    def forward(ctx, local, gate, glob, rows=False):
        lib = _lib.load()
        local, gate, glob = local.contiguous(), gate.contiguous(), glob.contiguous()
        N, H, W, C = _geom(local, rows)
        _, Hg, Wg, _ = _geom(gate, rows)
        out = torch.empty_like(local)
        _lib.check(lib.rtfs_tfar_combine_f32(_lib.ptr(local), _lib.ptr(gate), _lib.ptr(glob), _lib.ptr(out), N, H, W, Hg, Wg, C,
                                             _lib.stream_of(local)), "rtfs_tfar_combine_f32")
        ctx.save_for_backward(local, gate)
        ctx.geom = (N, H, W, Hg, Wg, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        local, gate = ctx.saved_tensors
        N, H, W, Hg, Wg, C = ctx.geom
        dout = dout.contiguous()
        dl, dg, de = torch.empty_like(local), torch.empty_like(gate), torch.empty_like(gate)
        _lib.check(lib.rtfs_tfar_combine_backward_f32(_lib.ptr(dout), _lib.ptr(local), _lib.ptr(gate), _lib.ptr(dl), _lib.ptr(dg), _lib.ptr(de),
                                                      N, H, W, Hg, Wg, C, _lib.stream_of(dout)), "rtfs_tfar_combine_backward_f32")
        return dl, dg, de, None


class _LayoutFn(torch.autograd.Function):
    """(B, C, *spatial) <-> rows (B, *spatial, C) on the tiled transpose kernel; the backward is the opposite change (and hands
    autograd a contiguous gradient, which a permuted view would not)."""

    @staticmethod
    def forward(ctx, x, to_rows):
        lib = _lib.load()
        x = x.contiguous()
        if to_rows:
            B, C, sp = x.shape[0], x.shape[1], tuple(x.shape[2:])
            y = torch.empty((B,) + sp + (C,), device=x.device, dtype=torch.float32)
        else:
            B, C, sp = x.shape[0], x.shape[-1], tuple(x.shape[1:-1])
            y = torch.empty((B, C) + sp, device=x.device, dtype=torch.float32)
        P = 1
        for d in sp:
            P *= d
        _lib.check(lib.rtfs_layout_f32(_lib.ptr(x), _lib.ptr(y), B, C, P, int(to_rows), _lib.stream_of(x)), "rtfs_layout_f32")
        ctx.to_rows = to_rows
        return y

    @staticmethod
    def backward(ctx, dy):
        return _LayoutFn.apply(dy, not ctx.to_rows), None


def adaptive_avg_pool(x, size, rows=False):
    return _AdaptivePoolFn.apply(x, tuple(size), rows)




class InjectionMultiSum(PackedModule):
    """reference layers/fusion.py:9-69.  2-D (audio) instances run on the HIP path; 1-D (video) ones on torch ops."""

    _pack_fn = staticmethod(packing.pack_tfar)

    def __init__(self, in_chan, kernel_size, norm_type="gLN", is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan, self.kernel_size, self.norm_type, self.is2d = in_chan, kernel_size, norm_type, is2d
        mk = lambda act=None: ConvNormAct(in_chan, in_chan, kernel_size, groups=in_chan, norm_type=norm_type, act_type=act, bias=False, is2d=is2d)
        self.local_embedding = mk()
        self.global_embedding = mk()
        self.global_gate = mk("Sigmoid")

    def _forward_train(self, loc, glo, rows=False):
        """Inside a training step: the three ConvNormActs on their training kernels + the combine kernel.  When the global map is
        not smaller than the local one the reference interpolates first (an identity at equal sizes, the only such case on the path).
        rows: inputs and output are (B, H, W, C) rows."""
        sl, sg = (loc.shape[1:-1], glo.shape[1:-1]) if rows else (loc.shape[2:], glo.shape[2:])
        if tuple(sl) != tuple(sg) and all(a <= b for a, b in zip(sl, sg)):
            raise RuntimeError("InjectionMultiSum: a global map larger than the local one does not occur on the RTFS path")
        if rows:
            r = (True, True)
            return _TfarCombineFn.apply(self.local_embedding._forward_train(loc, r), self.global_gate._forward_train(glo, r),
                                        self.global_embedding._forward_train(glo, r), True)
        return _TfarCombineFn.apply(self.local_embedding(loc), self.global_gate(glo), self.global_embedding(glo))

    def forward(self, local_features, global_features):
        if local_features.is_cuda and _recording(local_features, global_features, self):
            return self._forward_train(local_features, global_features)
        if not self.is2d:
            return self._forward_1d(local_features, global_features)
        self._guard(local_features, global_features)
        if self.in_chan != 64 or self.kernel_size != 4 or self.norm_type != "gLN":
            raise ValueError("MI355X TFAR kernel: in_chan 64, kernel 4, gLN")
        lib = _lib.load()
        loc, glo = local_features.contiguous(), global_features.contiguous()
        B, _, H, W = loc.shape
        Hg, Wg = glo.shape[-2:]
        out = torch.empty_like(loc)
        ws = _lib.workspace(lib.rtfs_tfar_workspace_bytes(B, H, W, Hg, Wg), loc.device)
        _lib.check(lib.rtfs_tfar_f32(_lib.ptr(loc), _lib.ptr(glo), _lib.ptr(self.pack()), _lib.ptr(out), B, H, W, Hg, Wg, _lib.ptr(ws),
                                     ws.numel(), _lib.stream_of(loc)), "rtfs_tfar_f32")
        return out

    def _forward_1d(self, loc, glo):
        n_new, n_old = loc.shape[-1], glo.shape[-1]
        le = self.local_embedding(loc)
        if n_new > n_old:
            ge = F.interpolate(self.global_embedding(glo), size=n_new, mode="nearest")
            gate = F.interpolate(self.global_gate(glo), size=n_new, mode="nearest")
        else:
            gi = F.interpolate(glo, size=n_new, mode="nearest")
            ge, gate = self.global_embedding(gi), self.global_gate(gi)
        return le * gate + ge


# ----------------------------------------------------------------------------- CAF cell
class _CafAttentionFn(torch.autograd.Function):
    """(B, 4C, Tv) -> mean over each group of 4 -> softmax over Tv (fusion.py:262-265), with its adjoint."""

    @staticmethod
    def forward(ctx, emb, C):
        lib = _lib.load()
        emb = emb.contiguous()
        B, _, Tv = emb.shape
        att = torch.empty(B, C, Tv, device=emb.device, dtype=torch.float32)
        _lib.check(lib.rtfs_caf_attention_f32(_lib.ptr(emb), _lib.ptr(att), B, C, Tv, _lib.stream_of(emb)), "rtfs_caf_attention_f32")
        ctx.save_for_backward(att)
        return att

    @staticmethod
    def backward(ctx, datt):
        lib = _lib.load()
        (att,) = ctx.saved_tensors
        B, C, Tv = att.shape
        datt = datt.contiguous()
        demb = torch.empty(B, 4 * C, Tv, device=att.device, dtype=torch.float32)
        _lib.check(lib.rtfs_caf_attention_backward_f32(_lib.ptr(att), _lib.ptr(datt), _lib.ptr(demb), B, C, Tv, _lib.stream_of(att)),
                   "rtfs_caf_attention_backward_f32")
        return demb, None


class _CafCombineFn(torch.autograd.Function):
    """key * up(resized) + up(att) * value (fusion.py:255-272), with the four adjoints."""

    @staticmethod
    def forward(ctx, key, value, resized, att, rows=False):
        lib = _lib.load()
        key, value, resized, att = key.contiguous(), value.contiguous(), resized.contiguous(), att.contiguous()
        Tv = resized.shape[-1]
        out = torch.empty_like(key)
        if rows:  # key, value (B, T, F, C)
            B, T, Fq, C = key.shape
            _lib.check(lib.rtfs_caf_combine_rows_f32(_lib.ptr(key), _lib.ptr(value), _lib.ptr(resized), _lib.ptr(att), _lib.ptr(out), B, T, Fq, C, Tv,
                                                     _lib.stream_of(key)), "rtfs_caf_combine_rows_f32")
        else:
            B, C, T, Fq = key.shape
            _lib.check(lib.rtfs_caf_combine_f32(_lib.ptr(key), _lib.ptr(value), _lib.ptr(resized), _lib.ptr(att), _lib.ptr(out), B * C, T, Fq, Tv,
                                                _lib.stream_of(key)), "rtfs_caf_combine_f32")
        ctx.save_for_backward(key, value, resized, att)
        ctx.rows = bool(rows)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        key, value, resized, att = ctx.saved_tensors
        Tv = resized.shape[-1]
        dout = dout.contiguous()
        dk, dv, dr, da = torch.empty_like(key), torch.empty_like(value), torch.empty_like(resized), torch.empty_like(att)
        if ctx.rows:
            B, T, Fq, C = key.shape
            _lib.check(lib.rtfs_caf_combine_rows_backward_f32(_lib.ptr(dout), _lib.ptr(key), _lib.ptr(value), _lib.ptr(resized), _lib.ptr(att),
                                                              _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dr), _lib.ptr(da), B, T, Fq, C, Tv,
                                                              _lib.stream_of(dout)), "rtfs_caf_combine_rows_backward_f32")
        else:
            B, C, T, Fq = key.shape
            _lib.check(lib.rtfs_caf_combine_backward_f32(_lib.ptr(dout), _lib.ptr(key), _lib.ptr(value), _lib.ptr(resized), _lib.ptr(att), _lib.ptr(dk),
                                                         _lib.ptr(dv), _lib.ptr(dr), _lib.ptr(da), B * C, T, Fq, Tv, _lib.stream_of(dout)),
                       "rtfs_caf_combine_backward_f32")
        return dk, dv, dr, da, None


class ATTNFusionCell(PackedModule):
    """reference layers/fusion.py:194-274 (the CAF block's arithmetic)."""

    _pack_fn = staticmethod(packing.pack_caf)

    def __init__(self, in_chan_a, in_chan_b, kernel_size=1, is2d=False, *args, **kwargs):
        super().__init__()
        self.in_chan_a, self.in_chan_b, self.kernel_size, self.is2d = in_chan_a, in_chan_b, kernel_size, is2d
        bn = "BatchNorm2d" if is2d else "BatchNorm1d"
        self.key_embed = ConvNormAct(in_chan_a, in_chan_a, 1, groups=in_chan_a, norm_type=bn, act_type="ReLU", bias=False, is2d=is2d)
        self.value_embed = ConvNormAct(in_chan_a, in_chan_a, 1, groups=in_chan_a, norm_type=bn, bias=False, is2d=is2d)
        self.attention_embed = ConvNormAct(in_chan_b, kernel_size * in_chan_a, 1, groups=in_chan_a, norm_type="gLN")
        self.resize = ConvNormAct(in_chan_b, in_chan_a, 1, groups=in_chan_a, norm_type="gLN")

    def _forward_train(self, a, v, rows=False):
        """Inside a training step (BatchNorm layers in eval mode = frozen statistics): the four ConvNormActs on their training kernels
        plus the attention / combine kernels; reference layers/fusion.py:252-274 line by line.  ``rows``: the audio tensor arrives and
        leaves as (B, T, F, C) rows (the video side stays (B, C, Tv))."""
        if self.kernel_size != 4:
            raise RuntimeError("CAF training kernels: kernel_size 4 (yaml fusion_params)")
        resized = self.resize(v)
        att = _CafAttentionFn.apply(self.attention_embed(v), self.in_chan_a)
        if rows:
            rr = (True, True)
            return _CafCombineFn.apply(self.key_embed._forward_train(a, rr), self.value_embed._forward_train(a, rr), resized, att, True)
        return _CafCombineFn.apply(self.key_embed(a), self.value_embed(a), resized, att)

    def forward(self, tensor_a, tensor_b):
        if tensor_a.is_cuda and _recording(tensor_a, tensor_b, self):
            _lib.need_gpu(tensor_a, tensor_b)
            return self._forward_train(tensor_a, tensor_b)
        self._guard(tensor_a, tensor_b)
        if not (self.is2d and self.in_chan_a == 256 and self.in_chan_b == 512 and self.kernel_size == 4):
            raise ValueError("MI355X CAF kernel: audio 256 ch (2-D), video 512 ch, kernel_size 4")
        lib = _lib.load()
        a, v = tensor_a.contiguous(), tensor_b.contiguous()
        B, _, T, Fq = a.shape
        Tv = v.shape[-1]
        out = torch.empty_like(a)
        ws = _lib.workspace(lib.rtfs_caf_workspace_bytes(B, Tv), a.device)
        _lib.check(lib.rtfs_caf_f32(_lib.ptr(a), _lib.ptr(v), _lib.ptr(self.pack()), _lib.ptr(out), B, T, Fq, Tv, _lib.ptr(ws), ws.numel(),
                                    _lib.stream_of(a)), "rtfs_caf_f32")
        return out


# ----------------------------------------------------------------------------- video-side attention (torch ops)
class PositionalEncoding(nn.Module):
    """reference attention.py:9-25: sinusoidal table kept as the persistent buffer ``pe`` (1, max_len, C)."""

    def __init__(self, channels, max_len=10000, *args, **kwargs):
        super().__init__()
        self.channels, self.max_len = channels, max_len
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        div = torch.exp(torch.arange(0, channels, 2).float() * -(torch.log(torch.tensor(max_len).float()) / channels))
        pe = torch.zeros(max_len, channels)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


class _LnRowsFn(torch.autograd.Function):
    """nn.LayerNorm over the last axis of (..., C) rows on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        lib = _lib.load()
        x = x.contiguous()
        C = x.shape[-1]
        N = x.numel() // C
        y = torch.empty_like(x)
        _lib.check(lib.rtfs_layernorm_rows_f32(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), N, C, _lib.stream_of(x)), "rtfs_layernorm_rows_f32")
        ctx.save_for_backward(x, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma = ctx.saved_tensors
        C = x.shape[-1]
        N = x.numel() // C
        dy = dy.contiguous()
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        _lib.check(lib.rtfs_layernorm_rows_backward_f32(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), N, C,
                                                        _lib.stream_of(x)), "rtfs_layernorm_rows_backward_f32")
        return dx, dg, db


class _LinearRowsFn(torch.autograd.Function):
    """nn.Linear on (..., K) rows: bf16x3 GEMMs forward and for both gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, w = x.contiguous(), weight.contiguous()
        N, K = w.shape
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
        _lib.check(lib.rtfs_linear_rows_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), M, N, K, _lib.stream_of(x)), "rtfs_linear_rows_f32")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        N, K = w.shape
        M = x.numel() // K
        dy = dy.contiguous()
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty(N, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        ws = _lib.workspace(N * K * 4, x.device)
        _lib.check(lib.rtfs_linear_rows_backward_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), M, N, K,
                                                     _lib.ptr(ws), ws.numel(), _lib.stream_of(x)), "rtfs_linear_rows_backward_f32")
        return dx, dw, db


class _MhaCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v per (batch, head) on packed projections (B, T, 3E); optional keep-mask on the attention weights."""

    @staticmethod
    def forward(ctx, qkv, n_head, pmask):
        lib = _lib.load()
        qkv = qkv.contiguous()
        B, T, E3 = qkv.shape
        E = E3 // 3
        o = torch.empty(B, T, E, device=qkv.device, dtype=torch.float32)
        pm = None if pmask is None else pmask.contiguous()
        _lib.check(lib.rtfs_mha_core_f32(_lib.ptr(qkv), _lib.ptr(pm), _lib.ptr(o), B, T, n_head, E // n_head, _lib.stream_of(qkv)), "rtfs_mha_core_f32")
        ctx.save_for_backward(qkv, pm)
        ctx.n_head = n_head
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        qkv, pm = ctx.saved_tensors
        B, T, E3 = qkv.shape
        E = E3 // 3
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        _lib.check(lib.rtfs_mha_core_backward_f32(_lib.ptr(qkv), _lib.ptr(pm), _lib.ptr(do), _lib.ptr(dqkv), B, T, ctx.n_head, E // ctx.n_head,
                                                  _lib.stream_of(qkv)), "rtfs_mha_core_backward_f32")
        return dqkv, None, None


def _drop_path(x, p, training):
    """timm's DropPath (stochastic depth per sample, scale_by_keep) as used at attention.py:54,71 and conv_layers.py:250,256-258."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
    if keep > 0.0:
        mask.div_(keep)
    return x * mask


class MultiHeadSelfAttention(nn.Module):
    """reference attention.py:28-73.  eval(): torch ops (the fused VP kernel covers the whole video block in inference); train() with
    autograd recording: LayerNorm / in_proj / attention core / out_proj on the HIP training kernels, dropout masks from torch's RNG."""

    def __init__(self, in_chan, n_head=8, dropout=0.1, positional_encoding=True, batch_first=True, *args, **kwargs):
        super().__init__()
        assert in_chan % n_head == 0, f"In channels: {in_chan} must be divisible by the number of heads: {n_head}"
        self.in_chan, self.n_head, self.dropout, self.positional_encoding, self.batch_first = in_chan, n_head, dropout, positional_encoding, batch_first
        self.norm1 = nn.LayerNorm(in_chan)
        self.pos_enc = PositionalEncoding(in_chan) if positional_encoding else nn.Identity()
        self.attention = nn.MultiheadAttention(in_chan, n_head, dropout, batch_first=batch_first)
        self.dropout_layer = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(in_chan)
        self.drop_path_layer = nn.Identity()

    def _forward_train(self, x):
        att = self.attention
        if not (self.batch_first and att._qkv_same_embed_dim and att.in_proj_bias is not None and att.bias_k is None and not att.add_zero_attn):
            raise RuntimeError("MultiHeadSelfAttention training kernels: batch_first self-attention with packed in_proj and biases")
        p = float(self.dropout) if self.training else 0.0  # (eval mode reaches here only under force_train_kernels)
        res = x
        y = x.transpose(1, 2).contiguous()                                   # (B, T, C)
        B, T, _ = y.shape
        y = _LnRowsFn.apply(y, self.norm1.weight, self.norm1.bias)
        y = self.pos_enc(y)
        qkv = _LinearRowsFn.apply(y, att.in_proj_weight, att.in_proj_bias)
        pmask = None
        if p > 0.0:  # nn.MultiheadAttention drops attention weights in train mode
            pmask = (torch.rand(B * self.n_head, T, T, device=x.device) >= p).to(torch.float32) / (1.0 - p)
        o = _MhaCoreFn.apply(qkv, self.n_head, pmask)
        o = _LinearRowsFn.apply(o, att.out_proj.weight, att.out_proj.bias)
        y = _LnRowsFn.apply(self.dropout_layer(o) + y, self.norm2.weight, self.norm2.bias)
        return _drop_path(y.transpose(2, 1), p, self.training) + res

    def forward(self, x):
        if x.is_cuda and _recording(x, self):
            return self._forward_train(x)
        res = x
        y = x.transpose(1, 2) if self.batch_first else x
        y = self.pos_enc(self.norm1(y))
        y = self.norm2(self.dropout_layer(self.attention(y, y, y, need_weights=False)[0]) + y)
        if self.batch_first:
            y = y.transpose(2, 1)
        return y + res


class GlobalAttention(nn.Module):
    """reference attention.py:192-220: MHSA followed by the FFN."""

    def __init__(self, in_chan, hid_chan=None, ffn_name="FeedForwardNetwork", kernel_size=5, n_head=8, dropout=0.1, pos_enc=True, *args, **kwargs):
        super().__init__()
        self.in_chan = in_chan
        self.hid_chan = hid_chan if hid_chan is not None else 2 * in_chan
        self.ffn_name, self.kernel_size, self.n_head, self.dropout, self.pos_enc = ffn_name, kernel_size, n_head, dropout, pos_enc
        self.MHSA = MultiHeadSelfAttention(in_chan, n_head, dropout, pos_enc)
        self.FFN = get(ffn_name)(in_chan, self.hid_chan, kernel_size, dropout=dropout)

    def forward(self, x):
        return self.FFN(self.MHSA(x))


_REGISTRY = {c.__name__: c for c in (ConvNormAct, ConvActNorm, FeedForwardNetwork, DualPathRNN, MultiHeadSelfAttention2D,
                                     MultiHeadSelfAttention, GlobalAttention, InjectionMultiSum, ATTNFusionCell)}


def get(identifier):
    """String -> layer class, like reference layers/__init__.py:19-30 (RTFS-Net path classes only)."""
    if identifier is None:
        return nn.Identity
    if callable(identifier):
        return identifier
    if isinstance(identifier, str) and identifier in _REGISTRY:
        return _REGISTRY[identifier]
    raise ValueError("Could not interpret normalization identifier: " + str(identifier))
