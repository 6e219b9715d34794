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

from rtfs_net_amd.configs import RTFS4_AUDIONET  # noqa: E402  (package data; checked against the reference yaml below)


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
    c["audio_params"]["layers"]["layer_1"]["rnn_type"] = "GRU"  # GRU in one sweep, SRU in the other: mixed cells are not on any yaml
    with pytest.raises(ValueError):
        R.AVNet(print_macs=False, **c)
    c["audio_params"]["layers"]["layer_2"]["rnn_type"] = "GRU"  # both sweeps GRU (SURVEY 8 row a8'): builds, with nn.GRU's parameter names
    g = R.AVNet(print_macs=False, **c)
    names = [k for k in g.state_dict() if ".globalatt.0.rnn." in k]
    assert len(names) == 32 and "refinement_module.audio_net.blocks.globalatt.0.rnn.weight_hh_l3_reverse" in names
    assert g.state_dict()["refinement_module.audio_net.blocks.globalatt.0.rnn.weight_ih_l0"].shape == (96, 512)
    c["audio_params"]["layers"]["layer_2"]["rnn_type"] = "SRU"
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
    with pytest.raises(ValueError):  # the video front-end stays frozen (as in the reference's yaml)
        R.System(audio_model=a0, train_video_model=True)
    assert R.System(audio_model=a0, optimizer=object()).optimizer is not None  # an optimizer is accepted since the backward exists
    with pytest.raises(RuntimeError):
        R.System(audio_model=a0).optimization_step(None)  # ... and required for optimization_step


# ---------------------------------------------------------------------------------------------- training side (host logic)
def test_sru_and_dualpath_training_pack_layouts_roundtrip():
    """pack_*_train re-orders the projections for the training kernels; unpack_*_grads must be the exact inverse
    re-ordering (a gradient laid out like the packed weight comes back in the module's parameter layout)."""
    import torch
    from rtfs_net_amd import packing, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(512, 256, generator=g)] + [torch.randn(64, 192, generator=g) for _ in range(3)]
    wcs = [torch.randn(128, generator=g) for _ in range(4)]
    bs = [torch.randn(128, generator=g) for _ in range(4)]
    tp = packing.pack_sru_train(ws, wcs, bs)
    assert tp.numel() == lib.rtfs_sru_train_pack_floats()
    # the Wp part of the pack (second copy of the projections) is exactly the gradient layout
    off = 256 * 512 + 3 * 192 * 64
    flat = torch.cat([tp[off:off + 512 * 256 + 3 * 64 * 192], torch.stack(wcs).reshape(-1), torch.stack(bs).reshape(-1)])
    assert flat.numel() == lib.rtfs_sru_grad_floats()
    dws, dwcs, dbs = packing.unpack_sru_grads(flat)
    for a, b in zip(dws + dwcs + dbs, ws + wcs + bs):
        assert torch.equal(a, b)
    # Wt is Wp transposed
    assert torch.equal(tp[:256 * 512].reshape(256, 512).t(), tp[off:off + 512 * 256].reshape(512, 256))
    gamma, beta = torch.randn(1, 64, 1, 1, generator=g), torch.randn(1, 64, 1, 1, generator=g)
    lw, lb = torch.randn(64, 64, 8, generator=g), torch.randn(64, generator=g)
    dp = packing.pack_dualpath_train(gamma, beta, ws, wcs, bs, lw, lb)
    assert dp.numel() == lib.rtfs_dualpath_train_pack_floats()
    sru_off = 128 + off
    wcf = dp[128 + lib.rtfs_sru_train_pack_floats():][:64 * 512].reshape(64, 512)
    gflat = torch.cat([dp[:128], dp[sru_off:sru_off + 512 * 256 + 3 * 64 * 192], torch.stack(wcs).reshape(-1), torch.stack(bs).reshape(-1),
                       wcf.t().reshape(-1), lb])
    assert gflat.numel() == lib.rtfs_dualpath_grad_floats()
    dg, db, dws, dwcs, dbs, dlw, dlb = packing.unpack_dualpath_grads(gflat)
    assert torch.equal(dg, gamma.reshape(64)) and torch.equal(db, beta.reshape(64)) and torch.equal(dlb, lb)
    for a, b in zip(dws + dwcs + dbs, ws + wcs + bs):
        assert torch.equal(a, b)
    assert torch.equal(dlw, lw)


def test_gradient_oracle_matches_numpy_oracle_and_finite_differences():
    """oracle/grad_oracle.py: forward equals the numpy restatement; its autograd gradient equals a central finite difference
    of the numpy restatement's own forward (so the backward kernels are checked against something that restates no backward)."""
    import numpy as np
    from oracle import rtfs_oracle as O, grad_oracle as G
    rng = np.random.default_rng(5)
    L, N = 6, 2
    x = rng.standard_normal((L, N, 512)).astype(np.float32)
    layers = []
    for i in range(4):
        din, k = (512, 4) if i == 0 else (64, 3)
        layers.append(((rng.standard_normal((din, 64 * k)) / np.sqrt(din)).astype(np.float32), rng.standard_normal(128).astype(np.float32),
                       (0.1 * rng.standard_normal(128)).astype(np.float32)))
    dh = rng.standard_normal((L, N, 64))
    h, dx, gl = G.sru_grads(x, layers, dh)
    assert np.abs(h - O.sru_forward(x, layers)).max() < 2e-6

    def loss(xv, lay):
        t = [tuple(torch.tensor(p, dtype=torch.float64) for p in l) for l in lay]
        return float((G.sru_forward_torch(torch.tensor(xv, dtype=torch.float64), t) * torch.tensor(dh)).sum())
    import torch
    eps = 1e-5
    for idx in [(0, 0, 3), (5, 1, 500), (2, 0, 77)]:
        xp, xm = x.astype(np.float64).copy(), x.astype(np.float64).copy()
        xp[idx] += eps
        xm[idx] -= eps
        fd = (loss(xp, layers) - loss(xm, layers)) / (2 * eps)
        assert abs(fd - dx[idx]) <= 1e-6 * max(1.0, abs(fd)), (idx, fd, dx[idx])
    for li, pi, idx in [(0, 0, (17, 200)), (2, 1, (70,)), (3, 2, (5,)), (1, 0, (63, 191))]:
        lp = [tuple(np.array(p, dtype=np.float64) for p in l) for l in layers]
        lm = [tuple(np.array(p, dtype=np.float64) for p in l) for l in layers]
        lp[li][pi][idx] += eps
        lm[li][pi][idx] -= eps
        fd = (loss(x, lp) - loss(x, lm)) / (2 * eps)
        assert abs(fd - gl[li][pi][idx]) <= 1e-6 * max(1.0, abs(fd)), (li, pi, idx, fd, gl[li][pi][idx])


def test_gradient_bundle_accumulates_over_applications():
    """layers._GradBundleFn / _apply_bundled (host logic, CPU tensors): a module applied three times through one bundle gets the same
    parameter gradients as plain autograd; the bundle is shared by the applications of one step and renewed after the parameters change;
    the shared zero constant used while packing is never written."""
    import rtfs_net_amd as R
    from rtfs_net_amd import layers as L, packing

    class Affine(torch.autograd.Function):  # y = x * w + b, flat gradient buffer [dw | db] like the C ABI's dparams
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return x * w + b

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            flat = torch.stack([(dy * x).sum(), dy.sum()])
            if getattr(ctx, "flat_grads", False):
                return dy * w, flat
            return dy * w, flat[0], flat[1]

    w, b = torch.nn.Parameter(torch.tensor(1.5)), torch.nn.Parameter(torch.tensor(-0.25))
    x = torch.linspace(-1, 2, 7, requires_grad=True)
    unpack = lambda flat: [flat[0], flat[1]]
    run = lambda t: L._apply_bundled(Affine, "affine", t, (), (w, b), 2, unpack)
    y = run(run(run(x)))
    assert L._grad_bundle("affine", (w, b), 2, unpack) is L._grad_bundle("affine", (w, b), 2, unpack)  # one node per step
    seen, todo = set(), [y.grad_fn]
    while todo:  # the three applications share ONE bundle node although nobody kept the bundle tensor
        node = todo.pop()
        if node is None or node in seen:
            continue
        seen.add(node)
        todo.extend(f for f, _ in node.next_functions)
    assert sum(type(nd).__name__.startswith("_GradBundleFn") for nd in seen) == 1, [type(nd).__name__ for nd in seen]
    y.square().sum().backward()
    got = (x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    f = lambda t: t * w + b
    f(f(f(x))).square().sum().backward()
    for g, r in zip(got, (x.grad, w.grad, b.grad)):
        assert torch.allclose(g, r, rtol=1e-6, atol=1e-6), (g, r)
    before = L._grad_bundle("affine", (w, b), 2, unpack)
    with torch.no_grad():
        w.add_(1.0)  # what an optimizer step does: the version counter moves, the next step gets a fresh node
    assert L._grad_bundle("affine", (w, b), 2, unpack) is not before
    # frozen parameters: no bundle, plain call
    w.requires_grad_(False), b.requires_grad_(False)
    assert torch.allclose(run(x), x * w + b)
    z = packing.zeros_view(5, torch.device("cpu"))
    packed = packing._cat([torch.ones(3), torch.ones(70)])
    assert packed.numel() == 64 + 128 and float(packed.sum()) == 73.0 and float(z.abs().sum()) == 0.0
    assert R is not None


def test_free_gpu_polling_on_amd_smi_output():
    """get_free_gpu_indices (reference src/utils/torch_utils.py:58-75 shells out to nvidia-smi; here amd-smi): canned output of
    `amd-smi process --json` in both document shapes, idle GPUs carry the "No running processes detected" placeholder."""
    from rtfs_net_amd import torch_utils as TU
    idle = {"process_info": "No running processes detected"}
    proc = {"process_info": {"name": "python3", "pid": 4242, "memory_usage": {"vram_mem": {"value": 1024, "unit": "B"}}}}
    doc = [{"gpu": 0, "process_list": [idle]}, {"gpu": 1, "process_list": [proc]}, {"gpu": 2, "process_list": [proc, proc]}, {"gpu": 3, "process_list": [idle]}]
    assert TU.parse_amd_smi_process(json.dumps(doc)) == ([0, 1, 2, 3], [1, 2])
    assert TU.get_free_gpu_indices(run=lambda cmd: json.dumps(doc)) == [0, 3]
    wrapped = "WARNING: something on the first line\n" + json.dumps({"gpu_data": doc})
    assert TU.get_free_gpu_indices(run=lambda cmd: wrapped) == [0, 3]
    assert TU.get_free_gpu_indices(run=lambda cmd: json.dumps([{"gpu": i, "process_list": [idle]} for i in range(8)])) == list(range(8))
    with pytest.raises(ValueError):
        TU.parse_amd_smi_process("ERROR:root:Unable to get devices")


def test_package_survives_relocation(tmp_path):
    """train.py:95 copies src/models into the experiment directory and test.py:33-36 / inference.py:31-34 import THAT copy as
    `<exp_name>.models`: the package (relative imports only, librtfs_amd.so found next to its own files) must work from there."""
    import importlib
    import shutil
    import sys
    exp = tmp_path / "exp_rtfs4"
    shutil.copytree(os.path.join(ROOT, "rtfs-net_amd"), exp / "models", ignore=shutil.ignore_patterns("csrc", "__pycache__"))
    (exp / "__init__.py").write_text("")
    sys.path.append(str(tmp_path))
    try:
        models = importlib.import_module("exp_rtfs4.models")
        assert os.path.dirname(models._lib.LIB_PATH) == str(exp / "models") and b"gfx950" in models._lib.load().rtfs_version()
        m = models.get("AVNet")(print_macs=False, **RTFS4_AUDIONET)
        assert sum(p.numel() for p in m.parameters()) == 739952
        assert type(m).__module__.startswith("exp_rtfs4.models")
    finally:
        sys.path.remove(str(tmp_path))
        for k in [k for k in sys.modules if k.startswith("exp_rtfs4")]:
            del sys.modules[k]


def test_pack_cache_survives_dot_data_updates_after_invalidate():
    """An in-place update through `.data` (EMA swap, weight clipping) has its own version counter and is invisible to the
    (data_ptr, _version) pack key; `rtfs_net_amd.invalidate_packs()` is the documented way to make the kernels see it."""
    import rtfs_net_amd as R
    m = build()
    dp = m.refinement_module.audio_net.blocks.globalatt[0]
    p0 = dp.pack()
    dp.linear.bias.data.add_(1.0)
    assert dp.pack() is p0  # the blind spot
    R.invalidate_packs()
    p1 = dp.pack()
    assert p1 is not p0
    off = 64 + 64 + 512 * 256 + 3 * 64 * 256 + 2 * 4 * 128 + 512 * 64
    assert torch.equal(p1[off:off + 64], dp.linear.bias.detach())


def test_no_hazardous_packed_f32_instructions_in_the_code_object():
    """tools/isa_check.py: no packed VALU instruction (`v_pk_*`, v_pk_mov_b32 included) with an op_sel that feeds a HIGH source half / dword to
    the LOW lane (the instruction form that was caught producing wrong results on MI355X under load in round 2; its cause was never
    established, so the guard covers the whole class; the library is built with -fno-slp-vectorize)."""
    import importlib.util
    from rtfs_net_amd import _lib
    spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.hazardous(_lib.LIB_PATH)
    assert not bad, bad[:5]
    bad = mod.mfma_read_hazards(_lib.LIB_PATH)  # VALU write -> matrix-instruction read needs 2 wait states; inline asm is not padded by the compiler
    assert not bad, bad[:5]
    bad = mod.mfma_result_hazards(_lib.LIB_PATH)  # matrix-instruction VGPR result -> any access needs 11 (hand-written instructions: fences)
    assert not bad, bad[:5]
    assert mod.PAT.search("v_pk_fma_f32 v[72:73], v[168:169], v[76:77], v[72:73] op_sel:[0,1,1]")
    assert not mod.PAT.search("v_pk_fma_f32 v[154:155], v[154:155], v[76:77], v[72:73] op_sel_hi:[1,0,0]")
    assert mod.PAT.search("v_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]")
    assert mod.PAT.search("v_pk_mul_f16 v1, v2, v3 op_sel:[0,1] op_sel_hi:[1,0]")
    assert not mod.PAT.search("v_fma_mixhi_f16 v1, v2, 1.0, -v3 op_sel:[0,0,1] op_sel_hi:[0,0,1]")  # not a packed op: one result lane


def test_isa_check_wait_state_rules_on_synthetic_streams(monkeypatch):
    """The two wait-state checks of tools/isa_check.py on hand-made instruction streams (the shipped library itself is checked above)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_check2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def run(fn, stream):
        monkeypatch.setattr(mod, "device_disassembly", lambda so: [("k", i) for i in stream])
        return fn("unused.so")

    mf = "v_mfma_f32_32x32x16_f16 a[0:15], v[64:67], v[8:11], a[0:15]"
    # VALU write of a source register directly / one instruction / two wait states in front of the matrix instruction
    assert len(run(mod.mfma_read_hazards, ["v_mov_b32_e32 v66, v65", mf])) == 1
    assert len(run(mod.mfma_read_hazards, ["v_mov_b32_e32 v66, v65", "s_nop 0", mf])) == 1
    assert not run(mod.mfma_read_hazards, ["v_mov_b32_e32 v66, v65", "s_nop 1", mf])
    assert not run(mod.mfma_read_hazards, ["v_mov_b32_e32 v66, v65", "s_waitcnt vmcnt(0)", "s_nop 0", mf])
    assert not run(mod.mfma_read_hazards, ["v_mov_b32_e32 v70, v65", mf])  # not a source
    # a hand-written (VGPR-form) matrix instruction's result: read or overwritten too early, or behind a fence
    mv = "v_mfma_f32_32x32x16_f16 v[0:15], v[134:137], v[88:91], v[0:15]"
    assert len(run(mod.mfma_result_hazards, [mv] + ["s_mov_b32 s1, s2"] * 9 + ["v_lshlrev_b32_e32 v0, 1, v202"])) == 1   # overwritten
    assert len(run(mod.mfma_result_hazards, [mv, "s_nop 7", "v_add_f32_e32 v20, v3, v3"])) == 1                            # read
    assert len(run(mod.mfma_result_hazards, [mv, "s_nop 7", "buffer_store_dword v3, v9, s[0:3], 0 offen"])) == 1           # stored
    assert not run(mod.mfma_result_hazards, [mv, "s_nop 15", "s_nop 3", "v_add_f32_e32 v20, v3, v3"])
    assert not run(mod.mfma_result_hazards, [mv] + [mf] * 12 + ["v_add_f32_e32 v20, v3, v3"])   # a dozen matrix instructions in between
    assert not run(mod.mfma_result_hazards, [mf, "v_accvgpr_read_b32 v1, a15"])                  # AGPR results are the compiler's business


def test_batch_split_setter_and_workspace_layout():
    """rtfs_set_batch_split: argument range, and the separator workspace follows the setting (parts of >= 8 mixtures only); the Python
    helper clears the memoised size query."""
    import rtfs_net_amd as R
    from rtfs_net_amd import _lib
    lib = _lib.load()
    try:
        assert lib.rtfs_set_batch_split(9) != 0 and lib.rtfs_set_batch_split(-1) != 0
        R.set_batch_split(1)
        one = [lib.rtfs_separator_workspace_bytes(B, 4096, 7) for B in (32, 17, 15, 9)]
        R.set_batch_split(2)
        two = [lib.rtfs_separator_workspace_bytes(B, 4096, 7) for B in (32, 17, 15, 9)]
        assert two[0] != one[0] and two[1] != one[1]      # 16 + 16, 8 + 9: two arenas, different padding
        assert two[2] == one[2] and two[3] == one[3]      # 15 and 9 mixtures are not split (a part would be < 8)
        assert all(0 <= t - o < 704 * 1024 for t, o in zip(two, one))  # per part: padding + the 32 KB encoder fragment image + two 288 KB padded weight images
    finally:
        R.set_batch_split(0)
