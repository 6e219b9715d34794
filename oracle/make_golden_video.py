#!/usr/bin/env python3
"""Golden vectors of the video front-end (CONTAINER ONLY: needs /root/reference).  Builds the reference's own
``FRCNNVideoModel(backbone_type="resnet", relu_type="prelu")`` (src/models/videomodels/frcnn_videomodel.py), loads the seeded
synthetic parameters of oracle.video_oracle.make_video_state_dict, runs it in eval mode on seeded inputs and stores the
output plus probes of internal tensors under tests/golden/video_cases.npz (inputs / parameters are regenerated from seeds)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import make_golden as G  # noqa: E402
from oracle import video_oracle as V  # noqa: E402


def main():
    G.import_reference()
    import importlib
    M = importlib.import_module("src.models.videomodels.frcnn_videomodel")
    model = M.FRCNNVideoModel(backbone_type="resnet", relu_type="prelu", print_macs=False)
    model.eval()  # (the reference's train() override returns None)
    ref_spec = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert ref_spec == [(n, tuple(s)) for n, s in V.video_state_spec()], "state_dict layout differs from the reference"
    sd = V.make_video_state_dict(0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    out = {}
    for k, (B, T) in enumerate([(1, 3), (2, 5)]):
        x = V.make_video_input(B, T, k)
        feats = {}
        hooks = [getattr(model.trunk, f"layer{li}").register_forward_hook(lambda m, i, o, li=li: feats.__setitem__(f"layer{li}", o.detach().numpy()))
                 for li in (1, 2, 3, 4)]
        hooks.append(model.frontend3D.register_forward_hook(lambda m, i, o: feats.__setitem__("stem", o.detach().numpy())))
        with torch.no_grad():
            y = model(torch.from_numpy(x)).numpy()
        for h in hooks:
            h.remove()
        out[f"c{k}_out"] = y
        st = feats["stem"]
        out[f"c{k}_stem"] = st.transpose(0, 2, 1, 3, 4).reshape(-1, 64, st.shape[3], st.shape[4])[:, ::8, ::3, ::3].copy()
        for li in (1, 2, 3, 4):
            out[f"c{k}_layer{li}"] = feats[f"layer{li}"][:, ::16].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "video_cases.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
