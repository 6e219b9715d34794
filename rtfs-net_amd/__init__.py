"""rtfs_net_amd -- MI355X-native forward pass of the RTFS-Net audio-visual speech separator.

Mirrors the reference's ``src/models`` surface for the RTFS-Net path (``AVNet`` / ``get`` / ``register_model``
plus the north_star names ``RTFSNet``, ``RTFSBlock``, ``CAFBlock``, ``S3Block``); all arithmetic on the audio
path runs in ``librtfs_amd.so`` (hand-written gfx950 HIP kernels behind the C ABI of ``include/rtfs_amd.h``).
"""
from . import _lib, configs, layers, losses, packing, system, torch_utils, videomodels  # noqa: F401
from .packing import invalidate_packs  # noqa: F401
from .system import System  # noqa: F401
from .videomodels import FRCNNVideoModel  # noqa: F401
from .models import (AVNet, ATTNFusion, BaseAVModel, CAFBlock, MaskGenerator, MultiModalFusion, RefinementModule, RTFSBlock, RTFSNet,  # noqa: F401
                     S3Block, STFTDecoder, STFTEncoder, TDANet, TDANetBlock, get, register_model)

__all__ = ["AVNet", "RTFSNet", "RTFSBlock", "CAFBlock", "S3Block", "get", "register_model"]
__version__ = "0.1"


def load_config(path):
    """Read one of the reference's yaml files (e.g. config/lrs2_RTFSNet_4_layer.yaml) -> dict."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
