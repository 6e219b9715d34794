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


def set_batch_split(n: int):
    """Throughput option of the fused separator call (``include/rtfs_amd.h: rtfs_set_batch_split``): n >= 2 runs the batch as n independent
    chains on internal side streams (batch 32: 14.2 -> 13.0 ms with n = 2); 1 = off (default), 0 = back to the default / ``RTFS_SPLIT``.
    Results per mixture do not depend on it."""
    lib = _lib.load()
    _lib.check(lib.rtfs_set_batch_split(int(n)), "rtfs_set_batch_split")
    lib.rtfs_separator_workspace_bytes.cache_clear()  # (the size queries are memoised; this one follows the setting)
    lib.rtfs_separator_workspace_bytes_ex.cache_clear()  # (split = 0 means this setting)


def load_config(path):
    """Read one of the reference's yaml files (e.g. config/lrs2_RTFSNet_4_layer.yaml) -> dict."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
