"""MI355X-native forward path for the AIGV-Assessor video-quality scorer (hot path only).

Host side: a Python mirror of the reference's ``InternVLChatModel`` API over a C-ABI shared library
(``csrc/`` -> ``libaigv_amd.so``, declared in ``include/aigv_amd.h``) of hand-written gfx950 HIP kernels.
"""
from .config import (InternLM2Config, InternVisionConfig, InternVLChatConfig, internvl2_8b,  # noqa: F401
                     internvl2_26b, tiny)

__all__ = ["InternVLChatConfig", "InternVisionConfig", "InternLM2Config", "internvl2_8b",
           "internvl2_26b", "tiny"]


def __getattr__(name):  # lazy: importing the package must not require torch.cuda / the built library
    if name == "InternVLChatModel":
        from .modeling import InternVLChatModel
        return InternVLChatModel
    if name in ("init_dist", "score_clips_dp"):
        from . import dist_utils
        return getattr(dist_utils, name)
    raise AttributeError(name)
