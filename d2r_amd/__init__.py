"""d2r_amd — MI355X-native (gfx950) implementation of the D2R dual-branch dynamic-routing hot path.

Host code (this package) mirrors the reference's ``UnimoModelF`` / ``MSDTrainer`` surface; every arithmetic op is
a hand-written HIP kernel reached through the C ABI of ``include/d2r_hip.h`` (``libd2r_hip.so``).  There is no CPU
or ATen fallback: without the built extension and a GPU the ops raise ``D2RError``.
"""
from ._lib import D2RError, LIB_PATH  # noqa: F401
from .config import TextConfig, VisionConfig, default_args  # noqa: F401

__all__ = ["D2RError", "LIB_PATH", "TextConfig", "VisionConfig", "default_args", "UnimoModelF", "UnimoModel",
           "MSDTrainer", "InteractionModule", "Reversed_InteractionModule"]


def __getattr__(name):  # lazy: importing the package must stay cheap (no torch import for symbol checks)
    if name in ("UnimoModelF", "UnimoModel", "InteractionModule", "Reversed_InteractionModule"):
        from . import modules
        return getattr(modules, name)
    if name == "MSDTrainer":
        from .train import MSDTrainer
        return MSDTrainer
    raise AttributeError(name)
