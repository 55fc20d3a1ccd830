"""d2r_amd — MI355X-native (gfx950) implementation of the D2R dual-branch dynamic-routing hot path.

Host code (this package) mirrors the reference's ``UnimoModelF`` / ``MSDTrainer`` surface; every arithmetic op is
a hand-written HIP kernel reached through the C ABI of ``include/d2r_hip.h`` (``libd2r_hip.so``).  There is no CPU
or ATen fallback: without the built extension and a GPU the ops raise ``D2RError``.
"""
from ._lib import D2RError, LIB_PATH  # noqa: F401
from .config import TextConfig, VisionConfig, default_args  # noqa: F401

__all__ = ["configure_runtime", "D2RError", "LIB_PATH", "TextConfig", "VisionConfig", "default_args", "UnimoModelF", "UnimoModel",
           "MSDTrainer", "InteractionModule", "Reversed_InteractionModule"]


def configure_runtime(single_thread_autograd: bool = True):
    """Process-wide runtime settings for the one-process-per-GPU design.  With one device per process the autograd
    engine's per-device worker thread only adds a thread hand-off and GIL ping-pong to every backward node (the nodes
    are Python functions that launch kernels): running the backward pass on the calling thread cut the host time of a
    C2 step from 25.3 to 19.0 ms.  Called by bench.py, d2r_amd.run and MSDTrainer; D2R_AUTOGRAD_THREADS=1 keeps
    torch's default."""
    import os

    import torch
    if single_thread_autograd and os.environ.get("D2R_AUTOGRAD_THREADS", "0") != "1":
        torch.autograd.set_multithreading_enabled(False)


def __getattr__(name):  # lazy: importing the package must stay cheap (no torch import for symbol checks)
    if name in ("UnimoModelF", "UnimoModel", "InteractionModule", "Reversed_InteractionModule"):
        from . import modules
        return getattr(modules, name)
    if name == "MSDTrainer":
        from .train import MSDTrainer
        return MSDTrainer
    raise AttributeError(name)
