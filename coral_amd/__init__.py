"""coral_amd — MI355X-native hot path (wav2vec2 CTC / Whisper) behind CoRal's config surface.

The compute lives in libcoral_amd.so (hand-written HIP for gfx950, C ABI in include/coral_amd.h);
this package is the Python host side mirroring the reference's ModelSetup boundary.
"""

import os as _os

# Kernel arguments in device memory instead of host memory: the first instruction of every kernel is a scalar load of its
# argument block, and a training step is ~1 000 dependent launches - with the block in host memory each of them starts
# with a PCIe round trip.  Measured on the XLS-R-2B step (same box, interleaved): 77.4 / 78.8 -> 75.3 / 75.7 ms.  The HIP
# runtime reads the variable when it initialises (the first HIP call of the process), so this has to run before any
# torch.cuda use: import coral_amd first, or export it in the job's environment.  An explicit
# setting in the environment wins.
if "HIP_FORCE_DEV_KERNARG" not in _os.environ:
    _os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
    import sys as _sys

    _torch = _sys.modules.get("torch")
    if _torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized():
        # too late for this process: the runtime read its environment when the caller first touched torch.cuda
        import warnings as _warnings

        _warnings.warn("coral_amd was imported after torch.cuda was initialised and HIP_FORCE_DEV_KERNARG was not in the "
                       "environment: kernel arguments stay in host memory (about 2 ms per XLS-R-2B step slower).  Import "
                       "coral_amd first, or export HIP_FORCE_DEV_KERNARG=1 (INTEGRATION.md).", RuntimeWarning, stacklevel=2)

__version__ = "0.1.0"
