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
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
