"""coral_amd — MI355X-native hot path (wav2vec2 CTC / Whisper) behind CoRal's config surface.

The compute lives in libcoral_amd.so (hand-written HIP for gfx950, C ABI in include/coral_amd.h);
this package is the Python host side mirroring the reference's ModelSetup boundary.
"""

__version__ = "0.1.0"
