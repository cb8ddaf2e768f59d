"""MI355X-native Tacotron inference hot path (drop-in for yweweler/single-speaker-tts).

The directory name contains a hyphen, so import it with::

    import importlib
    sstts = importlib.import_module('single-speaker-tts_amd')

Sub-packages mirror the reference's module surface for the hot path:
``tacotron.model`` / ``tacotron.inference`` / ``tacotron.params`` and
``audio.synthesis`` / ``audio.conversion`` / ``audio.features`` / ``audio.io``.
All arithmetic runs in libsstts_hip.so (hand-written HIP for gfx950, C ABI in
include/sstts_hip.h); there is no CPU fallback.
"""
__version__ = '0.1.0'

from ._hip import Engine, DeviceArray, TtsError, load_library, exported_symbols, LIB_PATH  # noqa: F401
