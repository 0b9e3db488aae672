"""astts -- MI355X-native host package for the AutoStyle-TTS inference hot path.

Python here is plumbing (device memory, streams, torch.distributed); the arithmetic is in
libastts.so (autostyle-tts_amd/csrc, hand-written HIP for gfx950) reached through the C ABI of
include/astts.h.  There is no CPU fallback anywhere in this package.
"""
__version__ = "0.1.0"
