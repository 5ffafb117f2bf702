"""sdvpcmdecoder_amd - MI355X-native decode engine for PCM-in-video (STC-007 path), drop-in for the
hot path of Fagear/SDVPCMdecoder.  The compute lives in libsdvpcm_hip.so (hand-written HIP for gfx950
behind the C-ABI of include/sdvpcm.h); this package is the thin Python plumbing used by tests and the
benchmark (device buffers via torch, ctypes calls).  There is NO CPU path: loading fails loudly when the
library or a HIP device is missing."""
from .engine import (Engine, load_library, LINE_DTYPE, STATS_DTYPE, DEINT_LINE_DTYPE, BLOCK_DTYPE,  # noqa: F401
                     DeintSettings, StitchSettings, StitchInfo, PAIR_DTYPE, Pcm1StitchSettings, Pcm16x0StitchSettings)
