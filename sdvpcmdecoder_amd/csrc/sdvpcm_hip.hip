/*
 * sdvpcm_hip.hip - the one translation unit of libsdvpcm_hip.so (product): HIP kernels for gfx950
 * plus the C-ABI engine.  Build: see sdvpcmdecoder_amd/build.py (hipcc --offload-arch=gfx950).
 */
#include <hip/hip_runtime.h>
#include "stc007_device.h"
#include "stc007_deint_device.h"
#include "stc007_stitch_device.h"
#include "pcm1_stitch_device.h"
#include "pcm1_bin_device.h"
#include "pcm1_frames_device.h"
#include "pcm16_bin_device.h"
#include "pcm16_frames_device.h"
#include "pcm16_stitch_device.h"
#include "audio_device.h"
#include "vis_device.h"
#include "engine.inc"
#include "stitch_engine.inc"
#include "pcm1_engine.inc"
#include "pcm1_frames_engine.inc"
#include "pcm16_frames_engine.inc"
#include "pcm16_engine.inc"
#include "audio_engine.inc"
#include "vis_engine.inc"
