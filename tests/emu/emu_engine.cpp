/*
 * emu_engine.cpp - TEST-ONLY build of the engine: the same kernel source (stc007_device.h) and the
 * same host logic (engine.inc) compiled with g++ on top of the SIMT emulator in hip_emu.h, so that
 * `-m "not gpu"` tests can run the device code lane by lane against the oracle in the GPU-less
 * container.  Produces tests/emu/libsdvpcm_emu.so; never loaded by the product package.
 */
#define SDV_EMU 1
#define SDV_P16_STITCH_BATCH 3         /* small batches: the hand-over between the batches of a call (histories, conv_queue's remainder) is exercised by every tape */
#include "hip_emu.h"
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
#include "../../sdvpcmdecoder_amd/csrc/stc007_device.h"
#include "../../sdvpcmdecoder_amd/csrc/stc007_deint_device.h"
#include "../../sdvpcmdecoder_amd/csrc/stc007_stitch_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_stitch_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_bin_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_frames_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_bin_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_frames_device.h"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_stitch_device.h"
#include "../../sdvpcmdecoder_amd/csrc/audio_device.h"
#include "../../sdvpcmdecoder_amd/csrc/engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/stitch_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_frames_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_frames_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/audio_engine.inc"
