/*
 * emu_engine.cpp - TEST-ONLY build of the engine: the same kernel source (stc007_device.h) and the
 * same host logic (engine.inc) compiled with g++ on top of the SIMT emulator in hip_emu.h, so that
 * `-m "not gpu"` tests can run the device code lane by lane against the oracle in the GPU-less
 * container.  Produces tests/emu/libsdvpcm_emu.so; never loaded by the product package.
 */
#define SDV_EMU 1
#define SDV_DEV_AIDS 1
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
#include "../../sdvpcmdecoder_amd/csrc/vis_device.h"
#include "../../sdvpcmdecoder_amd/csrc/engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/stitch_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm1_frames_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_frames_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/pcm16_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/audio_engine.inc"
#include "../../sdvpcmdecoder_amd/csrc/vis_engine.inc"

/* ---- test hooks (emulator build only) -------------------------------------------------------------------------------------------- */
/* bursts_word (64 blocks per call, mask arithmetic) against bursts_block (one block per call) on random flag sequences: returns the number of
 * sequences on which the two disagree.  Flags per block are independent bits with the given densities (per mille), in runs of random length so
 * that long silent / unchecked runs occur. */
extern "C" int sdv_emu_selftest_bursts(uint64_t seed, int iters)
{
    using namespace sdvp16;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        const bool ei = (rnd() & 1) != 0;
        const uint32_t max_sil = ei ? MAX_SIL_EI : MAX_SIL_SI, max_unch = ei ? MAX_UNCH_EI : MAX_UNCH_SI;
        const int n = ei ? 1 + (int)(rnd() % 760) : 35;
        std::vector<uint8_t> fl((size_t)n);
        uint32_t dens[4];
        for (int f = 0; f < 4; f++) { const uint32_t pick = (uint32_t)(rnd() % 6); dens[f] = pick == 0 ? 0u : pick == 1 ? 1000u : pick == 2 ? 20u : pick == 3 ? 980u : (uint32_t)(rnd() % 1000); }
        int i = 0;
        while (i < n) {                                 /* stretches with their own densities */
            const int len = 1 + (int)(rnd() % (ei ? 300 : 20));
            uint32_t d[4];
            for (int f = 0; f < 4; f++) d[f] = (rnd() % 3) ? dens[f] : ((rnd() & 1) ? 0u : 1000u);
            for (int k = 0; k < len && i < n; k++, i++) { uint8_t v = 0; for (int f = 0; f < 4; f++) if (rnd() % 1000 < d[f]) v |= (uint8_t)(1 << f); fl[(size_t)i] = v; }
        }
        /* the block-by-block bookkeeping on flags (bursts_block takes a decoded block; this is its body on the four questions it asks) */
        Bursts a = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int k = 0; k < n; k++) {
            const bool v = fl[(size_t)k] & 1, s = fl[(size_t)k] & 2, u = fl[(size_t)k] & 4, b = fl[(size_t)k] & 8;
            if (v) a.vc++; else if (a.vc > a.vm) a.vm = a.vc;
            if (s) { a.sc++; if (a.sc >= max_sil) a.vc = 0; } else { if (a.sc > a.sm) a.sm = a.sc; a.sc = 0; }
            if (u) { a.uc++; if (a.uc > max_unch) a.vc = 0; } else { if (a.uc > a.um) a.um = a.uc; a.uc = 0; }
            if (b) { a.bc++; if (a.bc >= MAX_BROKEN) a.vc = 0; } else { if (a.bc > a.bm) a.bm = a.bc; a.bc = 0; }
        }
        bursts_end(a);
        BurstsW w = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int k0 = 0; k0 < n;) {
            const int m = ei ? ((rnd() % 4) ? 64 : 1 + (int)(rnd() % 64)) : 35;
            const int cnt = n - k0 < m ? n - k0 : m;
            uint64_t V = 0, S = 0, U = 0, B = 0;
            for (int k = 0; k < cnt; k++) { const uint8_t f = fl[(size_t)(k0 + k)]; V |= (uint64_t)(f & 1) << k; S |= (uint64_t)((f >> 1) & 1) << k; U |= (uint64_t)((f >> 2) & 1) << k; B |= (uint64_t)((f >> 3) & 1) << k; }
            V |= rnd() << cnt % 64 ? (cnt < 64 ? (rnd() << cnt) : 0) : 0;         /* garbage above the count must not matter */
            bursts_word(w, V, S, U, B, (uint32_t)cnt, max_sil, max_unch);
            k0 += cnt;
        }
        bursts_end_w(w);
        if (w.vm != a.vm || w.sm != a.sm || w.um != a.um || w.bm != a.bm) bad++;
    }
    return bad;
}
