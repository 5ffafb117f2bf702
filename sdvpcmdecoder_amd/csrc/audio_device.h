/*
 * audio_device.h - device side of sdv_audio_process / sdv_wav_pack (include/sdvpcm.h): the reference's AudioProcessor
 * (audioprocessor.cpp, dropout masking on the PCMSamplePair stream) for gfx950.  Compiled by hipcc into the product and by g++
 * on the SIMT emulator for the CPU tests (tests/emu).
 *
 * The reference works through a stream window by window (512 pairs, three stay as look-behind), strictly in order, because
 * where a window starts depends on how many pairs the windows before it could put out.  What makes it parallel here:
 *   - a window without an invalid sample changes nothing and puts out all but three of its pairs, so the windows of a clean
 *     stretch need not be looked at: the prepare pass leaves one bit per pair ("some channel is invalid") in a three-level
 *     bitmap, and the wave that walks a file's windows jumps from one damaged place to the next in a few loads;
 *   - inside a window nothing is sequential: every invalid sample finds the valid samples around its run in the validity
 *     masks (eight 64-bit ballots per channel), decides from them which region of fixBadSamples it belongs to (ramp down,
 *     silence, ramp up, plain region, forced zero) and computes its own value in closed form;
 *   - the stretches between the NEW_FILE / END_FILE tags start from a purged window and are independent: one wave each.
 * Layout: the pairs of a call are laid out in a work array W, stretch after stretch, each stretch headed by what the window
 * held when it began (the silent pair purgePipeline leaves, or what waited from the call before); windows are worked on in
 * place in W.  W is the caller's output buffer where no purge shifts what follows it (a burst that continues or holds one file);
 * otherwise a buffer of the engine, and the emit pass copies it to the caller's without the pair every purge drops.
 */
#pragma once
#include <stdint.h>
#include "../../include/sdvpcm.h"

#ifndef SDV_AP_NT_STORES
#define SDV_AP_NT_STORES 1
#endif

namespace sdva {

enum { WIN = SDV_AP_BUF_SIZE, KEEP = SDV_AP_MIN_VALID_BEFORE, RAMP_DOWN = SDV_AP_MAX_RAMP_DOWN, RAMP_UP = SDV_AP_MAX_RAMP_UP,
       STRIDE = WIN - KEEP, SCAN_MIN = KEEP + RAMP_DOWN + RAMP_UP, PLAY_MIN = SCAN_MIN + 1, CALC_MULT = 16 };
enum { END_OPEN = 0, END_NEW_FILE = 1, END_END_FILE = 2 };      /* what ends a stretch: the end of the call, or a tag */
enum { HOW_MUTE = 0, HOW_HOLD = 1, HOW_LIN = 2 };
enum { MAX_TAGS = 65536 };
enum { RES_STALLED = 1, RES_PLAN_MISMATCH = 2 };
enum { MIN_ADVANCE = RAMP_DOWN - 1 };      /* a full window that does not stall puts out at least 191 pairs: what bounds the window list */

/* One stretch of the stream between two purges (host-built, audio_engine.inc). */
struct Stretch {
    uint32_t w_base;        /* where its pairs start in W */
    uint32_t v_len;         /* head + data pairs */
    uint32_t head;          /* pairs in front of the data: what the window held when the stretch began */
    uint32_t in_start;      /* its first data pair in the call's input */
    uint64_t out_base;      /* where its first pair goes in the output */
    uint8_t end_kind, head_is_carry, stop, _pad;
    uint32_t _pad2;
};
struct StretchResult { uint32_t popped, scanned_upto, flags, left; uint64_t masked; uint32_t n_win, _pad; };

__device__ __forceinline__ uint32_t a_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src); }
__device__ __forceinline__ uint64_t lane_read64(uint64_t v, int src)      /* the value lane `src` (the same for every lane) holds */
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

/* ---- tags: positions of the service pairs of the input ------------------------------------------------------------- */
struct TagArgs { const sdv_sample_pair *pairs; uint32_t n; uint32_t *count; uint64_t *list; /* (position << 8) | service_type, in no particular order */ };
__device__ inline void tags_body(const TagArgs &a, uint32_t blk, int lane)
{
    for (int r = 0; r < 16; r++) {
        const uint32_t i = (blk * 16u + (uint32_t)r) * 64u + (uint32_t)lane;
        const uint8_t srv = i < a.n ? a.pairs[i].service_type : (uint8_t)0;
        if (srv != 0) { const uint32_t k = atomicAdd(a.count, 1u); if (k < (uint32_t)MAX_TAGS) a.list[k] = ((uint64_t)i << 8) | srv; }
    }
}

/* ---- prepare: W and the bitmap ---------------------------------------------------------------------------------- */
struct PrepArgs {
    const sdv_sample_pair *pairs; const sdv_sample_pair *carry; const Stretch *st; uint32_t n_st; uint32_t total_w;
    sdv_sample_pair *w; uint64_t *bad1; uint8_t *bad2; uint64_t *bad3;   /* one bit per pair; one byte per 64 pairs (zeroed up to a multiple of 64 bytes); one bit per 4096 pairs */
    uint64_t *v0, *v1, *m0, *m1;        /* per channel: word_valid, word_masked of every pair of W (a set bit past the end of W in v0 / v1) */
    uint32_t *has_mi;                   /* set when some sample is masked without being valid (only foreign input can be) */
    uint8_t by_block, ignore;
};
__device__ inline uint32_t find_stretch(const Stretch *st, uint32_t n_st, uint32_t p)
{
    uint32_t lo = 0, hi = n_st;         /* the last stretch whose w_base <= p */
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (st[mid].w_base <= p) lo = mid; else hi = mid; }
    return lo;
}
/* a pair that is written once and read (if at all) by a later kernel */
__device__ __forceinline__ void store_pair_streaming(sdv_sample_pair *dst, const sdv_sample_pair &q)
{
#if defined(SDV_EMU) || !SDV_AP_NT_STORES
    *dst = q;
#else
    const uint32_t *src = (const uint32_t *)&q; uint32_t *d = (uint32_t *)dst;
    __builtin_nontemporal_store(src[0], d); __builtin_nontemporal_store(src[1], d + 1); __builtin_nontemporal_store(src[2], d + 2);
#endif
}
__device__ inline sdv_sample_pair silent_pair()
{
    sdv_sample_pair q;                  /* purgePipeline's setSamplePair(0, 0, true x4, false x2) on a cleared pair (:1734-1743) */
    q.audio_word[0] = q.audio_word[1] = 0; q.sample_flags[0] = q.sample_flags[1] = SDV_SF_BLOCK_OK | SDV_SF_WORD_VALID;
    q.sample_rate = 44056; q.emphasis = 0; q.service_type = 0; q._pad = 0;
    return q;
}
__device__ inline void prep_body(const PrepArgs &a, uint32_t blk, int lane)
{
    /* the stretch the wave's first pair lies in - almost always the stretch of all 256: looked up once, on the scalar unit */
    const uint32_t p_first = blk * 256u;
    const uint32_t s_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)find_stretch(a.st, a.n_st, p_first < a.total_w ? p_first : a.total_w - 1u));
    const Stretch t_first = a.st[s_first];
    const bool one = p_first + 256u <= t_first.w_base + t_first.v_len || s_first + 1u == a.n_st;
    /* all four loads of the wave first, then the work: four times the bytes in flight */
    sdv_sample_pair qs[4]; bool is_data[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t p = (blk * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
        qs[r] = silent_pair(); is_data[r] = false;
        if (p < a.total_w) {
            Stretch t = t_first;
            if (!one) t = a.st[find_stretch(a.st, a.n_st, p)];
            const uint32_t k = p - t.w_base;
            if (k >= t.head) { qs[r] = a.pairs[t.in_start + (k - t.head)]; is_data[r] = true; }
            else if (t.head_is_carry) qs[r] = a.carry[k];
        }
    }
    /* the four bitmap words of the wave go out together: lanes 0..3 store word 0..3 of each bitmap (one 32-byte store per bitmap) */
    uint64_t w_bad = 0, w_v0 = 0, w_v1 = 0, w_m0 = 0, w_m1 = 0; uint32_t any_mi = 0, bytes = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t p = (blk * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
        bool bad = false, ok0 = true, ok1 = true, mk0 = false, mk1 = false;
        if (p < a.total_w) {
            sdv_sample_pair q = qs[r];
            if (is_data[r]) {
                q._pad = 0;
                if (a.by_block) for (int ch = 0; ch < 2; ch++)      /* setValidityByBlock (:166-169) */
                    q.sample_flags[ch] = (uint8_t)((q.sample_flags[ch] & ~SDV_SF_WORD_VALID) | ((q.sample_flags[ch] & SDV_SF_BLOCK_OK) ? SDV_SF_WORD_VALID : 0));
            }
            store_pair_streaming(&a.w[p], q);
            ok0 = a.ignore || (q.sample_flags[0] & SDV_SF_WORD_VALID); ok1 = a.ignore || (q.sample_flags[1] & SDV_SF_WORD_VALID);
            mk0 = (q.sample_flags[0] & SDV_SF_WORD_MASKED) != 0; mk1 = (q.sample_flags[1] & SDV_SF_WORD_MASKED) != 0;
            bad = !(ok0 && ok1);
        }
        const uint64_t m = __ballot(bad), b0 = __ballot(ok0), b1 = __ballot(ok1), c0 = __ballot(mk0), c1 = __ballot(mk1);
        if (lane == r) { w_bad = m; w_v0 = b0; w_v1 = b1; w_m0 = c0; w_m1 = c1; }
        if ((c0 & ~b0) | (c1 & ~b1)) any_mi = 1;
        bytes |= (m ? 1u : 0u) << (8 * r);
    }
    const uint32_t wi = blk * 4u + (uint32_t)lane, n_words = (a.total_w + 63u) >> 6;
    if (lane < 4 && wi < n_words) { a.bad1[wi] = w_bad; a.v0[wi] = w_v0; a.v1[wi] = w_v1; a.m0[wi] = w_m0; a.m1[wi] = w_m1; }
    if (lane == 0) {
        /* the middle level: a byte per word (the array is zeroed up to a multiple of 64 bytes, whole groups of four lie inside it) */
        *(uint32_t *)&a.bad2[blk * 4u] = bytes;
        if (any_mi && !*a.has_mi) atomicOr(a.has_mi, 1u);
    }
}

/* the top level from the middle one: a lane per block of 64 bytes (no atomics anywhere: on a worn tape every wave of the prepare
 * pass has something to report, and thousands of atomics on a handful of words cost more than the pass itself) */
struct SummArgs { const uint8_t *bad2; uint32_t n_blocks; uint64_t *bad3; };
__device__ inline void summ_body(const SummArgs &a, uint32_t blk, int lane)
{
    const uint32_t g = blk * 64u + (uint32_t)lane;
    uint64_t acc = 0;
    if (g < a.n_blocks) for (int k = 0; k < 8; k++) acc |= ((const uint64_t *)a.bad2)[g * 8u + (uint32_t)k];
    const uint64_t m = __ballot(acc != 0);
    if (lane == 0) a.bad3[blk] = m;
}

/* The first position >= p with its bit set, `limit` if there is none below it.  A wave collective (lanes 0..7 read the eight
 * 64-bit words of a 64-byte block of the middle level). */
__device__ inline uint32_t next_bad(const uint64_t *bad1, const uint8_t *bad2, const uint64_t *bad3, uint32_t p, uint32_t limit, int lane)
{
    if (p >= limit) return limit;
    const uint32_t w = p >> 6;
    const uint64_t x = bad1[w] & (~0ull << (p & 63u));
    if (x) { const uint32_t q = w * 64u + (uint32_t)__ffsll((unsigned long long)x) - 1u; return q < limit ? q : limit; }
    uint32_t g = w >> 6, skip = (w & 63u) + 1u;          /* the block of 64 words, and how many of its bytes lie at or before w */
    for (;;) {
        uint64_t y = lane < 8 ? ((const uint64_t *)bad2)[g * 8u + (uint32_t)lane] : 0ull;
        const uint32_t first = (uint32_t)lane * 8u;     /* byte index of this lane's first byte in the block */
        if (skip > first) y &= skip - first >= 8u ? 0ull : (~0ull << ((skip - first) * 8u));
        const uint64_t m = __ballot(y != 0);
        if (m) {
            const int f = __ffsll((unsigned long long)m) - 1;
            const uint64_t yf = lane_read64(y, f);
            const uint32_t w1 = g * 64u + (uint32_t)f * 8u + (((uint32_t)__ffsll((unsigned long long)yf) - 1u) >> 3);
            const uint32_t q = w1 * 64u + (uint32_t)__ffsll((unsigned long long)bad1[w1]) - 1u;
            return q < limit ? q : limit;
        }
        /* nothing more in this block: the next block that holds something */
        uint32_t h = g >> 6;
        uint64_t z = bad3[h] & ((g & 63u) == 63u ? 0ull : (~0ull << ((g & 63u) + 1u)));
        const uint32_t h_end = ((limit - 1u) >> 18) + 1u;
        while (!z) { h++; if (h >= h_end) return limit; z = bad3[h]; }
        g = h * 64u + (uint32_t)__ffsll((unsigned long long)z) - 1u; skip = 0;
        if (g * 4096u >= limit) return limit;
    }
}

/* ---- one window ------------------------------------------------------------------------------------------------- */
struct WinLds { int16_t val[2][WIN]; uint8_t flg[2][WIN]; };

__device__ __forceinline__ int16_t fill_value(int how, int i, int a, int b, int va, int vb)
{
    /* rangeMute / rangeLevelHold / rangeLinearInterpolation (:511-737) for the sample i strictly inside (a, b) */
    if (how == HOW_MUTE) return 0;
    if (how == HOW_HOLD || va == vb) return (int16_t)va;
    const int32_t base = va * CALC_MULT, delta = vb * CALC_MULT - base, cnt = b - a;
    const int32_t step = (delta + cnt / 2) / cnt;
    return (int16_t)((step * (i - a) + base + CALC_MULT / 2) / CALC_MULT);
}

/* scanBuffer + fixBadSamples for both channels + the count outputAudio can put out, for the n pairs at w[0..n).
 * Returns the number of pairs that leave from the front; `masked` gets the guiAddMask count. */
__device__ inline uint32_t window_body(sdv_sample_pair *w, int n, bool file_end, int how, WinLds &lds, uint32_t &masked, int lane)
{
    uint32_t d0[8], d1[8];
    for (int j = 0; j < 8; j++) {
        const int i = lane + 64 * j;
        if (i < n) { const uint32_t *src = (const uint32_t *)&w[i]; d0[j] = src[0]; d1[j] = src[1]; } else { d0[j] = 0; d1[j] = 0; }
        lds.val[0][i] = (int16_t)(d0[j] & 0xFFFFu); lds.val[1][i] = (int16_t)(d0[j] >> 16);
        lds.flg[0][i] = (uint8_t)(d1[j] & 0xFFu); lds.flg[1][i] = (uint8_t)((d1[j] >> 8) & 0xFFu);
    }
    __syncthreads();
    uint32_t changed = 0;
    for (int ch = 0; ch < 2; ch++) {
        uint64_t vm[8];
        for (int j = 0; j < 8; j++) vm[j] = __ballot(lane + 64 * j < n && (lds.flg[ch][lane + 64 * j] & SDV_SF_WORD_VALID));
        /* the highest valid entry below word j, the lowest valid entry above word j */
        int top[8], bot[8];
        { int t = -1; for (int j = 0; j < 8; j++) { top[j] = t; if (vm[j]) t = 64 * j + 63 - __clzll((unsigned long long)vm[j]); } }
        { int t = -1; for (int j = 7; j >= 0; j--) { bot[j] = t; if (vm[j]) t = 64 * j + __ffsll((unsigned long long)vm[j]) - 1; } }
        int16_t nv[8]; uint8_t nf[8];
        for (int j = 0; j < 8; j++) {
            const int i = lane + 64 * j;
            int16_t v = lds.val[ch][i]; uint8_t f = lds.flg[ch][i];
            if (i < n && !(f & SDV_SF_WORD_VALID)) {
                const uint64_t lo = vm[j] & ((1ull << lane) - 1ull), hi = lane == 63 ? 0ull : (vm[j] & (~0ull << (lane + 1)));
                const int below = lo ? 64 * j + 63 - __clzll((unsigned long long)lo) : top[j];      /* good_end */
                const int above = hi ? 64 * j + __ffsll((unsigned long long)hi) - 1 : bot[j];       /* good_at_the_end */
                int a = -1, b = -1, va = 0, vb = 0; bool point = false;
                if (below >= 0) {
                    const int vbelow = lds.val[ch][below];
                    if (above < 0) {
                        /* nothing valid behind the run (:848-897) */
                        if (below < n - (RAMP_DOWN + RAMP_UP + 1)) {
                            const int down = below + RAMP_DOWN + 1;
                            if (i < down) { a = below; b = down; va = vbelow; vb = 0; } else if (i == down) point = true;
                        }
                    } else {
                        const int vabove = lds.val[ch][above];
                        const int leftover = above - below - 1;
                        const bool altered = (lds.flg[ch][below] & SDV_SF_WORD_MASKED) != 0 && vbelow == 0;     /* :919-920 */
                        const int up = above - RAMP_UP - 1, down = below + RAMP_DOWN + 1;
                        if (!altered && leftover > RAMP_DOWN + RAMP_UP) {
                            if (i < down) { a = below; b = down; va = vbelow; vb = 0; }
                            else if (i == down || i == up) point = true;
                            else if (i < up) { a = down; b = up; va = 0; vb = 0; }
                            else { a = up; b = above; va = 0; vb = vabove; }
                        } else if (altered && leftover > RAMP_UP) {
                            if (i < up) { a = below; b = up; va = vbelow; vb = 0; }
                            else if (i == up) point = true;
                            else { a = up; b = above; va = 0; vb = vabove; }
                        } else { a = below; b = above; va = vbelow; vb = vabove; }
                    }
                }
                if (point) { v = 0; f |= SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED; }         /* sampleMute (:495-508) */
                else if (a >= 0) {
                    const int16_t x = fill_value(how, i, a, b, va, vb);
                    if (x != v) { v = x; f |= SDV_SF_WORD_MASKED; changed++; }
                    f |= SDV_SF_WORD_VALID;
                }
            }
            nv[j] = v; nf[j] = f;
        }
        __syncthreads();
        for (int j = 0; j < 8; j++) { lds.val[ch][lane + 64 * j] = nv[j]; lds.flg[ch][lane + 64 * j] = nf[j]; }
        __syncthreads();
        if (file_end) {
            /* what is still invalid at the very end of the file ramps into a forced zero, linearly in every mode (:1122-1172) */
            uint64_t vm2[8];
            for (int j = 0; j < 8; j++) vm2[j] = __ballot(lane + 64 * j < n && (lds.flg[ch][lane + 64 * j] & SDV_SF_WORD_VALID));
            const int last = n - 1;
            const bool last_bad = last >= 1 && !((vm2[last >> 6] >> (last & 63)) & 1ull);
            if (last_bad) {
                int a = 0;
                for (int j = 0; j < 8; j++) { const uint64_t m = j == 0 ? (vm2[0] & ~1ull) : vm2[j]; if (m) a = 64 * j + 63 - __clzll((unsigned long long)m); }
                const int va = lds.val[ch][a];
                __syncthreads();
                for (int j = 0; j < 8; j++) {
                    const int i = lane + 64 * j;
                    if (i > a && i < last) {
                        const int16_t x = fill_value(HOW_LIN, i, a, last, va, 0);
                        if (x != lds.val[ch][i]) { lds.val[ch][i] = x; lds.flg[ch][i] |= SDV_SF_WORD_MASKED; changed++; }
                        lds.flg[ch][i] |= SDV_SF_WORD_VALID;
                    } else if (i == last) { lds.val[ch][i] = 0; lds.flg[ch][i] |= SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED; }
                }
                __syncthreads();
            }
        }
    }
    for (int d = 1; d < 64; d <<= 1) changed += a_shfl(changed, lane ^ d);
    masked = changed;
    /* back to W, and how many pairs can leave: the first four of the queue must be valid or masked (:1310-1337) */
    int first_wait = n;
    for (int j = 7; j >= 0; j--) {
        const int i = lane + 64 * j;
        const uint8_t f0 = lds.flg[0][i], f1 = lds.flg[1][i];
        const bool ready = (f0 & (SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED)) && (f1 & (SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED));
        const uint64_t m = __ballot(i < n && !ready);
        if (m) first_wait = 64 * j + __ffsll((unsigned long long)m) - 1;
        if (i < n) {
            uint32_t *dst = (uint32_t *)&w[i];
            const uint32_t n0 = (uint32_t)(uint16_t)lds.val[0][i] | ((uint32_t)(uint16_t)lds.val[1][i] << 16), n1 = (d1[j] & 0xFFFF0000u) | f0 | ((uint32_t)f1 << 8);
            if (n0 != d0[j]) dst[0] = n0;
            if (n1 != d1[j]) dst[1] = n1;
        }
    }
    __syncthreads();
    if (n < (file_end ? (int)KEEP : (int)PLAY_MIN)) return 0;
    int pops = first_wait - KEEP; if (pops < 0) pops = 0;
    if (pops > n - KEEP) pops = n - KEEP;
    return (uint32_t)pops;
}

/* ---- the plan: where the windows of a stretch start, from the validity bitmaps alone ------------------------------------ */
/* What a scan does to the *validity* of a window does not depend on the sample values: every run of invalid samples that has a
 * valid sample on both sides comes out valid (each of its samples is inside a region or is a forced zero), a run that reaches the
 * end of the window gets its first 193 samples valid if the ramps fit (fixBadSamples :848-897) and waits otherwise, a run that
 * starts at entry 0 stays, and at the end of a file the tail is ramped out (:1122-1172).  So one wave can walk a stretch window
 * after window on 512-bit masks - the only strictly sequential part - and leave a list of the windows that hold invalid samples;
 * the sample work of those windows then runs in parallel (exec_body).  A window depends on the one before it only when their
 * overlap (the pairs that stayed behind) holds a sample that was invalid in the input: such windows form a chain that one wave
 * works off in order, every other listed window is the head of a chain of its own. */
enum { CHUNK_WORDS = 3072, CHUNK_PAD = 16, LEAP_WORDS = (STRIDE * 63 + WIN + 127) / 64 + 1 };
/* (a leap reads nine words per lane, the lanes 509 bits = 7.95 words apart: eight-word steps land sixteen lanes on the same LDS banks, so the staged
 * words are kept nine apart - one unused word behind every eight) */
__device__ __forceinline__ uint32_t bmi(uint32_t i) { return i + (i >> 3); }
struct PlanLds { uint64_t bm[3][CHUNK_WORDS + CHUNK_PAD + (CHUNK_WORDS + CHUNK_PAD) / 8 + 1]; };    /* word_valid of the two channels, staged: word i at bmi(i); [2]: pairs with an invalid sample (what a leap asks) */
struct WinRec { uint32_t w_pos; uint32_t pops; uint16_t n; uint8_t file_end, head; };
struct PlanArgs { const Stretch *st; uint32_t n_st; const uint64_t *v0, *v1, *m0, *m1, *bad1; const uint8_t *bad2; const uint64_t *bad3; const uint32_t *has_mi; uint32_t n_words;
                  WinRec *wins; const uint32_t *win_base; StretchResult *res; };

__device__ __forceinline__ uint64_t word_range(int j, int a, int b)      /* the bits a..b (inclusive) that fall into word j */
{
    const int lo = a > 64 * j ? a : 64 * j, hi = b < 64 * j + 63 ? b : 64 * j + 63;
    if (lo > hi) return 0ull;
    return (~0ull >> (63 - (hi - lo))) << (lo - 64 * j);
}
/* A 512-bit mask spread over the wave: lane j < 8 holds bits 64j .. 64j+63, the other lanes hold 0.  One vector instruction then
 * works on all eight words, the questions about the whole mask are a ballot plus one lane read.  Every member is a wave collective:
 * call them from wave-uniform control flow only. */
__device__ __forceinline__ uint64_t lane_pull64(uint64_t v, int src)      /* the value lane `src` (own choice of every lane) holds */
{
    const uint32_t lo = a_shfl((uint32_t)v, src), hi = a_shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
struct Bits512 {
    uint64_t w;
    __device__ __forceinline__ static uint64_t range(int lane, int a, int b) { return lane < 8 ? word_range(lane, a, b) : 0ull; }
    __device__ __forceinline__ void keep_below(int n, int lane) { w &= n > 0 ? range(lane, 0, n - 1) : 0ull; }
    __device__ __forceinline__ void set_range(int a, int b, int lane) { if (a <= b) w |= range(lane, a, b); }
    __device__ __forceinline__ bool any() const { return __ballot(w != 0) != 0; }
    __device__ __forceinline__ int lowest() const
    {
        const uint64_t m = __ballot(w != 0);
        const int j = m ? __ffsll((unsigned long long)m) - 1 : 0;
        const uint64_t x = lane_read64(w, j);
        return m ? 64 * j + __ffsll((unsigned long long)x) - 1 : -1;
    }
    __device__ __forceinline__ int highest() const
    {
        const uint64_t m = __ballot(w != 0);
        const int j = m ? 63 - __clzll((unsigned long long)m) : 0;
        const uint64_t x = lane_read64(w, j);
        return m ? 64 * j + 63 - __clzll((unsigned long long)x) : -1;
    }
    __device__ __forceinline__ bool bit(int i) const { return (lane_read64(w, i >> 6) >> (i & 63)) & 1ull; }
    __device__ __forceinline__ void shift_down(int k, int lane)      /* bit i <- bit i + k */
    {
        const int q = k >> 6, r = k & 63;
        const int s0 = lane + q < 63 ? lane + q : 63, s1 = lane + q + 1 < 63 ? lane + q + 1 : 63;      /* lane 63 holds 0 */
        const uint64_t lo = lane_pull64(w, s0), hi = lane_pull64(w, s1);
        w = r ? (lo >> r) | (hi << (64 - r)) : lo;
    }
};
/* 512 bits of a staged bitmap from bit position `rel` on */
__device__ __forceinline__ void take512_staged(const uint64_t *bm, uint32_t rel, Bits512 &o, int lane)      /* ... of a bitmap staged in PlanLds */
{
    const uint32_t q = rel >> 6, r = rel & 63u;
    const uint64_t x = lane < 9 ? bm[bmi(q + (uint32_t)lane)] : 0ull;
    const uint64_t nx = lane_pull64(x, lane < 63 ? lane + 1 : 63);
    o.w = lane < 8 ? (r ? (x >> r) | (nx << (64u - r)) : x) : 0ull;
}
__device__ __forceinline__ void take512(const uint64_t *bm, uint32_t rel, Bits512 &o, int lane)
{
    const uint32_t q = rel >> 6, r = rel & 63u;
    const uint64_t x = lane < 9 ? bm[q + (uint32_t)lane] : 0ull;
    const uint64_t nx = lane_pull64(x, lane < 63 ? lane + 1 : 63);
    o.w = lane < 8 ? (r ? (x >> r) | (nx << (64u - r)) : x) : 0ull;
}

/* the validity of one channel after a scan of the window (n pairs) */
__device__ inline void scan_validity(Bits512 &v, int n, bool file_end, int lane)
{
    const int fv = v.lowest(), lv = v.highest();
    if (fv >= 0) {
        Bits512 o; o.w = 0;
        o.set_range(fv, lv, lane);                          /* the runs between valid samples are repaired */
        if (lv < n - (RAMP_DOWN + RAMP_UP + 1)) o.set_range(lv + 1, lv + RAMP_DOWN + 1, lane);      /* ramp down + its forced zero */
        v = o;
    }
    if (file_end && n >= 2 && !v.bit(n - 1)) {
        Bits512 x = v; if (lane == 0) x.w &= ~1ull;
        const int h = x.highest();
        v.set_range((h < 0 ? 0 : h) + 1, n - 1, lane);
    }
}

__device__ inline void plan_body(const PlanArgs &a, uint32_t s, PlanLds &lds, int lane)
{
    const Stretch t = a.st[s];
    const uint32_t total = t.v_len, limit = t.w_base + total;
    const bool has_mi = *a.has_mi != 0;
    WinRec *wins = a.wins + a.win_base[s];
    uint32_t S = 0, L = t.head, flags = 0, scanned = 0, n_win = 0;
    uint32_t chunk = 0xFFFFFFFFu;                           /* the first bitmap word staged in LDS */
    Bits512 ev0, ev1; bool have_ev = false, std_left = false;  /* std_left: they are all valid */                 /* validity of the pairs that stayed behind, as the scans left it */
    bool prev_adjacent = false;
    /* pf: the validity of the pairs that stayed behind is known in the short form the quick step below keeps it in - channel c valid exactly below entry pw[c] */
    bool pf = false; uint32_t pw0 = 0, pw1 = 0;
    bool stopper = false;       /* the window at S is the one the last leap stopped at: no use asking again */
    ev0.w = 0; ev1.w = 0;
#if defined(SDV_AP_STATS) && !defined(SDV_EMU)
    unsigned long long st_t[5] = { 0, 0, 0, 0, 0 }; uint32_t st_n[5] = { 0, 0, 0, 0, 0 }; unsigned long long st_mark = __builtin_readcyclecounter();
#define AP_STAT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_t[i] += now_ - st_mark; st_n[i]++; st_mark = now_; } while (0)
#else
#define AP_STAT(i) ((void)0)
#endif
    for (;;) {
        if (L >= (uint32_t)WIN) { flags |= RES_STALLED; break; }       /* a full window nothing can leave: the worker takes no more input */
        const uint32_t avail = total - (S + L);
        const uint32_t take = avail < (uint32_t)WIN - L ? avail : (uint32_t)WIN - L;
        const uint32_t n = L + take;
        const bool at_end = take == avail && n < (uint32_t)WIN;         /* the fill got as far as the tag (or the queue ran dry) */
        if (at_end && t.end_kind == END_NEW_FILE) break;                /* purged as it is, before any scan (:120-137) */
        if (at_end && t.end_kind == END_OPEN && take == 0) break;       /* nothing was added: the worker sleeps */
        const bool file_end = at_end && t.end_kind == END_END_FILE;
        const uint32_t gpos = t.w_base + S, gw = gpos >> 6;
        if (chunk == 0xFFFFFFFFu || gw < chunk || gw + (uint32_t)LEAP_WORDS > chunk + (uint32_t)CHUNK_WORDS) {
            __syncthreads();
            chunk = gw;
            /* (CHUNK_WORDS + CHUNK_PAD) / 64 = 48.25 rounds: sixteen rounds of loads in flight at a time */
            for (uint32_t base = 0; base < (uint32_t)(CHUNK_WORDS + CHUNK_PAD); base += 1024u) {
                uint64_t x0[16], x1[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t i = base + (uint32_t)u * 64u + (uint32_t)lane, wi = chunk + i;
                    const bool in = i < (uint32_t)(CHUNK_WORDS + CHUNK_PAD) && wi < a.n_words;
                    x0[u] = in ? a.v0[wi] : ~0ull; x1[u] = in ? a.v1[wi] : ~0ull;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t i = base + (uint32_t)u * 64u + (uint32_t)lane;
                    if (i < (uint32_t)(CHUNK_WORDS + CHUNK_PAD)) { const uint32_t at = bmi(i); lds.bm[0][at] = x0[u]; lds.bm[1][at] = x1[u]; lds.bm[2][at] = ~(x0[u] & x1[u]); }
                }
            }
            __syncthreads();
            AP_STAT(0);
        }
        const bool was_stopper = stopper; stopper = false;
        if (!was_stopper && L == (uint32_t)KEEP && have_ev && std_left && total - S >= (uint32_t)WIN + (uint32_t)STRIDE) {
            /* Leap: with three valid pairs behind it, a full window whose last pair is valid closes every run it holds - it comes out
             * valid throughout and 509 pairs leave, whatever else is in it.  So from here the windows start 509 apart up to the first
             * one whose last pair is invalid: 64 candidates are tested at once, a lane each; those with an invalid sample are listed. */
            const uint32_t cnt = (total - S - (uint32_t)WIN) / (uint32_t)STRIDE + 1u, nact = cnt < 64u ? cnt : 64u;
            const uint32_t P = gpos + (uint32_t)lane * (uint32_t)STRIDE, prel = P - chunk * 64u, pq = prel >> 6, pr = prel & 63u;
            bool last_bad = false, dirty = false, own3 = false;
            if ((uint32_t)lane < nact) {
                /* the nine words the window touches, read once: all three questions are answered from them */
                uint64_t w[9];
                const uint32_t b9 = bmi(pq), low = pq & 7u;
#pragma unroll
                for (int k = 0; k < 9; k++) w[k] = lds.bm[2][b9 + (uint32_t)k + ((low + (uint32_t)k) >> 3)];
                const uint64_t first = w[0] & (~0ull << pr), last = pr ? w[8] & ~(~0ull << pr) : 0ull;
                dirty = (first | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7] | last) != 0;
                last_bad = pr ? (w[8] >> (pr - 1u)) & 1ull : (w[7] >> 63) & 1ull;                          /* pair 511 of the window */
                own3 = (((w[0] >> pr) | (pr ? w[1] << (64u - pr) : 0ull)) & (uint64_t)((1u << KEEP) - 1u)) != 0;      /* its first three pairs */
            }
            const uint64_t evm = __ballot((uint32_t)lane < nact && last_bad);
            const uint32_t j_ev = evm ? (uint32_t)__ffsll((unsigned long long)evm) - 1u : nact;
            const uint64_t lm = __ballot((uint32_t)lane < j_ev && dirty);
            if ((uint32_t)lane < j_ev && dirty) {
                WinRec rec; rec.w_pos = P; rec.pops = STRIDE; rec.n = (uint16_t)WIN; rec.file_end = 0;
                rec.head = (own3 && (lane > 0 || prev_adjacent)) ? 0 : 1;
                wins[n_win + (uint32_t)__popcll((unsigned long long)(lm & (((uint64_t)1 << lane) - 1ull)))] = rec;
            }
            n_win += (uint32_t)__popcll((unsigned long long)lm);
            if (j_ev > 0) {
                S += j_ev * (uint32_t)STRIDE; scanned = S + KEEP; prev_adjacent = true;
                pf = true; pw0 = pw1 = KEEP;
                stopper = j_ev < nact;
                AP_STAT(1);
                continue;
            }
        }
        const uint32_t rel = gpos - chunk * 64u;
        if (pf && have_ev && !has_mi && n == (uint32_t)WIN && pw0 >= 1u && pw1 >= 1u) {
            /* Quick step, for the window a leap stops at and the ones behind it: a full window in the middle of a file whose pairs behind are valid up to
             * a point in either channel (no earlier hole), and whose last valid sample in either channel is too close to the end for a ramp.  Then a scan
             * repairs every run in front of a channel's last valid sample and nothing behind it (scan_validity: fv = 0, no ramp), the pairs wait from the
             * first channel's last valid sample on, and what stays behind is in the same short form again: four questions to the bitmaps instead of the
             * 512-bit walk below.  Anything else (a long run at the end, a stall, nothing invalid in reach) takes the walk. */
            const uint32_t q0 = rel >> 6;
            uint64_t x0 = 0, x1 = 0;
            if (lane < 9) { x0 = lds.bm[0][bmi(q0 + (uint32_t)lane)]; x1 = lds.bm[1][bmi(q0 + (uint32_t)lane)]; }
            const int wlo = (int)(rel & 63u);
            const uint64_t m_new = lane < 9 ? word_range(lane, wlo + (int)L, wlo + WIN - 1) : 0ull, m_beh = lane < 9 ? word_range(lane, wlo, wlo + (int)L - 1) : 0ull;
            const uint64_t v0 = x0 & m_new, v1 = x1 & m_new;
            const uint64_t b0 = __ballot(v0 != 0), b1 = __ballot(v1 != 0);
            int lv0 = (int)pw0 - 1, lv1 = (int)pw1 - 1;
            if (b0) { const int j = 63 - __clzll((unsigned long long)b0); lv0 = 64 * j + 63 - __clzll((unsigned long long)lane_read64(v0, j)) - wlo; }
            if (b1) { const int j = 63 - __clzll((unsigned long long)b1); lv1 = 64 * j + 63 - __clzll((unsigned long long)lane_read64(v1, j)) - wlo; }
            const bool any_inv = pw0 < L || pw1 < L || __ballot((~(x0 & x1) & m_new) != 0) != 0;
            const int wait0 = lv0 + 1, wait1 = lv1 + 1, first_wait = wait0 < wait1 ? wait0 : wait1;
            int pp = first_wait - KEEP; if (pp > WIN - KEEP) pp = WIN - KEEP;
            if (any_inv && lv0 >= WIN - (RAMP_DOWN + RAMP_UP + 1) && lv1 >= WIN - (RAMP_DOWN + RAMP_UP + 1) && pp > 0) {
                const bool dep_q = __ballot((~(x0 & x1) & m_beh) != 0) != 0;
                if (lane == 0) {
                    WinRec rec; rec.w_pos = gpos; rec.pops = (uint32_t)pp; rec.n = (uint16_t)WIN; rec.file_end = 0; rec.head = (prev_adjacent && dep_q) ? 0 : 1;
                    wins[n_win] = rec;
                }
                n_win++;
#ifdef SDV_EMU
                if (lane == 0) { static unsigned long hits = 0; static const bool tr = getenv("SDV_AP_QUICK_TRACE") != NULL; if (tr && (++hits % 1000) == 1) fprintf(stderr, "[ap quick step] %lu\n", hits); }
#endif
                scanned = S + n;
                prev_adjacent = true;
                S += (uint32_t)pp; L = (uint32_t)WIN - (uint32_t)pp;
                pw0 = (uint32_t)(wait0 - pp); pw1 = (uint32_t)(wait1 - pp);
                ev0.w = Bits512::range(lane, 0, (int)pw0 - 1); ev1.w = Bits512::range(lane, 0, (int)pw1 - 1);
                std_left = pw0 >= L && pw1 >= L;
                AP_STAT(4);
                continue;
            }
        }
        Bits512 o0, o1, c0, c1;
        take512_staged(lds.bm[0], rel, o0, lane); take512_staged(lds.bm[1], rel, o1, lane);
        o0.keep_below((int)n, lane); o1.keep_below((int)n, lane);
        c0 = o0; c1 = o1;
        const uint64_t behind = Bits512::range(lane, 0, (int)L - 1);      /* the pairs that stayed behind */
        if (have_ev) { c0.w = (o0.w & ~behind) | (ev0.w & behind); c1.w = (o1.w & ~behind) | (ev1.w & behind); }
        Bits512 inv; inv.w = ~(c0.w & c1.w);
        inv.keep_below((int)n, lane);
        /* does this window read what the one before it writes?  (the pairs that stayed behind, where they were invalid in the input) */
        const bool dep_now = __ballot((~(o0.w & o1.w) & behind) != 0) != 0;
        const bool scan = file_end ? n > 0 : n >= (uint32_t)SCAN_MIN;
        const uint32_t play_min = file_end ? (uint32_t)KEEP : (uint32_t)PLAY_MIN;
        uint32_t pops = 0;
        if (!inv.any()) {
            /* no invalid sample in reach: nothing changes, everything but the look-behind leaves; full windows in a row are skipped at once */
            prev_adjacent = false;
            if (!at_end) {
                const uint32_t q = next_bad(a.bad1, a.bad2, a.bad3, gpos + L, limit, lane) - t.w_base;
                const uint32_t end = q < total ? q : total;
                const uint32_t m = (end - S - (uint32_t)WIN) / (uint32_t)STRIDE + 1u;
                S += m * (uint32_t)STRIDE; L = KEEP;
                scanned = S + KEEP; have_ev = false; pf = false;
                AP_STAT(2);
                continue;
            }
            if (scan) scanned = S + n;
            pops = (scan && n >= play_min) ? n - KEEP : 0u;
            ev0 = c0; ev1 = c1;
        } else if (scan) {
            scanned = S + n;
            Bits512 r0 = c0, r1 = c1;
            scan_validity(r0, (int)n, file_end, lane); scan_validity(r1, (int)n, file_end, lane);
            Bits512 wait;       /* pairs that cannot leave: a channel neither valid nor masked (PCMSamplePair::isReadyForOutput) */
            if (has_mi) { Bits512 k0, k1; take512(a.m0 + (gpos >> 6), gpos & 63u, k0, lane); take512(a.m1 + (gpos >> 6), gpos & 63u, k1, lane); wait.w = ~((r0.w | k0.w) & (r1.w | k1.w)); }
            else wait.w = ~(r0.w & r1.w);
            wait.keep_below((int)n, lane);
            const int lw = wait.lowest();
            const int first_wait = lw >= 0 ? lw : (int)n;
            if (n >= play_min) {
                int pp = first_wait - KEEP; if (pp < 0) pp = 0;
                if (pp > (int)n - KEEP) pp = (int)n - KEEP;
                pops = (uint32_t)pp;
            }
            if (lane == 0) {
                WinRec rec; rec.w_pos = gpos; rec.pops = pops; rec.n = (uint16_t)n; rec.file_end = file_end ? 1 : 0; rec.head = (prev_adjacent && dep_now) ? 0 : 1;
                wins[n_win] = rec;
            }
            n_win++;
            if (pops == 0 && n == (uint32_t)WIN && !file_end) { L = n; flags |= RES_STALLED; break; }
            prev_adjacent = true;
            ev0 = r0; ev1 = r1;
        } else { ev0 = c0; ev1 = c1; prev_adjacent = false; }
        if (pops) { ev0.shift_down((int)pops, lane); ev1.shift_down((int)pops, lane); }
        have_ev = true;
        S += pops; L = n - pops;
        { Bits512 z; z.w = ~(ev0.w & ev1.w); z.keep_below((int)L, lane); std_left = !z.any(); }
        pf = std_left; pw0 = pw1 = L;
        AP_STAT(3);
        if (at_end) break;
    }
#if defined(SDV_AP_STATS) && !defined(SDV_EMU)
    if (lane == 0 && total > 100000u) printf("[ap plan] stretch %u: %u pairs, %u windows listed; chunk loads %u (%llu cycles), leaps %u (%llu), clean skips %u (%llu), windows walked %u (%llu), quick steps %u (%llu)\n", s, total, n_win,
                                             st_n[0], st_t[0], st_n[1], st_t[1], st_n[2], st_t[2], st_n[3], st_t[3], st_n[4], st_t[4]);
#endif
    if (lane == 0) {
        StretchResult r; r.popped = S; r.scanned_upto = scanned; r.flags = flags; r.left = total - S; r.masked = 0; r.n_win = n_win; r._pad = 0;
        a.res[s] = r;
    }
}

/* ---- the sample work of the listed windows ---------------------------------------------------------------------------- */
struct ExecArgs { const Stretch *st; uint32_t n_st; sdv_sample_pair *w; const WinRec *wins; const uint32_t *win_base; StretchResult *res; uint32_t n_slots; uint8_t how; };
__device__ inline void exec_body(const ExecArgs &a, uint32_t slot, WinLds &lds, int lane)
{
    if (slot >= a.n_slots) return;
    uint32_t lo = 0, hi = a.n_st;         /* the last stretch whose list starts at or before the slot */
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.win_base[mid] <= slot) lo = mid; else hi = mid; }
    const uint32_t s = lo, n_win = a.res[s].n_win;
    uint32_t k = slot - a.win_base[s];
    if (k >= n_win || !a.wins[slot].head) return;
    unsigned long long masked = 0;          /* of this wave's windows: one addition to the stretch's count at the end (every window adding to the one address stood in line) */
    for (;;) {
        const WinRec rec = a.wins[a.win_base[s] + k];
        uint32_t mk = 0;
        const uint32_t pops = window_body(a.w + rec.w_pos, (int)rec.n, rec.file_end != 0, a.how, lds, mk, lane);
        masked += mk;
        if (lane == 0 && pops != rec.pops) atomicOr(&a.res[s].flags, (uint32_t)RES_PLAN_MISMATCH);
        k++;
        if (k >= n_win || a.wins[a.win_base[s] + k].head) break;
    }
    if (lane == 0 && masked) atomicAdd((unsigned long long *)&a.res[s].masked, masked);
}

/* ---- emit: W -> the caller's buffer, and what waits for the next call ------------------------------------------------- */
struct EmitArgs { const Stretch *st; uint32_t n_st; const StretchResult *res; const sdv_sample_pair *w; uint32_t total_w; sdv_sample_pair *out; uint64_t out_cap;
                  sdv_sample_pair *carry_out; uint8_t ignore; };
__device__ inline void emit_body(const EmitArgs &a, uint32_t blk, int lane)
{
    const uint32_t p_first = blk * 256u;
    const uint32_t s_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)find_stretch(a.st, a.n_st, p_first < a.total_w ? p_first : a.total_w - 1u));
    const Stretch t_first = a.st[s_first];
    const StretchResult r_first = a.res[s_first];
    const bool one = p_first + 256u <= t_first.w_base + t_first.v_len || s_first + 1u == a.n_st;
    sdv_sample_pair qs[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t p = (blk * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
        if (p < a.total_w) qs[r] = a.w[p];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t p = (blk * 4u + (uint32_t)r) * 64u + (uint32_t)lane;
        if (p >= a.total_w) continue;
        Stretch t = t_first; uint32_t popped = r_first.popped, scanned = r_first.scanned_upto;
        if (!one) { const uint32_t s = find_stretch(a.st, a.n_st, p); t = a.st[s]; popped = a.res[s].popped; scanned = a.res[s].scanned_upto; }
        const uint32_t k = p - t.w_base;
        const bool closed = t.end_kind != END_OPEN || t.stop;
        const uint32_t n_out = closed ? (t.v_len > 0 ? t.v_len - 1u : 0u) : popped;     /* a purge drops the last pair (:1423-1434) */
        sdv_sample_pair q = qs[r];
        if (a.ignore && k < scanned) { q.sample_flags[0] |= SDV_SF_WORD_VALID; q.sample_flags[1] |= SDV_SF_WORD_VALID; }   /* clearInvalids (:1404-1409) */
        if (k < n_out) { const uint64_t o = t.out_base + k; if (o < a.out_cap) store_pair_streaming(&a.out[o], q); }
        else if (!closed && k - n_out < (uint32_t)WIN) a.carry_out[k - n_out] = q;      /* (a stalled window holds WIN pairs; what lies behind it was never taken) */
    }
}

/* ---- SamplesToWAV::saveAudio (samples2wav.cpp:306-323): four bytes per pair ------------------------------------------- */
struct WavArgs { const sdv_sample_pair *pairs; size_t n; uint32_t *pcm; };
__device__ inline void wav_body(const WavArgs &a, size_t i)
{
    if (i < a.n) { const sdv_sample_pair q = a.pairs[i]; a.pcm[i] = (uint32_t)(uint16_t)q.audio_word[0] | ((uint32_t)(uint16_t)q.audio_word[1] << 16); }
}

} // namespace sdva

__global__ void __launch_bounds__(64) sdv_k_ap_tags(sdva::TagArgs a) { sdva::tags_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_ap_prepare(sdva::PrepArgs a) { sdva::prep_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_ap_summary(sdva::SummArgs a) { sdva::summ_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_ap_plan(sdva::PlanArgs a)
{
    __shared__ sdva::PlanLds lds;
    sdva::plan_body(a, blockIdx.x, lds, (int)threadIdx.x);
}
__global__ void __launch_bounds__(64) sdv_k_ap_windows(sdva::ExecArgs a)
{
    __shared__ sdva::WinLds lds;
    sdva::exec_body(a, blockIdx.x, lds, (int)threadIdx.x);
}
__global__ void __launch_bounds__(64) sdv_k_ap_emit(sdva::EmitArgs a) { sdva::emit_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_wav_pack(sdva::WavArgs a) { sdva::wav_body(a, (size_t)blockIdx.x * 64u + threadIdx.x); }
