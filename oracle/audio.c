/*
 * audio.c - CPU restatement of the reference's AudioProcessor (dropout masking on the PCMSamplePair stream) and of the bytes
 * SamplesToWAV writes.  TEST INFRASTRUCTURE ONLY: the checker for sdv_audio_process / sdv_wav_pack / sdv_wav_header; nothing
 * of the product links or calls this.  Pinned against the real class (oracle/_ref, ref_audio_run) by tests/test_audio_oracle.py.
 *
 * All citations are into /root/reference (v0.99.7).  The reference keeps the window in deques of 88-byte objects and splits
 * them per channel for every scan; here the window is three flat arrays and a scan is a walk over the runs of invalid samples,
 * which is what the backwards state machine of fixBadSamples (audioprocessor.cpp:740-1178) enumerates.
 */
#include <stdlib.h>
#include <string.h>
#include "audio.h"

enum { WIN = SDV_AP_BUF_SIZE, KEEP = SDV_AP_MIN_VALID_BEFORE, RAMP_DOWN = SDV_AP_MAX_RAMP_DOWN, RAMP_UP = SDV_AP_MAX_RAMP_UP,
       CALC_MULT = 16 };                /* audioprocessor.h:62-80 */

struct orc_audio {
    sdv_sample_pair win[WIN];           /* prebuffer */
    uint64_t index[WIN];                /* PCMSample::index of each entry */
    int n;
    uint64_t sample_index;              /* master index of the current file */
    int mask_mode;
    int stalled;                        /* the window is full and nothing can leave: the worker takes no more input (:108) */
    int hit_unsupported;                /* something the product refuses has happened (see sdv_audio_process) */
    /* sinks of the current call */
    sdv_sample_pair *out; uint64_t *out_index; size_t out_cap, n_out;
    sdv_audio_purge *purges; size_t purges_cap, n_purges;
    uint64_t n_masked;
};

orc_audio *orc_audio_new(int mask_mode)
{
    orc_audio *h = (orc_audio *)calloc(1, sizeof(*h));      /* the constructor, :3-36: empty window, index 0 */
    h->mask_mode = mask_mode;
    return h;
}
void orc_audio_free(orc_audio *h) { free(h); }
void orc_audio_set_masking(orc_audio *h, int mode) { if (mode >= 0 && mode < SDV_DROP_MAX) h->mask_mode = mode; }   /* :1532-1574 */
size_t orc_audio_pending(const orc_audio *h) { return (size_t)h->n; }
int orc_audio_hit_unsupported(const orc_audio *h) { return h->hit_unsupported; }

/* outputWordPair (:1265-1284) */
static void put_out(orc_audio *h, int i)
{
    if (h->n_out < h->out_cap) {
        h->out[h->n_out] = h->win[i];
        if (h->out_index) h->out_index[h->n_out] = h->index[i];
    }
    h->n_out++;
}

static void pop_front(orc_audio *h, int count)
{
    memmove(h->win, h->win + count, (size_t)(h->n - count) * sizeof(h->win[0]));
    memmove(h->index, h->index + count, (size_t)(h->n - count) * sizeof(h->index[0]));
    h->n -= count;
}

/* purgePipeline (:1716-1745) with dumpBuffer (:1423-1434): everything but the last entry leaves as it is, then the window
 * holds one silent valid pair and the index starts over at 1 */
static void purge(orc_audio *h, int kind, uint32_t tag_index)
{
    for (int i = 0; i + 1 < h->n; i++) put_out(h, i);
    if (h->n_purges < h->purges_cap) {
        sdv_audio_purge *p = &h->purges[h->n_purges];
        memset(p, 0, sizeof(*p));
        p->first_pair = h->n_out; p->tag_index = tag_index; p->kind = (uint8_t)kind;
    }
    h->n_purges++;
    memset(&h->win[0], 0, sizeof(h->win[0]));
    h->win[0].sample_flags[0] = h->win[0].sample_flags[1] = SDV_SF_BLOCK_OK | SDV_SF_WORD_VALID;    /* setSamplePair(0, 0, true x4, false x2), :1742 */
    h->win[0].sample_rate = 44056;      /* PCMSamplePair::clear, pcmsamplepair.cpp:227-237 */
    h->index[0] = 0;
    h->n = 1;
    h->sample_index = 1;
}

/* one channel of the window */
typedef struct { int16_t val[WIN]; uint8_t flg[WIN]; int n; } chan;
static int c_valid(const chan *c, int i) { return (c->flg[i] & SDV_SF_WORD_VALID) != 0; }
static void sample_mute(chan *c, int i)     /* sampleMute (:495-508) */
{
    if (i < 0 || i > c->n - 1) return;
    c->flg[i] |= SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED;
    c->val[i] = 0;
}

/* rangeMute / rangeLevelHold / rangeLinearInterpolation (:511-737) on the samples strictly inside (a, b); how = 0 mute, 1 hold,
 * 2 linear.  Returns how many values changed. */
static unsigned fill_range(chan *c, int a, int b, int how)
{
    unsigned changed = 0;
    if (!(a < b) || a < 0 || b >= c->n) return 0;           /* CoordinatePair::areValid + the bounds checks */
    const int16_t va = c->val[a], vb = c->val[b];
    int32_t step = 0, base = 0;
    const int same = (va == vb);
    if (how == 2 && !same) {
        base = (int32_t)va * CALC_MULT;
        const int32_t delta = (int32_t)vb * CALC_MULT - base;
        const int32_t cnt = b - a;                          /* :671-673 */
        step = (delta + cnt / 2) / cnt;
    }
    for (int i = a + 1; i < b; i++) {
        int16_t v;
        if (how == 0) v = 0;
        else if (how == 1 || same) v = va;
        else v = (int16_t)((step * (int32_t)(i - a) + base + CALC_MULT / 2) / CALC_MULT);       /* :700-704 */
        if (c->val[i] != v) { c->val[i] = v; c->flg[i] |= SDV_SF_WORD_MASKED; changed++; }
        c->flg[i] |= SDV_SF_WORD_VALID;
    }
    return changed;
}

/* fixBadSamples (:740-1178) */
static unsigned fix_channel(chan *c, int mask_mode, int file_end)
{
    const int n = c->n;
    int reg_a[WIN], reg_b[WIN], n_reg = 0;
    if (n == 0) return 0;
    /* runs of invalid samples, last one first; `above` = the lowest valid sample above the run (good_at_the_end), -1 when the
     * run reaches the end of the window */
    int i = n - 1, above = -1;
    while (i >= 0) {
        if (c_valid(c, i)) { above = i; i--; continue; }
        while (i >= 0 && !c_valid(c, i)) i--;
        const int below = i;                                /* good_end */
        if (below < 0) break;                               /* a run that starts at entry 0 stays as it is (:1079-1082) */
        if (above < 0) {
            /* nothing valid behind the run (:848-897): wait while both ramps and a sample of silence do not fit yet,
             * otherwise ramp down from the last valid sample into a forced zero */
            if (below < (int)(int16_t)n - (int)(int16_t)(RAMP_DOWN + RAMP_UP + 1)) {
                sample_mute(c, below + RAMP_DOWN + 1);
                reg_a[n_reg] = below; reg_b[n_reg] = below + RAMP_DOWN + 1; n_reg++;
            }
        } else {
            const int leftover = above - below - 1;         /* :907 */
            const int start_altered = (c->flg[below] & SDV_SF_WORD_MASKED) != 0 && c->val[below] == 0;      /* :919-920 */
            if (!start_altered) {
                if (leftover > RAMP_DOWN + RAMP_UP) {
                    /* ramp down, silence, ramp up (:930-994); the list is filled back to front */
                    const int up = above - RAMP_UP - 1, down = below + RAMP_DOWN + 1;
                    reg_a[n_reg] = up; reg_b[n_reg] = above; n_reg++;
                    sample_mute(c, up);
                    if (up > down) { reg_a[n_reg] = down; reg_b[n_reg] = up; n_reg++; sample_mute(c, down); }
                    reg_a[n_reg] = below; reg_b[n_reg] = down; n_reg++;
                } else { reg_a[n_reg] = below; reg_b[n_reg] = above; n_reg++; }
            } else {
                /* the sample in front is a zero an earlier scan put there: no ramp down (:996-1050) */
                if (leftover > RAMP_UP) {
                    const int up = above - RAMP_UP - 1;
                    reg_a[n_reg] = up; reg_b[n_reg] = above; n_reg++;
                    sample_mute(c, up);
                    reg_a[n_reg] = below; reg_b[n_reg] = up; n_reg++;
                } else { reg_a[n_reg] = below; reg_b[n_reg] = above; n_reg++; }
            }
        }
        above = below;
    }
    unsigned masks = 0;
    const int how = (mask_mode == SDV_DROP_MUTE_BLOCK || mask_mode == SDV_DROP_MUTE_WORD) ? 0 :
                    (mask_mode == SDV_DROP_HOLD_BLOCK || mask_mode == SDV_DROP_HOLD_WORD) ? 1 : 2;
    for (int r = n_reg - 1; r >= 0; r--) masks += fill_range(c, reg_a[r], reg_b[r], how);     /* :1097-1117, front to back */
    if (file_end) {
        /* what is still invalid at the very end of a file ramps into a forced zero, always linearly (:1122-1172) */
        int q = n - 1, a = 0, bad = 0;
        while (q > 0) {
            if (!c_valid(c, q)) bad = 1;
            else { if (bad) a = q; break; }
            q--;
        }
        if (bad) { sample_mute(c, n - 1); masks += fill_range(c, a, n - 1, 2); }
    }
    return masks;
}

/* scanBuffer (:1360-1420) + fillBufferForOutput (:1200-1232) */
static int scan_window(orc_audio *h, int file_end)
{
    const int n = h->n;
    if (file_end) { if (n == 0) return 0; }
    else if (n < KEEP + RAMP_DOWN + RAMP_UP) return 0;      /* size_t arithmetic at :1367: the difference wraps below 227 */
    for (int ch = 0; ch < 2; ch++) {
        chan c;
        c.n = n;
        for (int i = 0; i < n; i++) { c.val[i] = h->win[i].audio_word[ch]; c.flg[i] = h->win[i].sample_flags[ch]; }
        if (h->mask_mode != SDV_DROP_IGNORE) {
            const unsigned m = fix_channel(&c, h->mask_mode, file_end);     /* remove_stray is never set: fixStraySamples is dead code */
            h->n_masked += m;                                               /* guiAddMask (:1174-1177) */
        } else for (int i = 0; i < n; i++) c.flg[i] |= SDV_SF_WORD_VALID;   /* clearInvalids over the whole window (:1404-1409) */
        for (int i = 0; i < n; i++) { h->win[i].audio_word[ch] = c.val[i]; h->win[i].sample_flags[ch] = c.flg[i]; }
    }
    return 1;
}

static int pair_ready(const sdv_sample_pair *p)     /* PCMSamplePair::isReadyForOutput, pcmsamplepair.cpp:424-435 */
{
    return (p->sample_flags[0] & (SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED)) != 0 && (p->sample_flags[1] & (SDV_SF_WORD_VALID | SDV_SF_WORD_MASKED)) != 0;
}

/* outputAudio (:1287-1357) */
static void output_window(orc_audio *h, int file_end, uint32_t tag_index)
{
    const int lim = file_end ? KEEP : KEEP + RAMP_DOWN + RAMP_UP + 1;
    if (h->n < lim) { if (file_end) h->hit_unsupported = 1; return; }      /* at the end of a file: no purge, no new source */
    int pops = 0;
    while (h->n - pops > KEEP) {
        int ok = 1;
        for (int k = 0; k <= KEEP; k++) ok = ok && pair_ready(&h->win[pops + k]);
        if (!ok) break;
        put_out(h, pops);
        pops++;
    }
    if (pops == 0 && h->n == WIN && !file_end) { h->stalled = 1; h->hit_unsupported = 1; }
    pop_front(h, pops);
    if (file_end) purge(h, SDV_AP_PURGE_END_FILE, tag_index);
}

/* The worker's loop (processAudio :1650-1709) over one burst: turns until the queue is dry.  Returns the number of pairs put
 * out by this call, or -1 when out_cap / purges_cap were too small. */
long orc_audio_process(orc_audio *h, const sdv_sample_pair *pairs, size_t n_pairs, int stop, sdv_sample_pair *out, uint64_t *out_index, size_t out_cap,
                       sdv_audio_purge *purges, size_t purges_cap, size_t *n_purges, uint64_t *n_masked)
{
    h->out = out; h->out_index = out_index; h->out_cap = out_cap; h->n_out = 0;
    h->purges = purges; h->purges_cap = purges_cap; h->n_purges = 0; h->n_masked = 0;
    size_t pos = 0;
    const int by_block = h->mask_mode == SDV_DROP_MUTE_BLOCK || h->mask_mode == SDV_DROP_HOLD_BLOCK || h->mask_mode == SDV_DROP_INTER_LIN_BLOCK;
    while (!h->stalled) {
        /* fillUntilBufferFull (:70-200) */
        size_t added = 0; int file_end = 0; uint32_t tag = 0;
        while (pos < n_pairs && h->n < WIN) {
            const sdv_sample_pair *p = &pairs[pos++];
            added++;
            if (p->service_type == SDV_PAIR_SRV_NEW_FILE) purge(h, SDV_AP_PURGE_NEW_FILE, (uint32_t)(pos - 1));
            else if (p->service_type == SDV_PAIR_SRV_END_FILE) { file_end = 1; tag = (uint32_t)(pos - 1); break; }
            else if (p->service_type == SDV_PAIR_SRV_NO) {
                sdv_sample_pair q = *p;
                q._pad = 0;
                if (by_block) for (int ch = 0; ch < 2; ch++)        /* setValidityByBlock (:166-169) */
                    q.sample_flags[ch] = (uint8_t)((q.sample_flags[ch] & ~SDV_SF_WORD_VALID) | ((q.sample_flags[ch] & SDV_SF_BLOCK_OK) ? SDV_SF_WORD_VALID : 0));
                h->win[h->n] = q; h->index[h->n] = h->sample_index++; h->n++;
            }
            /* a pair with any other tag is a service pair that is neither: taken and dropped (:117-153) */
        }
        if (added == 0) { if (pos < n_pairs) { h->stalled = 1; h->hit_unsupported = 1; } break; }
        if (scan_window(h, file_end)) output_window(h, file_end, tag);
    }
    if (stop) purge(h, SDV_AP_PURGE_STOP, (uint32_t)n_pairs);       /* stop() -> purgePipeline (:1655-1660) */
    if (n_purges) *n_purges = h->n_purges;
    if (n_masked) *n_masked = h->n_masked;
    return (h->n_out > out_cap || h->n_purges > purges_cap) ? -1 : (long)h->n_out;
}

/* Same shape as ref_audio_run of oracle/ref_audio_driver.cpp: a fresh worker, the bursts one after the other, stop() at the end
 * (what stop() puts out is reported only with `stop`). */
long orc_audio_run(const sdv_sample_pair *pairs, size_t n, const uint64_t *bursts, size_t n_bursts, int mask_mode, int stop,
                   sdv_sample_pair *out, uint64_t *out_index, size_t out_cap, sdv_audio_purge *purges, size_t purges_cap, size_t *n_purges,
                   uint64_t *n_masked, int *hit_unsupported)
{
    orc_audio *h = orc_audio_new(mask_mode);
    size_t from = 0, got = 0, got_p = 0; uint64_t masked = 0; int over = 0;
    for (size_t b = 0; b < n_bursts; b++) {
        const size_t to = bursts[b] < n ? (size_t)bursts[b] : n;
        size_t np = 0; uint64_t nm = 0;
        const int last = (b + 1 == n_bursts);
        const long r = orc_audio_process(h, pairs + from, to - from, last && stop, out + (got < out_cap ? got : out_cap), out_index ? out_index + (got < out_cap ? got : out_cap) : NULL,
                                         got < out_cap ? out_cap - got : 0, purges + (got_p < purges_cap ? got_p : purges_cap), got_p < purges_cap ? purges_cap - got_p : 0, &np, &nm);
        if (r < 0) { over = 1; break; }
        for (size_t k = 0; k < np && got_p + k < purges_cap; k++) { purges[got_p + k].first_pair += got; purges[got_p + k].tag_index += (uint32_t)from; }
        got += (size_t)r; got_p += np; masked += nm; from = to;
    }
    if (n_purges) *n_purges = got_p;
    if (n_masked) *n_masked = masked;
    if (hit_unsupported) *hit_unsupported = h->hit_unsupported;
    orc_audio_free(h);
    return over ? -1 : (long)got;
}

/* SamplesToWAV: the header as updateHeader leaves it (samples2wav.cpp:4-21, :111-206) and saveAudio's four bytes per pair (:306-323) */
void orc_wav_header(uint8_t hdr[44], uint64_t n_pairs, uint16_t last_sample_rate)
{
    static const uint8_t def[44] = { 'R', 'I', 'F', 'F', 0, 0, 0, 0, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 0x10, 0, 0, 0, 0x01, 0, 0x02, 0,
                                     0x44, 0xAC, 0, 0, 0x10, 0xB1, 0x02, 0, 0x04, 0, 0x10, 0, 'd', 'a', 't', 'a', 0, 0, 0, 0 };
    memcpy(hdr, def, 44);
    const uint32_t rate = last_sample_rate == 44056 ? 44056u : 44100u;         /* setSampleRate :257-289 */
    const uint32_t file_size = (uint32_t)(44u + 4u * n_pairs), riff = file_size - 8u, wave = file_size - 44u, brate = rate * 2u * 2u;
    for (int k = 0; k < 4; k++) {
        hdr[4 + k] = (uint8_t)(riff >> (8 * k)); hdr[24 + k] = (uint8_t)(rate >> (8 * k));
        hdr[28 + k] = (uint8_t)(brate >> (8 * k)); hdr[40 + k] = (uint8_t)(wave >> (8 * k));
    }
}
void orc_wav_pack(const sdv_sample_pair *pairs, size_t n, int16_t *pcm)
{
    for (size_t i = 0; i < n; i++) { pcm[2 * i] = pairs[i].audio_word[0]; pcm[2 * i + 1] = pairs[i].audio_word[1]; }
}
