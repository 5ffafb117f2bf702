/*
 * render.c - CPU restatement of RenderPCM's canvases of binarized lines (renderpcm.cpp).  TEST INFRASTRUCTURE ONLY: used by tests/,
 * __graft_entry__.smoke() and nothing else; pinned against the real RenderPCM (oracle/ref_render_driver.cpp) and the fixtures it made.
 *
 * The "binarized" visualiser of the reference (mainwindow.cpp:1949-1990) is one RenderPCM object that is handed every line VideoToDigital
 * queues, except service lines other than fillers (videotodigital.cpp:398-402, 452-456, 507-511), and a prepareNewFrame() per binarized frame
 * (:176-186): the canvas (a QImage::Format_RGB32) is copied out and the fill row goes back to 0 - the canvas itself is NOT cleared, rows a
 * frame does not reach keep what earlier frames drew.
 *
 *   STC-007   renderNewLine(STC007Line)     :939-1169    5 px per bit, 137 bits (START 1010, 128 data bits, STOP 01111), 685 x 650
 *   PCM-1     renderNewLine(PCM1Line)       :489-624     8 px per bit, 94 bits, 752 x 490
 *   PCM-16x0  renderNewLine(PCM16X0SubLine) :743-936     4 px per bit, three sub-lines of 64 bits and the control bit, 772 x 490
 *
 * VIS_BIT0_BLK is Qt::black - the GlobalColor enumerator (2), not a QRgb: the reference stores the value 2 in those pixels.
 */
#include <string.h>
#include "render.h"

enum { PX_BLK = 2u,                                           /* (QRgb)Qt::black */
       B0_GRY = 0xFF2D2D2Du, B1_GRY = 0xFF969696u, B0_YEL = 0xFF7F6E00u, B1_YEL = 0xFFFFDC00u, B0_GRN = 0xFF005F1Eu, B1_GRN = 0xFF00E146u,
       B0_RED = 0xFF8C0000u, B1_RED = 0xFFFF462Bu, B0_BLU = 0xFF005F7Fu, B1_BLU = 0xFF00BFFFu, B0_MGN = 0xFF8C008Cu, B1_MGN = 0xFFFF00FFu,
       B1_MARK = 0xFFFFFFFFu, LIM_OK = 0xFFFFFFFFu, LIM_MARK = 0xFFE0AAAAu };

void orc_vis_canvas_size(int kind, uint32_t *w, uint32_t *h)
{
    switch (kind) {
        case ORC_VIS_STC007: *w = 5 * 137; *h = 650; break;      /* startSTC007NTSCFrame + setLineCount(VID_UNKNOWN), mainwindow.cpp:1985-1986 */
        case ORC_VIS_PCM1: *w = 8 * 94; *h = 490; break;         /* startPCM1Frame :128-131 */
        case ORC_VIS_PCM16X0: *w = 4 * 193; *h = 490; break;     /* startPCM1600Frame :140-143 */
        case ORC_VIS_STC007_BLOCKS_NTSC: *w = 6 * (6 + 16 * 6 + 7); *h = 490; break;     /* startSTC007DBFrame :170-173 (+ setLineCount(VID_NTSC)) */
        case ORC_VIS_STC007_BLOCKS_PAL: *w = 6 * (6 + 16 * 6 + 7); *h = 588; break;      /* ... + setLineCount(VID_PAL), mainwindow.cpp:2093 */
        case ORC_VIS_STC007_ASM_NTSC: *w = 5 * 137; *h = 490; break;                     /* startSTC007NTSCFrame (+ setLineCount(VID_NTSC)), mainwindow.cpp:2032, 2046 */
        case ORC_VIS_STC007_ASM_PAL: *w = 5 * 137; *h = 588; break;
        case ORC_VIS_PCM1_BLOCKS: *w = 6 * (8 + 3 + 8 * 16 + 4); *h = (184 / 8) * 8 * 2; break;     /* startPCM1DBFrame :158-161 */
        case ORC_VIS_PCM1_ASM: *w = 8 * (94 - 16); *h = 490; break;                              /* startPCM1SubFrame :134-137 */
        case ORC_VIS_PCM16X0_BLOCKS: *w = 6 * (9 + 16 * 6 + 8); *h = 490; break;                 /* startPCM1600DBFrame :164-167 */
        default: *w = *h = 0;
    }
}

static bool drawn(uint8_t service_type) { return service_type == SDV_SRV_NO || service_type == SDV_SRV_FILLER; }

/* :939-1169 */
static void stc_line(const sdv_line_rec *r, uint32_t *px)
{
    uint16_t words[9];
    bool crc_valid, markers, forced, word_crc, word_valid;
    if (r->service_type == SDV_SRV_FILLER) {     /* a cleared line (STC007Line::clear): silent words, the read CRC is the inverse of CRC_SILENT */
        memset(words, 0, sizeof(words)); words[8] = (uint16_t)~0xA96A;
        crc_valid = markers = forced = word_crc = word_valid = false;
    } else {
        memcpy(words, r->words, sizeof(words));
        crc_valid = (r->flags & SDV_LF_CRC_VALID) != 0; forced = (r->flags & SDV_LF_FORCED_BAD) != 0;
        markers = r->mark_st_stage == 4 /* MARK_ST_BOT_2, stc007line.h:130 */ && r->mark_ed_stage == 3 /* MARK_ED_LEN_OK, :139 */;
        word_crc = (r->word_state & SDV_WS_WORD_CRC) != 0; word_valid = (r->word_state & SDV_WS_WORD_VALID) != 0;
    }
    for (int i = 0; i < 4; i++) {
        const bool one = (i % 2) == 0;
        const uint32_t c = (crc_valid || markers) ? (one ? B1_GRY : B0_GRY) : PX_BLK;
        for (int j = 0; j < 5; j++) *px++ = c;
    }
    for (int w = 0; w < 9; w++) {
        const int bits = w == 8 ? 16 : 14;
        for (int b = bits - 1; b >= 0; b--) {
            const bool one = ((words[w] >> b) & 1) != 0;
            uint32_t c;
            if (forced) c = one ? B1_MGN : B0_MGN;
            else if (word_crc) c = one ? B1_GRY : B0_GRY;
            else if (word_valid) c = one ? B1_GRN : B0_GRN;
            else if (markers) c = one ? B1_YEL : B0_YEL;
            else c = one ? B1_RED : B0_RED;
            for (int j = 0; j < 5; j++) *px++ = c;
        }
    }
    for (int i = 0; i < 5; i++) {
        uint32_t c = PX_BLK;
        if (i == 0) { if (crc_valid || markers) c = B0_GRY; }
        else if (crc_valid) c = B1_MARK;
        else if (markers) c = B1_GRY;
        for (int j = 0; j < 5; j++) *px++ = c;
    }
}

/* the colour of one data bit of a PCM-1 line / PCM-16x0 sub-line (:536-607, :803-876) */
static uint32_t bit_colour(bool one, bool crc_valid, bool forced, bool bw, bool picked)
{
    if (crc_valid) return picked ? (one ? B1_BLU : B0_BLU) : (one ? B1_GRY : PX_BLK);
    if (forced) return one ? B1_MGN : B0_MGN;
    if (bw) return one ? B1_YEL : B0_YEL;
    return one ? B1_RED : B0_RED;
}

/* :489-624 */
static void pcm1_line(const sdv_pcm1_bin_rec *r, uint32_t *px)
{
    uint16_t words[7];
    bool crc_valid = false, forced = false, bw = false;
    int pl = 0, pr = 0;
    if (r->service_type == SDV_SRV_FILLER) { for (int i = 0; i < 6; i++) words[i] = 1 << 12; words[6] = (uint16_t)~0xECBF; }
    else {
        memcpy(words, r->words, sizeof(words));
        crc_valid = (r->flags & SDV_LF_CRC_VALID) != 0; forced = (r->flags & SDV_LF_FORCED_BAD) != 0; bw = (r->flags & SDV_LF_BW_SET) != 0;
        pl = r->picked_bits_left; pr = r->picked_bits_right;
    }
    int line_bit = 0;
    for (int w = 0; w < 7; w++)
        for (int b = (w == 6 ? 16 : 13) - 1; b >= 0; b--, line_bit++) {
            const bool picked = line_bit < pl || line_bit > 94 - pr - 1;
            const uint32_t c = bit_colour(((words[w] >> b) & 1) != 0, crc_valid, forced, bw, picked);
            for (int j = 0; j < 8; j++) *px++ = c;
        }
}

/* :743-936; returns true when the row is finished (PART_RIGHT) */
static bool pcm16_subline(const sdv_pcm16x0_bin_rec *r, uint32_t *row)
{
    uint16_t words[4];
    bool crc_valid = false, forced = false, bw = false, coords = false, control = true;
    int pl = 0, pr = 0, part = 0;
    if (r->service_type == SDV_SRV_FILLER) { words[0] = words[1] = words[2] = 0; words[3] = (uint16_t)~0x0E10; }   /* PCM16X0SubLine::clear: PART_LEFT, control bit set */
    else {
        memcpy(words, r->words, sizeof(words));
        crc_valid = (r->flags & SDV_LF_CRC_VALID) != 0; forced = (r->flags & SDV_LF_FORCED_BAD) != 0; bw = (r->flags & SDV_LF_BW_SET) != 0;
        coords = (r->flags & SDV_LF_COORDS_SET) != 0; control = r->control_bit != 0;
        pl = r->picked_bits_left; pr = r->picked_bits_right; part = r->line_part;
    }
    int ofs = part > 2 ? 0 : part * 64;
    if (part == 2) ofs++;
    uint32_t *px = row + 4 * ofs;
    int line_bit = 0;
    for (int w = 0; w < 4; w++)
        for (int b = 15; b >= 0; b--, line_bit++) {
            const bool picked = line_bit < pl || line_bit > 64 - pr - 1;
            const uint32_t c = bit_colour(((words[w] >> b) & 1) != 0, crc_valid, forced, bw, picked);
            for (int j = 0; j < 4; j++) *px++ = c;
        }
    if (part == 1) {
        const uint32_t c = (!coords || !crc_valid) ? (control ? B1_RED : B0_RED) : (control ? B1_GRY : PX_BLK);
        for (int j = 0; j < 4; j++) *px++ = c;
    }
    return part == 2;
}

/* renderNewLine(PCM1SubLine) :626-741; returns true when the row is finished (PART_RIGHT) */
static bool pcm1_subline(const sdv_pcm1_asm_line_rec *r, uint32_t *row)
{
    const bool crc_valid = (r->flags & SDV_P1S_CRC_VALID) != 0, bw = (r->flags & SDV_P1S_BW_SET) != 0;
    int part = r->line_part; if (part > 2) part = 0;
    uint32_t *px = row + (size_t)part * 26 * 8;
    int line_bit = 0;
    for (int w = 0; w < 2; w++)
        for (int b = 12; b >= 0; b--, line_bit++) {
            const bool picked = line_bit < r->picked_bits_left && r->line_part == 0;
            const uint32_t c = bit_colour(((r->words[w] >> b) & 1) != 0, crc_valid, false, bw, picked);
            for (int j = 0; j < 8; j++) *px++ = c;
        }
    return r->line_part == 2;
}

long orc_vis_render_lines(int kind, const void *recs, size_t n_recs, uint32_t *canvas, uint32_t *out, size_t out_cap)
{
    uint32_t w, h, fill = 0;
    long frames = 0;
    orc_vis_canvas_size(kind, &w, &h);
    if (!w) return -1;
    if (kind == ORC_VIS_PCM1_ASM) {         /* the sub-lines the PCM-1 stitcher hands over, 1470 places per frame (newFrameAssembled -> prepareNewFrame behind them) */
        const sdv_pcm1_asm_line_rec *l = (const sdv_pcm1_asm_line_rec *)recs;
        for (size_t i = 0; i < n_recs; i++) {
            if (!(l[i].flags & SDV_P1S_SKIP) && fill < h && pcm1_subline(&l[i], canvas + (size_t)fill * w)) fill++;
            if ((i + 1) % 1470 == 0) {
                if ((size_t)frames < out_cap) memcpy(out + (size_t)frames * w * h, canvas, (size_t)w * h * 4);
                frames++; fill = 0;
            }
        }
        return frames;
    }
    for (size_t i = 0; i < n_recs; i++) {
        uint8_t srv;
        if (kind == ORC_VIS_STC007) srv = ((const sdv_line_rec *)recs)[i].service_type;
        else if (kind == ORC_VIS_PCM1) srv = ((const sdv_pcm1_bin_rec *)recs)[i].service_type;
        else srv = ((const sdv_pcm16x0_bin_rec *)recs)[i].service_type;
        if (srv == SDV_SRV_END_FRAME) {          /* newFrameBinarized -> prepareNewFrame (:176-186): the canvas goes out, the fill row back to 0 */
            if ((size_t)frames < out_cap) memcpy(out + (size_t)frames * w * h, canvas, (size_t)w * h * 4);
            frames++; fill = 0;
            continue;
        }
        if (!drawn(srv) || fill >= h) continue;
        uint32_t *row = canvas + (size_t)fill * w;
        if (kind == ORC_VIS_STC007) { stc_line(&((const sdv_line_rec *)recs)[i], row); fill++; }
        else if (kind == ORC_VIS_PCM1) { pcm1_line(&((const sdv_pcm1_bin_rec *)recs)[i], row); fill++; }
        else if (pcm16_subline(&((const sdv_pcm16x0_bin_rec *)recs)[i], row)) fill++;
    }
    return frames;
}

/* ---- the data blocks window (renderNewBlock(STC007DataBlock), renderpcm.cpp:1770-2051) ------------------------------------------------ */
static int16_t blk_sample(const sdv_block_rec *b, int w, bool m2)   /* STC007DataBlock::getSample, stc007datablock.cpp:507-562 */
{
    uint16_t v = b->words[w];
    if (!m2) return b->resolution == SDV_RES_16BIT ? (int16_t)v : (int16_t)(v << 2);
    if ((v & (1 << 13)) == 0) return (int16_t)(uint16_t)(v << 3);          /* higher range: value x 8 */
    {
        const bool positive = (v & (1 << 12)) == 0;
        v = (uint16_t)(v & ~(1 << 13));
        if (!positive) v |= (1 << 15) | (1 << 14) | (1 << 13);
        return (int16_t)v;
    }
}
static bool blk_near_silence(const sdv_block_rec *b, int w, bool m2)     /* isNearSilence :417-446 */
{
    const int16_t v = blk_sample(b, w, m2), lim = (b->resolution == SDV_RES_16BIT || m2) ? 4 : 16;
    return v < lim && v >= -lim;
}
static void stc_block(const sdv_block_rec *b, uint32_t *px, bool m2)
{
    const bool fix_p = b->audio_state == SDV_AUD_FIX_P, fix_q = b->audio_state == SDV_AUD_FIX_Q, broken = b->audio_state == SDV_AUD_BROKEN;
    const bool valid = (b->word_valid & 0x3F) == 0x3F;                                 /* isBlockValid: no audio word left invalid */
    const bool cwd_audio = (b->cwd_fixed & 0x3F) != 0;                                  /* isAudioAlteredByCWD */
    const bool almost_silent = (blk_near_silence(b, 0, m2) || blk_near_silence(b, 2, m2) || blk_near_silence(b, 4, m2)) &&
                               (blk_near_silence(b, 1, m2) || blk_near_silence(b, 3, m2) || blk_near_silence(b, 5, m2));
    const bool on_seam = b->w_line[0] > b->w_line[7];                                   /* getStartLine() > getStopLine() */
    for (int i = 0; i < 6; i++) {                                                        /* the status bar :1793-1861 */
        uint32_t c = PX_BLK;
        if (i == 0) { if (fix_p) c = B1_GRN; }
        else if (i == 1) { if (fix_q) c = B1_YEL; }
        else if (i == 2) { if (cwd_audio) c = valid ? B1_BLU : B0_BLU; }
        else if (i == 3) { if (!valid) c = B1_RED; }
        else if (i == 5) c = almost_silent ? LIM_MARK : LIM_OK;
        for (int j = 0; j < 6; j++) *px++ = c;
    }
    for (int w = 0; w < 6; w++) {                                                        /* the six samples, 16 bits each :1862-1993 */
        const uint16_t v = (uint16_t)blk_sample(b, w, m2);
        const bool crc = (b->line_crc >> w) & 1, cwd = (b->cwd_fixed >> w) & 1, wv = (b->word_valid >> w) & 1;
        for (int bit = 15; bit >= 0; bit--) {
            const bool one = (v >> bit) & 1;
            uint32_t c = one ? B1_GRY : PX_BLK;
            if (broken) { if (!crc) c = one ? B1_MGN : B0_MGN; }
            else if (fix_q) { if (cwd) c = one ? B1_BLU : B0_BLU; else if (!crc) c = one ? B1_YEL : B0_YEL; }
            else if (fix_p) { if (cwd) c = one ? B1_BLU : B0_BLU; else if (!crc) c = one ? B1_GRN : B0_GRN; }
            else if (cwd) c = one ? B1_BLU : B0_BLU;
            else if (!wv) c = one ? B1_RED : B0_RED;
            for (int j = 0; j < 6; j++) *px++ = c;
        }
    }
    for (int i = 0; i < 7; i++) {                                                        /* seam, emphasis (never set for STC-007, stc007datastitcher.cpp:6719), BROKEN :1995-2045 */
        uint32_t c = PX_BLK;
        if (i == 0) c = on_seam ? LIM_MARK : LIM_OK;
        else if (i == 4 || i == 5) { if (broken) c = B1_MGN; }
        for (int j = 0; j < 6; j++) *px++ = c;
    }
}

/* ---- renderNewBlock(PCM1DataBlock) :1171-1400: 23 rows of eight words; returns the rows drawn ------------------------------------------- */
static int16_t p1_sample(uint16_t w)        /* PCM1DataBlock::getSample, pcm1datablock.cpp:309-348 */
{
    if ((w & (1 << 12)) == 0) return (int16_t)(uint16_t)(w << 4);
    {
        const bool positive = (w & (1 << 11)) == 0;
        w = (uint16_t)(w & ~(1 << 12));
        w = (uint16_t)(w << 2);
        if (!positive) w |= (1 << 15) | (1 << 14);
        return (int16_t)w;
    }
}
static uint32_t pcm1_block(const sdv_pcm1_block_rec *b, uint32_t *canvas, uint32_t w_px, uint32_t fill, uint32_t h)
{
    const bool is_short = (b->flags & SDV_P1B_SHORT) != 0;
    const int count = is_short ? 182 : 184;
    bool valid = true, silent = true;
    uint32_t drawn_rows = 0;
    if (fill >= h) return 0;                                                            /* :1183-1189 */
    for (int i = 0; i < count; i++) {
        const int16_t v = p1_sample(b->words[i]);
        if (!(b->word_flags[i] & SDV_P1W_CRC_OK)) valid = false;                        /* isBlockValid: getErrorsAudio() == 0 */
        if (v >= 16 || v < -16) silent = false;                                         /* isAlmostSilent :229-244 */
    }
    for (int line = 0; line < 184 / 8; line++) {
        uint32_t *px = canvas + (size_t)fill * w_px;
        const int w0 = line * 8;
        for (int i = 0; i < 8 + 3; i++) {                                               /* the status bar :1213-1283 */
            uint32_t c = PX_BLK;
            if (i < 8) {
                const bool there = !is_short || (w0 + i) < 182;
                if (i % 2 == 0) { if (there) { if (b->word_flags[w0 + i] & SDV_P1W_PICKED_LEFT) c = B1_BLU; else if (b->word_flags[w0 + i] & SDV_P1W_PICKED_WORD) c = B0_BLU; } }
                else if (there && !(b->word_flags[w0 + i] & SDV_P1W_CRC_OK)) c = B1_YEL;
            } else if (i == 8) { if (!valid) c = B1_RED; }
            else if (i == 10) c = silent ? LIM_MARK : LIM_OK;
            for (int j = 0; j < 6; j++) *px++ = c;
        }
        for (int w = w0; w < w0 + 8; w++) {                                             /* the samples :1286-1352 */
            const uint16_t v = (uint16_t)p1_sample(b->words[w]);
            for (int bit = 15; bit >= 0; bit--) {
                const bool one = (v >> bit) & 1;
                uint32_t c;
                if (is_short && w >= 182) c = PX_BLK;
                else if (!(b->word_flags[w] & SDV_P1W_CRC_OK)) c = one ? B1_RED : B0_RED;
                else if (b->word_flags[w] & SDV_P1W_PICKED_LEFT) c = one ? B1_BLU : B0_BLU;
                else c = one ? B1_GRY : PX_BLK;
                for (int j = 0; j < 6; j++) *px++ = c;
            }
        }
        for (int i = 0; i < 4; i++) {                                                   /* parity of the block in its field, emphasis :1355-1387 */
            uint32_t c = PX_BLK;
            if (i == 0) c = (b->interleave_num % 2 == 0) ? LIM_OK : LIM_MARK;
            else if (i == 2) { if (b->flags & SDV_P1B_EMPHASIS) c = B0_GRN; }
            for (int j = 0; j < 6; j++) *px++ = c;
        }
        drawn_rows++;
        if (fill < h) fill++;
        if (fill >= h) break;               /* (scanLine past the image: the reference draws nowhere it can be seen; a block never straddles the end - 23 divides 368) */
    }
    return drawn_rows;
}

/* ---- renderNewBlock(PCM16X0DataBlock) :1403-1768: one row per block; the predicates are PCM16X0DataBlock's (pcm16x0datablock.cpp), asked of the record --- */
static int p16_line_of(const sdv_pcm16x0_block_rec *b, int blk, int word)      /* getWordToLine :1029-1155; word 0 = L, 1 = R */
{
    const bool l_first = ((blk & 1) != 0) != ((b->flags & SDV_P16B_EVEN_ORDER) != 0);
    return ((word == 0) == l_first) ? 0 : 2;
}
static bool p16_crc(const sdv_pcm16x0_block_rec *b, int blk, int line) { return (b->word_crc >> (3 * blk + line)) & 1; }
static bool p16_val(const sdv_pcm16x0_block_rec *b, int blk, int line) { return (b->word_valid >> (3 * blk + line)) & 1; }
static void pcm16_block(const sdv_pcm16x0_block_rec *b, uint32_t *px)
{
    bool valid = true, broken_any = false, silent = false;
    for (int i = 0; i < 3; i++) {
        if (!p16_val(b, i, 0) || !p16_val(b, i, 2)) valid = false;                     /* isBlockValid(): getErrorsFixedAudio() == 0, :510-517, :724-757 */
        if (b->audio_state[i] == 2) broken_any = true;                                  /* isDataBroken() :571-590 */
        {   /* isAlmostSilent :630-642: some sub-block with both samples within +-4 */
            const int16_t l = (int16_t)b->words[i][p16_line_of(b, i, 0)], r = (int16_t)b->words[i][p16_line_of(b, i, 1)];
            if (l < 4 && l >= -4 && r < 4 && r >= -4) silent = true;
        }
    }
    const bool pl1 = (b->picked_left & 1) || (b->picked_left & 4);                      /* hasPickedLeftBySub(SUBBLK_1) :311-322 */
    for (int i = 0; i < 9; i++) {                                                       /* the status bar :1423-1511 */
        uint32_t c = PX_BLK;
        const int sb = i / 2;
        if (i < 6) {
            const bool fixed_p = b->audio_state[sb] == 1;                               /* isDataFixedByP(blk) :520-539 */
            const bool pcrc = (b->picked_crc & 1) || (b->picked_crc & 4) || ((b->picked_crc & 2) && fixed_p);     /* hasPickedCRCBySub(blk) :335-357 */
            if (i % 2 == 0) { if (i == 0 && pl1) c = B1_BLU; else if (pcrc) c = B0_BLU; }
            else if (fixed_p) c = B1_GRN;
        } else if (i == 6) { if (!valid) c = B1_RED; }
        else if (i == 8) c = silent ? LIM_MARK : LIM_OK;
        for (int j = 0; j < 6; j++) *px++ = c;
    }
    for (int blk = 0; blk < 3; blk++)                                                   /* the six samples :1513-1712 */
        for (int word = 0; word < 2; word++) {
            const int line = p16_line_of(b, blk, word);
            const uint16_t v = b->words[blk][line];
            const bool crc = p16_crc(b, blk, line), wv = p16_val(b, blk, line), picked = blk == 0 && ((b->picked_left >> line) & 1);    /* hasPickedSample :408-434 */
            for (int bit = 15; bit >= 0; bit--) {
                const bool one = (v >> bit) & 1;
                uint32_t c = one ? B1_GRY : PX_BLK;
                if (!valid) {
                    if (b->audio_state[blk] != 2) { if (!crc) c = one ? B1_RED : B0_RED; else if (picked) c = one ? B1_BLU : B0_BLU; }
                    else if (!wv) c = one ? B1_MGN : B0_MGN;
                } else if (b->audio_state[blk] == 1) { if (!crc) c = one ? B1_GRN : B0_GRN; else if (picked) c = one ? B1_BLU : B0_BLU; }
                else if (!wv) c = one ? B1_RED : B0_RED;
                else if (picked) c = one ? B1_BLU : B0_BLU;
                for (int j = 0; j < 6; j++) *px++ = c;
            }
        }
    for (int i = 0; i < 8; i++) {                                                       /* format, emphasis, BROKEN :1714-1757 */
        uint32_t c = PX_BLK;
        if (i == 0) c = LIM_OK;
        else if (i == 2) { if (b->flags & SDV_P16B_EI_FORMAT) c = B1_BLU; }
        else if (i == 3) { if (b->flags & SDV_P16B_EMPHASIS) c = B0_GRN; }
        else if (i == 5 || i == 6) { if (broken_any) c = B1_MGN; }
        for (int j = 0; j < 6; j++) *px++ = c;
    }
}

long orc_vis_render_blocks(int kind, const void *blocks_, size_t n_blocks, const uint32_t *frame_blocks, size_t n_frames, uint32_t *canvas,
                           uint32_t *out, size_t out_cap)
{
    uint32_t w, h;
    const bool m2 = (kind & 0x100) != 0;           /* SDV_VIS_M2_SAMPLES */
    kind &= 0xFF;
    orc_vis_canvas_size(kind, &w, &h);
    if (kind == ORC_VIS_PCM1_BLOCKS) {
        const sdv_pcm1_block_rec *pb = (const sdv_pcm1_block_rec *)blocks_;
        size_t at1 = 0;
        for (size_t f = 0; f < n_frames; f++) {
            uint32_t fill = 0;
            for (uint32_t i = 0; i < frame_blocks[f] && at1 < n_blocks; i++, at1++) fill += pcm1_block(&pb[at1], canvas, w, fill, h);
            if (f < out_cap) memcpy(out + f * (size_t)w * h, canvas, (size_t)w * h * 4);
        }
        return (long)n_frames;
    }
    if (kind == ORC_VIS_PCM16X0_BLOCKS) {
        const sdv_pcm16x0_block_rec *pb = (const sdv_pcm16x0_block_rec *)blocks_;
        size_t at16 = 0;
        for (size_t f = 0; f < n_frames; f++) {
            for (uint32_t i = 0; i < frame_blocks[f] && at16 < n_blocks; i++, at16++)
                if (i < h) pcm16_block(&pb[at16], canvas + (size_t)i * w);
            if (f < out_cap) memcpy(out + f * (size_t)w * h, canvas, (size_t)w * h * 4);
        }
        return (long)n_frames;
    }
    const sdv_block_rec *blocks = (const sdv_block_rec *)blocks_;
    if (kind != ORC_VIS_STC007_BLOCKS_NTSC && kind != ORC_VIS_STC007_BLOCKS_PAL) return -1;
    size_t at = 0;
    for (size_t f = 0; f < n_frames; f++) {          /* newFrameAssembled -> prepareNewFrame: the canvas goes out, the fill row back to 0 */
        for (uint32_t i = 0; i < frame_blocks[f] && at < n_blocks; i++, at++)
            if (i < h) stc_block(&blocks[at], canvas + (size_t)i * w, m2);
        if (f < out_cap) memcpy(out + f * (size_t)w * h, canvas, (size_t)w * h * 4);
    }
    return (long)n_frames;
}

/* ---- the assembled-lines window: renderNewLine(STC007Line) (:939-1169) on the lines STC007DataStitcher hands over (newLineProcessed), whose words
 * carry their own states after the CWD pass ------------------------------------------------------------------------------------------------- */
static void stc_asm_line(const sdv_asm_line_rec *r, uint32_t *px)
{
    const bool crc_valid = (r->flags & SDV_AL_CRC_VALID) != 0, markers = (r->flags & SDV_AL_MARKERS) != 0, forced = (r->flags & SDV_AL_FORCED_BAD) != 0;
    for (int i = 0; i < 4; i++) {
        const bool one = (i % 2) == 0;
        const uint32_t c = (crc_valid || markers) ? (one ? B1_GRY : B0_GRY) : PX_BLK;
        for (int j = 0; j < 5; j++) *px++ = c;
    }
    for (int w = 0; w < 9; w++) {
        const bool wc = (r->word_crc_ok >> w) & 1, wv = (r->word_valid >> w) & 1;
        for (int b = (w == 8 ? 16 : 14) - 1; b >= 0; b--) {
            const bool one = ((r->words[w] >> b) & 1) != 0;
            uint32_t c;
            if (forced) c = one ? B1_MGN : B0_MGN;
            else if (wc) c = one ? B1_GRY : B0_GRY;
            else if (wv) c = one ? B1_GRN : B0_GRN;
            else if (markers) c = one ? B1_YEL : B0_YEL;
            else c = one ? B1_RED : B0_RED;
            for (int j = 0; j < 5; j++) *px++ = c;
        }
    }
    for (int i = 0; i < 5; i++) {
        uint32_t c = PX_BLK;
        if (i == 0) { if (crc_valid || markers) c = B0_GRY; }
        else if (crc_valid) c = B1_MARK;
        else if (markers) c = B1_GRY;
        for (int j = 0; j < 5; j++) *px++ = c;
    }
}
long orc_vis_render_asm_lines(int kind, const sdv_asm_line_rec *lines, size_t n_lines, const uint32_t *frame_lines, size_t n_frames, uint32_t *canvas,
                              uint32_t *out, size_t out_cap)
{
    uint32_t w, h;
    orc_vis_canvas_size(kind, &w, &h);
    if (kind != ORC_VIS_STC007_ASM_NTSC && kind != ORC_VIS_STC007_ASM_PAL) return -1;
    size_t at = 0;
    for (size_t f = 0; f < n_frames; f++) {
        for (uint32_t i = 0; i < frame_lines[f] && at < n_lines; i++, at++)
            if (i < h) stc_asm_line(&lines[at], canvas + (size_t)i * w);
        if (f < out_cap) memcpy(out + f * (size_t)w * h, canvas, (size_t)w * h * 4);
    }
    return (long)n_frames;
}
