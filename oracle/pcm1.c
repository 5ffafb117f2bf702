/*
 * pcm1.c - CPU restatement of the PCM-1 back half: PCM1SubLine (pcm1subline.cpp), PCM1DataBlock (pcm1datablock.cpp),
 * PCM1Deinterleaver::processBlock (pcm1deinterleaver.cpp:69-278) and PCM1DataStitcher (pcm1datastitcher.cpp:63-1772).
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).  PCM-1 frames are stitched one at a time and independently.
 */
#include "pcm1.h"
#include <stdlib.h>
#include <string.h>

enum { P1_BIT_RANGE = 1 << 12, P1_BIT_SIGN = 1 << 11, P1_WORD_MASK = (1 << 13) - 1, P1_CRC_SILENT = 0xECBF };
enum { P1_LINES_PF = 245, P1_SUBLINES_PF = 245 * 3, P1_MIN_GOOD = 245 * 4 / 5, P1_BUF_TRIM = 3 * 640 /* MAX_VLINE_QUEUE_SIZE, config.h:83 */ };
enum { P1_INT_BLK = 8, P1_MIN_DEINT = 735, P1_STRIPE_TWO = 46, P1_STRIPE_LEN = 46, P1_STRIPE_SHORT = 45, P1_WORD_CNT = 184, P1_WORD_CNT_SHORT = 182 };
enum { P1_ORDER_UNK = 0, P1_ORDER_TFF, P1_ORDER_BFF };

/* PCM1Line::calcCRC (pcm1line.cpp:158-171): CRC-16/CCITT over the six inverted 13-bit words, result inverted */
uint16_t orc_pcm1_crc_words(const uint16_t *w)
{
    uint16_t crc = ORC_CRC_INIT;
    for (int i = 0; i < 6; i++) crc = orc_crc16_update(crc, (uint16_t)~w[i], 13);
    return (uint16_t)~crc;
}

typedef struct { uint32_t frame_number; uint16_t line_number; uint8_t picked_left, picked_right, part; uint16_t words[2]; bool bw_set, crc; } p1_sub;
typedef struct {
    uint32_t frame_number; uint16_t start_line, stop_line; uint8_t interleave_num; uint16_t sample_rate; bool emphasis;
    uint16_t words[P1_WORD_CNT]; bool word_crc[P1_WORD_CNT], picked_left[P1_WORD_CNT], picked_crc[P1_WORD_CNT]; bool short_blk;
} p1_block;

static void sub_clear(p1_sub *s)   /* pcm1subline.cpp:47-64 */
{
    s->frame_number = 0; s->line_number = 0; s->picked_left = s->picked_right = 0; s->part = 0; s->bw_set = false; s->crc = false;
    s->words[0] = s->words[1] = P1_BIT_RANGE;
}
static void blk_clear(p1_block *b)   /* pcm1datablock.cpp:51-66 */
{
    b->frame_number = 0; b->start_line = b->stop_line = 0; b->interleave_num = 0; b->sample_rate = 44056; b->short_blk = b->emphasis = false;
    for (int i = 0; i < P1_WORD_CNT; i++) { b->words[i] = P1_BIT_RANGE; b->word_crc[i] = false; b->picked_left[i] = b->picked_crc[i] = false; }
}
static int blk_word_count(const p1_block *b) { return b->short_blk ? P1_WORD_CNT_SHORT : P1_WORD_CNT; }
static void blk_set_word(p1_block *b, int i, uint16_t w, bool valid, bool pl, bool pc)   /* :69-78 */
{
    if (i < blk_word_count(b)) { b->words[i] = w; b->word_crc[i] = valid; b->picked_left[i] = pl; b->picked_crc[i] = pc; }
}
static void blk_set_short(p1_block *b)   /* :87-98 */
{
    b->short_blk = true;
    for (int i = P1_WORD_CNT_SHORT - 1; i <= P1_WORD_CNT - 1; i++) { b->words[i] = P1_BIT_RANGE; b->word_crc[i] = false; b->picked_left[i] = b->picked_crc[i] = false; }
}
static int16_t blk_sample(const p1_block *b, int i)   /* :309-348 */
{
    if (i >= blk_word_count(b)) return 0;
    uint16_t w = b->words[i];
    if ((w & P1_BIT_RANGE) == 0) w = (uint16_t)(w << 4);
    else {
        bool pos = (w & P1_BIT_SIGN) == 0;
        w = (uint16_t)(w & ~P1_BIT_RANGE);
        w = (uint16_t)(w << 2);
        if (!pos) w |= (1 << 15) | (1 << 14);
    }
    return (int16_t)w;
}
static int blk_errors(const p1_block *b) { int n = 0; for (int i = 0; i < blk_word_count(b); i++) if (!b->word_crc[i]) n++; return n; }   /* :351-363 */
static bool blk_has_picked(const p1_block *b) { for (int i = 0; i < P1_WORD_CNT; i++) if (b->picked_left[i] || b->picked_crc[i]) return true; return false; }   /* :114-136 */

/* PCM1Deinterleaver::setWordData (pcm1deinterleaver.cpp:150-278), line_sh = 0 as processBlock calls it */
static void set_word_data(const p1_sub *q, p1_block *b, int itl, bool even_stripe, bool ignore_crc)
{
    bool even_itl = (itl % 2) == 0;
    int ofs = itl * (2 * P1_STRIPE_LEN), stripe_len;
    if (itl != P1_INT_BLK - 1) { stripe_len = P1_STRIPE_LEN; b->short_blk = false; }
    else { stripe_len = even_stripe ? P1_STRIPE_SHORT : P1_STRIPE_LEN; blk_set_short(b); }
    int one = ofs, two = ofs + P1_STRIPE_TWO;
    b->frame_number = q[one].frame_number; b->start_line = q[one].line_number;
    b->stop_line = q[two + stripe_len - 1].line_number;
    int word_ofs = even_stripe ? 2 : 0;
    for (int wp = 0; wp < stripe_len; wp++) {
        const p1_sub *s = &q[((even_itl == even_stripe) ? one : two) + wp];
        bool ok = ignore_crc ? s->bw_set : s->crc;
        blk_set_word(b, word_ofs, s->words[0], ok, s->picked_left > 0, s->picked_right > 0); word_ofs++;
        blk_set_word(b, word_ofs, s->words[1], ok, false, s->picked_right > 0); word_ofs++;
        word_ofs += 2;
    }
}

/* FrameAsmPCM1 with the base class fields the stitcher touches */
typedef struct {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop, odd_top_padding, odd_bottom_padding, even_top_padding, even_bottom_padding, blocks_fix_bp;
    uint8_t field_order, odd_ref, even_ref, service_type; bool order_preset, order_guessed, odd_emphasis, even_emphasis;
} p1_frasm;
static void frasm_clear(p1_frasm *f)   /* FrameAsmDescriptor::clear + FrameAsmPCM1::clearMisc (frametrimset.cpp:383-470, 727-752) */
{
    memset(f, 0, sizeof(*f));
    f->odd_bottom_data = f->even_bottom_data = 0xFFFF;
}
static void frasm_to_pod(const p1_frasm *f, sdv_frame_asm_pcm1 *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f->frame_number;
    o->odd_std_lines = f->odd_std_lines; o->even_std_lines = f->even_std_lines; o->odd_data_lines = f->odd_data_lines; o->even_data_lines = f->even_data_lines;
    o->odd_valid_lines = f->odd_valid_lines; o->even_valid_lines = f->even_valid_lines;
    o->odd_top_data = f->odd_top_data; o->odd_bottom_data = f->odd_bottom_data; o->even_top_data = f->even_top_data; o->even_bottom_data = f->even_bottom_data;
    o->odd_sample_rate = f->odd_sample_rate; o->even_sample_rate = f->even_sample_rate;
    o->blocks_total = f->blocks_total; o->blocks_drop = f->blocks_drop; o->samples_drop = f->samples_drop;
    o->odd_top_padding = f->odd_top_padding; o->odd_bottom_padding = f->odd_bottom_padding; o->even_top_padding = f->even_top_padding; o->even_bottom_padding = f->even_bottom_padding;
    o->blocks_fix_bp = f->blocks_fix_bp;
    o->field_order = f->field_order; o->odd_ref = f->odd_ref; o->even_ref = f->even_ref; o->service_type = f->service_type;
    o->flags = (uint8_t)((f->order_preset ? SDV_FA_ORDER_PRESET : 0) | (f->order_guessed ? SDV_FA_ORDER_GUESSED : 0) |
                         (f->odd_emphasis ? SDV_FA1_ODD_EMPHASIS : 0) | (f->even_emphasis ? SDV_FA1_EVEN_EMPHASIS : 0));
}

/* record accessors = the PCM1Line / PCMLine getters the stitcher uses */
static bool r_service(const sdv_pcm1_line_rec *r) { return r->service_type != SDV_SRV_NO; }
/* a service line is a cleared PCM1Line that keeps only its frame and line number (PCMLine::setServiceLine, pcmline.cpp:490-502):
 * no B/W levels, CRC invalid, reference level 0 - whatever else the record holds is ignored */
static bool r_crc_if(const sdv_pcm1_line_rec *r) { return !r_service(r) && r->calc_crc == r->words[6]; }
static bool r_crc(const sdv_pcm1_line_rec *r) { return !(r->flags & SDV_LF_FORCED_BAD) && r_crc_if(r); }
static bool r_bw(const sdv_pcm1_line_rec *r) { return !r_service(r) && (r->flags & SDV_LF_BW_SET) != 0; }

typedef struct {
    sdv_pcm1_stitch_settings st;
    p1_frasm f1; p1_sub odd[P1_SUBLINES_PF], even[P1_SUBLINES_PF]; p1_sub *q; size_t qn, qcap;
    bool header_present, emphasis_set, file_start, file_end;
    sdv_sample_pair *out; size_t out_n, out_cap; sdv_frame_asm_pcm1 *frames; size_t frames_n, frames_cap;
    /* the visualiser's feeds (newBlockProcessed / newLineProcessed), when asked for */
    sdv_pcm1_block_rec *vb; size_t vb_n, vb_cap; sdv_pcm1_asm_line_rec *vl; size_t vl_n, vl_cap;
} p1_stitcher;

static void q_push(p1_stitcher *s, const p1_sub *l)
{
    if (s->qn == s->qcap) { s->qcap = s->qcap ? s->qcap * 2 : 1024; s->q = (p1_sub *)realloc(s->q, s->qcap * sizeof(p1_sub)); }
    s->q[s->qn++] = *l;
}
static void out_pair(p1_stitcher *s, const sdv_sample_pair *p) { if (s->out_n < s->out_cap) s->out[s->out_n] = *p; s->out_n++; }
static void out_frasm(p1_stitcher *s, const p1_frasm *f) { if (s->frames_n < s->frames_cap) frasm_to_pod(f, &s->frames[s->frames_n]); s->frames_n++; }
static void out_service(p1_stitcher *s, uint8_t srv)
{
    p1_frasm d; frasm_clear(&d); d.service_type = srv; out_frasm(s, &d);
    sdv_sample_pair p; memset(&p, 0, sizeof(p)); p.sample_rate = 44056; p.service_type = srv; out_pair(s, &p);
}

/* splitLineToSubline (:571-606); a NULL record = a cleared PCM1Line (what a filler of the odd field is turned into, :713-717) */
static void split_line(const sdv_pcm1_line_rec *r, uint32_t frame, uint16_t line, p1_sub *o, uint8_t part)
{
    sub_clear(o);
    o->frame_number = frame; o->line_number = line; o->part = part;
    if (r) {
        o->bw_set = r_bw(r);
        o->words[0] = (uint16_t)(r->words[2 * part] & P1_WORD_MASK); o->words[1] = (uint16_t)(r->words[2 * part + 1] & P1_WORD_MASK);
        if (part == 0) o->picked_left = r->picked_bits_left;
        o->picked_right = r->picked_bits_right;
        o->crc = r_crc(r);
    } else {
        /* PCM1Line::clear(): silent words, calc_crc = CRC_SILENT, read CRC inverted: invalid; no B/W levels; nothing picked */
        o->bw_set = false; o->words[0] = o->words[1] = P1_BIT_RANGE; o->crc = false;
    }
}

static void add_padding(p1_stitcher *s, uint32_t frame, uint16_t line_cnt, uint16_t *last_line)   /* addFieldPadding :1019-1073 */
{
    p1_sub e; sub_clear(&e);
    for (uint16_t i = 0; i < line_cnt; i++) {
        e.frame_number = frame; e.line_number = *last_line; *last_line = (uint16_t)(*last_line + 2);
        for (uint8_t part = 0; part < 3; part++) { e.part = part; q_push(s, &e); }
    }
}
static void add_lines(p1_stitcher *s, p1_sub *field, uint16_t start, uint16_t count, uint16_t *last_line)   /* addLinesFromField :952-1016 */
{
    if (!(P1_SUBLINES_PF >= start && P1_SUBLINES_PF >= (int)start + (int)count)) return;
    for (uint16_t i = start; i < (uint16_t)(start + count); i++) {
        field[i].line_number = *last_line;
        q_push(s, &field[i]);
        if (field[i].part == 2) *last_line = (uint16_t)(field[i].line_number + 2);
    }
}
/* the block as outputDataBlock hands it to the visualiser (newBlockProcessed, :1333) */
static void vis_block(p1_stitcher *s, const p1_block *b)
{
    if (s->vb && s->vb_n < s->vb_cap) {
        sdv_pcm1_block_rec *o = &s->vb[s->vb_n];
        memset(o, 0, sizeof(*o));
        o->frame_number = b->frame_number; o->start_line = b->start_line; o->stop_line = b->stop_line; o->interleave_num = b->interleave_num;
        /* the last block of a field: setWordData reads its stop line one sub-line past the end of the queue (:204-211, stripe_len 46) - undefined in
         * the reference; here the number that line would have had */
        if (b->interleave_num == P1_INT_BLK - 1 && s->qn >= P1_MIN_DEINT) o->stop_line = (uint16_t)(s->q[P1_MIN_DEINT - 1].line_number + 2);
        o->flags = (uint8_t)((b->short_blk ? SDV_P1B_SHORT : 0) | (b->emphasis ? SDV_P1B_EMPHASIS : 0));
        o->sample_rate = b->sample_rate;
        for (int i = 0; i < P1_WORD_CNT; i++) {
            o->words[i] = b->words[i];
            o->word_flags[i] = (uint8_t)((b->word_crc[i] ? SDV_P1W_CRC_OK : 0) | (b->picked_left[i] ? SDV_P1W_PICKED_LEFT : 0) |
                                         ((b->picked_left[i] || b->picked_crc[i]) ? SDV_P1W_PICKED_WORD : 0));
        }
    }
    s->vb_n++;
}
static void perform_deinterleave(p1_stitcher *s)   /* :1382-1453 */
{
    p1_block b;
    p1_frasm *f = &s->f1;
    /* the whole queue to the visualiser (newLineProcessed, :1392-1407): the sub-lines of this frame; 735 places per field here */
    for (size_t i = 0; i < (size_t)P1_MIN_DEINT; i++) {
        if (s->vl && s->vl_n < s->vl_cap) {
            sdv_pcm1_asm_line_rec *o = &s->vl[s->vl_n];
            memset(o, 0, sizeof(*o));
            o->frame_number = f->frame_number; o->words[0] = o->words[1] = P1_BIT_RANGE; o->flags = SDV_P1S_SKIP;
            if (i < s->qn) {
                const p1_sub *l = &s->q[i];
                o->line_number = l->line_number; o->line_part = l->part;
                if (l->frame_number == f->frame_number) {
                    o->words[0] = l->words[0]; o->words[1] = l->words[1]; o->picked_bits_left = l->picked_left; o->picked_bits_right = l->picked_right;
                    o->flags = (uint8_t)((l->bw_set ? SDV_P1S_BW_SET : 0) | (l->crc ? SDV_P1S_CRC_VALID : 0));
                }
            }
        }
        s->vl_n++;
    }
    for (int iblk = 0; iblk < P1_INT_BLK; iblk++) {
        blk_clear(&b);
        if (s->qn >= P1_MIN_DEINT) {            /* processBlock: DI_RET_NO_DATA leaves the cleared block */
            b.interleave_num = (uint8_t)iblk;
            set_word_data(s->q, &b, iblk, true, !s->st.use_ecc);
            set_word_data(s->q, &b, iblk, false, !s->st.use_ecc);
        }
        f->blocks_total++;
        b.emphasis = s->emphasis_set; f->odd_emphasis = f->even_emphasis = s->emphasis_set;
        b.sample_rate = 44100; f->odd_sample_rate = f->even_sample_rate = 44100;
        int errs = blk_errors(&b);
        bool valid = errs == 0;
        if (!valid) { f->blocks_drop++; f->samples_drop = (uint16_t)(f->samples_drop + (uint8_t)errs); }
        if (valid && blk_has_picked(&b)) f->blocks_fix_bp++;
        for (int w = 0; w < blk_word_count(&b); w += 2) {   /* outputDataBlock :1271-1334 */
            sdv_sample_pair p; memset(&p, 0, sizeof(p));
            p.sample_rate = 44100; p.emphasis = b.emphasis;
            p.audio_word[0] = blk_sample(&b, w); p.audio_word[1] = blk_sample(&b, w + 1);
            p.sample_flags[0] = (uint8_t)((valid ? SDV_SF_BLOCK_OK : 0) | (b.word_crc[w] ? SDV_SF_WORD_VALID : 0));
            p.sample_flags[1] = (uint8_t)((valid ? SDV_SF_BLOCK_OK : 0) | (((w + 1) < blk_word_count(&b) && b.word_crc[w + 1]) ? SDV_SF_WORD_VALID : 0));
            out_pair(s, &p);
        }
        vis_block(s, &b);
    }
}

/* one turn of doFrameReassemble (:1609-1750) for the frame recs[lo..hi) (END_FRAME excluded) */
static void stitch_frame(p1_stitcher *s, const sdv_pcm1_line_rec *recs, size_t lo, size_t hi, uint32_t frame)
{
    p1_frasm *f = &s->f1;
    const sdv_pcm1_stitch_settings *st = &s->st;
    size_t n = hi - lo;
    const sdv_pcm1_line_rec *t = recs + lo;
    if (n > P1_BUF_TRIM) {      /* fillUntilFullFrame (:150-196) keeps the first BUF_SIZE_TRIM lines that carry the frame's number; the rest is popped unseen */
        size_t kept = 0;
        for (size_t i = 0; i < n; i++) if (t[i].frame_number == frame && ++kept == P1_BUF_TRIM) { n = i + 1; break; }
    }
    f->frame_number = frame;
    /* findFrameTrim (:202-568) */
    uint16_t o_good = 0, e_good = 0;
    bool e_top = false, o_top = false, o_skip = false, e_skip = false, ds_odd = false, ds_even = false;
    s->header_present = s->emphasis_set = s->file_start = s->file_end = false;
    f->even_top_data = f->even_bottom_data = f->odd_top_data = f->odd_bottom_data = 0;
    for (size_t i = 0; i < n; i++) {
        const sdv_pcm1_line_rec *r = &t[i];
        if (r->frame_number != frame) continue;
        bool even = (r->line_number % 2) == 0;
        if (!r_service(r)) {
            if (r_crc(r)) {
                if (even) { ds_even = true; e_good++; if (e_good > P1_MIN_GOOD) e_skip = true; }
                else { ds_odd = true; o_good++; if (o_good > P1_MIN_GOOD) o_skip = true; }
            }
        } else if (r->service_type == SDV_SRV_HEADER_LINE) { if (even ? !ds_even : !ds_odd) s->header_present = true; }
        else if (r->service_type == SDV_SRV_NEW_FILE) s->file_start = true;
        else if (r->service_type == SDV_SRV_END_FILE) s->file_end = true;
    }
    ds_odd = ds_even = false;
    for (size_t i = n; i > 0;) {
        i--;
        const sdv_pcm1_line_rec *r = &t[i];
        bool even = (r->line_number % 2) == 0;
        if (!r_service(r)) {
            if (r->frame_number == frame && r_crc(r)) {
                if (even) { ds_even = true; if (ds_odd) break; } else { ds_odd = true; if (ds_even) break; }
            }
        } else if (r->service_type == SDV_SRV_HEADER_LINE) {
            if (r->frame_number == frame && (even ? !ds_even : !ds_odd)) s->emphasis_set = true;
        }
    }
    if (!st->auto_offset) {
        o_top = e_top = true;
        f->odd_top_data = st->odd_offset > 0 ? (uint16_t)(2 * st->odd_offset + 1) : 1;
        f->even_top_data = st->even_offset > 0 ? (uint16_t)(2 * st->even_offset + 2) : 2;
    }
    for (size_t i = 0; i < n; i++) {
        const sdv_pcm1_line_rec *r = &t[i];
        if (r_service(r) && r->service_type != SDV_SRV_FILLER) continue;
        if (r->frame_number != frame) continue;
        bool even = (r->line_number % 2) == 0;
        bool skip = even ? e_skip : o_skip;
        if ((!skip && r_bw(r)) || (skip && r_crc_if(r))) {
            if (even) { if (!e_top) { f->even_top_data = r->line_number; e_top = true; } f->even_bottom_data = r->line_number; }
            else { if (!o_top) { f->odd_top_data = r->line_number; o_top = true; } f->odd_bottom_data = r->line_number; }
        }
    }
    if (s->file_start) { s->qn = 0; /* resetState :63-76: header/emphasis flags are NOT restored afterwards */ s->header_present = s->emphasis_set = false;
                         uint32_t fn = f->frame_number; uint16_t a = f->odd_top_data, b2 = f->odd_bottom_data, c = f->even_top_data, d = f->even_bottom_data;
                         frasm_clear(f); f->frame_number = fn; f->odd_top_data = a; f->odd_bottom_data = b2; f->even_top_data = c; f->even_bottom_data = d; }
    if (!s->file_end) {
        /* splitFrameToFields (:609-806) */
        uint32_t ref_o = 0, ref_e = 0, ref_ob = 0, ref_eb = 0;
        for (size_t i = 0; i < n; i++) {
            const sdv_pcm1_line_rec *r = &t[i];
            if (r_service(r) && r->service_type != SDV_SRV_FILLER) continue;
            if (r->frame_number != frame) continue;
            uint16_t ln = r->line_number;
            if ((ln % 2) == 0) {
                if (((f->even_top_data != f->even_bottom_data) || (f->even_top_data != 0)) && ln >= f->even_top_data && ln <= f->even_bottom_data)
                    for (uint8_t sub = 0; sub < 3; sub++)
                        if (f->even_data_lines < P1_SUBLINES_PF) {
                            const sdv_pcm1_line_rec *src = r_service(r) ? NULL : r;   /* a filler keeps frame and line number here */
                            split_line(src, r->frame_number, ln, &s->even[f->even_data_lines], sub); f->even_data_lines++;
                            if (src) ref_eb += r->ref_level;
                            if (r_crc(r)) { f->even_valid_lines++; ref_e += r->ref_level; }
                        }
            } else if (ln >= f->odd_top_data && ln <= f->odd_bottom_data) {
                const sdv_pcm1_line_rec *src = r_service(r) ? NULL : r;           /* a filler is cleared first (:713-717) */
                for (uint8_t sub = 0; sub < 3; sub++)
                    if (f->odd_data_lines < P1_SUBLINES_PF) {
                        /* PCMLine::clear() also zeroes frame and line number of the cleared filler */
                        split_line(src, src ? r->frame_number : 0, src ? ln : 0, &s->odd[f->odd_data_lines], sub); f->odd_data_lines++;
                        uint8_t ref = src ? r->ref_level : 0;
                        ref_ob += ref;
                        if (src && r_crc(r)) { f->odd_valid_lines++; ref_o += ref; }
                    }
            }
        }
        f->odd_ref = f->odd_valid_lines > 0 ? (uint8_t)(ref_o / f->odd_valid_lines) : (f->odd_data_lines > 0 ? (uint8_t)(ref_ob / f->odd_data_lines) : 0);
        f->even_ref = f->even_valid_lines > 0 ? (uint8_t)(ref_e / f->even_valid_lines) : (f->even_data_lines > 0 ? (uint8_t)(ref_eb / f->even_data_lines) : 0);
        /* findFramePadding (:809-923) */
        if (st->auto_offset) {
            uint16_t po = (uint16_t)((P1_SUBLINES_PF - f->odd_data_lines) / 3), pe = (uint16_t)((P1_SUBLINES_PF - f->even_data_lines) / 3);
            if (!s->header_present) { f->odd_bottom_padding = f->even_bottom_padding = 0; f->odd_top_padding = po; f->even_top_padding = pe; }
            else { f->odd_top_padding = f->even_top_padding = 0; f->odd_bottom_padding = po; f->even_bottom_padding = pe; }
        } else {
            f->odd_top_padding = st->odd_offset > 0 ? 0 : (uint16_t)(0 - st->odd_offset);
            f->even_top_padding = st->even_offset > 0 ? 0 : (uint16_t)(0 - st->even_offset);
            f->odd_bottom_padding = (uint16_t)((f->odd_bottom_data - f->odd_top_data) / 2 + 1);
            f->odd_bottom_padding = (uint16_t)(f->odd_bottom_padding + f->odd_top_padding);
            if (f->odd_bottom_padding > P1_LINES_PF) {
                f->odd_bottom_padding = (uint16_t)(f->odd_bottom_padding - P1_LINES_PF);
                f->odd_bottom_data = (uint16_t)(f->odd_bottom_data - (f->odd_bottom_padding * 2));
                f->odd_data_lines = (uint16_t)((f->odd_bottom_data - f->odd_top_data) / 2 + 1);
                f->odd_data_lines = (uint16_t)(f->odd_data_lines * 3);
            }
            f->even_bottom_padding = (uint16_t)((f->even_bottom_data - f->even_top_data) / 2 + 1);
            f->even_bottom_padding = (uint16_t)(f->even_bottom_padding + f->even_top_padding);
            if (f->even_bottom_padding > P1_LINES_PF) {
                f->even_bottom_padding = (uint16_t)(f->even_bottom_padding - P1_LINES_PF);
                f->even_bottom_data = (uint16_t)(f->even_bottom_data - (f->even_bottom_padding * 2));
                f->even_data_lines = (uint16_t)((f->even_bottom_data - f->even_top_data) / 2 + 1);
                f->even_data_lines = (uint16_t)(f->even_data_lines * 3);
            }
            f->odd_bottom_padding = (uint16_t)((P1_SUBLINES_PF - f->odd_data_lines) / 3 - f->odd_top_padding);
            f->even_bottom_padding = (uint16_t)((P1_SUBLINES_PF - f->even_data_lines) / 3 - f->even_top_padding);
        }
        f->order_preset = true; f->order_guessed = false; f->field_order = st->field_order == P1_ORDER_BFF ? P1_ORDER_BFF : P1_ORDER_TFF;
        if (s->file_start) out_service(s, 1);
        for (int field = 0; field < 2; field++) {      /* fillFirstFieldForOutput / fillSecondFieldForOutput (:1076-1218) */
            bool odd = (f->field_order == P1_ORDER_TFF) == (field == 0);
            uint16_t last_line = (uint16_t)((f->field_order == P1_ORDER_TFF) == (field == 0) ? 1 : 2);
            add_padding(s, f->frame_number, odd ? f->odd_top_padding : f->even_top_padding, &last_line);
            add_lines(s, odd ? s->odd : s->even, 0, odd ? f->odd_data_lines : f->even_data_lines, &last_line);
            add_padding(s, f->frame_number, odd ? f->odd_bottom_padding : f->even_bottom_padding, &last_line);
            perform_deinterleave(s);
            s->qn = 0;
        }
        f->odd_data_lines /= 3; f->even_data_lines /= 3; f->odd_valid_lines /= 3; f->even_valid_lines /= 3;
        f->odd_std_lines = f->even_std_lines = P1_LINES_PF;
        out_frasm(s, f);
    } else {
        out_service(s, 2);
        s->qn = 0; s->header_present = s->emphasis_set = false;
    }
    frasm_clear(f);
}

void orc_default_pcm1_stitch_settings(sdv_pcm1_stitch_settings *st)
{
    memset(st, 0, sizeof(*st));
    st->field_order = P1_ORDER_TFF; st->auto_offset = 1; st->use_ecc = 1;
}

long orc_pcm1_stitch_run(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                         sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames)
{
    return orc_pcm1_stitch_run_vis(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL, 0, NULL, NULL, 0, NULL);
}
/* ... with the visualiser's feeds: blocks (16 per frame) and sub-lines (1470 per frame) next to the pairs */
long orc_pcm1_stitch_run_vis(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                             sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm1_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                             sdv_pcm1_asm_line_rec *lines, size_t lines_cap, size_t *n_lines)
{
    p1_stitcher *s = (p1_stitcher *)calloc(1, sizeof(p1_stitcher));
    s->st = *st; s->out = out; s->out_cap = out_cap; s->frames = frames; s->frames_cap = frames_cap;
    for (int i = 0; i < P1_SUBLINES_PF; i++) { sub_clear(&s->odd[i]); sub_clear(&s->even[i]); }     /* the field buffers of a new stitcher: cleared sub-lines (silent words) */
    s->vb = blocks; s->vb_cap = blocks_cap; s->vl = lines; s->vl_cap = lines_cap;
    frasm_clear(&s->f1);
    size_t lo = 0;
    for (size_t i = 0; i < n_recs; i++)
        if (recs[i].service_type == SDV_SRV_END_FRAME) { stitch_frame(s, recs, lo, i, recs[i].frame_number); lo = i + 1; }
    long n = s->out_n > out_cap ? -1 : (long)s->out_n;
    if (n_frames) *n_frames = s->frames_n;
    if (n_blocks) *n_blocks = s->vb_n;
    if (n_lines) *n_lines = s->vl_n;
    free(s->q); free(s);
    return n;
}
