/*
 * stitcher.c - CPU restatement of STC007DataStitcher (stc007datastitcher.cpp:3-7488): two-frame window,
 * trim detection, field split, audio-resolution / video-standard / field-order detection, padding search
 * between fields, frame assembly with filler lines, CWD pre-scan, final deinterleave with seam masking and
 * PCMSamplePair output.  TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).
 * The Qt worker loop, mutexes, sleeps and GUI signals are not restated; orc_stitcher_step() is one turn of
 * doFrameReassemble (stc007datastitcher.cpp:7273-7475).
 */
#include "stitcher.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small containers */
static void dq_clear(orc_line_deque *q) { q->head = 0; q->n = 0; }
static void dq_reserve(orc_line_deque *q, size_t need)
{
    if (q->head + q->n + need <= q->cap) return;
    if (q->head > 0) { memmove(q->v, q->v + q->head, q->n * sizeof(orc_stc_line)); q->head = 0; }
    if (q->n + need > q->cap) {
        size_t nc = q->cap ? q->cap * 2 : 1024;
        while (nc < q->n + need) nc *= 2;
        q->v = (orc_stc_line *)realloc(q->v, nc * sizeof(orc_stc_line));
        q->cap = nc;
    }
}
static void dq_push_back(orc_line_deque *q, const orc_stc_line *l) { dq_reserve(q, 1); q->v[q->head + q->n] = *l; q->n++; }
static void dq_pop_front(orc_line_deque *q) { q->head++; q->n--; }
static void dq_pop_back(orc_line_deque *q) { q->n--; }
static orc_stc_line *dq_at(orc_line_deque *q, size_t i) { return &q->v[q->head + i]; }

static void circ_clear(orc_circ65 *c) { c->fill_cnt = c->head_i = c->tail_i = 0; c->is_full = false; }   /* circbuffer.h:48-52 */
static void circ_push(orc_circ65 *c, uint8_t v)   /* circbuffer.h:80-103 */
{
    c->data[c->head_i] = v;
    if (c->is_full) { c->tail_i = (c->tail_i + 1) % ORC_STATS_DEPTH; c->head_i = c->tail_i; }
    else { c->fill_cnt++; c->head_i = (c->head_i + 1) % ORC_STATS_DEPTH; c->is_full = (c->head_i == c->tail_i); }
}
static void circ_fill(orc_circ65 *c, uint8_t v) { circ_clear(c); while (!c->is_full) circ_push(c, v); }
static uint8_t circ_get(const orc_circ65 *c, size_t i) { return c->data[(i + c->tail_i) % ORC_STATS_DEPTH]; }

/* ------------------------------------------------------------------ FrameAsmSTC007 (frametrimset.cpp:383-959) */
static void frasm_base_clear_asm_stats(orc_frasm *f) { f->odd_ref = f->even_ref = 0; f->blocks_total = f->blocks_drop = f->samples_drop = 0; }
static void frasm_base_clear_misc(orc_frasm *f)
{
    f->odd_std_lines = f->even_std_lines = f->odd_data_lines = f->even_data_lines = f->odd_valid_lines = f->even_valid_lines = 0;
    f->odd_sample_rate = f->even_sample_rate = 0;
    f->field_order = ORC_ORDER_UNK;
    f->odd_emphasis = f->even_emphasis = false;
    f->order_preset = f->order_guessed = false;
    f->drawn = false; f->service_type = 0;
    frasm_base_clear_asm_stats(f);
}
static void frasm_clear_asm_stats(orc_frasm *f)
{
    frasm_base_clear_asm_stats(f);
    f->blocks_broken_field = f->blocks_broken_seam = f->blocks_fix_p = f->blocks_fix_q = f->blocks_fix_cwd = 0;
}
static void frasm_clear_misc(orc_frasm *f)
{
    frasm_base_clear_misc(f);
    f->video_standard = ORC_VID_UNKNOWN;
    f->tff_cnt = f->bff_cnt = 0; f->odd_resolution = f->even_resolution = 0;
    f->inner_padding = f->outer_padding = 0;
    f->trim_ok = false; f->inner_padding_ok = f->outer_padding_ok = false;
    f->inner_silence = f->outer_silence = true;
    f->vid_std_preset = f->vid_std_guessed = false;
    f->ctrl_index = f->ctrl_hour = f->ctrl_minute = f->ctrl_second = f->ctrl_field = -1;
    frasm_clear_asm_stats(f);
}
static void frasm_clear(orc_frasm *f)
{
    f->frame_number = 0; f->odd_top_data = 0; f->odd_bottom_data = 0xFFFF; f->even_top_data = 0; f->even_bottom_data = 0xFFFF;
    frasm_base_clear_misc(f);
    frasm_clear_misc(f);
}
static bool frasm_order_set(const orc_frasm *f) { return f->field_order == ORC_ORDER_TFF || f->field_order == ORC_ORDER_BFF; }
static void frasm_preset_tff(orc_frasm *f) { f->order_preset = true; f->order_guessed = false; f->field_order = ORC_ORDER_TFF; }
static void frasm_preset_bff(orc_frasm *f) { f->order_preset = true; f->order_guessed = false; f->field_order = ORC_ORDER_BFF; }
static void frasm_set_order_unknown(orc_frasm *f) { if (!f->order_preset) { f->field_order = ORC_ORDER_UNK; f->order_guessed = false; } }
static void frasm_set_order_tff(orc_frasm *f) { if (!f->order_preset) f->field_order = ORC_ORDER_TFF; }
static void frasm_set_order_bff(orc_frasm *f) { if (!f->order_preset) f->field_order = ORC_ORDER_BFF; }
static void frasm_update_vid_std_soft(orc_frasm *f, uint8_t s) { if (!f->vid_std_preset && s < ORC_VID_MAX) f->video_standard = s; }

/* ------------------------------------------------------------------ line accessors */
static bool ln_service(const orc_stc_line *l) { return l->service_type != ORC_SRV_NO; }
static bool ln_word_valid(const orc_stc_line *l, int i) { return l->forced_bad ? false : l->word_valid[i]; }
static bool ln_word_crc_ok(const orc_stc_line *l, int i) { return l->forced_bad ? false : l->word_crc[i]; }

/* ------------------------------------------------------------------ lifetime */
static void reset_state(orc_stitcher *s)   /* :69-89 */
{
    circ_fill(&s->stats_field_order, ORC_ORDER_UNK);
    circ_fill(&s->stats_resolution, ORC_SAMPLE_RES_UNKNOWN);
    dq_clear(&s->conv_queue);
    s->last_pad_counter = 0xFF;
    s->broken_countdown = 0;
    frasm_clear(&s->frasm_f0);
    frasm_clear_misc(&s->frasm_f1);
    frasm_clear_misc(&s->frasm_f2);
}

void orc_stitcher_init(orc_stitcher *s)   /* :3-38, :7228-7236 */
{
    memset(s, 0, sizeof(*s));
    s->trim_buf = (orc_stc_line *)calloc(ORC_BUF_SIZE_TRIM, sizeof(orc_stc_line));
    s->frame1_even = (orc_stc_line *)calloc(ORC_BUF_SIZE_FIELD, sizeof(orc_stc_line));
    s->frame1_odd = (orc_stc_line *)calloc(ORC_BUF_SIZE_FIELD, sizeof(orc_stc_line));
    s->frame2_even = (orc_stc_line *)calloc(ORC_BUF_SIZE_FIELD, sizeof(orc_stc_line));
    s->frame2_odd = (orc_stc_line *)calloc(ORC_BUF_SIZE_FIELD, sizeof(orc_stc_line));
    s->preset_video_mode = ORC_VID_UNKNOWN; s->preset_field_order = ORC_ORDER_UNK; s->preset_audio_res = ORC_SAMPLE_RES_UNKNOWN;
    s->preset_sample_rate = 1;      /* PCMSamplePair::SAMPLE_RATE_AUTO */
    s->enable_P_code = false; s->enable_Q_code = false; s->enable_CWD = true; s->mode_m2 = false;
    orc_deint_init(&s->pad_checker); orc_deint_init(&s->lines_to_block);
    orc_block_clear(&s->padding_block);
    reset_state(s);
    s->max_unchecked_14b_blocks = ORC_MAX_BURST_UNCH_14BIT; s->max_unchecked_16b_blocks = ORC_MAX_BURST_UNCH_16BIT;
    s->ignore_CRC = false; s->fix_cut_above = false; s->mask_seams = true; s->broken_mask_dur = ORC_UNCH_MASK_DURATION;
    /* doFrameReassemble prologue :7257-7258 */
    s->broken_countdown = 0;
    s->frasm_f0.video_standard = s->frasm_f1.video_standard = s->frasm_f2.video_standard = ORC_VID_UNKNOWN;
}
void orc_stitcher_free(orc_stitcher *s)
{
    free(s->trim_buf); free(s->frame1_even); free(s->frame1_odd); free(s->frame2_even); free(s->frame2_odd);
    free(s->in_lines.v); free(s->padding_queue.v); free(s->conv_queue.v); free(s->out); free(s->frames); free(s->blocks); free(s->asm_lines); free(s->asm_frame_n);
    memset(s, 0, sizeof(*s));
}
void orc_stitcher_push_line(orc_stitcher *s, const orc_stc_line *l) { dq_push_back(&s->in_lines, l); }

/* ------------------------------------------------------------------ waitForTwoFrames / fillUntilTwoFrames */
static bool wait_for_two_frames(orc_stitcher *s)   /* :92-186 */
{
    bool f1 = false, f2 = false;
    frasm_clear(&s->frasm_f1); frasm_clear(&s->frasm_f2);
    for (size_t i = 0; i < s->in_lines.n; i++) {
        orc_stc_line *l = dq_at(&s->in_lines, i);
        if (!f1) { if (l->service_type == ORC_SRV_END_FRAME) { f1 = true; s->frasm_f1.frame_number = l->frame_number; } }
        else if (!f2) { if (l->service_type == ORC_SRV_END_FRAME) { f2 = true; s->frasm_f2.frame_number = l->frame_number; } }
        if (f1 && f2) return true;
    }
    return false;
}
static void fill_until_two_frames(orc_stitcher *s)   /* :189-256 */
{
    uint8_t frames_cnt = 0;
    s->trim_fill = 0;
    for (size_t i = 0; i < s->in_lines.n; i++) {
        orc_stc_line *l = dq_at(&s->in_lines, i);
        if (l->service_type == ORC_SRV_END_FRAME) { frames_cnt++; if (frames_cnt >= 2) break; }
        else if (s->trim_fill < ORC_BUF_SIZE_TRIM) { s->trim_buf[s->trim_fill] = *l; s->trim_fill++; }
    }
}

/* ------------------------------------------------------------------ findFramesTrim (:259-734) */
static int8_t ctrl_index(const orc_stc_line *l) { return (int8_t)((l->words[5] >> 8) & 0x3F); }     /* stc007line.cpp:375-388 */
static int8_t ctrl_hour(const orc_stc_line *l) { return (int8_t)((l->words[5] >> 4) & 0x0F); }
static int8_t ctrl_minute(const orc_stc_line *l) { uint16_t t = (uint16_t)((l->words[6] >> 12) & 0x03); t = (uint16_t)(t + ((l->words[5] & 0x0F) << 2)); return (int8_t)t; }
static int8_t ctrl_second(const orc_stc_line *l) { return (int8_t)((l->words[6] >> 6) & 0x3F); }
static int8_t ctrl_field(const orc_stc_line *l) { return (int8_t)(l->words[6] & 0x3F); }

static void trim_track(const orc_stc_line *l, bool skip_bad, bool *top, bool *bottom, uint16_t *top_data, uint16_t *bottom_data)
{
    if ((!skip_bad && orc_stc_has_markers(l)) || (!skip_bad && orc_stc_crc_valid_ignore_forced(l)) || (skip_bad && orc_stc_crc_valid_ignore_forced(l))) {
        if (!*top) { *top_data = l->line_number; *top = true; }
        *bottom_data = l->line_number; *bottom = true;
    }
}
static void find_frames_trim(orc_stitcher *s)
{
    uint16_t f1o_good = 0, f1e_good = 0, f2o_good = 0, f2e_good = 0;
    bool f1e_top = false, f1e_bottom = false, f1o_top = false, f1o_bottom = false, f2e_top = false, f2e_bottom = false, f2o_top = false, f2o_bottom = false;
    bool f1o_skip = false, f1e_skip = false, f2o_skip = false, f2e_skip = false;
    orc_frasm *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    if (f1->trim_ok) f1e_top = f1o_top = f1e_bottom = f1o_bottom = true;
    else f1->even_top_data = f1->even_bottom_data = f1->odd_top_data = f1->odd_bottom_data = 0;
    if (f2->trim_ok) f2e_top = f2o_top = f2e_bottom = f2o_bottom = true;
    else f2->even_top_data = f2->even_bottom_data = f2->odd_top_data = f2->odd_bottom_data = 0;
    for (uint16_t i = 0; i < s->trim_fill; i++) {
        const orc_stc_line *l = &s->trim_buf[i];
        if (l->frame_number == f1->frame_number) {
            if (!ln_service(l)) {
                if (orc_stc_crc_valid(l)) {
                    if ((l->line_number % 2) == 0) { f1e_good++; if (f1e_good > ORC_MIN_GOOD_LINES_PF) f1e_skip = true; }
                    else { f1o_good++; if (f1o_good > ORC_MIN_GOOD_LINES_PF) f1o_skip = true; }
                }
            } else if (l->service_type == ORC_SRV_NEW_FILE) s->file_start = true;
            else if (l->service_type == ORC_SRV_END_FILE) s->file_end = true;
            else if (l->service_type == ORC_SRV_CTRL_BLOCK) {
                if (f1e_good == 0 && f1o_good == 0) { f1->ctrl_index = ctrl_index(l); f1->ctrl_hour = ctrl_hour(l); f1->ctrl_minute = ctrl_minute(l); f1->ctrl_second = ctrl_second(l); f1->ctrl_field = ctrl_field(l); }
            }
        } else if (l->frame_number == f2->frame_number) {
            if (!ln_service(l)) {
                if (orc_stc_crc_valid(l)) {
                    if ((l->line_number % 2) == 0) { f2e_good++; if (f2e_good > ORC_MIN_GOOD_LINES_PF) f2e_skip = true; }
                    else { f2o_good++; if (f2o_good > ORC_MIN_GOOD_LINES_PF) f2o_skip = true; }
                }
            } else if (l->service_type == ORC_SRV_END_FILE) s->file_end = true;
            else if (l->service_type == ORC_SRV_CTRL_BLOCK) {
                if (f2e_good == 0 && f2o_good == 0) { f2->ctrl_index = ctrl_index(l); f2->ctrl_hour = ctrl_hour(l); f2->ctrl_minute = ctrl_minute(l); f2->ctrl_second = ctrl_second(l); f2->ctrl_field = ctrl_field(l); }
            }
        }
    }
    for (uint16_t i = 0; i < s->trim_fill; i++) {
        const orc_stc_line *l = &s->trim_buf[i];
        if (f1->trim_ok && f2->trim_ok) break;
        if (ln_service(l) && l->service_type != ORC_SRV_FILLER) continue;
        if (l->frame_number == f1->frame_number && !f1->trim_ok) {
            if ((l->line_number % 2) == 0) trim_track(l, f1e_skip, &f1e_top, &f1e_bottom, &f1->even_top_data, &f1->even_bottom_data);
            else trim_track(l, f1o_skip, &f1o_top, &f1o_bottom, &f1->odd_top_data, &f1->odd_bottom_data);
        } else if (l->frame_number == f2->frame_number && !f2->trim_ok) {
            if ((l->line_number % 2) == 0) trim_track(l, f2e_skip, &f2e_top, &f2e_bottom, &f2->even_top_data, &f2->even_bottom_data);
            else trim_track(l, f2o_skip, &f2o_top, &f2o_bottom, &f2->odd_top_data, &f2->odd_bottom_data);
        }
    }
    if (f1e_top && f1o_top && f1e_bottom && f1o_bottom) if (!f1->trim_ok) f1->trim_ok = true;
    if (f2e_top && f2o_top && f2e_bottom && f2o_bottom) if (!f2->trim_ok) f2->trim_ok = true;
}

/* ------------------------------------------------------------------ splitFramesToFields (:737-985) */
static void split_frames_to_fields(orc_stitcher *s)
{
    uint32_t ref_odd = 0, ref_even = 0, ref_odd_bad = 0, ref_even_bad = 0;
    orc_frasm *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    s->f1_max_line = s->f2_max_line = 0;
    f1->odd_data_lines = f1->even_data_lines = f2->odd_data_lines = f2->even_data_lines = 0;
    f1->odd_valid_lines = f1->even_valid_lines = f2->odd_valid_lines = f2->even_valid_lines = 0;
    frasm_clear_asm_stats(f1);
    for (uint16_t i = 0; i < s->trim_fill; i++) {
        const orc_stc_line *l = &s->trim_buf[i];
        uint16_t line_num = l->line_number;
        if (ln_service(l)) if (l->service_type != ORC_SRV_FILLER) continue;
        if (l->frame_number == f1->frame_number) {
            if (s->f1_max_line < line_num) s->f1_max_line = line_num;
            if ((line_num % 2) == 0) {
                if ((f1->even_top_data != f1->even_bottom_data) || (f1->even_top_data != 0))
                    if (line_num >= f1->even_top_data && line_num <= f1->even_bottom_data)
                        if (f1->even_data_lines < ORC_BUF_SIZE_FIELD) {
                            s->frame1_even[f1->even_data_lines] = *l; f1->even_data_lines++;
                            ref_even_bad += l->ref_level;
                            if (orc_stc_crc_valid(l)) { f1->even_valid_lines++; ref_even += l->ref_level; }
                        }
            } else if (line_num >= f1->odd_top_data && line_num <= f1->odd_bottom_data) {
                if (f1->odd_data_lines < ORC_BUF_SIZE_FIELD) {
                    s->frame1_odd[f1->odd_data_lines] = *l; f1->odd_data_lines++;
                    ref_odd_bad += l->ref_level;
                    if (orc_stc_crc_valid(l)) { f1->odd_valid_lines++; ref_odd += l->ref_level; }
                }
            }
        } else if (l->frame_number == f2->frame_number) {
            if (s->f2_max_line < line_num) s->f2_max_line = line_num;
            if ((line_num % 2) == 0) {
                if ((f2->even_top_data != f2->even_bottom_data) || (f2->even_top_data != 0))
                    if (line_num >= f2->even_top_data && line_num <= f2->even_bottom_data)
                        if (f2->even_data_lines < ORC_BUF_SIZE_FIELD) {
                            s->frame2_even[f2->even_data_lines] = *l; f2->even_data_lines++;
                            if (orc_stc_crc_valid(l)) f2->even_valid_lines++;
                        }
            } else if (line_num >= f2->odd_top_data && line_num <= f2->odd_bottom_data) {
                if (f2->odd_data_lines < ORC_BUF_SIZE_FIELD) {
                    s->frame2_odd[f2->odd_data_lines] = *l; f2->odd_data_lines++;
                    if (orc_stc_crc_valid(l)) f2->odd_valid_lines++;
                }
            }
        }
    }
    if (f1->odd_valid_lines > 0) f1->odd_ref = (uint8_t)(ref_odd / f1->odd_valid_lines);
    else if (f1->odd_data_lines > 0) f1->odd_ref = (uint8_t)(ref_odd_bad / f1->odd_data_lines);
    else f1->odd_ref = 0;
    if (f1->even_valid_lines > 0) f1->even_ref = (uint8_t)(ref_even / f1->even_valid_lines);
    else if (f1->even_data_lines > 0) f1->even_ref = (uint8_t)(ref_even_bad / f1->even_data_lines);
    else f1->even_ref = 0;
}

/* ------------------------------------------------------------------ resolution helpers (:996-1414) */
static uint8_t get_field_resolution(orc_stitcher *s, const orc_stc_line *field, uint16_t f_size)
{
    uint16_t test_size = 0, res14 = 0, res16 = 0;
    if (s->preset_audio_res == ORC_SAMPLE_RES_14BIT) return ORC_SAMPLE_RES_14BIT;
    if (s->preset_audio_res == ORC_SAMPLE_RES_16BIT) return ORC_SAMPLE_RES_16BIT;
    if (ORC_BUF_SIZE_FIELD < f_size) return ORC_SAMPLE_RES_UNKNOWN;
    if (f_size > ORC_MIN_DEINT_DATA) test_size = (uint16_t)(f_size - ORC_MIN_DEINT_DATA); else return ORC_SAMPLE_RES_UNKNOWN;
    orc_deint *pc = &s->pad_checker;
    pc->ignore_crc = false; pc->force_ecc_check = true;
    orc_deint_set_p(pc, true); orc_deint_set_q(pc, false);
    for (uint16_t i = 0; i < test_size; i++) {
        /* the field buffers are vectors of fixed size BUF_SIZE_FIELD (:13-16): that is the size processBlock sees */
        pc->data_res_mode = ORC_RES_MODE_14BIT;
        orc_deint_process_block(pc, field, ORC_BUF_SIZE_FIELD, i, &s->padding_block);
        s->padding_block.m2_format = s->mode_m2;
        if (orc_block_is_valid(&s->padding_block) && orc_block_can_force_check(&s->padding_block) && !orc_block_is_silent(&s->padding_block)) res14++;
        else if (s->padding_block.audio_state == ORC_AUD_BROKEN && res14 > 0) res14--;
        pc->data_res_mode = ORC_RES_MODE_16BIT;
        orc_deint_process_block(pc, field, ORC_BUF_SIZE_FIELD, i, &s->padding_block);
        s->padding_block.m2_format = s->mode_m2;
        if (orc_block_is_valid(&s->padding_block) && orc_block_can_force_check(&s->padding_block) && !orc_block_is_silent(&s->padding_block)) res16++;
        else if (s->padding_block.audio_state == ORC_AUD_BROKEN && res16 > 0) res16--;
    }
    pc->ignore_crc = s->ignore_CRC;
    uint8_t result = ORC_SAMPLE_RES_UNKNOWN;
    if (res14 > (ORC_INTERLEAVE_OFS * 2)) {
        test_size = (uint16_t)(res16 * 128);
        test_size = test_size / res14;
        result = (test_size > 32) ? ORC_SAMPLE_RES_16BIT : ORC_SAMPLE_RES_14BIT;
    }
    return result;
}
static uint8_t res_mode_for_seam(uint8_t r1, uint8_t r2)   /* :1214-1253 */
{
    uint8_t fin = ORC_RES_MODE_16BIT_AUTO;
    if (r1 == r2) { fin = r1; if (r1 == ORC_RES_MODE_14BIT_AUTO) fin = ORC_RES_MODE_14BIT; else if (r1 == ORC_RES_MODE_16BIT_AUTO) fin = ORC_RES_MODE_16BIT; }
    else if (r1 == ORC_RES_MODE_14BIT) { if (r2 == ORC_RES_MODE_14BIT_AUTO) fin = ORC_RES_MODE_14BIT_AUTO; }
    else if (r1 == ORC_RES_MODE_14BIT_AUTO) { if (r2 == ORC_RES_MODE_14BIT) fin = ORC_RES_MODE_14BIT_AUTO; }
    else if (r1 == ORC_RES_MODE_16BIT) { if (r2 == ORC_RES_MODE_14BIT) fin = ORC_RES_MODE_14BIT_AUTO; }
    return fin;
}
static uint8_t res_for_seam(uint8_t r1, uint8_t r2)   /* :1256-1269 */
{
    uint8_t fin = res_mode_for_seam(r1, r2);
    return (fin == ORC_RES_MODE_16BIT || fin == ORC_RES_MODE_16BIT_AUTO) ? ORC_RES_16BIT : ORC_RES_14BIT;
}
static uint8_t line_res(const orc_stitcher *s, const orc_stc_line *l, uint8_t dflt)
{
    bool even = (l->line_number % 2) == 0;
    if (l->frame_number == s->frasm_f2.frame_number) return even ? s->frasm_f2.even_resolution : s->frasm_f2.odd_resolution;
    if (l->frame_number == s->frasm_f1.frame_number) return even ? s->frasm_f1.even_resolution : s->frasm_f1.odd_resolution;
    if (l->frame_number == s->frasm_f0.frame_number) return even ? s->frasm_f0.even_resolution : s->frasm_f0.odd_resolution;
    return dflt;
}
static uint8_t get_data_block_resolution(const orc_stitcher *s, orc_line_deque *q, uint16_t line_sh)   /* :1272-1414 */
{
    if (s->mode_m2) return ORC_RES_MODE_14BIT;
    if (q->n <= (size_t)(line_sh + ORC_MIN_DEINT_DATA)) return ORC_RES_MODE_14BIT_AUTO;
    uint8_t first = line_res(s, dq_at(q, 0 + line_sh), ORC_RES_MODE_14BIT);
    uint8_t last = line_res(s, dq_at(q, ORC_MIN_DEINT_DATA + line_sh), ORC_RES_MODE_14BIT);
    return res_mode_for_seam(first, last);
}

/* ------------------------------------------------------------------ tryPadding / findPadding (:1417-2054) */
static void stats_clear(orc_stitch_stats *t) { t->index = t->valid = 0; t->silent = t->unchecked = t->broken = 0xFF; }
static bool stats_lt(const orc_stitch_stats *a, const orc_stitch_stats *b)   /* frametrimset.cpp:312-371 */
{
    if (a->broken < b->broken) return true;
    if (a->broken == b->broken) {
        if (a->valid > b->valid) return true;
        if (a->valid == b->valid) {
            if (a->unchecked < b->unchecked) return true;
            if (a->unchecked == b->unchecked) {
                if (a->silent < b->silent) return true;
                if (a->silent == b->silent) return a->index < b->index;
            }
        }
    }
    return false;
}
static int stats_cmp(const void *a, const void *b)
{
    if (stats_lt((const orc_stitch_stats *)a, (const orc_stitch_stats *)b)) return -1;
    if (stats_lt((const orc_stitch_stats *)b, (const orc_stitch_stats *)a)) return 1;
    return 0;
}
static void make_empty_pad_line(const orc_stitcher *s, orc_stc_line *e)   /* :1479-1484 */
{
    orc_stc_clear(e);
    e->coords.data_start = e->coords.data_stop = 0; e->coords.from_doubled = e->coords.not_sure = false;
    e->m2_format = s->mode_m2;
    orc_stc_set_silent(e);
}

static uint8_t try_padding(orc_stitcher *s, const orc_stc_line *field1, uint16_t f1_size, const orc_stc_line *field2, uint16_t f2_size,
                           uint16_t padding, orc_stitch_stats *st)
{
    uint16_t line_count, line_num, valid_cnt = 0, silence_cnt = 0, uncheck_cnt = 0, broken_count = 0;
    uint16_t valid_max = 0, silence_max = 0, uncheck_max = 0;
    uint32_t frame_num;
    uint8_t unchecked_lim;
    orc_stc_line empty_line;
    if (field1 == NULL || field2 == NULL) return ORC_DS_RET_NO_DATA;
    if (ORC_BUF_SIZE_FIELD < f1_size || ORC_BUF_SIZE_FIELD < f2_size) return ORC_DS_RET_NO_DATA;
    dq_clear(&s->padding_queue);
    if ((int)f1_size > (ORC_MIN_DEINT_DATA + ORC_INTERLEAVE_OFS / 2 - (int)padding)) line_count = (uint16_t)(f1_size - (ORC_MIN_DEINT_DATA + ORC_INTERLEAVE_OFS / 2 - padding));
    else line_count = 0;
    for (uint16_t i = line_count; i < f1_size; i++) dq_push_back(&s->padding_queue, &field1[i]);
    line_num = field1[f1_size - 1].line_number;
    frame_num = field1[f1_size - 1].frame_number;
    make_empty_pad_line(s, &empty_line);
    for (uint16_t p = 0; p < padding; p++) {
        empty_line.frame_number = frame_num;
        line_num = (uint16_t)(line_num + 2);
        empty_line.line_number = line_num;
        dq_push_back(&s->padding_queue, &empty_line);
    }
    if (f2_size > (ORC_MIN_DEINT_DATA + ORC_INTERLEAVE_OFS / 2)) line_count = (ORC_MIN_DEINT_DATA + ORC_INTERLEAVE_OFS / 2);
    else line_count = f2_size;
    for (uint16_t i = 0; i < line_count; i++) dq_push_back(&s->padding_queue, &field2[i]);
    if (s->padding_queue.n < ORC_MIN_DEINT_DATA) return ORC_DS_RET_NO_DATA;
    unchecked_lim = s->max_unchecked_14b_blocks;
    if (!s->enable_Q_code) unchecked_lim = s->max_unchecked_16b_blocks;
    orc_deint *pc = &s->pad_checker;
    pc->data_res_mode = get_data_block_resolution(s, &s->padding_queue, 0);
    pc->ignore_crc = s->ignore_CRC; pc->force_ecc_check = true;
    orc_deint_set_p(pc, s->enable_P_code); orc_deint_set_q(pc, s->enable_Q_code); pc->en_cwd = false;
    size_t buf_size = 0;
    for (;;) {
        if (orc_deint_process_block(pc, dq_at(&s->padding_queue, 0), s->padding_queue.n, (uint16_t)buf_size, &s->padding_block) != ORC_DI_RET_OK) break;
        orc_stc_block *b = &s->padding_block;
        b->m2_format = s->mode_m2;
        if (orc_block_is_valid(b) && !orc_block_is_silent(b) && orc_block_can_force_check(b)) valid_cnt++;
        else if (valid_cnt > valid_max) valid_max = valid_cnt;
        if (orc_block_is_silent(b)) { silence_cnt++; if (silence_cnt >= ORC_MAX_BURST_SILENCE) valid_cnt = 0; }
        else { if (silence_cnt > silence_max) silence_max = silence_cnt; silence_cnt = 0; }
        if ((s->enable_Q_code && (!orc_block_can_force_check(b) || b->audio_state == ORC_AUD_FIX_Q)) || (!s->enable_Q_code && b->audio_state == ORC_AUD_FIX_P)) {
            uncheck_cnt++;
            if (uncheck_cnt >= unchecked_lim) valid_cnt = 0;
        } else { if (uncheck_cnt > uncheck_max) uncheck_max = uncheck_cnt; uncheck_cnt = 0; }
        if (b->audio_state == ORC_AUD_BROKEN) { broken_count++; if (broken_count >= ORC_MAX_BURST_BROKEN) valid_cnt = 0; }
        buf_size++;
    }
    dq_clear(&s->padding_queue);
    if (valid_cnt > valid_max) valid_max = valid_cnt;
    if (silence_cnt > silence_max) silence_max = silence_cnt;
    if (uncheck_cnt > uncheck_max) uncheck_max = uncheck_cnt;
    /* run_lock is only ever assigned `true` in the reference (:1570); an optimising build treats it as true */
    if (st != NULL) { st->index = padding; st->valid = valid_max; st->silent = silence_max; st->unchecked = uncheck_max; st->broken = broken_count; }
    if (broken_count >= ORC_MAX_BURST_BROKEN) return ORC_DS_RET_BROKE;
    if (silence_max > ORC_MAX_BURST_SILENCE) return ORC_DS_RET_SILENCE;
    if (uncheck_max > unchecked_lim) return ORC_DS_RET_NO_PAD;
    if (valid_max == 0) return ORC_DS_RET_NO_PAD;
    return ORC_DS_RET_OK;
}

static uint8_t find_padding(orc_stitcher *s, const orc_stc_line *field1, uint16_t f1_size, const orc_stc_line *field2, uint16_t f2_size,
                            uint8_t in_std, uint8_t in_resolution, uint16_t *padding)
{
    uint16_t pad, max_padding, min_broken;
    uint8_t unchecked_lim, no_brk_idx, stitch_res = ORC_DS_RET_NO_PAD;
    if (field1 == NULL || field2 == NULL || padding == NULL) return stitch_res;
    pad = f1_size;
    if (in_std == ORC_VID_PAL) *padding = (pad > ORC_LINES_PF_PAL) ? 0 : (uint16_t)(ORC_LINES_PF_PAL - pad);
    else if (in_std == ORC_VID_NTSC) *padding = (pad > ORC_LINES_PF_NTSC) ? 0 : (uint16_t)(ORC_LINES_PF_NTSC - pad);
    else *padding = 0;
    max_padding = ORC_MAX_PADDING_14BIT; unchecked_lim = s->max_unchecked_14b_blocks;
    if (in_resolution == ORC_RES_16BIT || !s->enable_Q_code) { max_padding = ORC_MAX_PADDING_16BIT; unchecked_lim = s->max_unchecked_16b_blocks; }
    s->last_pad_counter = 0xFF;
    if (s->enable_P_code || s->enable_Q_code) {
        orc_stitch_stats sd[ORC_MAX_PADDING_14BIT];
        for (int i = 0; i < max_padding; i++) stats_clear(&sd[i]);
        min_broken = 0xFFFF; no_brk_idx = 0;
        for (pad = 0; pad < max_padding; pad++) {
            try_padding(s, field1, f1_size, field2, f2_size, pad, &sd[pad]);
            if (min_broken > sd[pad].broken) { min_broken = sd[pad].broken; if (min_broken == 0) no_brk_idx = (uint8_t)pad; }
            else if (min_broken == 0) {
                if (sd[no_brk_idx].valid > 0 && sd[no_brk_idx].unchecked < unchecked_lim && sd[pad].broken > 0) break;
            }
        }
        qsort(sd, max_padding, sizeof(sd[0]), stats_cmp);
        s->last_pad_counter = (uint8_t)sd[0].broken;
        if (sd[0].silent < ORC_MAX_BURST_SILENCE) {
            if (sd[0].unchecked < unchecked_lim) {
                if (sd[0].broken < 2 && sd[0].broken < sd[1].broken) { stitch_res = ORC_DS_RET_OK; *padding = sd[0].index; }
                else if ((((int16_t)sd[0].valid - (int16_t)sd[1].valid) > ORC_MAX_BURST_UNCH_DELTA) && sd[0].broken == 0) { stitch_res = ORC_DS_RET_OK; *padding = sd[0].index; }
            } else {
                for (pad = 0; pad < max_padding; pad++) { sd[pad].broken = min_broken; if (sd[pad].unchecked >= unchecked_lim) sd[pad].broken = 0xFF; }
                qsort(sd, max_padding, sizeof(sd[0]), stats_cmp);
                if (sd[0].unchecked < unchecked_lim)
                    if (((int16_t)sd[0].valid - (int16_t)sd[1].valid) > ORC_MAX_BURST_UNCH_DELTA) { stitch_res = ORC_DS_RET_OK; *padding = sd[0].index; }
            }
        } else stitch_res = ORC_DS_RET_SILENCE;
    }
    return stitch_res;
}

/* ------------------------------------------------------------------ stats (:2057-2198) */
static uint8_t probable_field_order(const orc_stitcher *s)
{
    uint8_t t = 0, b = 0;
    for (uint8_t i = 0; i < ORC_STATS_DEPTH; i++) { uint8_t v = circ_get(&s->stats_field_order, i); if (v == ORC_ORDER_TFF) t++; else if (v == ORC_ORDER_BFF) b++; }
    if (t > 0 || b > 0) return (t < b) ? ORC_ORDER_BFF : ORC_ORDER_TFF;
    return ORC_ORDER_UNK;
}
static uint8_t probable_resolution(const orc_stitcher *s)
{
    uint8_t c14 = 0, c16 = 0;
    for (uint8_t i = 0; i < ORC_STATS_DEPTH; i++) { uint8_t v = circ_get(&s->stats_resolution, i); if (v == ORC_SAMPLE_RES_14BIT) c14++; if (v == ORC_SAMPLE_RES_16BIT) c16++; }
    if (c14 > 0 || c16 > 0) return (c14 < c16) ? ORC_SAMPLE_RES_16BIT : ORC_SAMPLE_RES_14BIT;
    return ORC_SAMPLE_RES_UNKNOWN;
}

/* ------------------------------------------------------------------ detectAudioResolution (:2207-2763) */
static void set_pair(uint8_t known_res, uint8_t *known, uint8_t *other)
{
    /* one field detected, its sibling unknown: fixed mode for the known one, AUTO of the same kind for the other */
    if (known_res == ORC_SAMPLE_RES_16BIT) { *known = ORC_RES_MODE_16BIT; *other = ORC_RES_MODE_16BIT_AUTO; }
    else { *known = ORC_RES_MODE_14BIT; *other = ORC_RES_MODE_14BIT_AUTO; }
}
static void detect_audio_resolution(orc_stitcher *s)
{
    orc_frasm *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    if (s->mode_m2) { f1->odd_resolution = f1->even_resolution = ORC_RES_MODE_14BIT; f2->odd_resolution = f2->even_resolution = ORC_RES_MODE_14BIT; return; }
    uint8_t f1o = get_field_resolution(s, s->frame1_odd, f1->odd_data_lines);
    uint8_t f1e = get_field_resolution(s, s->frame1_even, f1->even_data_lines);
    uint8_t f2o = get_field_resolution(s, s->frame2_odd, f2->odd_data_lines);
    uint8_t f2e = get_field_resolution(s, s->frame2_even, f2->even_data_lines);
    if (f1o == ORC_SAMPLE_RES_14BIT || f1o == ORC_SAMPLE_RES_16BIT) circ_push(&s->stats_resolution, f1o);
    if (f1e == ORC_SAMPLE_RES_14BIT || f1e == ORC_SAMPLE_RES_16BIT) circ_push(&s->stats_resolution, f1e);
    if (f1o == ORC_SAMPLE_RES_UNKNOWN && f1e == ORC_SAMPLE_RES_UNKNOWN) {
        if (f2o == ORC_SAMPLE_RES_UNKNOWN && f2e == ORC_SAMPLE_RES_UNKNOWN) {
            uint8_t m = (probable_resolution(s) == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT_AUTO : ORC_RES_MODE_14BIT_AUTO;
            f1->odd_resolution = f1->even_resolution = f2->odd_resolution = f2->even_resolution = m;
        } else if (f2o == ORC_SAMPLE_RES_UNKNOWN) {
            if (f2e == ORC_SAMPLE_RES_16BIT) { f2->even_resolution = ORC_RES_MODE_16BIT; f1->odd_resolution = f1->even_resolution = f2->odd_resolution = ORC_RES_MODE_16BIT_AUTO; }
            else { f2->even_resolution = ORC_RES_MODE_14BIT; f1->odd_resolution = f1->even_resolution = f2->odd_resolution = ORC_RES_MODE_14BIT_AUTO; }
        } else if (f2e == ORC_SAMPLE_RES_UNKNOWN) {
            if (f2o == ORC_SAMPLE_RES_16BIT) { f2->odd_resolution = ORC_RES_MODE_16BIT; f1->odd_resolution = f1->even_resolution = f2->even_resolution = ORC_RES_MODE_16BIT_AUTO; }
            else { f2->odd_resolution = ORC_RES_MODE_14BIT; f1->odd_resolution = f1->even_resolution = f2->even_resolution = ORC_RES_MODE_14BIT_AUTO; }
        } else {
            if (f2o == f2e && f2o == ORC_SAMPLE_RES_16BIT) { f2->odd_resolution = f2->even_resolution = ORC_RES_MODE_16BIT; f1->odd_resolution = f1->even_resolution = ORC_RES_MODE_16BIT_AUTO; }
            else {
                f2->odd_resolution = (f2o == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
                f2->even_resolution = (f2e == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
                f1->odd_resolution = f1->even_resolution = ORC_RES_MODE_14BIT_AUTO;
            }
        }
    } else {
        if (f1o == ORC_SAMPLE_RES_UNKNOWN) set_pair(f1e, &f1->even_resolution, &f1->odd_resolution);
        else if (f1e == ORC_SAMPLE_RES_UNKNOWN) set_pair(f1o, &f1->odd_resolution, &f1->even_resolution);
        else {
            f1->odd_resolution = (f1o == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
            f1->even_resolution = (f1e == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
        }
        if (f2o == ORC_SAMPLE_RES_UNKNOWN && f2e == ORC_SAMPLE_RES_UNKNOWN) {
            uint8_t m = (probable_resolution(s) == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT_AUTO : ORC_RES_MODE_14BIT_AUTO;
            f2->odd_resolution = f2->even_resolution = m;
        } else if (f2o == ORC_SAMPLE_RES_UNKNOWN) set_pair(f2e, &f2->even_resolution, &f2->odd_resolution);
        else if (f2e == ORC_SAMPLE_RES_UNKNOWN) set_pair(f2o, &f2->odd_resolution, &f2->even_resolution);
        else {
            f2->odd_resolution = (f2o == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
            f2->even_resolution = (f2e == ORC_SAMPLE_RES_16BIT) ? ORC_RES_MODE_16BIT : ORC_RES_MODE_14BIT;
        }
    }
}

/* ------------------------------------------------------------------ detectVideoStandard (:2773-2925) */
static void detect_video_standard(orc_stitcher *s)
{
    orc_frasm *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    f1->video_standard = ORC_VID_UNKNOWN;
    f1->odd_std_lines = f1->even_std_lines = 0;
    if (s->preset_video_mode == ORC_VID_UNKNOWN) {
        f1->vid_std_preset = false;
        if (f1->odd_data_lines > ORC_LINES_PF_MAX_PAL || f1->even_data_lines > ORC_LINES_PF_MAX_PAL || f2->odd_data_lines > ORC_LINES_PF_MAX_PAL || f2->even_data_lines > ORC_LINES_PF_MAX_PAL)
            f1->video_standard = ORC_VID_UNKNOWN;
        else if (f1->odd_data_lines > ORC_LINES_PF_MAX_NTSC || f1->even_data_lines > ORC_LINES_PF_MAX_NTSC || f2->odd_data_lines > ORC_LINES_PF_MAX_NTSC || f2->even_data_lines > ORC_LINES_PF_MAX_NTSC)
            f1->video_standard = ORC_VID_PAL;
        else f1->video_standard = (s->f1_max_line <= ((ORC_LINES_PF_PAL - ORC_INTERLEAVE_OFS) * 2)) ? ORC_VID_NTSC : ORC_VID_PAL;
    } else { f1->vid_std_preset = true; f1->video_standard = s->preset_video_mode; }
    if (f1->video_standard == ORC_VID_UNKNOWN) f1->video_standard = s->frasm_f0.video_standard;
    if (f1->video_standard == ORC_VID_NTSC) f1->odd_std_lines = f1->even_std_lines = ORC_LINES_PF_NTSC;
    else if (f1->video_standard == ORC_VID_PAL) f1->odd_std_lines = f1->even_std_lines = ORC_LINES_PF_PAL;
    if (s->preset_field_order == ORC_ORDER_TFF) { frasm_preset_tff(f1); frasm_preset_tff(f2); }
    else if (s->preset_field_order == ORC_ORDER_BFF) { frasm_preset_bff(f1); frasm_preset_bff(f2); }
    else { f2->order_preset = false; frasm_set_order_unknown(f2); }
}

/* ------------------------------------------------------------------ findFieldStitching (:2929-4275) */
enum { STG_TRY_PREVIOUS = 0, STG_TRY_TFF_TO_TFF, STG_TRY_BFF_TO_BFF, STG_A_PREPARE, STG_A_PAD_TFF, STG_A_PAD_BFF, STG_AB_UNK_PREPARE,
       STG_AB_TFF_TO_TFF, STG_AB_TFF_TO_BFF, STG_AB_BFF_TO_BFF, STG_AB_BFF_TO_TFF, STG_PAD_NO_GOOD, STG_PAD_SILENCE, STG_PAD_OK, STG_PAD_MAX };

static uint8_t find_field_stitching(orc_stitcher *s)
{
    bool en_sw_order = true;
    uint8_t proc_state = STG_TRY_PREVIOUS, stage_count = 0, stitch_resolution, f_res;
    orc_frasm *f0 = &s->frasm_f0, *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    detect_audio_resolution(s);
    detect_video_standard(s);
    do {
        stage_count++;
        if (proc_state == STG_TRY_PREVIOUS) {
            proc_state = STG_A_PREPARE;
            if (f0->odd_data_lines == f1->odd_data_lines && f0->even_data_lines == f1->even_data_lines && f0->inner_padding_ok && f0->outer_padding_ok) {
                if (!f1->order_preset || f0->field_order == f1->field_order) {
                    f1->inner_silence = f1->outer_silence = f2->inner_silence = f2->outer_silence = true;
                    f2->inner_padding_ok = f2->outer_padding_ok = false;
                    f2->inner_padding = f2->outer_padding = 0;
                    if (f1->odd_data_lines < ORC_MIN_FILL_LINES_PF && f1->even_data_lines < ORC_MIN_FILL_LINES_PF) {
                        frasm_set_order_unknown(f1);
                        f1->inner_padding_ok = f1->outer_padding_ok = false; f1->inner_padding = f1->outer_padding = 0;
                        proc_state = STG_PAD_NO_GOOD;
                    } else {
                        f_res = ORC_DS_RET_NO_PAD;
                        if (f0->field_order == ORC_ORDER_TFF) f_res = try_padding(s, s->frame1_odd, f1->odd_data_lines, s->frame1_even, f1->even_data_lines, f0->inner_padding, NULL);
                        else if (f0->field_order == ORC_ORDER_BFF) f_res = try_padding(s, s->frame1_even, f1->even_data_lines, s->frame1_odd, f1->odd_data_lines, f0->inner_padding, NULL);
                        if (f_res == ORC_DS_RET_OK) {
                            frasm_update_vid_std_soft(f1, f0->video_standard);
                            f1->field_order = f0->field_order;
                            f1->inner_padding = f0->inner_padding; f1->inner_padding_ok = true; f1->inner_silence = false;
                            if (f1->field_order == ORC_ORDER_TFF) { f1->tff_cnt = s->last_pad_counter; proc_state = STG_TRY_TFF_TO_TFF; }
                            else { f1->bff_cnt = s->last_pad_counter; proc_state = STG_TRY_BFF_TO_BFF; }
                        }
                    }
                }
            }
        } else if (proc_state == STG_TRY_TFF_TO_TFF) {
            f_res = ORC_DS_RET_NO_PAD;
            if (f2->odd_data_lines >= ORC_MIN_FILL_LINES_PF) f_res = try_padding(s, s->frame1_even, f1->even_data_lines, s->frame2_odd, f2->odd_data_lines, f0->outer_padding, NULL);
            if (f_res == ORC_DS_RET_OK) { f1->outer_padding = f0->outer_padding; f1->outer_padding_ok = true; frasm_set_order_tff(f2); f1->outer_silence = false; proc_state = STG_PAD_OK; }
            else { proc_state = STG_AB_TFF_TO_TFF; en_sw_order = false; }
        } else if (proc_state == STG_TRY_BFF_TO_BFF) {
            f_res = ORC_DS_RET_NO_PAD;
            if (f2->even_data_lines >= ORC_MIN_FILL_LINES_PF) f_res = try_padding(s, s->frame1_odd, f1->odd_data_lines, s->frame2_even, f2->even_data_lines, f0->outer_padding, NULL);
            if (f_res == ORC_DS_RET_OK) { f1->outer_padding = f0->outer_padding; f1->outer_padding_ok = true; frasm_set_order_bff(f2); f1->outer_silence = false; proc_state = STG_PAD_OK; }
            else { proc_state = STG_AB_BFF_TO_BFF; en_sw_order = false; }
        } else if (proc_state == STG_A_PREPARE) {
            f1->inner_padding_ok = f1->outer_padding_ok = false; f1->inner_padding = f1->outer_padding = 0; f1->tff_cnt = f1->bff_cnt = 0;
            if (f1->odd_data_lines < ORC_MIN_FILL_LINES_PF && f1->even_data_lines < ORC_MIN_FILL_LINES_PF) {
                if (!f1->order_preset) frasm_set_order_unknown(f1);
                proc_state = STG_PAD_NO_GOOD;
            } else if (f1->even_data_lines < ORC_MIN_FILL_LINES_PF) {
                if (f1->field_order == ORC_ORDER_TFF) { f1->outer_padding_ok = false; f1->outer_padding = 0; proc_state = STG_PAD_NO_GOOD; }
                else { proc_state = STG_AB_BFF_TO_BFF; en_sw_order = false; }
            } else if (f1->odd_data_lines < ORC_MIN_FILL_LINES_PF) {
                if (f1->field_order == ORC_ORDER_BFF) { f1->outer_padding_ok = false; f1->outer_padding = 0; proc_state = STG_PAD_NO_GOOD; }
                else { proc_state = STG_AB_TFF_TO_TFF; en_sw_order = false; }
            } else {
                if (f1->field_order == ORC_ORDER_BFF) { proc_state = STG_A_PAD_BFF; en_sw_order = false; }
                else if (f1->field_order == ORC_ORDER_TFF) { proc_state = STG_A_PAD_TFF; en_sw_order = false; }
                else {
                    f_res = probable_field_order(s);
                    proc_state = (f_res == ORC_ORDER_BFF) ? STG_A_PAD_BFF : STG_A_PAD_TFF;
                    en_sw_order = true;
                }
            }
        } else if (proc_state == STG_A_PAD_TFF || proc_state == STG_A_PAD_BFF) {
            bool tff = proc_state == STG_A_PAD_TFF;
            f1->inner_padding = 0;
            if (tff) {
                stitch_resolution = res_for_seam(f1->odd_resolution, f1->even_resolution);
                f_res = find_padding(s, s->frame1_odd, f1->odd_data_lines, s->frame1_even, f1->even_data_lines, f1->video_standard, stitch_resolution, &f1->inner_padding);
                f1->tff_cnt = s->last_pad_counter;
            } else {
                stitch_resolution = res_for_seam(f1->even_resolution, f1->odd_resolution);
                f_res = find_padding(s, s->frame1_even, f1->even_data_lines, s->frame1_odd, f1->odd_data_lines, f1->video_standard, stitch_resolution, &f1->inner_padding);
                f1->bff_cnt = s->last_pad_counter;
            }
            f1->inner_silence = false;
            if (f_res == ORC_DS_RET_OK) {
                if (tff) frasm_set_order_tff(f1); else frasm_set_order_bff(f1);
                f1->inner_padding_ok = true;
                proc_state = tff ? STG_AB_TFF_TO_TFF : STG_AB_BFF_TO_BFF; en_sw_order = false;
            } else if (f_res == ORC_DS_RET_SILENCE) {
                f1->inner_silence = true; f1->outer_silence = true; f1->inner_padding_ok = false; f1->inner_padding = 0;
                proc_state = STG_PAD_SILENCE;
            } else {
                f1->inner_padding = 0;
                if ((tff && f1->field_order == ORC_ORDER_TFF) || (!tff && f1->field_order == ORC_ORDER_BFF)) {
                    f1->inner_padding_ok = false;
                    proc_state = tff ? STG_AB_TFF_TO_TFF : STG_AB_BFF_TO_BFF; en_sw_order = false;
                } else if (en_sw_order) { proc_state = tff ? STG_A_PAD_BFF : STG_A_PAD_TFF; en_sw_order = false; }
                else proc_state = STG_AB_UNK_PREPARE;
            }
        } else if (proc_state == STG_AB_UNK_PREPARE) {
            f1->inner_padding = 0; f1->inner_padding_ok = false; frasm_set_order_unknown(f1);
            f_res = probable_field_order(s);
            proc_state = (f_res == ORC_ORDER_BFF) ? STG_AB_BFF_TO_BFF : STG_AB_TFF_TO_TFF;
            en_sw_order = true;
        } else if (proc_state == STG_AB_TFF_TO_TFF || proc_state == STG_AB_BFF_TO_BFF) {
            bool tt = proc_state == STG_AB_TFF_TO_TFF;
            uint16_t need = tt ? f2->odd_data_lines : f2->even_data_lines;      /* first field of frame B in this order */
            uint16_t other = tt ? f2->even_data_lines : f2->odd_data_lines;
            if (f2->odd_data_lines < ORC_MIN_FILL_LINES_PF && f2->even_data_lines < ORC_MIN_FILL_LINES_PF) {
                f1->outer_padding = 0; f1->outer_padding_ok = false; f2->inner_padding_ok = false; proc_state = STG_PAD_NO_GOOD;
            } else if (need < ORC_MIN_FILL_LINES_PF) {
                if (!f1->order_preset) proc_state = tt ? STG_AB_TFF_TO_BFF : STG_AB_BFF_TO_TFF;
                else { f1->outer_padding = 0; f1->outer_padding_ok = false; f2->inner_padding_ok = false; proc_state = STG_PAD_NO_GOOD; }
            } else {
                if (tt) {
                    stitch_resolution = res_for_seam(f1->even_resolution, f2->odd_resolution);
                    f_res = find_padding(s, s->frame1_even, f1->even_data_lines, s->frame2_odd, f2->odd_data_lines, f1->video_standard, stitch_resolution, &f1->outer_padding);
                } else {
                    stitch_resolution = res_for_seam(f1->odd_resolution, f2->even_resolution);
                    f_res = find_padding(s, s->frame1_odd, f1->odd_data_lines, s->frame2_even, f2->even_data_lines, f1->video_standard, stitch_resolution, &f1->outer_padding);
                }
                f1->outer_silence = false;
                if (f_res == ORC_DS_RET_OK) {
                    f1->outer_padding_ok = true;
                    if (tt) frasm_set_order_tff(f2); else frasm_set_order_bff(f2);
                    proc_state = STG_PAD_OK;
                    if (!frasm_order_set(f1)) { if (tt) frasm_set_order_tff(f1); else frasm_set_order_bff(f1); }
                    else if ((tt && f1->field_order == ORC_ORDER_BFF) || (!tt && f1->field_order == ORC_ORDER_TFF)) { f1->outer_padding_ok = false; proc_state = STG_PAD_NO_GOOD; }
                } else if (f_res == ORC_DS_RET_SILENCE) {
                    f1->outer_silence = true; f1->outer_padding = 0; f1->outer_padding_ok = false; proc_state = STG_PAD_SILENCE;
                } else {
                    if (other < ORC_MIN_FILL_LINES_PF) { f1->outer_padding = 0; f1->outer_padding_ok = false; f2->inner_padding_ok = false; proc_state = STG_PAD_NO_GOOD; }
                    else if (!f1->order_preset) proc_state = tt ? STG_AB_TFF_TO_BFF : STG_AB_BFF_TO_TFF;
                    else { f1->outer_padding = 0; f1->outer_padding_ok = false; proc_state = STG_PAD_NO_GOOD; }
                }
            }
        } else if (proc_state == STG_AB_TFF_TO_BFF || proc_state == STG_AB_BFF_TO_TFF) {
            bool tb = proc_state == STG_AB_TFF_TO_BFF;
            if (tb) {
                stitch_resolution = res_for_seam(f1->even_resolution, f2->even_resolution);
                f_res = find_padding(s, s->frame1_even, f1->even_data_lines, s->frame2_even, f2->even_data_lines, f1->video_standard, stitch_resolution, &f1->outer_padding);
            } else {
                stitch_resolution = res_for_seam(f1->odd_resolution, f2->odd_resolution);
                f_res = find_padding(s, s->frame1_odd, f1->odd_data_lines, s->frame2_odd, f2->odd_data_lines, f1->video_standard, stitch_resolution, &f1->outer_padding);
            }
            f1->outer_silence = false;
            if (f_res == ORC_DS_RET_OK) {
                f1->outer_padding_ok = true;
                if (tb) frasm_set_order_bff(f2); else frasm_set_order_tff(f2);
                proc_state = STG_PAD_OK;
                if (!frasm_order_set(f1)) { if (tb) frasm_set_order_tff(f1); else frasm_set_order_bff(f1); }
                else if ((tb && f1->field_order == ORC_ORDER_BFF) || (!tb && f1->field_order == ORC_ORDER_TFF)) { f1->outer_padding_ok = false; proc_state = STG_PAD_NO_GOOD; }
            } else if (f_res == ORC_DS_RET_SILENCE) {
                f1->outer_silence = true; f1->outer_padding = 0; f1->outer_padding_ok = false; f2->inner_padding_ok = false; proc_state = STG_PAD_SILENCE;
            } else {
                f1->outer_padding = 0; f1->outer_padding_ok = false; f2->inner_padding_ok = false;
                if (en_sw_order && f1->even_data_lines >= ORC_MIN_FILL_LINES_PF) { proc_state = tb ? STG_AB_BFF_TO_BFF : STG_AB_TFF_TO_TFF; en_sw_order = false; }
                else proc_state = STG_PAD_NO_GOOD;
            }
        } else break;      /* STG_PAD_OK / STG_PAD_SILENCE / STG_PAD_NO_GOOD */
        if (stage_count > STG_PAD_MAX) return ORC_DS_RET_NO_PAD;
    } while (1);
    if (proc_state == STG_PAD_OK) return ORC_DS_RET_OK;
    if (proc_state == STG_PAD_SILENCE) return ORC_DS_RET_SILENCE;
    return ORC_DS_RET_NO_PAD;
}

/* ------------------------------------------------------------------ assembly helpers (:4278-4585) */
static uint8_t get_assembly_field_order(orc_stitcher *s)
{
    orc_frasm *f0 = &s->frasm_f0, *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    uint8_t cur = ORC_ORDER_UNK;
    if (frasm_order_set(f1)) { cur = f1->field_order; if (!f1->order_preset) circ_push(&s->stats_field_order, cur); }
    else {
        if (f2->order_preset && frasm_order_set(f2)) cur = f2->field_order;
        else if (frasm_order_set(f0) && f0->outer_padding_ok) cur = f0->field_order;
    }
    if (cur != ORC_ORDER_TFF && cur != ORC_ORDER_BFF) {
        uint8_t last_good = probable_field_order(s);
        if (last_good == ORC_ORDER_TFF || last_good == ORC_ORDER_BFF) cur = last_good;
        else if (f1->tff_cnt < f1->bff_cnt) cur = ORC_ORDER_TFF;
        else if (f1->tff_cnt > f1->bff_cnt) cur = ORC_ORDER_BFF;
        else cur = ORC_ORDER_TFF;         /* FLD_ORDER_DEFAULT */
    }
    if (!frasm_order_set(f1)) { f1->field_order = cur; if (!f1->order_preset) f1->order_guessed = true; }
    return cur;
}
static uint16_t first_field_line_num(uint8_t order) { return order == ORC_ORDER_TFF ? 1 : 2; }
static uint16_t second_field_line_num(uint8_t order) { return order == ORC_ORDER_TFF ? 2 : 1; }

static uint16_t add_lines_from_field(orc_stitcher *s, const orc_stc_line *field, uint16_t ind_start, uint16_t count, uint16_t *last_line_num)   /* :4452-4518 */
{
    uint16_t cnt = 0;
    if (ORC_BUF_SIZE_FIELD >= ind_start && ORC_BUF_SIZE_FIELD >= (ind_start + count)) {
        for (uint16_t i = ind_start; i < (ind_start + count); i++) {
            dq_push_back(&s->conv_queue, &field[i]);
            if (last_line_num) { *last_line_num = field[i].line_number; *last_line_num = (uint16_t)(*last_line_num + 2); }
            cnt++;
        }
    }
    return cnt;
}
static uint16_t add_field_padding(orc_stitcher *s, uint32_t in_frame, uint16_t line_cnt, uint16_t *last_line_num)   /* :4521-4571 */
{
    uint16_t cnt = 0;
    for (uint16_t i = 0; i < line_cnt; i++) {
        orc_stc_line e; orc_stc_clear(&e);
        e.frame_number = in_frame;
        if (last_line_num) { e.line_number = *last_line_num; *last_line_num = (uint16_t)(*last_line_num + 2); }
        dq_push_back(&s->conv_queue, &e);
        cnt++;
    }
    return cnt;
}
static bool is_block_no_report(const orc_stitcher *s, const orc_stc_block *b)
{
    return (s->file_start && b->w_frame[0] == s->frasm_f0.frame_number) || (s->file_end && b->w_frame[7] == s->frasm_f2.frame_number);
}

/* ------------------------------------------------------------------ fillFrameForOutput (:4588-5387) */
static void fill_frame_for_output(orc_stitcher *s)
{
    orc_frasm *f0 = &s->frasm_f0, *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    uint16_t c1, c2, last_line = 0, lines_to_fill = 0, added_inner = 0, added_outer = 0;
    const orc_stc_line *p1, *p2;
    uint8_t order = get_assembly_field_order(s);
    if (order == ORC_ORDER_TFF) {
        c1 = f1->odd_data_lines; c2 = f1->even_data_lines; p1 = s->frame1_odd; p2 = s->frame1_even;
        if (frasm_order_set(f0) && f0->field_order != ORC_ORDER_TFF) f0->outer_padding_ok = false;
    } else {
        c1 = f1->even_data_lines; c2 = f1->odd_data_lines; p1 = s->frame1_even; p2 = s->frame1_odd;
        if (frasm_order_set(f0) && f0->field_order != ORC_ORDER_BFF) f0->outer_padding_ok = false;
    }
    int16_t target = (f1->video_standard == ORC_VID_PAL) ? ORC_LINES_PF_PAL : ORC_LINES_PF_NTSC;
    if (c1 > target) c1 = (uint16_t)target;
    if (c2 > target) c2 = (uint16_t)target;
    bool insert_top_line = s->fix_cut_above;
    const uint32_t fr = f1->frame_number;
#define FIRST()  (last_line = first_field_line_num(order))
#define SECOND() (last_line = second_field_line_num(order))
#define LINES(p, st, cnt) add_lines_from_field(s, (p), (uint16_t)(st), (uint16_t)(cnt), &last_line)
#define PAD(cnt) add_field_padding(s, fr, (uint16_t)(cnt), &last_line)
    if (s->file_start) {
        f0->frame_number = 0;
        f0->even_resolution = f0->odd_resolution = (order == ORC_ORDER_TFF) ? f1->odd_resolution : f1->even_resolution;
        last_line = (f1->video_standard == ORC_VID_PAL) ? ORC_LINES_PF_PAL : ORC_LINES_PF_NTSC;
        uint8_t add_count = 80;               /* STC007DataBlock::LINE_R2 */
        last_line = (uint16_t)((last_line * 2) - (add_count * 2));
        add_field_padding(s, 0, add_count, &last_line);
        last_line = 0;
    }
    if (f0->outer_padding_ok) {
        if (f1->inner_padding_ok) {
            if (f1->outer_padding_ok) {
                lines_to_fill = (uint16_t)(c1 + c2 + f1->inner_padding + f1->outer_padding);
                if ((target * 2) == lines_to_fill) {
                    FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                    SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
                } else if ((target * 2) > lines_to_fill) {
                    lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                    FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                    SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding); added_outer = (uint16_t)(added_outer + PAD(lines_to_fill));
                    f1->outer_padding_ok = false; frasm_set_order_unknown(f2);
                } else {
                    lines_to_fill = (uint16_t)(c1 + c2 + f1->inner_padding);
                    if ((target * 2) >= lines_to_fill) {
                        lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                        FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                        SECOND(); LINES(p2, 0, c2); added_outer = PAD(lines_to_fill);
                    } else {
                        lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                        FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                        SECOND(); LINES(p2, 0, c2 - lines_to_fill);
                    }
                    f1->outer_padding_ok = false; frasm_set_order_unknown(f2);
                }
            } else {
                lines_to_fill = (uint16_t)(c1 + c2 + f1->inner_padding);
                if ((target * 2) >= lines_to_fill) {
                    lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                    FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                    SECOND(); LINES(p2, 0, c2); added_outer = PAD(lines_to_fill);
                } else {
                    lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                    FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                    SECOND(); LINES(p2, 0, c2 - lines_to_fill);
                }
            }
        } else if (f1->outer_padding_ok) {
            lines_to_fill = (uint16_t)(c1 + c2 + f1->outer_padding);
            if ((target * 2) >= lines_to_fill) {
                lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                FIRST(); LINES(p1, 0, c1); added_inner = PAD(lines_to_fill);
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
            } else {
                lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                FIRST(); LINES(p1, 0, c1);
                SECOND(); LINES(p2, lines_to_fill, c2 - lines_to_fill); added_outer = PAD(f1->outer_padding);
            }
        } else {
            lines_to_fill = (uint16_t)(c1 + c2);
            if ((target * 2) >= lines_to_fill) {
                FIRST(); LINES(p1, 0, c1); added_inner = PAD(target - c1);
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(target - c2);
            } else {
                lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                FIRST(); LINES(p1, 0, c1);
                SECOND(); LINES(p2, 0, c2 - lines_to_fill);
            }
        }
    } else if (f1->inner_padding_ok) {
        if (f1->outer_padding_ok) {
            lines_to_fill = (uint16_t)(c1 + c2 + f1->inner_padding + f1->outer_padding);
            if ((target * 2) >= lines_to_fill) {
                lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                FIRST(); added_inner = PAD(lines_to_fill); LINES(p1, 0, c1); added_inner = (uint16_t)(added_inner + PAD(f1->inner_padding));
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
            } else {
                lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                FIRST(); LINES(p1, lines_to_fill, c1 - lines_to_fill); added_inner = PAD(f1->inner_padding);
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
            }
        } else {
            lines_to_fill = (uint16_t)(c1 + c2 + f1->inner_padding);
            if ((target * 2) >= lines_to_fill) {
                lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(lines_to_fill);
            } else {
                lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                FIRST(); LINES(p1, 0, c1); added_inner = PAD(f1->inner_padding);
                SECOND(); LINES(p2, 0, c2 - lines_to_fill);
            }
        }
    } else if (f1->outer_padding_ok) {
        lines_to_fill = (uint16_t)(c1 + c2 + f1->outer_padding);
        if ((target * 2) >= lines_to_fill) {
            lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
            FIRST(); LINES(p1, 0, c1); added_inner = PAD(lines_to_fill);
            SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
        } else {
            lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
            FIRST(); LINES(p1, 0, c1 - lines_to_fill);
            SECOND(); LINES(p2, 0, c2); added_outer = PAD(f1->outer_padding);
        }
    } else {
        lines_to_fill = (uint16_t)(c1 + c2);
        if ((target * 2) >= lines_to_fill) {
            FIRST();
            if (insert_top_line && c1 > 0 && c2 > 0) {
                if (order == ORC_ORDER_BFF) {
                    added_outer = PAD(1); LINES(p1, 0, c1); c1++; added_inner = PAD(target - c1);
                    SECOND(); LINES(p2, 0, c2); added_outer = (uint16_t)(added_outer + PAD(target - c2));
                } else {
                    LINES(p1, 0, c1); added_inner = PAD(target - c1 + 1);
                    SECOND(); LINES(p2, 0, c2); c2++; added_outer = PAD(target - c2);
                }
            } else {
                LINES(p1, 0, c1); added_inner = PAD(target - c1);
                SECOND(); LINES(p2, 0, c2); added_outer = PAD(target - c2);
            }
        } else {
            FIRST();
            if (c1 < target) { LINES(p1, 0, c1); added_inner = PAD(target - c1); } else LINES(p1, 0, target);
            SECOND();
            if (c2 < target) { LINES(p2, 0, c2); added_outer = PAD(target - c2); } else LINES(p2, 0, target);
        }
    }
#undef FIRST
#undef SECOND
#undef LINES
#undef PAD
    if (s->file_end) { last_line = 1; add_field_padding(s, f2->frame_number, ORC_MIN_DEINT_DATA, &last_line); }
    f1->inner_padding = added_inner;
    f1->outer_padding = added_outer;
}

/* ------------------------------------------------------------------ CWD pre-scan (:5390-5456, 5905-6452) */
static bool fill_next_field_for_cwd(orc_stitcher *s)
{
    orc_frasm *f1 = &s->frasm_f1, *f2 = &s->frasm_f2;
    if (f1->outer_padding_ok && frasm_order_set(f1)) {
        uint16_t last_line = first_field_line_num(f1->field_order), cnt;
        const orc_stc_line *p;
        if (f1->field_order == ORC_ORDER_TFF) { p = s->frame2_odd; cnt = f2->odd_data_lines; } else { p = s->frame2_even; cnt = f2->even_data_lines; }
        if (cnt > ORC_MIN_DEINT_DATA) cnt = ORC_MIN_DEINT_DATA;
        add_lines_from_field(s, p, 0, cnt, &last_line);
        return true;
    }
    return false;
}
static void remove_next_field_after_cwd(orc_stitcher *s)
{
    while (s->conv_queue.n > 0 && dq_at(&s->conv_queue, s->conv_queue.n - 1)->frame_number == s->frasm_f2.frame_number) dq_pop_back(&s->conv_queue);
}
static void ln_set_word(orc_stc_line *l, int i, uint16_t w, bool valid)   /* stc007line.cpp:158-173 */
{
    l->words[i] = (i == ORC_STC_WORD_CRC) ? w : (uint16_t)(w & ORC_STC_WORD_MASK);
    l->word_crc[i] = l->word_valid[i] = valid;
}
static void cwd_after_patch(orc_stc_line *l, uint16_t *line_fix_cnt)
{
    if (orc_stc_crc_valid_ignore_forced(l)) {
        for (int w = 0; w <= ORC_STC_WORD_CRC; w++) l->word_valid[w] = true;
        (*line_fix_cnt)++;
    }
}
static uint16_t perform_cwd(orc_stitcher *s, orc_line_deque *q)
{
    orc_stc_block b;
    orc_deint *d = &s->lines_to_block;
    uint16_t line_fix_cnt = 0, buf_ofs = 0;
    static const int s_ofs[7] = { 12, 10, 8, 6, 4, 2, 0 };
    d->ignore_crc = s->ignore_CRC; d->force_ecc_check = !s->ignore_CRC;
    orc_deint_set_p(d, s->enable_P_code); orc_deint_set_q(d, s->enable_Q_code); d->en_cwd = true;
    d->data_res_mode = get_data_block_resolution(s, q, 0);
    uint16_t buf_size = (uint16_t)q->n;
    while ((int)buf_ofs < ((int)buf_size - ORC_MIN_DEINT_DATA)) {
        orc_block_clear(&b);
        if (orc_deint_process_block(d, dq_at(q, 0), q->n, buf_ofs, &b) != ORC_DI_RET_OK) break;
        uint8_t max_fixable = (!s->enable_Q_code || b.resolution == ORC_RES_16BIT) ? ORC_BLK_WORD_P0 : ORC_BLK_WORD_Q0;
        bool data_fixed = false;
        for (int i = 0; i <= 7; i++) if (!b.line_crc[i] && b.word_valid[i]) data_fixed = true;     /* isDataFixed, stc007datablock.cpp:371-384 */
        if (orc_block_is_valid(&b) && data_fixed) {
            for (uint8_t wi = 0; wi <= max_fixable; wi++) {
                if (b.line_crc[wi]) continue;
                orc_stc_line *l = dq_at(q, (size_t)buf_ofs + (size_t)wi * ORC_INTERLEAVE_OFS);
                if (!orc_stc_crc_valid_ignore_forced(l) && orc_coords_valid(&l->coords) && !l->forced_bad && l->frame_number != s->frasm_f2.frame_number) {
                    if (b.resolution == ORC_RES_14BIT) {
                        if (l->words[wi] != b.words[wi]) {
                            ln_set_word(l, wi, b.words[wi], ln_word_crc_ok(l, wi));
                            orc_stc_calc_crc(l);
                            l->word_valid[wi] = true;
                            cwd_after_patch(l, &line_fix_cnt);
                        } else l->word_valid[wi] = true;
                        if (!orc_stc_crc_valid_ignore_forced(l)) {
                            bool all_fixed = true;
                            for (int w = 0; w <= ORC_STC_WORD_Q; w++) if (!ln_word_valid(l, w)) { all_fixed = false; break; }
                            if (all_fixed) { orc_stc_calc_crc(l); l->words[ORC_STC_WORD_CRC] = l->calc_crc; l->word_valid[ORC_STC_WORD_CRC] = true; line_fix_cnt++; }
                        }
                    } else {
                        uint16_t old_word = l->words[wi], old_bitword = l->words[ORC_STC_WORD_Q];
                        uint16_t new_word = b.words[wi], new_bitword = (uint16_t)(new_word & 3);
                        new_word = (uint16_t)(new_word >> 2);
                        int ofs = s_ofs[wi];                                   /* wi <= WORD_P0 here (max_fixable in 16-bit mode) */
                        new_bitword = (uint16_t)(new_bitword << ofs);
                        old_bitword = (uint16_t)(old_bitword & (3 << ofs));
                        if (old_word != new_word) {
                            ln_set_word(l, wi, new_word, ln_word_crc_ok(l, wi));
                            orc_stc_calc_crc(l);
                            l->word_valid[wi] = true;
                            cwd_after_patch(l, &line_fix_cnt);
                        }
                        if (!orc_stc_crc_valid_ignore_forced(l)) {
                            if (old_bitword != new_bitword) {
                                old_bitword = l->words[ORC_STC_WORD_Q];
                                old_bitword = (uint16_t)(old_bitword & ~(3 << ofs));
                                ln_set_word(l, ORC_STC_WORD_Q, (uint16_t)(old_bitword | new_bitword), ln_word_crc_ok(l, ORC_STC_WORD_Q));
                                orc_stc_calc_crc(l);
                                cwd_after_patch(l, &line_fix_cnt);
                            }
                        }
                    }
                } else if (orc_stc_crc_valid(l)) {
                    if (b.resolution == ORC_RES_14BIT && l->words[wi] != b.words[wi]) l->forced_bad = true;
                }
            }
        }
        buf_ofs++;
    }
    return line_fix_cnt;
}
static void prescan_frame(orc_stitcher *s)
{
    if (s->enable_CWD) {
        bool next = fill_next_field_for_cwd(s);
        uint16_t fix_per_run;
        do { fix_per_run = perform_cwd(s, &s->conv_queue); } while (fix_per_run != 0);
        if (next) remove_next_field_after_cwd(s);
    }
}

/* ------------------------------------------------------------------ output (:6455-6672) */
static void out_push(orc_stitcher *s, const orc_sample_pair *p)
{
    if (s->out_n == s->out_cap) { s->out_cap = s->out_cap ? s->out_cap * 2 : 8192; s->out = (orc_sample_pair *)realloc(s->out, s->out_cap * sizeof(*p)); }
    s->out[s->out_n++] = *p;
}
static void pair_clear(orc_sample_pair *p) { memset(p, 0, sizeof(*p)); p->sample_rate = 44056; }
static void output_service(orc_stitcher *s, uint8_t srv) { orc_sample_pair p; pair_clear(&p); p.service_type = srv; out_push(s, &p); }
static void frames_push(orc_stitcher *s, const orc_frasm *f)
{
    if (s->frames_n == s->frames_cap) { s->frames_cap = s->frames_cap ? s->frames_cap * 2 : 256; s->frames = (orc_frasm *)realloc(s->frames, s->frames_cap * sizeof(*f)); }
    s->frames[s->frames_n++] = *f;
}
static void output_sample_pair(orc_stitcher *s, const orc_stc_block *b, uint8_t il, uint8_t ir)   /* :6525-6569 */
{
    orc_sample_pair p; pair_clear(&p);
    bool block_state, wl, wr, fl, fr;
    p.emphasis = b->emphasis;
    if (b->sample_rate < 44101) p.sample_rate = b->sample_rate;        /* setSampleRate: in_rate < SAMPLE_RATE_MAX */
    if (b->audio_state != ORC_AUD_BROKEN) {
        block_state = orc_block_is_valid(b);
        if (!block_state) fl = fr = false; else { fl = b->line_crc[il]; fr = b->line_crc[ir]; }
        wl = b->word_valid[il]; wr = b->word_valid[ir];
    } else { block_state = false; wl = wr = false; fl = fr = false; }
    p.audio_word[0] = orc_block_get_sample(b, il); p.audio_word[1] = orc_block_get_sample(b, ir);
    p.data_block_ok[0] = p.data_block_ok[1] = block_state;
    p.word_valid[0] = wl; p.word_valid[1] = wr; p.word_fixed[0] = fl; p.word_fixed[1] = fr;
    out_push(s, &p);
}

/* ------------------------------------------------------------------ performDeinterleave (:6675-6885) */
static void perform_deinterleave(orc_stitcher *s)
{
    orc_stc_block b;
    orc_deint *d = &s->lines_to_block;
    orc_frasm *f0 = &s->frasm_f0, *f1 = &s->frasm_f1;
    d->ignore_crc = s->ignore_CRC; d->force_ecc_check = !s->ignore_CRC;
    orc_deint_set_p(d, s->enable_P_code); orc_deint_set_q(d, s->enable_Q_code); d->en_cwd = s->enable_CWD;
    if (s->keep_blocks) {       /* "dump the whole line buffer out (for visualization)", :6689-6704: the lines of frame A and frame B */
        size_t made = 0;
        for (size_t i = 0; i < s->conv_queue.n; i++) {
            const orc_stc_line *l = dq_at(&s->conv_queue, i);
            if (l->frame_number != f1->frame_number && l->frame_number != s->frasm_f2.frame_number) continue;
            if (s->asm_n == s->asm_cap) { s->asm_cap = s->asm_cap ? s->asm_cap * 2 : 2048; s->asm_lines = (orc_stc_line *)realloc(s->asm_lines, s->asm_cap * sizeof(*l)); }
            s->asm_lines[s->asm_n++] = *l; made++;
        }
        if (s->asm_frames == s->asm_frames_cap) { s->asm_frames_cap = s->asm_frames_cap ? s->asm_frames_cap * 2 : 64; s->asm_frame_n = (size_t *)realloc(s->asm_frame_n, s->asm_frames_cap * sizeof(size_t)); }
        s->asm_frame_n[s->asm_frames++] = made;
    }
    while (s->conv_queue.n > ORC_MIN_DEINT_DATA) {
        bool already_unsafe = false;
        orc_block_clear(&b);
        d->data_res_mode = get_data_block_resolution(s, &s->conv_queue, 0);
        orc_deint_process_block(d, dq_at(&s->conv_queue, 0), s->conv_queue.n, 0, &b);
        f1->blocks_total++;
        /* setBlockSampleRate :6455-6480 */
        if (s->preset_sample_rate == 44100 || s->preset_sample_rate == 44056) b.sample_rate = s->preset_sample_rate;
        else b.sample_rate = (f1->video_standard == ORC_VID_NTSC) ? 44056 : 44100;
        f1->odd_sample_rate = f1->even_sample_rate = b.sample_rate;
        b.m2_format = s->mode_m2;
        dq_pop_front(&s->conv_queue);
        if (!orc_block_is_silent(&b)) {
            if (s->mask_seams) {
                if (!f1->inner_padding_ok && !f1->inner_silence)
                    if ((b.w_line[0] > b.w_line[7]) && b.w_frame[0] == f1->frame_number && b.w_frame[0] == b.w_frame[7]) { orc_block_mark_unsafe(&b); already_unsafe = true; }
                if (!f0->outer_padding_ok && !f0->outer_silence)
                    if (b.w_frame[0] != b.w_frame[7] && b.w_frame[0] == f0->frame_number && b.w_frame[7] == f1->frame_number) { orc_block_mark_unsafe(&b); already_unsafe = true; }
            }
            if (!already_unsafe) {
                if (s->broken_mask_dur > 0 && s->broken_countdown == 0) if (b.audio_state == ORC_AUD_BROKEN) s->broken_countdown = s->broken_mask_dur;
                if (s->broken_countdown != 0) orc_block_mark_unsafe(&b);
            }
        }
        if (!is_block_no_report(s, &b)) {
            if (orc_block_is_valid(&b)) {
                if (b.audio_state == ORC_AUD_FIX_P) f1->blocks_fix_p++;
                else if (b.audio_state == ORC_AUD_FIX_Q) f1->blocks_fix_q++;
                bool altered = false; for (int i = 0; i <= 7; i++) if (b.cwd_fixed[i]) altered = true;
                if (b.cwd_applied && altered) f1->blocks_fix_cwd++;
            } else {
                f1->blocks_drop++;
                f1->samples_drop = (uint16_t)(f1->samples_drop + orc_block_errors_audio_fixed(&b));
                if (b.audio_state == ORC_AUD_BROKEN) f1->blocks_broken_field++;
            }
        }
        if (s->broken_countdown > 0) s->broken_countdown--;
        output_sample_pair(s, &b, 0, 1); output_sample_pair(s, &b, 2, 3); output_sample_pair(s, &b, 4, 5);
        if (s->keep_blocks) {                                          /* emit newBlockProcessed(*in_block), :6626 */
            if (s->blocks_n == s->blocks_cap) { s->blocks_cap = s->blocks_cap ? s->blocks_cap * 2 : 1024; s->blocks = (orc_stc_block *)realloc(s->blocks, s->blocks_cap * sizeof(b)); }
            s->blocks[s->blocks_n++] = b;
        }
    }
}

/* ------------------------------------------------------------------ one turn of doFrameReassemble (:7284-7457) */
bool orc_stitcher_step(orc_stitcher *s)
{
    if (s->in_lines.n == 0) return false;
    if (!wait_for_two_frames(s)) return false;
    fill_until_two_frames(s);
    while (s->in_lines.n > 0) {           /* remove Frame A from the input queue */
        if (dq_at(&s->in_lines, 0)->frame_number <= s->frasm_f1.frame_number) dq_pop_front(&s->in_lines); else break;
    }
    find_frames_trim(s);
    if (s->file_start) reset_state(s);
    if (s->in_lines.n > 0 && s->file_end) {
        /* :7380-7400 - the reference copies the front line ONCE and pops while that copy's frame number matches */
        uint32_t fn = dq_at(&s->in_lines, 0)->frame_number;
        while (fn == s->frasm_f2.frame_number) { dq_pop_front(&s->in_lines); if (s->in_lines.n == 0) break; }
    }
    split_frames_to_fields(s);
    find_field_stitching(s);
    if (s->file_start) { orc_frasm sd; frasm_clear(&sd); sd.service_type = 1; frames_push(s, &sd); output_service(s, 1); }
    fill_frame_for_output(s);
    prescan_frame(s);
    perform_deinterleave(s);
    frames_push(s, &s->frasm_f1);
    s->frasm_f0 = s->frasm_f1;
    s->frasm_f1 = s->frasm_f2;
    s->frasm_f2.trim_ok = false; s->frasm_f2.inner_padding_ok = false; s->frasm_f2.outer_padding_ok = false;
    if (s->file_end) { orc_frasm sd; frasm_clear(&sd); sd.service_type = 2; frames_push(s, &sd); output_service(s, 2); reset_state(s); }
    s->file_start = s->file_end = false;
    return true;
}
