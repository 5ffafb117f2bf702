/*
 * api.c - ctypes-friendly entry points of the CPU oracle (liborc.so), mirroring the entry
 * points of oracle/ref_driver.cpp one for one so tests can run both on the same inputs.
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).
 */
#include "sdv_oracle.h"
#include "../include/sdvpcm.h"
#include <stdlib.h>
#include <string.h>

void orc_line_to_rec(const orc_stc_line *l, sdv_line_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l->frame_number;
    r->line_number = l->line_number;
    for (int i = 0; i < 9; i++) r->words[i] = l->words[i];
    r->calc_crc = l->calc_crc;
    r->data_start = l->coords.data_start; r->data_stop = l->coords.data_stop;
    r->marker_start_bg_coord = l->marker_start_bg_coord;
    r->marker_start_ed_coord = l->marker_start_ed_coord;
    r->marker_stop_ed_coord = l->marker_stop_ed_coord;
    r->black_level = l->black_level; r->white_level = l->white_level;
    r->ref_low = l->ref_low; r->ref_level = l->ref_level; r->ref_high = l->ref_high;
    r->hysteresis_depth = l->hysteresis_depth; r->shift_stage = l->shift_stage;
    r->service_type = l->service_type;
    r->mark_st_stage = l->mark_st_stage; r->mark_ed_stage = l->mark_ed_stage;
    uint8_t f = 0;
    if (l->ref_level_sweeped) f |= SDV_LF_REF_SWEEPED;
    if (l->coords_sweeped) f |= SDV_LF_COORDS_SWEEPED;
    if (l->data_by_ext_tune) f |= SDV_LF_BY_EXT_TUNE;
    if (l->blk_wht_set) f |= SDV_LF_BW_SET;
    if (l->coords_set) f |= SDV_LF_COORDS_SET;
    if (l->forced_bad) f |= SDV_LF_FORCED_BAD;
    if (orc_stc_crc_valid(l)) f |= SDV_LF_CRC_VALID;
    if (l->coords.from_doubled) f |= SDV_LF_FROM_DOUBLED;
    r->flags = f;
    uint8_t ws = 0;                         /* getters mask by forced_bad: stc007line.cpp:656-680 */
    if (!l->forced_bad && l->word_crc[0]) ws |= SDV_WS_WORD_CRC;
    if (!l->forced_bad && l->word_valid[0]) ws |= SDV_WS_WORD_VALID;
    r->word_state = ws;
}

typedef struct { orc_binarizer bin; orc_video_line vl; orc_stc_line out; } orc_bin_handle;

uint16_t orc_crc_stc007(const uint16_t *words8) { return orc_stc_crc_words(words8); }

void *orc_bin_new(void)
{
    orc_bin_handle *h = (orc_bin_handle *)calloc(1, sizeof(*h));
    orc_binarizer_init(&h->bin);
    orc_stc_clear(&h->out);
    return h;
}
void orc_bin_free(void *h) { free(h); }
void orc_bin_set_mode(void *h, int mode) { orc_binarizer_set_mode(&((orc_bin_handle *)h)->bin, (uint8_t)mode); }
void orc_bin_set_coord_search(void *h, int on) { ((orc_bin_handle *)h)->bin.do_coord_search = on != 0; }
void orc_bin_set_preset(void *hh, const sdv_bin_preset *p)
{
    orc_bin_preset *s = &((orc_bin_handle *)hh)->bin.digi_set;
    orc_bin_preset_reset(s);
    s->max_black_lvl = p->max_black_lvl; s->min_white_lvl = p->min_white_lvl; s->min_contrast = p->min_contrast;
    s->min_ref_lvl = p->min_ref_lvl; s->max_ref_lvl = p->max_ref_lvl; s->min_valid_crcs = p->min_valid_crcs;
    s->mark_max_dist = p->mark_max_dist; s->left_bit_pick = p->left_bit_pick; s->right_bit_pick = p->right_bit_pick;
    s->en_force_coords = p->en_force_coords; s->en_coord_search = p->en_coord_search;
    s->en_first_line_dup = p->en_first_line_dup; s->en_good_no_marker = p->en_good_no_marker;
    s->horiz_coords.data_start = p->horiz_start; s->horiz_coords.data_stop = p->horiz_stop;
}
void orc_bin_reset_good(void *h) { orc_binarizer_set_good_parameters(&((orc_bin_handle *)h)->bin, NULL); }
void orc_bin_set_good_from_last(void *hh) { orc_bin_handle *h = (orc_bin_handle *)hh; orc_binarizer_set_good_parameters(&h->bin, &h->out); }
void orc_bin_set_state(void *hh, const sdv_bin_state *s)
{
    orc_bin_handle *h = (orc_bin_handle *)hh;
    orc_binarizer_set_reference_level(&h->bin, s->in_def_reference);
    orc_coords c; orc_coords_clear(&c);
    c.data_start = s->in_def_start; c.data_stop = s->in_def_stop; c.from_doubled = s->in_def_from_doubled != 0;
    orc_binarizer_set_data_coordinates(&h->bin, c);
    orc_binarizer_set_bw_levels(&h->bin, s->in_def_black, s->in_def_white);
}
/* ... and the sticky do_ref_lvl_sweep member (no setter in the reference: it stays as the last line left it, binarizer.cpp:1104-1128; read by the level
 * detection of the next line, :3409): the per-line entry of the engine (sdv_binarize_lines) takes it from the state */
void orc_bin_set_state_full(void *hh, const sdv_bin_state *s) { orc_bin_set_state(hh, s); ((orc_bin_handle *)hh)->bin.do_ref_lvl_sweep = s->do_ref_lvl_sweep != 0; }
int orc_bin_process(void *hh, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                    sdv_line_rec *out)
{
    orc_bin_handle *h = (orc_bin_handle *)hh;
    h->vl.frame_number = frame; h->vl.line_number = line;
    h->vl.pixels = px; h->vl.length = (uint16_t)len;
    h->vl.service_type = (uint8_t)service;
    /* VideoLine::setServ*() mark the line empty (videoline.cpp:84-118) */
    h->vl.empty = (service != SDV_SRV_NO) ? true : (empty != 0);
    h->vl.doubled = (service == SDV_SRV_NO) ? (doubled != 0) : false;
    h->bin.video_line = &h->vl;
    h->bin.out_pcm_line = &h->out;
    int ret = orc_binarizer_process_line(&h->bin);
    orc_line_to_rec(&h->out, out);
    return ret;
}

/* ------------------------------------------------------------------ PCM-1 front half (bin_pcm1.c) */
#include "bin_pcm1.h"
typedef struct { orc_binarizer bin; orc_video_line vl; orc_p1_line out; } orc_bin1_handle;

void orc_p1_line_to_rec(const orc_p1_line *l, sdv_pcm1_bin_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l->frame_number; r->line_number = l->line_number;
    for (int i = 0; i < 7; i++) r->words[i] = l->words[i];
    r->calc_crc = l->calc_crc;
    r->data_start = l->coords.data_start; r->data_stop = l->coords.data_stop;
    r->black_level = l->black_level; r->white_level = l->white_level;
    r->ref_low = l->ref_low; r->ref_level = l->ref_level; r->ref_high = l->ref_high;
    r->hysteresis_depth = l->hysteresis_depth; r->shift_stage = l->shift_stage;
    r->service_type = l->service_type;
    r->picked_bits_left = l->picked_bits_left; r->picked_bits_right = l->picked_bits_right;
    r->flags = (uint8_t)((l->ref_level_sweeped ? SDV_LF_REF_SWEEPED : 0) | (l->coords_sweeped ? SDV_LF_COORDS_SWEEPED : 0) |
                         (l->data_by_ext_tune ? SDV_LF_BY_EXT_TUNE : 0) | (l->blk_wht_set ? SDV_LF_BW_SET : 0) |
                         (l->coords_set ? SDV_LF_COORDS_SET : 0) | (l->forced_bad ? SDV_LF_FORCED_BAD : 0) |
                         (orc_p1_crc_valid(l) ? SDV_LF_CRC_VALID : 0) | (l->coords.from_doubled ? SDV_LF_FROM_DOUBLED : 0));
}

void *orc_bin1_new(void)
{
    orc_bin1_handle *h = (orc_bin1_handle *)calloc(1, sizeof(*h));
    orc_binarizer_init(&h->bin);
    orc_p1_clear(&h->out);
    return h;
}
void orc_bin1_free(void *h) { free(h); }
void orc_bin1_set_mode(void *h, int mode) { orc_binarizer_set_mode(&((orc_bin1_handle *)h)->bin, (uint8_t)mode); }
void orc_bin1_set_coord_search(void *h, int on) { ((orc_bin1_handle *)h)->bin.do_coord_search = on != 0; }
void orc_bin1_set_preset(void *hh, const sdv_bin_preset *p) { orc_bin_set_preset(hh, p); }      /* the binarizer is the first member of both handles */
void orc_bin1_reset_good(void *h) { orc_binarizer_set_good_parameters_p1(&((orc_bin1_handle *)h)->bin, NULL); }
void orc_bin1_set_good_from_last(void *hh) { orc_bin1_handle *h = (orc_bin1_handle *)hh; orc_binarizer_set_good_parameters_p1(&h->bin, &h->out); }
/* the flag has no setter in the reference (the member stays as the last line left it); the per-line entry of the engine takes it from the state */
void orc_bin1_set_state(void *hh, const sdv_bin_state *s) { orc_bin_set_state(hh, s); ((orc_bin1_handle *)hh)->bin.do_ref_lvl_sweep = s->do_ref_lvl_sweep != 0; }
int orc_bin1_scan_done(void *h) { return ((orc_bin1_handle *)h)->bin.p1_scan_done ? 1 : 0; }
int orc_bin1_process(void *hh, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                     sdv_pcm1_bin_rec *out)
{
    orc_bin1_handle *h = (orc_bin1_handle *)hh;
    h->vl.frame_number = frame; h->vl.line_number = line;
    h->vl.pixels = px; h->vl.length = (uint16_t)len;
    h->vl.service_type = (uint8_t)service;
    h->vl.empty = (service != SDV_SRV_NO) ? true : (empty != 0);
    h->vl.doubled = (service == SDV_SRV_NO) ? (doubled != 0) : false;
    h->bin.video_line = &h->vl;
    h->bin.out_pcm_line = NULL;
    int ret = orc_binarizer_process_line_p1(&h->bin, &h->out);
    orc_p1_line_to_rec(&h->out, out);
    return ret;
}

/* ------------------------------------------------------------------ PCM-16x0 front half (bin_pcm16.c) */
#include "bin_pcm16.h"
typedef struct { orc_binarizer bin; orc_video_line vl; orc_p16_line out; } orc_bin16_handle;

void orc_p16_line_to_rec(const orc_p16_line *l, sdv_pcm16x0_bin_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l->frame_number; r->line_number = l->line_number;
    for (int i = 0; i < 4; i++) r->words[i] = l->words[i];
    r->calc_crc = l->calc_crc;
    r->data_start = l->coords.data_start; r->data_stop = l->coords.data_stop;
    r->queue_order = l->queue_order;
    r->black_level = l->black_level; r->white_level = l->white_level;
    r->ref_low = l->ref_low; r->ref_level = l->ref_level; r->ref_high = l->ref_high;
    r->hysteresis_depth = l->hysteresis_depth; r->shift_stage = l->shift_stage;
    r->service_type = l->service_type;
    r->picked_bits_left = l->picked_bits_left; r->picked_bits_right = l->picked_bits_right;
    r->flags = (uint8_t)((l->ref_level_sweeped ? SDV_LF_REF_SWEEPED : 0) | (l->coords_sweeped ? SDV_LF_COORDS_SWEEPED : 0) |
                         (l->data_by_ext_tune ? SDV_LF_BY_EXT_TUNE : 0) | (l->blk_wht_set ? SDV_LF_BW_SET : 0) |
                         (l->coords_set ? SDV_LF_COORDS_SET : 0) | (l->forced_bad ? SDV_LF_FORCED_BAD : 0) |
                         (orc_p16_crc_valid(l) ? SDV_LF_CRC_VALID : 0) | (l->coords.from_doubled ? SDV_LF_FROM_DOUBLED : 0));
    r->line_part = l->line_part; r->control_bit = l->control_bit ? 1 : 0;
}

void *orc_bin16_new(void)
{
    orc_bin16_handle *h = (orc_bin16_handle *)calloc(1, sizeof(*h));
    orc_binarizer_init(&h->bin);
    orc_p16_clear(&h->out);
    return h;
}
void orc_bin16_free(void *h) { free(h); }
void orc_bin16_set_mode(void *h, int mode) { orc_binarizer_set_mode(&((orc_bin16_handle *)h)->bin, (uint8_t)mode); }
void orc_bin16_set_coord_search(void *h, int on) { ((orc_bin16_handle *)h)->bin.do_coord_search = on != 0; }
void orc_bin16_set_preset(void *hh, const sdv_bin_preset *p) { orc_bin_set_preset(hh, p); }
void orc_bin16_reset_good(void *h) { orc_binarizer_set_good_parameters_p16(&((orc_bin16_handle *)h)->bin, NULL); }
void orc_bin16_set_good_from_last(void *hh) { orc_bin16_handle *h = (orc_bin16_handle *)hh; orc_binarizer_set_good_parameters_p16(&h->bin, &h->out); }
void orc_bin16_set_state(void *hh, const sdv_bin_state *s) { orc_bin_set_state(hh, s); ((orc_bin16_handle *)hh)->bin.do_ref_lvl_sweep = s->do_ref_lvl_sweep != 0; }
int orc_bin16_scan_done(void *h) { return ((orc_bin16_handle *)h)->vl.scan_done ? 1 : 0; }
/* one pass over a video line: part = Binarizer::PART_PCM16X0_LEFT / _MIDDLE / _RIGHT (1 / 2 / 3; 0 = FULL_LINE); new_line: the
 * VideoLine is a fresh one (scan_done cleared), else the same line as in the pass before */
int orc_bin16_process(void *hh, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                      int part, int new_line, sdv_pcm16x0_bin_rec *out)
{
    orc_bin16_handle *h = (orc_bin16_handle *)hh;
    if (new_line) h->vl.scan_done = false;
    h->vl.frame_number = frame; h->vl.line_number = line;
    h->vl.pixels = px; h->vl.length = (uint16_t)len;
    h->vl.service_type = (uint8_t)service;
    h->vl.empty = (service != SDV_SRV_NO) ? true : (empty != 0);
    h->vl.doubled = (service == SDV_SRV_NO) ? (doubled != 0) : false;
    h->bin.video_line = &h->vl;
    h->bin.out_pcm_line = NULL;
    h->bin.line_part_mode = (uint8_t)part;
    int ret = orc_binarizer_process_line_p16(&h->bin, &h->out);
    orc_p16_line_to_rec(&h->out, out);
    return ret;
}
uint16_t orc_pcm16x0_crc(const uint16_t *w3) { return orc_p16_crc_words(w3); }

/* ------------------------------------------------------------------ VideoToDigital level */
#include "v2d.h"

static void stats_to_pod(const orc_frame_stats *q, sdv_frame_stats *s)
{
    memset(s, 0, sizeof(*s));
    s->frame_id = q->frame_id; s->line_length = q->line_length;
    s->lines_odd = q->lines_odd; s->lines_even = q->lines_even;
    s->lines_pcm_odd = q->lines_pcm_odd; s->lines_pcm_even = q->lines_pcm_even;
    s->lines_bad_odd = q->lines_bad_odd; s->lines_bad_even = q->lines_bad_even;
    s->lines_dup_odd = q->lines_dup_odd; s->lines_dup_even = q->lines_dup_even;
    s->data_start = q->data_coord.data_start; s->data_stop = q->data_coord.data_stop;
    s->data_from_doubled = q->data_coord.from_doubled; s->data_not_sure = q->data_coord.not_sure;
}

void *orc_v2d_new(void) { orc_v2d *v = (orc_v2d *)malloc(sizeof(orc_v2d)); orc_v2d_init(v); return v; }
void orc_v2d_delete(void *v) { orc_v2d_free((orc_v2d *)v); free(v); }
void orc_v2d_set_mode(void *v, int mode) { if (mode >= 0 && mode < 4) ((orc_v2d *)v)->binarization_mode = (uint8_t)mode; }
void orc_v2d_set_check_line_dup(void *v, int on) { ((orc_v2d *)v)->check_line_copy = on != 0; }
void orc_v2d_set_m2(void *v, int on) { ((orc_v2d *)v)->m2_format = on != 0; }
void orc_v2d_set_preset(void *vv, const sdv_bin_preset *p)
{
    orc_bin_preset s; orc_bin_preset_reset(&s);
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    orc_v2d_set_fine_settings((orc_v2d *)vv, &s);
}

/* dropped frames of the next *_run call: mask[f] != 0 = frame f of that call is a frame of empty lines (consumed by the call) */
int orc_g_empty_frame = 0;
static uint8_t *g_empty_mask = NULL; static size_t g_empty_n = 0;
void orc_set_empty_frames(const uint8_t *mask, size_t n)
{
    free(g_empty_mask); g_empty_mask = NULL; g_empty_n = 0;
    if (mask && n) { g_empty_mask = (uint8_t *)malloc(n); memcpy(g_empty_mask, mask, n); g_empty_n = n; }
}
static int empty_frame_at(int f) { return (size_t)f < g_empty_n && g_empty_mask[f] != 0; }

/* Runs n_frames consecutive frames (frame numbers first_frame_no..). Records: per frame height+3
 * (+1 NEW_FILE before the first frame when new_file). Returns total records written. */
long orc_v2d_run(void *vv, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                 int new_file, int doubled, sdv_line_rec *out, sdv_frame_stats *stats)
{
    orc_v2d *v = (orc_v2d *)vv;
    long n = 0;
    for (int f = 0; f < n_frames; f++) {
        orc_frame_stats q;
        orc_g_empty_frame = empty_frame_at(f);
        n += orc_v2d_frame(v, luma + (size_t)f * stride * (size_t)height, stride, width, height, first_frame_no + (uint32_t)f,
                           (new_file & 1) && f == 0, doubled != 0, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[f]);
    }
    orc_g_empty_frame = 0; orc_set_empty_frames(NULL, 0);
    if (new_file & 2) {     /* bit 1: the file ends here (out needs height + 4 more records, stats one more row) */
        orc_frame_stats q;
        n += orc_v2d_end_file_frame(v, height, first_frame_no + (uint32_t)n_frames, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[n_frames]);
    }
    return n;
}

/* Export of the frame-to-frame chain state (what the HIP engine speculates on, include/sdvpcm.h). */
void orc_v2d_get_state(void *vv, sdv_v2d_state *s)
{
    orc_v2d *v = (orc_v2d *)vv;
    memset(s, 0, sizeof(*s));
    s->bin.in_def_black = v->line_converter.in_def_black; s->bin.in_def_white = v->line_converter.in_def_white;
    s->bin.in_def_reference = v->line_converter.in_def_reference;
    s->bin.in_def_start = v->line_converter.in_def_coord.data_start; s->bin.in_def_stop = v->line_converter.in_def_coord.data_stop;
    s->bin.in_def_from_doubled = v->line_converter.in_def_coord.from_doubled;
    s->do_ref_lvl_sweep = v->line_converter.do_ref_lvl_sweep;
    s->reset_stats = v->reset_stats;
    s->n_last_valid = (uint8_t)v->last_valid_coord_list.n;
    s->n_long_valid = (uint8_t)v->long_valid_coords.n;
    uint16_t m = 0;
    for (int i = 0; i < v->last_valid_coord_list.n && i < 9; i++) {
        s->last_valid[i].data_start = v->last_valid_coord_list.v[i].data_start;
        s->last_valid[i].data_stop = v->last_valid_coord_list.v[i].data_stop;
        if (v->last_valid_coord_list.v[i].from_doubled) m |= (uint16_t)(1u << i);
    }
    s->last_valid_doubled_mask_lo = (uint8_t)(m & 0xFF); s->last_valid_doubled_mask_hi = (uint8_t)(m >> 8);
    m = 0;
    for (int i = 0; i < v->long_valid_coords.n && i < 16; i++) {
        s->long_valid[i].data_start = v->long_valid_coords.v[i].data_start;
        s->long_valid[i].data_stop = v->long_valid_coords.v[i].data_stop;
        if (v->long_valid_coords.v[i].from_doubled) m |= (uint16_t)(1u << i);
    }
    s->long_valid_doubled_mask = m;
}

/* ------------------------------------------------------------------ VideoToDigital level, PCM-1 (v2d_p1.c) */
#include "v2d_p1.h"
void *orc_v2d1_new(void) { orc_v2d1 *v = (orc_v2d1 *)malloc(sizeof(orc_v2d1)); orc_v2d1_init(v); return v; }
void orc_v2d1_delete(void *v) { orc_v2d1_free((orc_v2d1 *)v); free(v); }
void orc_v2d1_set_mode(void *v, int mode) { if (mode >= 0 && mode < 4) ((orc_v2d1 *)v)->binarization_mode = (uint8_t)mode; }
void orc_v2d1_set_check_line_dup(void *v, int on) { ((orc_v2d1 *)v)->check_line_copy = on != 0; }
void orc_v2d1_set_preset(void *vv, const sdv_bin_preset *p)
{
    orc_bin_preset s; orc_bin_preset_reset(&s);
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    orc_v2d1_set_fine_settings((orc_v2d1 *)vv, &s);
}
/* n_frames consecutive PCM-1 frames through the worker; new_file bit 0: NEW_FILE line first, bit 1: the filler frame + END_FILE
 * follow (height + 4 more records, one more stats row).  Returns the records written. */
long orc_v2d1_run(void *vv, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                  int new_file, int doubled, sdv_pcm1_bin_rec *out, sdv_frame_stats *stats)
{
    orc_v2d1 *v = (orc_v2d1 *)vv;
    long n = 0;
    for (int f = 0; f < n_frames; f++) {
        orc_frame_stats q;
        orc_g_empty_frame = empty_frame_at(f);
        n += orc_v2d1_frame(v, luma + (size_t)f * stride * (size_t)height, stride, width, height, first_frame_no + (uint32_t)f,
                            (new_file & 1) && f == 0, doubled != 0, false, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[f]);
    }
    orc_g_empty_frame = 0; orc_set_empty_frames(NULL, 0);
    if (new_file & 2) {
        orc_frame_stats q;
        n += orc_v2d1_frame(v, NULL, 0, width, height, first_frame_no + (uint32_t)n_frames, false, false, true, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[n_frames]);
    }
    return n;
}

/* ------------------------------------------------------------------ VideoToDigital level, PCM-16x0 (v2d_p16.c) */
#include "v2d_p16.h"
void *orc_v2d16_new(void) { orc_v2d16 *v = (orc_v2d16 *)malloc(sizeof(orc_v2d16)); orc_v2d16_init(v); return v; }
void orc_v2d16_delete(void *v) { orc_v2d16_free((orc_v2d16 *)v); free(v); }
void orc_v2d16_set_mode(void *v, int mode) { if (mode >= 0 && mode < 4) ((orc_v2d16 *)v)->binarization_mode = (uint8_t)mode; }
void orc_v2d16_set_check_line_dup(void *v, int on) { ((orc_v2d16 *)v)->check_line_copy = on != 0; }
void orc_v2d16_set_preset(void *vv, const sdv_bin_preset *p)
{
    orc_bin_preset s; orc_bin_preset_reset(&s);
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    orc_v2d16_set_fine_settings((orc_v2d16 *)vv, &s);
}
/* n_frames consecutive PCM-16x0 frames through the worker (flags as orc_v2d1_run).  Returns the records written:
 * per frame 3 * height + 3, + 1 for NEW_FILE, + height + 4 for the filler frame. */
long orc_v2d16_run(void *vv, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                   int new_file, int doubled, sdv_pcm16x0_bin_rec *out, sdv_frame_stats *stats)
{
    orc_v2d16 *v = (orc_v2d16 *)vv;
    long n = 0;
    for (int f = 0; f < n_frames; f++) {
        orc_frame_stats q;
        orc_g_empty_frame = empty_frame_at(f);
        n += orc_v2d16_frame(v, luma + (size_t)f * stride * (size_t)height, stride, width, height, first_frame_no + (uint32_t)f,
                             (new_file & 1) && f == 0, doubled != 0, false, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[f]);
    }
    orc_g_empty_frame = 0; orc_set_empty_frames(NULL, 0);
    if (new_file & 2) {
        orc_frame_stats q;
        n += orc_v2d16_frame(v, NULL, 0, width, height, first_frame_no + (uint32_t)n_frames, false, false, true, out + n, &q);
        if (stats) stats_to_pod(&q, &stats[n_frames]);
    }
    return n;
}

/* ------------------------------------------------------------------ deinterleaver level */
#include "deint.h"

static void deint_line_to_orc(const sdv_deint_line *in, orc_stc_line *l)
{
    orc_stc_clear(l);
    l->frame_number = in->frame_number; l->line_number = in->line_number;
    for (int i = 0; i < 8; i++) { l->words[i] = in->words[i]; l->word_crc[i] = l->word_valid[i] = (in->word_crc_ok >> i) & 1; }
    orc_stc_calc_crc(l);
    l->words[8] = l->calc_crc;                       /* CRC state itself is only read through isFixedByCWD() */
    if (in->flags & SDV_DL_FIXED_BY_CWD) {
        /* a line repaired by CWD: CRC valid, at least one word with word_crc false but word_valid true */
        int done = 0;
        for (int i = 0; i < 8 && !done; i++) if (!l->word_crc[i]) { l->word_valid[i] = true; done = 1; }
        if (!done) { l->word_crc[0] = false; l->word_valid[0] = true; }
    } else {
        /* make sure isFixedByCWD() is false */
        bool any = false;
        for (int i = 0; i < 8; i++) if (!l->word_crc[i]) any = true;
        if (any) l->words[8] = (uint16_t)~l->calc_crc;
    }
    if (in->flags & SDV_DL_COORDS_BW_OK) { l->blk_wht_set = true; orc_coords_set(&l->coords, 10, 700); }
}

static void block_to_rec(const orc_stc_block *b, sdv_block_rec *r)
{
    memset(r, 0, sizeof(*r));
    for (int i = 0; i < 8; i++) {
        r->w_frame[i] = b->w_frame[i]; r->w_line[i] = b->w_line[i]; r->words[i] = b->words[i];
        if (b->line_crc[i]) r->line_crc |= (uint8_t)(1u << i);
        if (b->cwd_fixed[i]) r->cwd_fixed |= (uint8_t)(1u << i);
        if (b->word_valid[i]) r->word_valid |= (uint8_t)(1u << i);
    }
    r->resolution = b->resolution; r->audio_state = b->audio_state; r->cwd_applied = b->cwd_applied; r->sample_rate = b->sample_rate;
}

/* processBlock(line_shift) for every shift 0..n_blocks-1 over the same line buffer */
int orc_deint_run(const sdv_deint_line *lines, size_t n_lines, const sdv_deint_settings *st, sdv_block_rec *out, size_t n_blocks)
{
    orc_stc_line *l = (orc_stc_line *)malloc(n_lines * sizeof(orc_stc_line));
    for (size_t i = 0; i < n_lines; i++) deint_line_to_orc(&lines[i], &l[i]);
    orc_deint d; orc_deint_init(&d);
    d.data_res_mode = st->res_mode; d.ignore_crc = st->ignore_crc; d.force_ecc_check = st->force_ecc_check;
    d.en_p_code = st->en_p_code; d.en_q_code = st->en_q_code; d.en_cwd = st->en_cwd;
    int rc = ORC_DI_RET_OK;
    for (size_t s = 0; s < n_blocks; s++) {
        orc_stc_block b;
        orc_block_clear(&b);
        rc = orc_deint_process_block(&d, l, n_lines, (uint16_t)s, &b);
        if (rc != ORC_DI_RET_OK) break;
        block_to_rec(&b, &out[s]);
    }
    free(l);
    return rc;
}
uint16_t orc_q_code(const uint16_t *w6) { return orc_calc_q(w6); }
uint16_t orc_p_code(const uint16_t *w6) { return orc_calc_p(w6); }

/* ------------------------------------------------------------------ stitcher level */
#include "stitcher.h"

/* re-hydrates an STC007Line from a binarizer record (what INTEGRATION.md's toLine() does on the reference side) */
void orc_rec_to_line(const sdv_line_rec *r, orc_stc_line *l)
{
    orc_stc_clear(l);
    l->frame_number = r->frame_number; l->line_number = r->line_number;
    if (r->service_type != SDV_SRV_NO && r->service_type != SDV_SRV_CTRL_BLOCK) { orc_stc_set_service(l, r->service_type); return; }
    for (int i = 0; i < 9; i++) l->words[i] = r->words[i];
    l->calc_crc = r->calc_crc;
    l->black_level = r->black_level; l->white_level = r->white_level;
    l->ref_low = r->ref_low; l->ref_level = r->ref_level; l->ref_high = r->ref_high;
    l->coords.data_start = r->data_start; l->coords.data_stop = r->data_stop; l->coords.from_doubled = (r->flags & SDV_LF_FROM_DOUBLED) != 0;
    l->hysteresis_depth = r->hysteresis_depth; l->shift_stage = r->shift_stage;
    l->ref_level_sweeped = (r->flags & SDV_LF_REF_SWEEPED) != 0; l->data_by_ext_tune = (r->flags & SDV_LF_BY_EXT_TUNE) != 0;
    l->blk_wht_set = (r->flags & SDV_LF_BW_SET) != 0; l->coords_set = (r->flags & SDV_LF_COORDS_SET) != 0;
    l->forced_bad = (r->flags & SDV_LF_FORCED_BAD) != 0;
    l->mark_st_stage = r->mark_st_stage; l->mark_ed_stage = r->mark_ed_stage;
    l->marker_start_bg_coord = r->marker_start_bg_coord; l->marker_start_ed_coord = r->marker_start_ed_coord; l->marker_stop_ed_coord = r->marker_stop_ed_coord;
    bool v = orc_stc_crc_valid(l);                   /* applyCRCStatePerWord */
    for (int i = 0; i < 9; i++) l->word_crc[i] = l->word_valid[i] = v;
    l->service_type = r->service_type;               /* SDV_SRV_NO or CTRL_BLOCK */
}

static void frasm_to_pod(const orc_frasm *f, sdv_frame_asm *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f->frame_number;
    o->odd_std_lines = f->odd_std_lines; o->even_std_lines = f->even_std_lines; o->odd_data_lines = f->odd_data_lines; o->even_data_lines = f->even_data_lines;
    o->odd_valid_lines = f->odd_valid_lines; o->even_valid_lines = f->even_valid_lines;
    o->odd_top_data = f->odd_top_data; o->odd_bottom_data = f->odd_bottom_data; o->even_top_data = f->even_top_data; o->even_bottom_data = f->even_bottom_data;
    o->odd_sample_rate = f->odd_sample_rate; o->even_sample_rate = f->even_sample_rate;
    o->blocks_total = f->blocks_total; o->blocks_drop = f->blocks_drop; o->samples_drop = f->samples_drop;
    o->inner_padding = f->inner_padding; o->outer_padding = f->outer_padding;
    o->blocks_broken_field = f->blocks_broken_field; o->blocks_broken_seam = f->blocks_broken_seam;
    o->blocks_fix_p = f->blocks_fix_p; o->blocks_fix_q = f->blocks_fix_q; o->blocks_fix_cwd = f->blocks_fix_cwd;
    o->field_order = f->field_order; o->odd_ref = f->odd_ref; o->even_ref = f->even_ref; o->service_type = f->service_type;
    o->video_standard = f->video_standard; o->tff_cnt = f->tff_cnt; o->bff_cnt = f->bff_cnt; o->odd_resolution = f->odd_resolution; o->even_resolution = f->even_resolution;
    o->flags = (uint8_t)((f->order_preset ? SDV_FA_ORDER_PRESET : 0) | (f->order_guessed ? SDV_FA_ORDER_GUESSED : 0) | (f->trim_ok ? SDV_FA_TRIM_OK : 0) |
                         (f->inner_padding_ok ? SDV_FA_INNER_OK : 0) | (f->outer_padding_ok ? SDV_FA_OUTER_OK : 0) | (f->inner_silence ? SDV_FA_INNER_SILENCE : 0) |
                         (f->outer_silence ? SDV_FA_OUTER_SILENCE : 0) | (f->vid_std_preset ? SDV_FA_VID_STD_PRESET : 0));
    o->flags2 = (uint8_t)((f->odd_emphasis ? SDV_FA2_ODD_EMPHASIS : 0) | (f->even_emphasis ? SDV_FA2_EVEN_EMPHASIS : 0) | (f->vid_std_guessed ? SDV_FA2_VID_STD_GUESSED : 0));
    o->ctrl_index = f->ctrl_index; o->ctrl_hour = f->ctrl_hour; o->ctrl_minute = f->ctrl_minute; o->ctrl_second = f->ctrl_second; o->ctrl_field = f->ctrl_field;
}

void orc_default_stitch_settings(sdv_stitch_settings *st)
{
    memset(st, 0, sizeof(*st));
    st->enable_p = 0; st->enable_q = 0; st->enable_cwd = 1;       /* ctor :28-30 (the GUI switches P/Q on) */
    st->max_unch_14 = ORC_MAX_BURST_UNCH_14BIT; st->max_unch_16 = ORC_MAX_BURST_UNCH_16BIT;
    st->use_ecc = 1; st->mask_seams = 1; st->broke_mask = ORC_UNCH_MASK_DURATION; st->top_line_fix = 0; st->sample_rate_preset = 1;
}

/* feeds all records, runs the stitcher until the queue holds less than two frames; returns number of pairs
 * (<0: output buffer too small) */
static long stitch_run(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                       sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames, sdv_block_rec *blocks, size_t blocks_cap, size_t *n_blocks);
long orc_stitch_run(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                    sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames)
{
    return stitch_run(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL, 0, NULL);
}
/* ... the assembled lines the stitcher hands to the visualiser (newLineProcessed) of the last orc_stitch_run_blocks call: kept here until asked for */
static sdv_asm_line_rec *g_asm; static size_t g_asm_n; static uint32_t *g_asm_frames; static size_t g_asm_frames_n;
static void asm_to_rec(const orc_stc_line *l, sdv_asm_line_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l->frame_number; r->line_number = l->line_number; r->calc_crc = l->calc_crc;
    for (int i = 0; i < 9; i++) {
        r->words[i] = l->words[i];
        if (!l->forced_bad && l->word_crc[i]) r->word_crc_ok |= (uint16_t)(1u << i);        /* isWordCRCOk / isWordValid: false on a forced-bad line */
        if (!l->forced_bad && l->word_valid[i]) r->word_valid |= (uint16_t)(1u << i);
    }
    r->flags = (uint8_t)((l->forced_bad ? SDV_AL_FORCED_BAD : 0) | (orc_stc_has_markers(l) ? SDV_AL_MARKERS : 0) | (orc_stc_crc_valid(l) ? SDV_AL_CRC_VALID : 0));
}
/* the lines (n_lines of them, up to cap copied) and how many each stitcher turn made (n_turns entries, up to turns_cap copied) */
void orc_stitch_last_asm_lines(sdv_asm_line_rec *out, size_t cap, size_t *n_lines, uint32_t *per_turn, size_t turns_cap, size_t *n_turns)
{
    for (size_t i = 0; i < g_asm_n && i < cap; i++) out[i] = g_asm[i];
    for (size_t i = 0; i < g_asm_frames_n && i < turns_cap; i++) per_turn[i] = g_asm_frames[i];
    if (n_lines) *n_lines = g_asm_n;
    if (n_turns) *n_turns = g_asm_frames_n;
}
/* ... and the data blocks the stitcher hands to the visualiser (newBlockProcessed), one per three sample pairs */
long orc_stitch_run_blocks(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                           sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames, sdv_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    return stitch_run(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, blocks, blocks_cap, n_blocks);
}
static long stitch_run(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                       sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames, sdv_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    orc_stitcher *s = (orc_stitcher *)malloc(sizeof(orc_stitcher));
    orc_stitcher_init(s);
    s->keep_blocks = blocks != NULL;
    if (st->video_standard < ORC_VID_MAX) s->preset_video_mode = st->video_standard;
    if (st->field_order < ORC_ORDER_MAX) s->preset_field_order = st->field_order;
    s->enable_P_code = st->enable_p; s->enable_Q_code = st->enable_q; s->enable_CWD = st->enable_cwd; s->mode_m2 = st->m2_format;
    if (st->resolution_preset < ORC_SAMPLE_RES_MAX) s->preset_audio_res = st->resolution_preset;
    s->preset_sample_rate = st->sample_rate_preset;
    s->max_unchecked_14b_blocks = st->max_unch_14; s->max_unchecked_16b_blocks = st->max_unch_16;
    s->ignore_CRC = !st->use_ecc; s->mask_seams = st->mask_seams; s->broken_mask_dur = st->broke_mask; s->fix_cut_above = st->top_line_fix;
    orc_stc_line l;
    for (size_t i = 0; i < n_recs; i++) { orc_rec_to_line(&recs[i], &l); orc_stitcher_push_line(s, &l); }
    while (orc_stitcher_step(s)) {}
    long n = (long)s->out_n;
    if (s->out_n > out_cap) n = -1;
    else for (size_t i = 0; i < s->out_n; i++) {
        const orc_sample_pair *p = &s->out[i];
        sdv_sample_pair *o = &out[i];
        memset(o, 0, sizeof(*o));
        for (int c = 0; c < 2; c++) {
            o->audio_word[c] = p->audio_word[c];
            o->sample_flags[c] = (uint8_t)((p->data_block_ok[c] ? SDV_SF_BLOCK_OK : 0) | (p->word_valid[c] ? SDV_SF_WORD_VALID : 0) |
                                           (p->word_fixed[c] ? SDV_SF_WORD_FIXED : 0) | (p->word_masked[c] ? SDV_SF_WORD_MASKED : 0));
        }
        o->sample_rate = p->sample_rate; o->emphasis = p->emphasis; o->service_type = p->service_type;
    }
    size_t nf = s->frames_n < frames_cap ? s->frames_n : frames_cap;
    for (size_t i = 0; i < nf; i++) frasm_to_pod(&s->frames[i], &frames[i]);
    if (n_frames) *n_frames = s->frames_n;
    if (blocks) {
        for (size_t i = 0; i < s->blocks_n && i < blocks_cap; i++) block_to_rec(&s->blocks[i], &blocks[i]);
        free(g_asm); free(g_asm_frames);
        g_asm = (sdv_asm_line_rec *)malloc((s->asm_n + 1) * sizeof(*g_asm)); g_asm_n = s->asm_n;
        for (size_t i = 0; i < s->asm_n; i++) asm_to_rec(&s->asm_lines[i], &g_asm[i]);
        g_asm_frames = (uint32_t *)malloc((s->asm_frames + 1) * sizeof(uint32_t)); g_asm_frames_n = s->asm_frames;
        for (size_t i = 0; i < s->asm_frames; i++) g_asm_frames[i] = (uint32_t)s->asm_frame_n[i];
    }
    if (n_blocks) *n_blocks = s->blocks_n;
    orc_stitcher_free(s);
    free(s);
    return n;
}
