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
