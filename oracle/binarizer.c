/*
 * binarizer.c - CPU restatement of the per-line STC-007 binarizer of SDVPCMdecoder.
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).  Restates, with the reference's integer
 * widths and control flow:
 *   PCMLine            pcmline.cpp:96-519
 *   STC007Line         stc007line.cpp:64-680, 1051-1057
 *   CoordinatePair     frametrimset.cpp:61-164
 *   Binarizer          binarizer.cpp:48-8055 (STC-007 branches only)
 */
#include "sdv_oracle.h"
#include "bin_internal.h"
#include <string.h>

/* ------------------------------------------------------------------ CRC-16 CCITT-FALSE */
/* pcmline.cpp:461-487 getCalcCRC16: bit-serial, MSB first over the low bit_cnt bits. */
uint16_t orc_crc16_update(uint16_t crc, uint16_t in_data, uint8_t bit_cnt)
{
    for (uint8_t i = 0; i < bit_cnt; i++) {
        bool msb = (crc & 0x8000) != 0;
        bool inb = (in_data & (1 << (bit_cnt - 1))) != 0;
        crc = (uint16_t)(crc << 1);
        if (msb != inb) crc ^= ORC_CRC_POLY;
        in_data = (uint16_t)(in_data << 1);
    }
    return crc;
}

/* stc007line.cpp:245-251 */
uint16_t orc_stc_crc_words(const uint16_t *w)
{
    uint16_t crc = ORC_CRC_INIT;
    for (int i = 0; i <= ORC_STC_WORD_Q; i++) crc = orc_crc16_update(crc, w[i], ORC_STC_BITS_PER_WORD);
    return crc;
}

/* pcmline.h:88-97: check value 0x29B1 for "123456789" */
uint16_t orc_crc16_bytes(const uint8_t *data, size_t n)
{
    uint16_t crc = ORC_CRC_INIT;
    for (size_t i = 0; i < n; i++) crc = orc_crc16_update(crc, data[i], 8);
    return crc;
}

/* ------------------------------------------------------------------ CoordinatePair */
void orc_coords_clear(orc_coords *c)            /* frametrimset.cpp:101-106 */
{
    c->reference = 0; c->data_start = ORC_NO_COORD_LEFT; c->data_stop = ORC_NO_COORD_RIGHT;
    c->from_doubled = c->not_sure = false;
}
bool orc_coords_set(orc_coords *c, int16_t start, int16_t stop)   /* :109-118 */
{
    if (stop > start) { c->data_start = start; c->data_stop = stop; return true; }
    return false;
}
bool orc_coords_valid(const orc_coords *c)      /* :153-156 */
{
    return (c->data_start != ORC_NO_COORD_LEFT) && (c->data_stop != ORC_NO_COORD_RIGHT) && (c->data_start < c->data_stop);
}
bool orc_coords_lt(const orc_coords *a, const orc_coords *b)   /* :63-98 */
{
    if (a->data_start < b->data_start) return true;
    if (a->data_start == b->data_start) {
        if (a->data_stop > b->data_stop) return true;
        if (a->data_stop == b->data_stop) return a->reference < b->reference;
    }
    return false;
}
bool coords_ne(const orc_coords *a, const orc_coords *b)  /* :39-46 */
{
    return (a->data_start != b->data_start) || (a->data_stop != b->data_stop) || (a->from_doubled != b->from_doubled);
}

/* ------------------------------------------------------------------ PCMLine / STC007Line */
static void pcmline_clear(orc_stc_line *l)      /* pcmline.cpp:96-116 */
{
    l->frame_number = 0; l->line_number = 0;
    l->black_level = l->white_level = 0;
    l->ref_low = l->ref_level = l->ref_high = 0;
    orc_coords_clear(&l->coords);
    l->hysteresis_depth = l->shift_stage = 0;
    l->ref_level_sweeped = l->coords_sweeped = l->data_by_ext_tune = false;
    l->calc_crc = 0;
    l->blk_wht_set = l->coords_set = l->forced_bad = false;
    l->service_type = ORC_SRV_NO;
    l->pixel_start = 0; l->pixel_stop = 1; l->pixel_start_offset = 0;
    l->pixel_size_mult = ORC_INT_CALC_MULT;
    l->halfpixel_size_mult = l->pixel_size_mult / 2;
}

void orc_stc_calc_crc(orc_stc_line *l) { l->calc_crc = orc_stc_crc_words(l->words); }

void orc_stc_set_silent(orc_stc_line *l)        /* stc007line.cpp:138-155 */
{
    for (int i = 0; i <= ORC_STC_WORD_Q; i++) l->words[i] = l->m2_format ? (1 << 13) : 0;
    orc_stc_calc_crc(l);
}

void orc_stc_set_invalid_crc(orc_stc_line *l)   /* pcmline.cpp:189-193 */
{
    l->words[ORC_STC_WORD_CRC] = (uint16_t)~l->calc_crc;
}

void orc_stc_clear(orc_stc_line *l)             /* stc007line.cpp:64-93 */
{
    pcmline_clear(l);
    l->mark_st_stage = ORC_MARK_ST_START;
    l->mark_ed_stage = ORC_MARK_ED_START;
    l->marker_start_bg_coord = l->marker_start_ed_coord = l->marker_stop_ed_coord = 0;
    l->m2_format = false;
    memset(l->pixel_coordinates, 0, sizeof(l->pixel_coordinates));
    orc_stc_set_silent(l);
    for (int i = 0; i < ORC_STC_WORD_CNT; i++) l->word_crc[i] = l->word_valid[i] = false;
    l->calc_crc = ORC_STC_CRC_SILENT;
    orc_stc_set_invalid_crc(l);
}

/* pcmline.cpp:489-503 setServiceLine + :118-171 setServ*: keeps frame/line numbers.
 * NOTE PCMLine::clear() is non-virtual and called on the base, so the STC007Line
 * members (words, markers, flags) are NOT reset by a service conversion. */
void orc_stc_set_service(orc_stc_line *l, uint8_t service_type)
{
    uint32_t frame = l->frame_number; uint16_t line = l->line_number;
    pcmline_clear(l);
    l->frame_number = frame; l->line_number = line;
    l->service_type = service_type;
}

void orc_stc_set_serv_ctrl_blk(orc_stc_line *l)  /* stc007line.cpp:96-129 */
{
    uint16_t id_word = l->words[4], a1 = l->words[5], a2 = l->words[6], ctrl = l->words[7];
    uint32_t frame = l->frame_number; uint16_t line = l->line_number;
    orc_stc_clear(l);
    l->frame_number = frame; l->line_number = line;
    l->words[4] = id_word; l->words[5] = a1; l->words[6] = a2; l->words[7] = ctrl;
    orc_stc_calc_crc(l);
    l->words[ORC_STC_WORD_CRC] = l->calc_crc;
    l->service_type = ORC_SRV_CTRL_BLOCK;
}

static void stc_set_word(orc_stc_line *l, uint8_t index, uint16_t w, bool valid)  /* :158-173 */
{
    if (index < ORC_STC_WORD_CNT) {
        l->words[index] = (index == ORC_STC_WORD_CRC) ? w : (uint16_t)(w & ORC_STC_WORD_MASK);
        l->word_crc[index] = l->word_valid[index] = valid;
    }
}

bool orc_stc_crc_valid_ignore_forced(const orc_stc_line *l) { return l->calc_crc == l->words[ORC_STC_WORD_CRC]; } /* :506-513 */
bool orc_stc_crc_valid(const orc_stc_line *l) { return !l->forced_bad && orc_stc_crc_valid_ignore_forced(l); }     /* pcmline.cpp:367-374 */
static bool stc_has_start(const orc_stc_line *l) { return l->mark_st_stage == ORC_MARK_ST_BOT_2; }   /* :447-454 */
static bool stc_has_stop(const orc_stc_line *l) { return l->mark_ed_stage == ORC_MARK_ED_LEN_OK; }   /* :457-464 */
bool orc_stc_has_markers(const orc_stc_line *l) { return stc_has_start(l) && stc_has_stop(l); }      /* :467-474 */

bool orc_stc_has_control_block(const orc_stc_line *l)   /* :493-504 */
{
    return l->words[0] == 0x3333 && l->words[1] == 0x0CCC && l->words[2] == 0x3333 && l->words[3] == 0x0CCC
        && l->words[4] == 0x0000 && (l->words[7] & 0x0FF0) == 0;
}

void orc_stc_apply_crc_state_per_word(orc_stc_line *l)  /* :198-204 */
{
    bool v = orc_stc_crc_valid(l);
    for (int i = 0; i < ORC_STC_WORD_CNT; i++) l->word_crc[i] = l->word_valid[i] = v;
}

int16_t orc_stc_get_sample(const orc_stc_line *l, uint8_t index)   /* :282-326 */
{
    if (index > 5) return 0;
    uint16_t w = l->words[index];
    if (!l->m2_format) {
        w = (uint16_t)(w << 2);
    } else {
        if ((w & (1 << 13)) == 0) {
            w = (uint16_t)(w << 3);
        } else {
            bool pos = (w & (1 << 12)) == 0;
            w = (uint16_t)(w & ~(1 << 13));
            if (!pos) w |= (1 << 15) | (1 << 14) | (1 << 13);
        }
    }
    return (int16_t)w;
}

uint8_t orc_stc_words_diff_bit_count(const orc_stc_line *l, const orc_stc_line *o)  /* :329-357 */
{
    if (o == NULL) return 0;
    uint8_t bit_cnt = 0;
    for (int i = 0; i <= ORC_STC_WORD_Q; i++) {
        uint8_t diff_mask = (uint8_t)(l->words[i] ^ o->words[i]);      /* uint8_t truncation is the reference's */
        if (diff_mask != 0)
            for (uint8_t bit = 0; bit <= 16; bit++)
                if ((diff_mask & (1 << bit)) != 0) bit_cnt++;
    }
    return bit_cnt;
}

static bool stc_near_silence(const orc_stc_line *l, uint8_t index)   /* :568-582 */
{
    int16_t s = orc_stc_get_sample(l, index);
    if (s >= (int16_t)(1 << 4)) return false;
    if (s < (0 - (int16_t)(1 << 4))) return false;
    return true;
}
bool orc_stc_is_almost_silent(const orc_stc_line *l)   /* :585-599 */
{
    uint8_t n = 0;
    for (uint8_t i = 0; i <= 5; i++) if (stc_near_silence(l, i)) n++;
    return n >= 2;
}
bool orc_stc_is_silent(const orc_stc_line *l)          /* :602-612 */
{
    for (uint8_t i = 0; i <= 5; i++) if (orc_stc_get_sample(l, i) != 0) return false;
    return true;
}

static void stc_set_source_pixels(orc_stc_line *l, uint16_t in_start, uint16_t in_stop)  /* pcmline.cpp:202-213 */
{
    if (in_stop > in_start)
        if ((3 + ORC_STC_BITS_DATA + 1) <= (in_stop - in_start)) { l->pixel_start = in_start; l->pixel_stop = in_stop; }
}

/* pcmline.cpp:506-519 setPPB; getBitsBetweenDataCoordinates = 132 (stc007line.cpp:219-223) */
static void stc_set_ppb(orc_stc_line *l, orc_coords c)
{
    uint8_t bit_count = 3 + ORC_STC_BITS_DATA + 1;
    l->pixel_size_mult = (uint32_t)(c.data_stop - c.data_start);
    l->pixel_size_mult = (l->pixel_size_mult * ORC_INT_CALC_MULT + bit_count / 2) / bit_count;
    l->pixel_start_offset = c.data_start;
    l->halfpixel_size_mult = (l->pixel_size_mult + 2 / 2) / 2;
}

/* pcmline.cpp:249-311 getVideoPixeBylCalc */
static const int8_t PIX_SH_BG_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:63-66 */
static const int8_t PIX_SH_ED_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:68-71 */
static uint16_t stc_pixel_by_calc(const orc_stc_line *l, uint8_t pcm_bit, uint8_t in_shift, uint8_t bit_ofs)
{
    int32_t video_pixel;
    pcm_bit = (uint8_t)(pcm_bit + bit_ofs);
    if (pcm_bit >= ORC_STC_BITS_IN_LINE) pcm_bit = ORC_STC_BITS_IN_LINE - 1;
    video_pixel = (int32_t)((pcm_bit * l->pixel_size_mult) + l->halfpixel_size_mult);
    video_pixel = video_pixel / ORC_INT_CALC_MULT;
    video_pixel = video_pixel + l->pixel_start_offset;
    int8_t bg = PIX_SH_BG_TBL[in_shift], ed = PIX_SH_ED_TBL[in_shift];
    if (bg == ed) video_pixel += bg;
    else if (pcm_bit < ORC_STC_BITS_LEFT_SHIFT) video_pixel += bg;
    else if (pcm_bit > ORC_STC_BITS_RIGHT_SHIFT) video_pixel += ed;
    if (video_pixel < l->pixel_start) video_pixel = l->pixel_start;
    else if (video_pixel >= l->pixel_stop) video_pixel = l->pixel_stop - 1;
    return (uint16_t)video_pixel;
}

void orc_stc_calc_ppb(orc_stc_line *l, orc_coords c)    /* pcmline.cpp:223-232 + stc007line.cpp:1051-1057 */
{
    stc_set_ppb(l, c);
    for (uint8_t s = 0; s < ORC_PS_STAGES; s++)
        for (uint8_t bit = 0; bit < ORC_STC_BITS_DATA; bit++)
            l->pixel_coordinates[s][bit] = stc_pixel_by_calc(l, bit, s, ORC_STC_BITS_START - 1);
}

/* ------------------------------------------------------------------ Binarizer: settings */
void orc_bin_preset_reset(orc_bin_preset *p)    /* binarizer.cpp:48-65 */
{
    p->max_black_lvl = 160; p->min_white_lvl = 28; p->min_contrast = 10; p->min_ref_lvl = 7; p->max_ref_lvl = 240;
    p->min_valid_crcs = 5; p->mark_max_dist = 6; p->left_bit_pick = 4; p->right_bit_pick = 2;
    orc_coords_clear(&p->horiz_coords);
    p->horiz_coords.data_start = p->horiz_coords.data_stop = 0;
    p->en_force_coords = false; p->en_coord_search = true; p->en_first_line_dup = true; p->en_good_no_marker = true;
}

void reset_crc_stats(orc_crc_handler *a, uint16_t count, uint8_t *valid_cnt)   /* :1771-1786 */
{
    for (uint16_t i = 0; i < count; i++) {
        a[i].result = 0; a[i].data_start = a[i].data_stop = 0; a[i].crc = 0; a[i].hyst_dph = a[i].shift_stg = 0x0f;
    }
    if (valid_cnt) *valid_cnt = 0;
}

void orc_binarizer_set_mode(orc_binarizer *b, uint8_t in_mode)   /* :120-152 */
{
    if (in_mode == ORC_MODE_DRAFT) { b->bin_mode = in_mode; b->in_max_hysteresis_depth = ORC_HYST_DEPTH_SAFE; b->in_max_shift_stages = ORC_SHIFT_STAGES_MIN; }
    else if (in_mode == ORC_MODE_FAST) { b->bin_mode = in_mode; b->in_max_hysteresis_depth = 7; b->in_max_shift_stages = ORC_SHIFT_STAGES_SAFE; }
    else if (in_mode == ORC_MODE_INSANE) { b->bin_mode = in_mode; b->in_max_hysteresis_depth = ORC_HYST_DEPTH_MAX; b->in_max_shift_stages = ORC_SHIFT_STAGES_MAX; }
    else { b->bin_mode = ORC_MODE_NORMAL; b->in_max_hysteresis_depth = ORC_HYST_DEPTH_SAFE; b->in_max_shift_stages = ORC_SHIFT_STAGES_SAFE; }
}

void orc_binarizer_init(orc_binarizer *b)       /* :67-99 */
{
    memset(b, 0, sizeof(*b));
    orc_bin_preset_reset(&b->digi_set);
    b->video_line = NULL; b->out_pcm_line = NULL;
    b->in_def_black = b->in_def_reference = b->in_def_white = 0;
    orc_coords_clear(&b->in_def_coord);
    b->in_max_hysteresis_depth = ORC_HYST_DEPTH_SAFE;
    b->in_max_shift_stages = ORC_SHIFT_STAGES_MIN;
    b->do_coord_search = true; b->do_start_mark_sweep = true; b->do_ref_lvl_sweep = false; b->force_bit_picker = true;
    b->proc_state = ORC_STG_REF_FIND;
    orc_binarizer_set_mode(b, ORC_MODE_FAST);
    b->line_part_mode = 0;
    b->hysteresis_depth_lim = ORC_HYST_DEPTH_MIN; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN;
    b->line_length = 0; b->scan_start = b->scan_end = 0; b->mark_start_max = 0; b->mark_end_min = 0xFFFF;
    b->estimated_ppb = 0; b->was_BW_scanned = false;
    reset_crc_stats(b->shift_crcs, ORC_SHIFT_STAGES_MAX + 1, NULL);
    reset_crc_stats(b->hyst_crcs, ORC_HYST_DEPTH_MAX + 1, NULL);
    reset_crc_stats(b->crc_stats, ORC_MAX_COLL_CRCS + 1, NULL);
}

void orc_binarizer_set_bw_levels(orc_binarizer *b, uint8_t in_black, uint8_t in_white)   /* :240-273 */
{
    if ((in_black < in_white) && (in_black < b->digi_set.max_black_lvl) && (in_white > b->digi_set.min_white_lvl) && (in_white != 0)) {
        b->in_def_black = in_black; b->in_def_white = in_white;
    } else {
        b->in_def_black = b->in_def_white = 0;
    }
}
void orc_binarizer_set_reference_level(orc_binarizer *b, uint8_t ref) { b->in_def_reference = ref; }   /* :276-298 */
void orc_binarizer_set_data_coordinates(orc_binarizer *b, orc_coords c)   /* :323-350 */
{
    if (orc_coords_valid(&c)) b->in_def_coord = c; else orc_coords_clear(&b->in_def_coord);
}
void orc_binarizer_set_data_coordinates2(orc_binarizer *b, int16_t s, int16_t e)   /* :301-320 */
{
    orc_coords tmp; orc_coords_clear(&tmp);
    if ((s < e) && (e != 0) && (s != ORC_NO_COORD_LEFT) && (e != ORC_NO_COORD_RIGHT)) orc_coords_set(&tmp, s, e);
    orc_binarizer_set_data_coordinates(b, tmp);
}
void orc_binarizer_set_good_parameters(orc_binarizer *b, const orc_stc_line *l)   /* :353-377 */
{
    if (l == NULL) {
        orc_binarizer_set_reference_level(b, 0);
        orc_binarizer_set_data_coordinates2(b, 0, 0);
        orc_binarizer_set_bw_levels(b, 0, 0);
    } else if (orc_stc_crc_valid_ignore_forced(l)) {
        orc_binarizer_set_reference_level(b, l->ref_level);
        orc_binarizer_set_data_coordinates(b, l->coords);
        orc_binarizer_set_bw_levels(b, l->black_level, l->white_level);
    }
}
bool is_ref_level_preset(const orc_binarizer *b) { return b->in_def_reference >= b->digi_set.min_ref_lvl; }  /* :428-438 */
bool are_bw_levels_preset(const orc_binarizer *b)   /* :406-425 */
{
    if ((b->in_def_white > b->digi_set.min_white_lvl) && (b->in_def_black < b->digi_set.max_black_lvl)) {
        if (is_ref_level_preset(b))
            if ((b->in_def_reference <= b->in_def_black) || (b->in_def_reference >= b->in_def_white)) return false;
        return true;
    }
    return false;
}

/* ------------------------------------------------------------------ CRC statistics */
void update_crc_stats(orc_crc_handler *a, orc_crc_handler in, uint8_t *valid_cnt)   /* :1789-1826 */
{
    bool found = false;
    if (*valid_cnt >= ORC_MAX_COLL_CRCS) *valid_cnt = ORC_MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= *valid_cnt; i++)
        if (a[i].crc == in.crc) { a[i].result++; found = true; break; }
    if (!found) {
        (*valid_cnt)++;
        if (*valid_cnt < ORC_MAX_COLL_CRCS) {
            a[*valid_cnt].crc = in.crc; a[*valid_cnt].hyst_dph = in.hyst_dph; a[*valid_cnt].shift_stg = in.shift_stg;
            a[*valid_cnt].result++;
        }
    }
}

void find_most_frequent_crc(orc_crc_handler *a, uint8_t *valid_cnt, bool skip_equal)   /* :1829-1928 */
{
    a[0].result = 0; a[0].data_start = 0; a[0].data_stop = 0; a[0].hyst_dph = 0; a[0].shift_stg = 0;
    if (*valid_cnt >= ORC_MAX_COLL_CRCS) *valid_cnt = ORC_MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= *valid_cnt; i++)
        if (a[i].result > a[0].result) {
            a[0].result = a[i].result; a[0].crc = a[i].crc; a[0].hyst_dph = a[i].hyst_dph; a[0].shift_stg = a[i].shift_stg;
            a[0].data_start = i;
        }
    if (skip_equal)
        for (uint8_t i = 1; i <= *valid_cnt; i++)
            if (a[0].data_start != i)
                if (a[0].result <= (2 * a[i].result)) { a[0].result = 0; a[0].hyst_dph = 0; a[0].shift_stg = 0; break; }
    if (a[0].result == 0) *valid_cnt = 0;
}

void invalidate_non_frequent_crcs(orc_crc_handler *a, uint8_t low_level, uint8_t high_level, uint8_t valid_cnt, uint16_t target_crc)  /* :1931-1982 */
{
    uint8_t index = high_level;
    while (index >= low_level) {
        if (a[index].result == ORC_REF_CRC_OK)
            if ((valid_cnt == 0) || (a[index].crc != target_crc)) a[index].result = ORC_REF_CRC_COLL;
        if (index == low_level) break;
        index--;
    }
}

uint8_t pick_level_by_crc_stats(const orc_crc_handler *crcs, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                       uint8_t target_result, uint8_t max_hyst, uint8_t max_shift)   /* :1985-2140 */
{
    bool good_ref_det = false, range_lock = false, second_start_lock = false;
    uint8_t index, low_depth = 0xFF, low_shift = 0xFF;
    uint8_t low_ref = 0, high_ref = 0, tst_low_ref = 0, tst_high_ref = 0, picked_ref;
    if (ref_result == NULL) return ORC_SPAN_NOT_FOUND;
    index = high_lvl;
    while (index >= low_lvl) {
        if ((crcs[index].result == target_result) && (crcs[index].hyst_dph <= max_hyst) && (crcs[index].shift_stg <= max_shift)) {
            good_ref_det = true;
            if (crcs[index].hyst_dph < low_depth) { low_depth = crcs[index].hyst_dph; low_shift = crcs[index].shift_stg; high_ref = index; }
            else if (crcs[index].hyst_dph == low_depth)
                if (crcs[index].shift_stg < low_shift) { low_shift = crcs[index].shift_stg; high_ref = index; }
        }
        if (index == low_lvl) break;
        index--;
    }
    if (!good_ref_det) return ORC_SPAN_NOT_FOUND;
    index = high_ref;
    while (index >= low_lvl) {
        if ((crcs[index].result == target_result) && (crcs[index].hyst_dph == low_depth) && (crcs[index].shift_stg == low_shift)) {
            if (!range_lock) low_ref = index;
            else {
                if (!second_start_lock) { tst_high_ref = index; second_start_lock = true; }
                tst_low_ref = index;
            }
        } else {
            range_lock = true;
            if (second_start_lock) {
                second_start_lock = false;
                if ((tst_high_ref - tst_low_ref) >= (high_ref - low_ref)) { low_ref = tst_low_ref; high_ref = tst_high_ref; }
            }
        }
        if (index == low_lvl) break;
        index--;
    }
    picked_ref = (uint8_t)(high_ref - low_ref);
    picked_ref = picked_ref / 2;
    picked_ref = (uint8_t)(low_ref + picked_ref);
    *ref_result = picked_ref;
    return ORC_SPAN_OK;
}

uint8_t pick_level_by_crc_stats_opt(const orc_binarizer *b, const orc_crc_handler *crcs, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                           uint8_t target_result, uint8_t max_hyst, uint8_t max_shift)   /* :2143-2383 */
{
    bool range_lock = false, good_ref_det = false;
    uint8_t index, hold_cnt, same_cnt, low_depth, low_shift = 0, high_shift = 0;
    uint8_t low_ref = 0, high_ref = 0, picked_ref;
    if (ref_result == NULL) return ORC_SPAN_NOT_FOUND;
    index = high_lvl;
    while (index >= low_lvl) {
        if ((crcs[index].result == target_result) && (crcs[index].hyst_dph <= max_hyst) && (crcs[index].shift_stg <= max_shift)) {
            if (!good_ref_det) { good_ref_det = true; low_ref = high_ref = index; }
            else {
                low_ref = index;
                if (low_ref == low_lvl) { low_shift = low_ref; high_shift = high_ref; range_lock = true; }
            }
        } else {
            if (good_ref_det) {
                if ((high_ref - low_ref + 1) >= (high_shift - low_shift + 1)) { low_shift = low_ref; high_shift = high_ref; range_lock = true; }
                good_ref_det = false;
            }
        }
        if (index == low_lvl) break;
        index--;
    }
    if (range_lock) { high_lvl = high_shift; low_lvl = low_shift; }
    good_ref_det = false;
    hold_cnt = low_shift = high_shift = 0;
    low_depth = low_shift = 255;
    same_cnt = ORC_MIN_VALID_CRCS;
    low_ref = high_ref = picked_ref = b->digi_set.max_ref_lvl;
    index = high_lvl;
    while (index >= low_lvl) {
        if ((crcs[index].result == target_result) && (crcs[index].hyst_dph <= max_hyst) && (crcs[index].shift_stg <= max_shift)) {
            good_ref_det = true;
            if (low_depth > crcs[index].hyst_dph) {
                low_depth = crcs[index].hyst_dph; low_shift = crcs[index].shift_stg; low_ref = high_ref = index; hold_cnt = ORC_MIN_VALID_CRCS;
            } else if (low_depth == crcs[index].hyst_dph) {
                if (low_shift > crcs[index].shift_stg) {
                    low_shift = crcs[index].shift_stg; low_ref = high_ref = index; same_cnt = ORC_MIN_VALID_CRCS; hold_cnt = ORC_MIN_VALID_CRCS;
                } else if (low_shift == crcs[index].shift_stg) {
                    low_ref = index; same_cnt--;
                    if (same_cnt == 0) { hold_cnt = 0; break; }
                } else {
                    hold_cnt--;
                    if (hold_cnt == 0) break;
                }
            } else {
                hold_cnt--;
                if (hold_cnt == 0) break;
            }
        }
        if (index == low_lvl) break;
        index--;
    }
    (void)picked_ref;
    if (good_ref_det) {
        picked_ref = (uint8_t)(high_ref - low_ref);
        picked_ref = picked_ref / 2;
        picked_ref = (uint8_t)(low_ref + picked_ref);
        *ref_result = picked_ref;
        return ORC_SPAN_OK;
    }
    return ORC_SPAN_NOT_FOUND;
}

/* ------------------------------------------------------------------ AGC: BLACK / WHITE */
#define PIX(b, x) ((b)->video_line->pixels[(x)])

uint16_t most_frequent_brightness_count(const uint16_t *s)   /* :2450-2468 */
{
    uint16_t hf = 0;
    for (int lev = 255; lev >= 0; lev--) if (s[lev] > hf) hf = s[lev];
    return hf;
}

uint8_t usefull_low_level(const orc_binarizer *b, const uint16_t *s)   /* :2471-2513 */
{
    bool filtered_found = false;
    uint8_t brt_lev = 0, lowest_lev = 0;
    uint16_t min_freq = most_frequent_brightness_count(s) / 64;
    while (brt_lev < b->digi_set.max_black_lvl) {
        if (s[brt_lev] > min_freq) { lowest_lev = brt_lev; filtered_found = true; break; }
        brt_lev++;
    }
    if (!filtered_found)
        while (brt_lev < b->digi_set.max_black_lvl) {
            if (s[brt_lev] > 0) { lowest_lev = brt_lev; break; }
            brt_lev++;
        }
    return lowest_lev;
}

uint8_t usefull_high_level(const orc_binarizer *b, const uint16_t *s)  /* :2516-2557 */
{
    bool filtered_found = false;      /* never set in the reference */
    uint8_t brt_lev = 255, highest_lev = 255;
    uint16_t min_freq = most_frequent_brightness_count(s) / 64;
    while (brt_lev >= b->digi_set.min_white_lvl) {
        if (s[brt_lev] > min_freq) { highest_lev = brt_lev; break; }
        brt_lev--;
    }
    if (!filtered_found)
        while (brt_lev >= b->digi_set.min_white_lvl) {
            if (s[brt_lev] > 0) { highest_lev = brt_lev; break; }
            brt_lev--;
        }
    return highest_lev;
}

uint8_t get_low_level(uint8_t in_lvl, uint8_t diff)  { return (in_lvl > diff) ? (uint8_t)(in_lvl - diff) : 1; }          /* :3476-3487 */
uint8_t get_high_level(uint8_t in_lvl, uint8_t diff) { return (in_lvl < (255 - diff)) ? (uint8_t)(in_lvl + diff) : 254; } /* :3490-3501 */

uint8_t pick_center_ref_level(const orc_binarizer *b, uint8_t lvl_black, uint8_t lvl_white)   /* :3504-3548 */
{
    uint8_t br_delta = (uint8_t)(lvl_white - lvl_black), res_lvl;
    if (br_delta >= b->digi_set.min_contrast) {
        br_delta = br_delta / 2;
        res_lvl = (uint8_t)(br_delta + lvl_black);
        if (res_lvl < b->digi_set.min_ref_lvl) res_lvl = b->digi_set.min_ref_lvl;
        else if (res_lvl > b->digi_set.max_ref_lvl) res_lvl = b->digi_set.max_ref_lvl;
    } else {
        res_lvl = (lvl_white < b->digi_set.max_ref_lvl) ? b->digi_set.max_ref_lvl : b->digi_set.min_ref_lvl;
    }
    return res_lvl;
}

static void find_stc007_bw(orc_binarizer *b, uint16_t *sprd)   /* :2684-3070 */
{
    uint8_t pixel_val, brt_lev, marker_detect_stage;
    uint8_t br_mark_white, useful_low, useful_high;
    uint8_t low_scan_limit, high_scan_limit, range_limit;
    uint8_t bin_level, bin_low, bin_high;
    uint16_t pixel, pixel_limit, mark_ed_bit_start, mark_ed_bit_end, white_lvl_count, search_lim;
    uint32_t temp_calc;
    bool white_level_detected;
    orc_stc_line *line = b->out_pcm_line;

    search_lim = (uint16_t)(b->scan_start + b->estimated_ppb * 10);
    for (uint16_t p = b->scan_start; p < search_lim; p++) sprd[PIX(b, p)]++;
    search_lim = (uint16_t)(b->scan_end - b->estimated_ppb * 20);
    for (uint16_t p = search_lim; p <= b->scan_end; p++) sprd[PIX(b, p)]++;

    useful_low = low_scan_limit = usefull_low_level(b, sprd);
    useful_high = high_scan_limit = br_mark_white = usefull_high_level(b, sprd);
    range_limit = (uint8_t)(high_scan_limit - low_scan_limit);
    high_scan_limit = (uint8_t)(high_scan_limit - (range_limit / 4));
    bin_high = range_limit / 8;
    brt_lev = useful_high;
    white_lvl_count = 0;
    white_level_detected = false;
    while (brt_lev >= high_scan_limit) {
        if (sprd[brt_lev] > white_lvl_count) { white_lvl_count = sprd[brt_lev]; br_mark_white = brt_lev; white_level_detected = true; }
        if (white_level_detected)
            if ((br_mark_white - brt_lev) >= bin_high) break;
        brt_lev--;
    }
    pixel_limit = (uint16_t)(b->scan_end - b->scan_start);
    temp_calc = pixel_limit / 8;
    pixel_limit = (uint16_t)(b->scan_start + (uint16_t)temp_calc);
    search_lim = (uint16_t)(b->scan_end - (uint16_t)temp_calc);
    memset(sprd, 0, 256 * sizeof(uint16_t));
    for (uint16_t p = pixel_limit; p < search_lim; p++) sprd[PIX(b, p)]++;

    marker_detect_stage = ORC_MARK_ED_START;
    mark_ed_bit_start = mark_ed_bit_end = 0;
    if (white_level_detected) {
        bin_level = pick_center_ref_level(b, useful_low, br_mark_white);
        bin_high = bin_low = bin_level;
        if (b->mark_end_min > (b->estimated_ppb * 6)) pixel_limit = (uint16_t)(b->mark_end_min - b->estimated_ppb * 6);
        else pixel_limit = 0;
        pixel = b->scan_end;
        while (pixel > pixel_limit) {
            pixel_val = PIX(b, pixel);
            if (marker_detect_stage == ORC_MARK_ED_START) {
                if (pixel < b->mark_end_min) break;
                if (pixel_val >= bin_low) { mark_ed_bit_end = (uint16_t)(pixel + 1); marker_detect_stage = ORC_MARK_ED_TOP; }
            } else if (marker_detect_stage == ORC_MARK_ED_TOP) {
                if (pixel_val < bin_high) {
                    mark_ed_bit_start = (uint16_t)(pixel + 1);
                    marker_detect_stage = ORC_MARK_ED_BOT;
                    if ((mark_ed_bit_end - mark_ed_bit_start) >= (b->estimated_ppb * 2)) { marker_detect_stage = ORC_MARK_ED_LEN_OK; break; }
                    else marker_detect_stage = ORC_MARK_ED_START;
                }
            }
            pixel--;
        }
        line->mark_ed_stage = marker_detect_stage;
        line->coords.data_stop = (int16_t)mark_ed_bit_start;
        line->marker_stop_ed_coord = mark_ed_bit_end;
        if (stc_has_stop(line)) {
            uint16_t sprd_cnt;
            search_lim = (uint16_t)(b->estimated_ppb * 64);
            if (search_lim > mark_ed_bit_start) search_lim = b->mark_start_max;
            else search_lim = (uint16_t)(mark_ed_bit_start - search_lim);
            memset(sprd, 0, 256 * sizeof(uint16_t));
            sprd_cnt = 0;
            for (uint16_t p = (uint16_t)(mark_ed_bit_start - 1); p > search_lim; p--) { sprd[PIX(b, p)]++; sprd_cnt++; }
            if (sprd_cnt < 32) {
                pixel_limit = (uint16_t)(b->scan_end - b->scan_start);
                pixel_limit = pixel_limit / 8;
                search_lim = (uint16_t)(b->scan_end - pixel_limit);
                for (uint16_t p = pixel_limit; p < search_lim; p++) sprd[PIX(b, p)]++;
            }
        }
    }
}

static bool find_black_white(orc_binarizer *b)   /* :3116-3473 (STC-007 branch of the type switch) */
{
    uint8_t brt_lev, br_black = 0, br_white = 255, useful_low, useful_high;
    uint8_t low_scan_limit, high_scan_limit, range_limit, bin_low, bin_high;
    uint16_t black_lvl_count, white_lvl_count, search_lim;
    uint32_t temp_calc;
    uint16_t sprd[256];
    bool black_level_detected, white_level_detected;
    orc_stc_line *line = b->out_pcm_line;

    memset(sprd, 0, sizeof(sprd));
    find_stc007_bw(b, sprd);

    useful_low = low_scan_limit = br_black = usefull_low_level(b, sprd);
    useful_high = high_scan_limit = br_white = usefull_high_level(b, sprd);
    range_limit = (uint8_t)(high_scan_limit - low_scan_limit);
    low_scan_limit = (uint8_t)(low_scan_limit + (range_limit / 3));
    high_scan_limit = (uint8_t)(high_scan_limit - (range_limit / 3));
    temp_calc = range_limit; temp_calc = temp_calc * 10 / 100; bin_low = (uint8_t)temp_calc;
    temp_calc = range_limit; temp_calc = temp_calc * 12 / 100; bin_high = (uint8_t)temp_calc;
    search_lim = most_frequent_brightness_count(sprd);
    search_lim = search_lim / 64;

    brt_lev = useful_low; black_lvl_count = 0; black_level_detected = false;
    while (brt_lev <= low_scan_limit) {
        if (sprd[brt_lev] > black_lvl_count) {
            black_lvl_count = sprd[brt_lev];
            if (black_lvl_count > search_lim) { br_black = brt_lev; black_level_detected = true; }
        }
        if (black_level_detected)
            if ((brt_lev - br_black) >= bin_low) break;
        brt_lev++;
    }
    brt_lev = useful_high; white_lvl_count = 0; white_level_detected = false;
    if (black_level_detected) {
        while (brt_lev >= high_scan_limit) {
            if (brt_lev < (br_black + b->digi_set.min_contrast)) break;
            if (sprd[brt_lev] > white_lvl_count) {
                white_lvl_count = sprd[brt_lev];
                if (white_lvl_count > search_lim) { br_white = brt_lev; white_level_detected = true; }
            }
            if (white_level_detected)
                if ((br_white - brt_lev) >= bin_high) break;
            brt_lev--;
        }
    }
    if (black_level_detected && white_level_detected) {
        bool invalidate = false;
        if (br_white < br_black) invalidate = true;
        else if ((br_white - br_black) < b->digi_set.min_contrast) invalidate = true;
        else if (b->do_ref_lvl_sweep && ((br_white - br_black) < b->digi_set.min_valid_crcs)) invalidate = true;
        else if (br_black > b->digi_set.max_black_lvl) invalidate = true;
        else if (br_white < b->digi_set.min_white_lvl) invalidate = true;
        if (invalidate) { black_level_detected = white_level_detected = false; br_black = useful_low; br_white = useful_high; }
    }
    b->was_BW_scanned = true;
    line->black_level = br_black;
    line->white_level = br_white;
    if (!black_level_detected || !white_level_detected) { line->blk_wht_set = false; return false; }
    line->blk_wht_set = true;
    return true;
}

/* ------------------------------------------------------------------ Macro-TBC: markers */
static void search_stc007_markers(orc_binarizer *b, orc_stc_line *stc, uint8_t hyst_lvl)   /* :5275-5595 */
{
    uint8_t marker_detect_stage = ORC_MARK_ST_START, pixel_val, bin_level, bin_low, bin_high;
    uint16_t pixel, pixel_limit;
    uint16_t st1s = 0, st1e = 0, st3s = 0, st3e = 0, ed_start, ed_end;

    bin_level = stc->ref_level;
    bin_low = get_low_level(bin_level, hyst_lvl);
    if (bin_low < b->digi_set.min_ref_lvl) bin_low = b->digi_set.min_ref_lvl;
    bin_high = bin_level;
    pixel_limit = (uint16_t)(b->mark_start_max + b->estimated_ppb * 5);
    if (pixel_limit > b->line_length) pixel_limit = b->line_length;
    pixel = b->scan_start;
    while (pixel < pixel_limit) {
        pixel_val = PIX(b, pixel);
        if (marker_detect_stage == ORC_MARK_ST_START) {
            if (pixel > b->mark_start_max) break;
            if (pixel_val >= bin_low) { st1s = pixel; marker_detect_stage = ORC_MARK_ST_TOP_1; }
        } else if (marker_detect_stage == ORC_MARK_ST_TOP_1) {
            if (pixel_val < bin_low) { st1e = pixel; marker_detect_stage = ORC_MARK_ST_BOT_1; }
        } else if (marker_detect_stage == ORC_MARK_ST_BOT_1) {
            if (pixel_val >= bin_high) {
                st3s = pixel;
                if (((st3s - st1e) > (b->estimated_ppb * 2)) || ((st3s - st1e) < (b->estimated_ppb / 2))) marker_detect_stage = ORC_MARK_ST_START;
                else marker_detect_stage = ORC_MARK_ST_TOP_2;
            }
        } else if (marker_detect_stage == ORC_MARK_ST_TOP_2) {
            if (pixel_val < bin_high) {
                st3e = pixel;
                if (((st3e - st3s) > (b->estimated_ppb * 2)) || ((st3e - st3s) < (b->estimated_ppb / 2))) marker_detect_stage = ORC_MARK_ST_START;
                else { marker_detect_stage = ORC_MARK_ST_BOT_2; break; }
            }
        }
        pixel++;
    }
    stc->mark_st_stage = marker_detect_stage;
    stc->marker_start_bg_coord = st1s;
    stc->marker_start_ed_coord = st3e;

    marker_detect_stage = ORC_MARK_ED_START;
    ed_start = ed_end = 0;
    if (stc_has_start(stc)) {
        bin_low = bin_level;
        if (b->mark_end_min > (b->estimated_ppb * 6)) pixel_limit = (uint16_t)(b->mark_end_min - b->estimated_ppb * 6);
        else pixel_limit = 0;
        pixel = b->scan_end;
        while (pixel > pixel_limit) {
            pixel_val = PIX(b, pixel);
            if (marker_detect_stage == ORC_MARK_ED_START) {
                if (pixel < b->mark_end_min) break;
                if (pixel_val >= bin_low) { ed_end = (uint16_t)(pixel + 1); marker_detect_stage = ORC_MARK_ED_TOP; }
            } else if (marker_detect_stage == ORC_MARK_ED_TOP) {
                if (pixel_val < bin_high) {
                    ed_start = (uint16_t)(pixel + 1);
                    marker_detect_stage = ORC_MARK_ED_BOT;
                    if (((ed_end - ed_start) >= (b->estimated_ppb * 2)) && ((ed_end - ed_start) <= (b->estimated_ppb * 5))) { marker_detect_stage = ORC_MARK_ED_LEN_OK; break; }
                    else marker_detect_stage = ORC_MARK_ED_START;
                }
            }
            pixel--;
        }
        stc->mark_ed_stage = marker_detect_stage;
    }
    orc_coords_set(&stc->coords, (int16_t)st1e, (int16_t)ed_start);
    stc->marker_stop_ed_coord = ed_end;
    stc->coords_set = orc_stc_has_markers(stc);
}

static void find_stc007_coordinates(orc_binarizer *b, orc_stc_line *stc)   /* :6047-6113 */
{
    uint8_t best_hyst = 0;
    if (b->do_start_mark_sweep) {
        orc_stc_line temp_line = *stc;
        orc_coords best; bool have = false;
        for (uint8_t h = 0; h < 24; h++) {
            search_stc007_markers(b, &temp_line, h);
            if (orc_stc_has_markers(&temp_line)) {
                orc_coords t = temp_line.coords; t.reference = h; t.not_sure = false;
                /* std::sort + [0]  ==  minimum under CoordinatePair::operator< (keys distinct by reference) */
                if (!have || orc_coords_lt(&t, &best)) { best = t; have = true; }
            }
        }
        if (have) best_hyst = best.reference;
    }
    search_stc007_markers(b, stc, best_hyst);
}

/* ------------------------------------------------------------------ bit extraction */
static uint8_t fill_stc007(orc_binarizer *b, orc_stc_line *l, uint8_t shift_stg)   /* :7322-7445 */
{
    bool prev_high = false;
    uint8_t pcm_bit = 0, pixel_val, low_ref = l->ref_low, high_ref = l->ref_high;
    uint8_t word_bit_pos = ORC_STC_BITS_PER_WORD - 1, word_index = 0;
    uint16_t pcm_word = 0;
    while (pcm_bit <= (ORC_STC_BITS_DATA - 1)) {
        pixel_val = PIX(b, l->pixel_coordinates[shift_stg][pcm_bit]);
        if (!prev_high) {
            if (pixel_val > low_ref) { pcm_word |= (uint16_t)(1 << word_bit_pos); prev_high = true; }
        } else {
            if (pixel_val >= high_ref) pcm_word |= (uint16_t)(1 << word_bit_pos);
            else prev_high = false;
        }
        if (word_bit_pos == 0) {
            if (pcm_bit > (ORC_STC_BITS_DATA - 16 - 1)) l->words[ORC_STC_WORD_CRC] = pcm_word;
            else stc_set_word(l, word_index, pcm_word, false);
            pcm_word = 0;
            word_index++;
            if (pcm_bit > (ORC_STC_BITS_DATA - 16 - 1)) break;
            else if (pcm_bit == (ORC_STC_BITS_DATA - 16 - 1)) word_bit_pos = 16;
            else word_bit_pos = ORC_STC_BITS_PER_WORD;
        }
        word_bit_pos--;
        pcm_bit++;
    }
    orc_stc_calc_crc(l);
    return ORC_STG_DATA_OK;
}

static uint8_t fill_data_words(orc_binarizer *b, orc_stc_line *l, uint8_t ref_delta, uint8_t shift_stg)   /* :7560-7691 */
{
    uint8_t low_ref, high_ref;
    if (ref_delta > ORC_HYST_DEPTH_MAX) return ORC_STG_NO_GOOD;
    if (shift_stg > ORC_SHIFT_STAGES_MAX) return ORC_STG_NO_GOOD;
    low_ref = get_low_level(l->ref_level, ref_delta);
    high_ref = get_high_level(l->ref_level, ref_delta);
    l->ref_low = low_ref; l->ref_high = high_ref;
    if (low_ref <= l->black_level) { orc_stc_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    if (high_ref >= l->white_level) { orc_stc_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    l->hysteresis_depth = ref_delta;
    l->shift_stage = shift_stg;
    return fill_stc007(b, l, shift_stg);
}

static void read_pcm_data(orc_binarizer *b, orc_stc_line *l)   /* :7695-8055 */
{
    bool invalid_hyst;
    uint8_t hyst_cnt, shift_try_cnt, valid_crcs_hyst, valid_crcs_shift, hyst_good_cnt;
    uint8_t valid_delta, valid_shift;

    orc_stc_calc_ppb(l, l->coords);
    if (b->hysteresis_depth_lim > ORC_HYST_DEPTH_MAX) b->hysteresis_depth_lim = ORC_HYST_DEPTH_MAX;
    if (b->shift_stages_lim > ORC_SHIFT_STAGES_MAX) b->shift_stages_lim = ORC_SHIFT_STAGES_MAX;

    if (!l->ref_level_sweeped) {
        hyst_cnt = (uint8_t)(b->hysteresis_depth_lim + 1);
        while (hyst_cnt > 0) { hyst_cnt--; b->hyst_crcs[hyst_cnt].result = ORC_REF_BAD_CRC; }
        valid_delta = hyst_good_cnt = 0;
        hyst_cnt = 0;
        do {
            invalid_hyst = false;
            reset_crc_stats(b->crc_stats, ORC_MAX_COLL_CRCS + 1, &valid_crcs_shift);
            b->crc_stats[0].hyst_dph = 0; b->crc_stats[0].shift_stg = 0;
            shift_try_cnt = (uint8_t)(b->shift_stages_lim + 1);
            while (shift_try_cnt > 0) { shift_try_cnt--; b->shift_crcs[shift_try_cnt].result = ORC_REF_BAD_CRC; }
            shift_try_cnt = 0;
            do {
                b->shift_crcs[shift_try_cnt].hyst_dph = hyst_cnt;
                b->shift_crcs[shift_try_cnt].shift_stg = shift_try_cnt;
                if (fill_data_words(b, l, hyst_cnt, shift_try_cnt) != ORC_STG_DATA_OK) { invalid_hyst = true; break; }
                else {
                    b->shift_crcs[shift_try_cnt].crc = l->calc_crc;
                    if (orc_stc_crc_valid(l)) {
                        b->shift_crcs[shift_try_cnt].result = ORC_REF_CRC_OK;
                        update_crc_stats(b->crc_stats, b->shift_crcs[shift_try_cnt], &valid_crcs_shift);
                        break;
                    }
                }
                shift_try_cnt++;
            } while (shift_try_cnt <= b->shift_stages_lim);
            if (valid_crcs_shift > 0) {
                find_most_frequent_crc(b->crc_stats, &valid_crcs_shift, true);
                invalidate_non_frequent_crcs(b->shift_crcs, 0, b->shift_stages_lim, valid_crcs_shift, b->crc_stats[0].crc);
            }
            b->hyst_crcs[hyst_cnt].shift_stg = b->crc_stats[0].shift_stg;
            b->hyst_crcs[hyst_cnt].crc = b->crc_stats[0].crc;
            if (valid_crcs_shift > 0) {
                b->hyst_crcs[hyst_cnt].hyst_dph = b->crc_stats[0].hyst_dph;
                b->hyst_crcs[hyst_cnt].result = ORC_REF_CRC_OK;
                hyst_good_cnt++;
                break;
            } else {
                b->hyst_crcs[hyst_cnt].hyst_dph = hyst_cnt;
                if (hyst_good_cnt > 0) break;
            }
            if (invalid_hyst) break;
            hyst_cnt++;
        } while (hyst_cnt <= b->hysteresis_depth_lim);

        reset_crc_stats(b->crc_stats, ORC_MAX_COLL_CRCS, &valid_crcs_hyst);
        b->crc_stats[0].hyst_dph = 0; b->crc_stats[0].shift_stg = 0;
        if (hyst_good_cnt > 0) {
            for (uint8_t i = 0; i <= hyst_cnt; i++)
                if (b->hyst_crcs[i].result == ORC_REF_CRC_OK) update_crc_stats(b->crc_stats, b->hyst_crcs[i], &valid_crcs_hyst);
            if (valid_crcs_hyst > 0) {
                find_most_frequent_crc(b->crc_stats, &valid_crcs_hyst, true);
                /* reference passes hyst_cnt+1 as the high index (one past the last used element) */
                invalidate_non_frequent_crcs(b->hyst_crcs, 0, (uint8_t)(hyst_cnt + 1), valid_crcs_hyst, b->crc_stats[0].crc);
            }
        }
        valid_delta = b->crc_stats[0].hyst_dph;
        valid_shift = b->crc_stats[0].shift_stg;
    } else {
        valid_delta = b->hysteresis_depth_lim;
        valid_shift = b->shift_stages_lim;
    }
    fill_data_words(b, l, valid_delta, valid_shift);
}

/* ------------------------------------------------------------------ reference level sweep */
void calc_forced_coords(const orc_binarizer *b, orc_coords *fc)   /* :631-641, :3586-3596, :3853-3863 */
{
    orc_coords_clear(fc);
    if (b->digi_set.en_force_coords) {
        fc->data_start = (int16_t)((int16_t)b->scan_start + b->digi_set.horiz_coords.data_start);
        fc->data_stop = (int16_t)((int16_t)b->scan_end - b->digi_set.horiz_coords.data_stop);
        if (b->video_line->doubled) {
            fc->data_start = (int16_t)(fc->data_start + b->digi_set.horiz_coords.data_start);
            fc->data_stop = (int16_t)(fc->data_stop - b->digi_set.horiz_coords.data_stop);
        }
    }
}

static void sweep_ref_level(orc_binarizer *b, orc_stc_line *pcm_line, orc_crc_handler *crc_res)   /* :3551-3817 */
{
    bool skip_bin;
    uint8_t low_lvl, high_lvl, read_result;
    uint16_t ref_index;
    orc_stc_line temp_stc;
    orc_coords forced_coords;

    orc_stc_clear(&temp_stc);
    calc_forced_coords(b, &forced_coords);
    low_lvl = pcm_line->black_level; high_lvl = pcm_line->white_level;
    low_lvl = (uint8_t)(low_lvl + 1); high_lvl = (uint8_t)(high_lvl - 1);
    if (b->digi_set.min_ref_lvl > low_lvl) low_lvl = b->digi_set.min_ref_lvl;
    if (b->digi_set.max_ref_lvl < high_lvl) high_lvl = b->digi_set.max_ref_lvl;
    ref_index = high_lvl;
    while (ref_index >= low_lvl) {
        /* :3629 dummy_line->clear() goes through a PCMLine* and clear() is NOT virtual: only the base
         * fields are reset; words, marker stages and coordinate tables persist from the previous level. */
        pcmline_clear(&temp_stc);
        stc_set_source_pixels(&temp_stc, 0, (uint16_t)(b->video_line->length - 1));
        temp_stc.coords.from_doubled = b->video_line->doubled;
        temp_stc.black_level = low_lvl; temp_stc.white_level = high_lvl;
        temp_stc.ref_level = (uint8_t)ref_index;
        if (!orc_coords_valid(&forced_coords)) {
            if (orc_coords_valid(&b->in_def_coord)) {
                skip_bin = false;
                if (b->digi_set.en_good_no_marker) {
                    find_stc007_coordinates(b, &temp_stc);
                    if (!orc_stc_has_markers(&temp_stc)) skip_bin = true;
                }
                if (skip_bin) {
                    temp_stc.coords = b->in_def_coord;
                    read_pcm_data(b, &temp_stc);
                }
            }
        }
        if (!orc_stc_crc_valid(&temp_stc)) {
            if (!orc_coords_valid(&forced_coords)) find_stc007_coordinates(b, &temp_stc);
            else { temp_stc.coords = forced_coords; temp_stc.coords_set = true; }
            if (temp_stc.coords_set) read_pcm_data(b, &temp_stc);
        }
        if (temp_stc.hysteresis_depth > 0x0F) temp_stc.hysteresis_depth = 0x0F;
        read_result = ORC_REF_NO_PCM;
        if (orc_stc_crc_valid(&temp_stc) && orc_coords_valid(&temp_stc.coords)) {
            read_result = ORC_REF_CRC_OK;
        } else if (temp_stc.coords_set) {
            read_result = ORC_REF_BAD_CRC;
        }
        if (read_result != ORC_REF_NO_PCM) {
            crc_res[ref_index].result = read_result;
            crc_res[ref_index].data_start = temp_stc.coords.data_start;
            crc_res[ref_index].data_stop = temp_stc.coords.data_stop;
            crc_res[ref_index].hyst_dph = temp_stc.hysteresis_depth;
            crc_res[ref_index].shift_stg = temp_stc.shift_stage;
            crc_res[ref_index].crc = temp_stc.calc_crc;
        }
        ref_index--;
    }
}

static void calc_ref_level_by_sweep(orc_binarizer *b, orc_stc_line *pcm_line)   /* :3821-4120 */
{
    uint8_t fast_ref, bin_level, valid_crc_cnt, span_res;
    orc_crc_handler scan_sweep_crcs[256];
    orc_coords forced_coords;

    fast_ref = pick_center_ref_level(b, pcm_line->black_level, pcm_line->white_level);
    b->hysteresis_depth_lim = b->in_max_hysteresis_depth;
    b->shift_stages_lim = b->in_max_shift_stages;
    calc_forced_coords(b, &forced_coords);
    reset_crc_stats(scan_sweep_crcs, 256, NULL);
    sweep_ref_level(b, pcm_line, scan_sweep_crcs);
    span_res = ORC_SPAN_NOT_FOUND;
    reset_crc_stats(b->crc_stats, ORC_MAX_COLL_CRCS + 1, &valid_crc_cnt);
    b->crc_stats[0].hyst_dph = 0; b->crc_stats[0].shift_stg = 0;
    for (bin_level = (uint8_t)(pcm_line->white_level - 1); bin_level > pcm_line->black_level; bin_level--)
        if (scan_sweep_crcs[bin_level].result == ORC_REF_CRC_OK) update_crc_stats(b->crc_stats, scan_sweep_crcs[bin_level], &valid_crc_cnt);
    if (valid_crc_cnt > 0) {
        find_most_frequent_crc(b->crc_stats, &valid_crc_cnt, true);
        invalidate_non_frequent_crcs(scan_sweep_crcs, (uint8_t)(pcm_line->black_level + 1), (uint8_t)(pcm_line->white_level - 1), valid_crc_cnt, b->crc_stats[0].crc);
        if (valid_crc_cnt > 0) {
            if (b->crc_stats[0].result < b->digi_set.min_valid_crcs) span_res = ORC_SPAN_TOO_NARROW;
            else span_res = pick_level_by_crc_stats(scan_sweep_crcs, &pcm_line->ref_level, (uint8_t)(pcm_line->black_level + 1), (uint8_t)(pcm_line->white_level - 1),
                                                    ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX);
        }
    }
    if (span_res == ORC_SPAN_OK) {
        orc_crc_handler t = scan_sweep_crcs[pcm_line->ref_level];
        pcm_line->ref_level_sweeped = true;
        orc_coords_set(&pcm_line->coords, t.data_start, t.data_stop);
        pcm_line->coords_set = true;
        if (!orc_coords_valid(&forced_coords)) find_stc007_coordinates(b, pcm_line);
        b->hysteresis_depth_lim = t.hyst_dph;
        if (b->hysteresis_depth_lim > ORC_HYST_DEPTH_MAX) b->hysteresis_depth_lim = ORC_HYST_DEPTH_MAX;
        b->shift_stages_lim = t.shift_stg;
    } else {
        if (span_res == ORC_SPAN_TOO_NARROW) {
            span_res = pick_level_by_crc_stats_opt(b, scan_sweep_crcs, &pcm_line->ref_level, (uint8_t)(pcm_line->black_level + 1), (uint8_t)(pcm_line->white_level - 1),
                                                   ORC_REF_CRC_OK, b->hysteresis_depth_lim, b->shift_stages_lim);
            pcm_line->forced_bad = true;
        } else {
            span_res = pick_level_by_crc_stats(scan_sweep_crcs, &pcm_line->ref_level, (uint8_t)(pcm_line->black_level + 1), (uint8_t)(pcm_line->white_level - 1),
                                               ORC_REF_BAD_CRC, 0xFF, 0xFF);
        }
        if (span_res == ORC_SPAN_OK) {
            orc_crc_handler t = scan_sweep_crcs[pcm_line->ref_level];
            orc_coords_set(&pcm_line->coords, t.data_start, t.data_stop);
            pcm_line->coords_set = true;
            if (!orc_coords_valid(&forced_coords)) find_stc007_coordinates(b, pcm_line);
        } else if (is_ref_level_preset(b)) {
            pcm_line->ref_level = b->in_def_reference;
            if (orc_coords_valid(&b->in_def_coord)) pcm_line->coords = b->in_def_coord;
        } else {
            pcm_line->ref_level = fast_ref;
            if (!orc_coords_valid(&b->in_def_coord))
                orc_coords_set(&pcm_line->coords, (int16_t)(b->scan_start + b->estimated_ppb), (int16_t)(b->scan_end - (4 * b->estimated_ppb)));
            else pcm_line->coords = b->in_def_coord;
        }
        b->hysteresis_depth_lim = ORC_HYST_DEPTH_MIN;
        b->shift_stages_lim = ORC_SHIFT_STAGES_MIN;
    }
}

/* ------------------------------------------------------------------ processLine */
uint8_t orc_binarizer_process_line(orc_binarizer *b)   /* :443-1724 */
{
    uint8_t stage_count;
    uint32_t tmp_calc;
    orc_coords forced_coords;
    const orc_video_line *vl = b->video_line;
    orc_stc_line *out = b->out_pcm_line;

    if (vl == NULL) return ORC_LB_RET_NULL_VIDEO;
    if (out == NULL) return ORC_LB_RET_NULL_PCM;
    orc_stc_clear(out);
    out->frame_number = vl->frame_number;
    out->line_number = vl->line_number;

    if (vl->service_type != ORC_SRV_NO) {
        /* :539-568 service passthrough; tag values are identical in VideoLine and PCMLine for these five */
        if (vl->service_type >= ORC_SRV_NEW_FILE && vl->service_type <= ORC_SRV_END_FRAME)
            orc_stc_set_service(out, vl->service_type);
    } else if (!vl->empty) {
        b->line_length = vl->length;
        out->coords.from_doubled = vl->doubled;
        b->scan_start = 0;
        b->scan_end = (uint16_t)(b->line_length - 1);
        stc_set_source_pixels(out, b->scan_start, b->scan_end);
        if (b->line_length < ORC_STC_BITS_IN_LINE) return ORC_LB_RET_SHORT_LINE;
        b->mark_start_max = (uint16_t)(b->line_length * b->digi_set.mark_max_dist);
        b->mark_start_max = b->mark_start_max / 100;
        b->mark_end_min = (uint16_t)(b->scan_end - b->mark_start_max);
        b->mark_start_max = (uint16_t)(b->scan_start + b->mark_start_max);
        tmp_calc = (uint32_t)(b->line_length * ORC_INT_CALC_MULT);
        tmp_calc = tmp_calc / ORC_STC_BITS_IN_LINE;
        b->estimated_ppb = (uint16_t)((tmp_calc + (ORC_INT_CALC_MULT / 2)) / ORC_INT_CALC_MULT);
        orc_coords_set(&out->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
        calc_forced_coords(b, &forced_coords);
        if (b->digi_set.en_force_coords && orc_coords_valid(&forced_coords)) { out->coords = forced_coords; out->coords_set = true; }

        b->proc_state = ORC_STG_REF_FIND;
        b->was_BW_scanned = false;
        if (are_bw_levels_preset(b)) { out->black_level = b->in_def_black; out->white_level = b->in_def_white; out->blk_wht_set = true; }
        if (is_ref_level_preset(b)) b->proc_state = orc_coords_valid(&b->in_def_coord) ? ORC_STG_INPUT_ALL : ORC_STG_INPUT_LEVEL;
        b->hysteresis_depth_lim = b->in_max_hysteresis_depth;
        b->shift_stages_lim = b->in_max_shift_stages;

        stage_count = 0;
        do {
            stage_count++;
            if (b->proc_state == ORC_STG_INPUT_ALL) {           /* :774-931 */
                if (!out->blk_wht_set) find_black_white(b);
                if (!orc_coords_valid(&forced_coords)) out->coords = b->in_def_coord;
                out->ref_level = b->in_def_reference;
                out->ref_level_sweeped = false;
                if (!out->blk_wht_set) b->proc_state = ORC_STG_NO_GOOD;
                else if ((b->in_def_reference >= out->white_level) || (b->in_def_reference <= out->black_level)) b->proc_state = ORC_STG_REF_FIND;
                else {
                    bool force_level_find = false;
                    if (b->do_coord_search && !orc_coords_valid(&forced_coords)) {
                        if (!b->digi_set.en_good_no_marker) {
                            find_stc007_coordinates(b, out);
                            out->coords = b->in_def_coord;
                            if (!orc_stc_has_markers(out)) force_level_find = true;
                        }
                    }
                    read_pcm_data(b, out);
                    if (orc_stc_crc_valid(out)) {
                        if (!force_level_find) { out->data_by_ext_tune = true; b->proc_state = ORC_STG_DATA_OK; }
                        else b->proc_state = ORC_STG_REF_FIND;
                    } else {
                        if (!orc_coords_valid(&forced_coords)) b->proc_state = ORC_STG_INPUT_LEVEL;
                        else b->proc_state = ORC_STG_REF_FIND;
                    }
                }
            } else if (b->proc_state == ORC_STG_INPUT_LEVEL) {  /* :932-1072 */
                if (!b->was_BW_scanned) find_black_white(b);
                if (!orc_coords_valid(&forced_coords)) orc_coords_set(&out->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
                out->ref_level = b->in_def_reference;
                out->ref_level_sweeped = false;
                if (!out->blk_wht_set) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    b->proc_state = ORC_STG_REF_FIND;
                    if ((b->in_def_reference < out->white_level) && (b->in_def_reference > out->black_level)) {
                        if (!b->do_coord_search) {
                            if (!orc_coords_valid(&b->in_def_coord))
                                orc_coords_set(&out->coords, (int16_t)(b->scan_start + b->estimated_ppb), (int16_t)(b->scan_end - (4 * b->estimated_ppb)));
                            else out->coords = b->in_def_coord;
                        } else find_stc007_coordinates(b, out);
                        if (orc_stc_has_markers(out)) {
                            if (!orc_coords_valid(&b->in_def_coord) || coords_ne(&out->coords, &b->in_def_coord)) {
                                read_pcm_data(b, out);
                                if (orc_stc_crc_valid(out)) { out->data_by_ext_tune = true; b->proc_state = ORC_STG_DATA_OK; }
                            }
                        }
                    }
                }
            } else if (b->proc_state == ORC_STG_REF_FIND) {     /* :1073-1390 */
                if (!b->was_BW_scanned) find_black_white(b);
                if (!out->blk_wht_set) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    b->do_ref_lvl_sweep = false;
                    if ((b->bin_mode == ORC_MODE_NORMAL) || (b->bin_mode == ORC_MODE_INSANE)) b->do_ref_lvl_sweep = true;
                    if (b->do_ref_lvl_sweep) b->proc_state = ORC_STG_REF_SWEEP_RUN;
                    else {
                        b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE;
                        b->shift_stages_lim = ORC_SHIFT_STAGES_MIN;
                        b->proc_state = ORC_STG_READ_PCM;
                        out->ref_level = pick_center_ref_level(b, out->black_level, out->white_level);
                        if (orc_coords_valid(&forced_coords)) { out->coords = forced_coords; out->coords_set = true; }
                        else {
                            if (!b->do_coord_search) {
                                if (!orc_coords_valid(&b->in_def_coord))
                                    orc_coords_set(&out->coords, (int16_t)(b->scan_start + b->estimated_ppb), (int16_t)(b->scan_end - (4 * b->estimated_ppb)));
                                else out->coords = b->in_def_coord;
                            } else find_stc007_coordinates(b, out);
                            if (!orc_stc_has_markers(out)) { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                            else { b->hysteresis_depth_lim = b->in_max_hysteresis_depth; b->shift_stages_lim = b->in_max_shift_stages; }
                        }
                    }
                }
            } else if (b->proc_state == ORC_STG_REF_SWEEP_RUN) { /* :1391-1400 */
                calc_ref_level_by_sweep(b, out);
                b->proc_state = ORC_STG_READ_PCM;
            } else if (b->proc_state == ORC_STG_READ_PCM) {     /* :1401-1533 */
                if (orc_coords_valid(&forced_coords)) { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                if (out->coords_set) read_pcm_data(b, out);
                if (orc_stc_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                if (b->proc_state != ORC_STG_DATA_OK) {
                    if (orc_coords_valid(&b->in_def_coord) && !orc_coords_valid(&forced_coords) && !b->do_ref_lvl_sweep
                        && !out->forced_bad && !out->coords_set) {
                        if (coords_ne(&out->coords, &b->in_def_coord)) {
                            out->coords = b->in_def_coord;
                            out->marker_start_bg_coord = 0; out->marker_start_ed_coord = 0; out->marker_stop_ed_coord = 0;
                            read_pcm_data(b, out);
                            if (orc_stc_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                        }
                    }
                    if (b->proc_state != ORC_STG_DATA_OK) b->proc_state = ORC_STG_NO_GOOD;
                }
            } else if (b->proc_state == ORC_STG_DATA_OK) {      /* :1534-1621 */
                if (out->forced_bad) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    if (!b->digi_set.en_good_no_marker) {
                        if (!orc_stc_has_markers(out)) { out->forced_bad = true; b->proc_state = ORC_STG_NO_GOOD; continue; }
                    }
                    orc_stc_apply_crc_state_per_word(out);
                    if (orc_stc_has_control_block(out)) orc_stc_set_serv_ctrl_blk(out);
                    out->coords.from_doubled = vl->doubled;
                    break;
                }
            } else if (b->proc_state == ORC_STG_NO_GOOD) {      /* :1622-1669 */
                if (orc_stc_crc_valid(out)) orc_stc_set_invalid_crc(out);
                orc_stc_apply_crc_state_per_word(out);
                if (orc_coords_valid(&forced_coords) && out->blk_wht_set) { out->mark_st_stage = ORC_MARK_ST_BOT_2; out->mark_ed_stage = ORC_MARK_ED_LEN_OK; }
                out->coords.from_doubled = vl->doubled;
                break;
            } else break;
            if (stage_count > ORC_STG_MAX) break;
        } while (1);
    } else {
        orc_stc_set_silent(out);
        orc_stc_set_invalid_crc(out);
    }
    return ORC_LB_RET_OK;
}
