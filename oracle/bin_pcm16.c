/* TEST INFRASTRUCTURE ONLY - never part of the product path.
 *
 * CPU restatement of the PCM-16x0 (Sony PCM-1610/1620/1630) branch of the reference's line binarizer: SURVEY.md section 8 row a9
 * with the PCM-16x0 parts of a2/a3/a5/a6/a7/a10.  One video line is read three times, once per third (Binarizer::setLinePartMode),
 * each pass filling one PCM16X0SubLine:
 *   findPCM16X0BW (binarizer.cpp:2603-2681) + findBlackWhite (:3116-3473), findPCM16X0Coordinates (:5819-6042),
 *   searchPCM16X0Data (:4514-5271: 21 x 21 coordinate pairs, all three parts read at each, votes per part and combined),
 *   fillPCM16X0 (:7134-7319), pickCutBitsUpPCM16X0 (:6599-7013, the Bit Picker: left bits of the left part, right bits of the
 *   right part), fillDataWords (:7560-7691), readPCMdata (:7695-8055), the PCM-16x0 paths of processLine (:443-1724) and the
 *   PCM16X0SubLine object (pcm16x0subline.cpp, pcmline.cpp).
 * Not restated: the reference level sweep, which PCM-16x0 runs in MODE_INSANE only (:1113-1120) - ORC_LB_RET_UNSUPPORTED here.
 * Pinned against the reference itself (oracle/_ref, ref_bin16_*) by tests/test_pcm16_front.py. */
#include "bin_pcm16.h"
#include "bin_internal.h"
#include <string.h>

#define PIX(b, x) ((b)->video_line->pixels[(x)])

enum { P16_BITS = ORC_P16_BITS_IN_LINE, P16_DATA = ORC_P16_BITS_PCM_DATA, P16_BITS_PER_WORD = 16, P16_BITS_PER_CRC = 16,
       P16_LEFT_SHIFT = 34, P16_RIGHT_SHIFT = 107, P16_WORD_CRCC = 3, P16_CRC_SILENT = 0x0E10 };
enum { P16_SEARCH_STEP_DIV = 2, P16_SEARCH_MAX_OFS = 10, P16_SEARCH_STEP_CNT = (P16_SEARCH_MAX_OFS + 1) * 2 };   /* binarizer.h:262-264 */

/* ------------------------------------------------------------------ PCM16X0SubLine : PCMLine */
static void p16_base_clear(orc_p16_line *l)    /* PCMLine::clear, pcmline.cpp:96-116 */
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
uint16_t orc_p16_crc_words(const uint16_t *w)  /* pcm16x0subline.cpp:158-170: CRC-16/CCITT-FALSE over the three 16-bit words */
{
    uint16_t crc = ORC_CRC_INIT;
    for (int i = 0; i < 3; i++) crc = orc_crc16_update(crc, w[i], 16);
    return crc;
}
static void p16_calc_crc(orc_p16_line *l) { l->calc_crc = orc_p16_crc_words(l->words); }
static void p16_set_invalid_crc(orc_p16_line *l) { l->words[P16_WORD_CRCC] = (uint16_t)~l->calc_crc; }   /* pcmline.cpp:194-198 */
static void p16_set_silent(orc_p16_line *l) { l->words[0] = l->words[1] = l->words[2] = 0; p16_calc_crc(l); }   /* pcm16x0subline.cpp:89-96 */
void orc_p16_clear(orc_p16_line *l)            /* pcm16x0subline.cpp:59-79 */
{
    p16_base_clear(l);
    l->control_bit = true;
    l->line_part = 0;
    l->picked_bits_left = l->picked_bits_right = 0;
    l->queue_order = 0;
    memset(l->pixel_coordinates, 0, sizeof(l->pixel_coordinates));
    p16_set_silent(l);
    l->calc_crc = P16_CRC_SILENT;
    p16_set_invalid_crc(l);
}
static void p16_set_word(orc_p16_line *l, uint8_t index, uint16_t w) { if (index < 4) l->words[index] = w; }   /* :99-113 */
bool orc_p16_crc_valid_ignore_forced(const orc_p16_line *l) { return l->calc_crc == l->words[P16_WORD_CRCC]; }           /* :281-288 */
bool orc_p16_crc_valid(const orc_p16_line *l) { return !l->forced_bad && orc_p16_crc_valid_ignore_forced(l); }            /* pcmline.cpp:360-367 */
/* PCMLine::setServiceLine (pcmline.cpp:490-502) calls the base clear(): words, part, control bit and the coordinate table stay */
static void p16_set_service(orc_p16_line *l, uint8_t service_type)
{
    uint32_t frame = l->frame_number; uint16_t line = l->line_number;
    p16_base_clear(l);
    l->frame_number = frame; l->line_number = line;
    l->service_type = service_type;
}
static void p16_set_source_pixels(orc_p16_line *l, uint16_t in_start, uint16_t in_stop)   /* pcmline.cpp:207-219 */
{
    if (in_stop > in_start)
        if (P16_BITS <= (in_stop - in_start)) { l->pixel_start = in_start; l->pixel_stop = in_stop; }
}
static void p16_set_ppb(orc_p16_line *l, orc_coords c)   /* pcmline.cpp:506-519, 193 bits between the data coordinates */
{
    uint8_t bit_count = P16_BITS;
    l->pixel_size_mult = (uint32_t)(c.data_stop - c.data_start);
    l->pixel_size_mult = (l->pixel_size_mult * ORC_INT_CALC_MULT + bit_count / 2) / bit_count;
    l->pixel_start_offset = c.data_start;
    l->halfpixel_size_mult = (l->pixel_size_mult + 2 / 2) / 2;
}
static uint8_t p16_get_ppb(const orc_p16_line *l) { return (uint8_t)(l->pixel_size_mult / ORC_INT_CALC_MULT); }   /* pcmline.cpp:235-238 */
static const int8_t P16_SH_BG_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:63-66 */
static const int8_t P16_SH_ED_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:68-71 */
static uint16_t p16_pixel_by_calc(const orc_p16_line *l, uint8_t pcm_bit, uint8_t in_shift)   /* pcmline.cpp:249-311, bit_ofs = 0 */
{
    int32_t video_pixel;
    if (pcm_bit >= P16_BITS) pcm_bit = P16_BITS - 1;
    video_pixel = (int32_t)((pcm_bit * l->pixel_size_mult) + l->halfpixel_size_mult);
    video_pixel = video_pixel / ORC_INT_CALC_MULT;
    video_pixel = video_pixel + l->pixel_start_offset;
    int8_t bg = P16_SH_BG_TBL[in_shift], ed = P16_SH_ED_TBL[in_shift];
    if (bg == ed) video_pixel += bg;
    else if (pcm_bit < P16_LEFT_SHIFT) video_pixel += bg;
    else if (pcm_bit > P16_RIGHT_SHIFT) video_pixel += ed;
    if (video_pixel < l->pixel_start) video_pixel = l->pixel_start;
    else if (video_pixel >= l->pixel_stop) video_pixel = l->pixel_stop - 1;
    return (uint16_t)video_pixel;
}
static void p16_calc_ppb(orc_p16_line *l, orc_coords c)   /* pcmline.cpp:223-232 + pcm16x0subline.cpp:654-660 */
{
    p16_set_ppb(l, c);
    for (uint8_t s = 0; s < ORC_PS_STAGES; s++)
        for (uint8_t bit = 0; bit < P16_BITS; bit++) l->pixel_coordinates[s][bit] = p16_pixel_by_calc(l, bit, s);
}

/* ------------------------------------------------------------------ AGC: BLACK / WHITE */
static void find_pcm16x0_bw(const orc_binarizer *b, uint16_t *sprd)   /* :2603-2681: three windows, in each third of the line */
{
    uint16_t pixel_limit, search_lim;
    uint32_t temp_calc;
    pixel_limit = (uint16_t)(b->scan_end - b->scan_start);
    temp_calc = pixel_limit / 8;
    pixel_limit = pixel_limit / 5;
    search_lim = (uint16_t)(pixel_limit + (uint16_t)temp_calc);
    for (uint16_t pixel = pixel_limit; pixel < search_lim; pixel++) sprd[PIX(b, pixel)]++;
    pixel_limit = (uint16_t)((uint16_t)temp_calc * 4);
    pixel_limit = (uint16_t)(pixel_limit + (uint16_t)temp_calc / 2);
    search_lim = (uint16_t)(pixel_limit + (uint16_t)temp_calc);
    for (uint16_t pixel = pixel_limit; pixel < search_lim; pixel++) sprd[PIX(b, pixel)]++;
    pixel_limit = (uint16_t)(b->scan_end - b->scan_start);
    search_lim = (uint16_t)(b->scan_end - pixel_limit / 64);
    pixel_limit = (uint16_t)(search_lim - (uint16_t)temp_calc);
    for (uint16_t pixel = pixel_limit; pixel < search_lim; pixel++) sprd[PIX(b, pixel)]++;
}

static bool find_black_white(orc_binarizer *b, orc_p16_line *line)   /* :3116-3473 (PCM-16x0 branch of the type switch) */
{
    uint8_t brt_lev, br_black = 0, br_white = 255, useful_low, useful_high;
    uint8_t low_scan_limit, high_scan_limit, range_limit, bin_low, bin_high;
    uint16_t black_lvl_count, white_lvl_count, search_lim;
    uint32_t temp_calc;
    uint16_t sprd[256];
    bool black_level_detected, white_level_detected;

    memset(sprd, 0, sizeof(sprd));
    find_pcm16x0_bw(b, sprd);

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

/* ------------------------------------------------------------------ data fill, Bit Picker */
static uint8_t fill_pcm16x0(orc_binarizer *b, orc_p16_line *l, uint8_t shift_stg)   /* :7134-7319 */
{
    bool prev_high = false;
    uint8_t pcm_bit, pixel_val, bit_count = 0, start_bit, stop_bit, low_ref = l->ref_low, high_ref = l->ref_high;
    uint8_t word_bit_pos = P16_BITS_PER_WORD - 1, word_index = 0;
    uint16_t pcm_word = 0;
    if (b->line_part_mode == ORC_PART_PCM16X0_LEFT) { start_bit = 0; stop_bit = P16_DATA - 1; }
    else if (b->line_part_mode == ORC_PART_PCM16X0_MIDDLE) { start_bit = P16_DATA; stop_bit = 2 * P16_DATA - 1; }
    else if (b->line_part_mode == ORC_PART_PCM16X0_RIGHT) { start_bit = 2 * P16_DATA + 1; stop_bit = 3 * P16_DATA; }     /* behind the control bit */
    else return ORC_STG_NO_GOOD;
    pcm_bit = start_bit;
    while (pcm_bit <= stop_bit) {
        pixel_val = PIX(b, l->pixel_coordinates[shift_stg][pcm_bit]);
        if (!prev_high) {
            if (pixel_val > low_ref) { pcm_word |= (uint16_t)(1 << word_bit_pos); prev_high = true; }
        } else {
            if (pixel_val >= high_ref) pcm_word |= (uint16_t)(1 << word_bit_pos);
            else prev_high = false;
        }
        if (word_bit_pos == 0) {
            p16_set_word(l, word_index, pcm_word);
            pcm_word = 0;
            word_index++;
            if (bit_count > (P16_DATA - P16_BITS_PER_CRC - 1)) break;
            else if (bit_count == (P16_DATA - P16_BITS_PER_CRC - 1)) word_bit_pos = P16_BITS_PER_CRC;
            else word_bit_pos = P16_BITS_PER_WORD;
        }
        word_bit_pos--;
        bit_count++;
        pcm_bit++;
    }
    p16_calc_crc(l);
    l->control_bit = true;
    if (orc_p16_crc_valid(l)) {
        pixel_val = PIX(b, l->pixel_coordinates[shift_stg][2 * P16_DATA]);
        if (pixel_val < l->ref_level) l->control_bit = false;
    }
    return ORC_STG_DATA_OK;
}

/* Bits whose sampling pixel was clamped to the edge of the picture were cut off by the capture: the left bits of the left part
 * and the right bits (of the CRCC) of the right part; every value of them is tried and the one and only combination that gives a
 * valid CRC is kept (:6599-7013) */
static uint8_t pick_cut_bits_up_pcm16x0(orc_binarizer *b, orc_p16_line *l)
{
    bool patch_found = false, coll_lock = false;
    uint8_t max_cut_bits, bit_count = 0;
    uint16_t index, first_pixel_coord, current_pixel_coord, orig_word, clean_word, patch_word, fix_word = 0, rep_limit;

    l->picked_bits_left = 0; l->picked_bits_right = 0;
    if (b->line_part_mode == ORC_PART_PCM16X0_LEFT) {
        first_pixel_coord = b->scan_start;
        max_cut_bits = b->digi_set.left_bit_pick;
        if (b->bin_mode == ORC_MODE_DRAFT) max_cut_bits = max_cut_bits / 2;
        for (index = 0; index < max_cut_bits; index++) {
            current_pixel_coord = l->pixel_coordinates[0][index];
            if ((current_pixel_coord - first_pixel_coord) >= ((p16_get_ppb(l) + 1) / 2)) break;
            if (index == 0) first_pixel_coord = current_pixel_coord;
            bit_count = (uint8_t)(index + 1);
        }
        if (b->force_bit_picker && orc_p16_crc_valid(l)) { l->picked_bits_left = bit_count; return ORC_STG_DATA_OK; }
        if (bit_count == 0) return ORC_STG_NO_GOOD;
        orig_word = l->words[0];
        rep_limit = (uint16_t)(1 << bit_count);
        clean_word = (uint16_t)((rep_limit - 1) << (P16_BITS_PER_WORD - bit_count));
        clean_word = (uint16_t)~clean_word;
        clean_word = orig_word & clean_word;
        for (index = 0; index < rep_limit; index++) {
            patch_word = (uint16_t)(index << (P16_BITS_PER_WORD - bit_count));
            p16_set_word(l, 0, clean_word | patch_word);
            p16_calc_crc(l);
            if (orc_p16_crc_valid(l)) {
                if (patch_found) { coll_lock = true; break; }
                patch_found = true; fix_word = patch_word;
            }
        }
        if (coll_lock) { p16_set_word(l, 0, orig_word); p16_calc_crc(l); l->forced_bad = true; return ORC_STG_NO_GOOD; }
        else if (!patch_found) { p16_set_word(l, 0, orig_word); p16_calc_crc(l); return ORC_STG_NO_GOOD; }
        p16_set_word(l, 0, clean_word | fix_word); p16_calc_crc(l);
        l->picked_bits_left = bit_count;
        return ORC_STG_DATA_OK;
    } else if (b->line_part_mode == ORC_PART_PCM16X0_RIGHT) {
        first_pixel_coord = b->scan_end;
        max_cut_bits = b->digi_set.right_bit_pick;
        if (b->bin_mode == ORC_MODE_DRAFT) max_cut_bits = max_cut_bits / 2;
        for (index = 0; index < max_cut_bits; index++) {
            current_pixel_coord = l->pixel_coordinates[0][P16_BITS - 1 - index];
            if ((first_pixel_coord - current_pixel_coord) >= ((p16_get_ppb(l) + 1) / 2)) break;
            if (index == 0) first_pixel_coord = current_pixel_coord;
            bit_count = (uint8_t)(index + 1);
        }
        if (b->force_bit_picker && orc_p16_crc_valid(l)) { l->picked_bits_right = bit_count; return ORC_STG_DATA_OK; }
        if (bit_count == 0) return ORC_STG_NO_GOOD;
        orig_word = l->words[P16_WORD_CRCC];
        rep_limit = (uint16_t)(1 << bit_count);
        clean_word = (uint16_t)(rep_limit - 1);
        clean_word = (uint16_t)~clean_word;
        clean_word = orig_word & clean_word;
        for (index = 0; index < rep_limit; index++) {
            patch_word = index;
            p16_set_word(l, P16_WORD_CRCC, clean_word | patch_word);
            p16_calc_crc(l);
            if (orc_p16_crc_valid(l)) {
                if (patch_found) { coll_lock = true; break; }
                patch_found = true; fix_word = patch_word;
            }
        }
        if (coll_lock) { p16_set_word(l, P16_WORD_CRCC, orig_word); p16_calc_crc(l); l->forced_bad = true; return ORC_STG_NO_GOOD; }
        else if (!patch_found) { p16_set_word(l, P16_WORD_CRCC, orig_word); p16_calc_crc(l); return ORC_STG_NO_GOOD; }
        p16_set_word(l, P16_WORD_CRCC, clean_word | fix_word); p16_calc_crc(l);
        l->picked_bits_right = bit_count;
        return ORC_STG_DATA_OK;
    }
    return ORC_STG_NO_GOOD;
}

static uint8_t fill_data_words(orc_binarizer *b, orc_p16_line *l, uint8_t ref_delta, uint8_t shift_stg)   /* :7560-7670 */
{
    uint8_t low_ref, high_ref, bin_res;
    if (ref_delta > ORC_HYST_DEPTH_MAX) return ORC_STG_NO_GOOD;
    if (shift_stg > ORC_SHIFT_STAGES_MAX) return ORC_STG_NO_GOOD;
    low_ref = get_low_level(l->ref_level, ref_delta);
    high_ref = get_high_level(l->ref_level, ref_delta);
    l->ref_low = low_ref; l->ref_high = high_ref;
    if (low_ref <= l->black_level) { p16_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    if (high_ref >= l->white_level) { p16_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    l->hysteresis_depth = ref_delta;
    l->shift_stage = shift_stg;
    bin_res = fill_pcm16x0(b, l, shift_stg);
    if (bin_res == ORC_STG_DATA_OK)
        if ((!orc_p16_crc_valid(l) && (l->ref_level > b->digi_set.min_white_lvl) && ((b->digi_set.left_bit_pick != 0) || (b->digi_set.right_bit_pick != 0)))
            || b->force_bit_picker)
            pick_cut_bits_up_pcm16x0(b, l);
    return bin_res;
}

static void read_pcm_data(orc_binarizer *b, orc_p16_line *l)   /* :7695-8055 */
{
    bool invalid_hyst;
    uint8_t hyst_cnt, shift_try_cnt, valid_crcs_hyst, valid_crcs_shift, hyst_good_cnt;
    uint8_t valid_delta, valid_shift;

    p16_calc_ppb(l, l->coords);
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
                    if (orc_p16_crc_valid(l)) {
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
                /* the reference passes hyst_cnt+1 as the high index (one past the last used element, see sdv_oracle.h) */
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

/* ------------------------------------------------------------------ Macro-TBC without markers */
/* one part of the line read at one coordinate pair (:4680-4778) */
static void search_read_part(orc_binarizer *b, orc_p16_line *l, uint8_t part_mode, orc_crc_handler *res, int16_t start_ofs, int16_t stop_ofs,
                             uint8_t picked_penalty_left, uint8_t picked_penalty_right)
{
    b->line_part_mode = part_mode;
    read_pcm_data(b, l);
    res->crc = l->words[P16_WORD_CRCC];
    res->hyst_dph = l->hysteresis_depth;
    res->shift_stg = l->shift_stage;
    res->data_start = start_ofs; res->data_stop = stop_ofs;
    res->result = ORC_REF_BAD_CRC;
    if (picked_penalty_left && l->picked_bits_left != 0) { res->hyst_dph = (uint8_t)(res->hyst_dph + picked_penalty_left); if (res->hyst_dph > 0x0F) res->hyst_dph = 0x0F; }
    if (picked_penalty_right && l->picked_bits_right != 0) { res->hyst_dph = (uint8_t)(res->hyst_dph + picked_penalty_right); if (res->hyst_dph > 0x0F) res->hyst_dph = 0x0F; }
}

static uint8_t search_pcm16x0_data(orc_binarizer *b, orc_p16_line *l, orc_coords data_loc)   /* :4514-5271 */
{
    bool lock_right, lock_left, lock_min, bitpick_previous;
    uint8_t right_ofs = 0xFF, left_ofs = 0xFF;
    uint8_t step_min, step_max;
    uint8_t stat_left_idx, stat_right_idx;
    uint8_t valid_crcs = 0, valid_left_crcs, valid_right_crcs, valid_right_p0_crcs, valid_right_p1_crcs, valid_right_p2_crcs;
    uint16_t scan_step = 1, step_span;
    orc_coords left_coord, right_coord;
    orc_crc_handler scan_left_res[P16_SEARCH_STEP_CNT], scan_left_crcs[ORC_MAX_COLL_CRCS];
    orc_crc_handler scan_right_res[P16_SEARCH_STEP_CNT], scan_right_p0_res[P16_SEARCH_STEP_CNT], scan_right_p1_res[P16_SEARCH_STEP_CNT], scan_right_p2_res[P16_SEARCH_STEP_CNT];
    orc_crc_handler scan_right_crcs[ORC_MAX_COLL_CRCS], scan_right_p0_crcs[ORC_MAX_COLL_CRCS], scan_right_p1_crcs[ORC_MAX_COLL_CRCS], scan_right_p2_crcs[ORC_MAX_COLL_CRCS];

    lock_right = lock_left = false;
    orc_coords_clear(&left_coord); orc_coords_clear(&right_coord);
    step_max = 2;
    while (step_max > 0) {
        p16_calc_ppb(l, data_loc);
        scan_step = p16_get_ppb(l);
        if (scan_step >= P16_SEARCH_STEP_DIV) scan_step = scan_step / P16_SEARCH_STEP_DIV; else scan_step = 1;
        step_span = (uint16_t)(scan_step * P16_SEARCH_MAX_OFS);
        left_coord.data_start = (int16_t)(data_loc.data_start - step_span);
        left_coord.data_stop = (int16_t)(data_loc.data_start + step_span);
        right_coord.data_start = (int16_t)(data_loc.data_stop - step_span);
        right_coord.data_stop = (int16_t)(data_loc.data_stop + step_span);
        if (((left_coord.data_start < b->scan_start) && (left_coord.data_stop < b->scan_start))
            || ((left_coord.data_start > b->scan_start) && (left_coord.data_stop > b->scan_start))
            || ((right_coord.data_start < b->scan_end) && (right_coord.data_stop < b->scan_end))
            || ((right_coord.data_start > b->scan_end) && (right_coord.data_stop > b->scan_end))) {
            data_loc.data_start = (int16_t)b->scan_start;
            data_loc.data_stop = (int16_t)b->scan_end;
        } else break;
        step_max--;
    }
    bitpick_previous = b->force_bit_picker;
    b->force_bit_picker = true;
    b->hysteresis_depth_lim = 0;
    if (b->bin_mode == ORC_MODE_DRAFT || b->bin_mode == ORC_MODE_FAST) b->shift_stages_lim = 0;
    else if (b->bin_mode == ORC_MODE_NORMAL || b->bin_mode == ORC_MODE_INSANE) b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE;

    reset_crc_stats(scan_left_res, P16_SEARCH_STEP_CNT, NULL);
    reset_crc_stats(scan_left_crcs, ORC_MAX_COLL_CRCS, &valid_left_crcs);
    stat_left_idx = 0;
    for (int16_t start_ofs = left_coord.data_start; start_ofs <= left_coord.data_stop; start_ofs = (int16_t)(start_ofs + scan_step)) {
        reset_crc_stats(scan_right_res, P16_SEARCH_STEP_CNT, NULL);
        reset_crc_stats(scan_right_p0_res, P16_SEARCH_STEP_CNT, NULL);
        reset_crc_stats(scan_right_p1_res, P16_SEARCH_STEP_CNT, NULL);
        reset_crc_stats(scan_right_p2_res, P16_SEARCH_STEP_CNT, NULL);
        reset_crc_stats(scan_right_crcs, ORC_MAX_COLL_CRCS, &valid_right_crcs);
        reset_crc_stats(scan_right_p0_crcs, ORC_MAX_COLL_CRCS, &valid_right_p0_crcs);
        reset_crc_stats(scan_right_p1_crcs, ORC_MAX_COLL_CRCS, &valid_right_p1_crcs);
        reset_crc_stats(scan_right_p2_crcs, ORC_MAX_COLL_CRCS, &valid_right_p2_crcs);
        stat_right_idx = 0;
        lock_right = false;
        lock_min = false;
        step_min = 0;
        step_max = P16_SEARCH_STEP_CNT;
        for (int16_t stop_ofs = right_coord.data_stop; stop_ofs >= right_coord.data_start; stop_ofs = (int16_t)(stop_ofs - scan_step)) {
            orc_coords_set(&l->coords, start_ofs, stop_ofs);
            search_read_part(b, l, ORC_PART_PCM16X0_LEFT, &scan_right_p0_res[stat_right_idx], start_ofs, stop_ofs, 0x02, 0);
            if (orc_p16_crc_valid(l)) {
                scan_right_p0_res[stat_right_idx].result = ORC_REF_CRC_OK;
                update_crc_stats(scan_right_p0_crcs, scan_right_p0_res[stat_right_idx], &valid_right_p0_crcs);
                if (!lock_min) { step_min = stat_right_idx; lock_min = true; }
                step_max = stat_right_idx;
            }
            search_read_part(b, l, ORC_PART_PCM16X0_MIDDLE, &scan_right_p1_res[stat_right_idx], start_ofs, stop_ofs, 0, 0);
            if (orc_p16_crc_valid(l)) {
                scan_right_p1_res[stat_right_idx].result = ORC_REF_CRC_OK;
                update_crc_stats(scan_right_p1_crcs, scan_right_p1_res[stat_right_idx], &valid_right_p1_crcs);
                if (!lock_min) { step_min = stat_right_idx; lock_min = true; }
                step_max = stat_right_idx;
            }
            search_read_part(b, l, ORC_PART_PCM16X0_RIGHT, &scan_right_p2_res[stat_right_idx], start_ofs, stop_ofs, 0, 0x03);
            if (orc_p16_crc_valid(l)) {
                scan_right_p2_res[stat_right_idx].result = ORC_REF_CRC_OK;
                update_crc_stats(scan_right_p2_crcs, scan_right_p2_res[stat_right_idx], &valid_right_p2_crcs);
                if (!lock_min) { step_min = stat_right_idx; lock_min = true; }
                step_max = stat_right_idx;
            }
            /* past the window where all three parts read: the first miss of all three ends the sweep of the right coordinate */
            if (lock_right && scan_right_p0_res[stat_right_idx].result != ORC_REF_CRC_OK && scan_right_p1_res[stat_right_idx].result != ORC_REF_CRC_OK
                && scan_right_p2_res[stat_right_idx].result != ORC_REF_CRC_OK) break;
            if (!lock_right && scan_right_p0_res[stat_right_idx].result == ORC_REF_CRC_OK && scan_right_p1_res[stat_right_idx].result == ORC_REF_CRC_OK
                && scan_right_p2_res[stat_right_idx].result == ORC_REF_CRC_OK) lock_right = true;
            stat_right_idx++;
            if (stat_right_idx >= P16_SEARCH_STEP_CNT) break;
        }
        if (valid_right_p0_crcs > 0) {
            find_most_frequent_crc(scan_right_p0_crcs, &valid_right_p0_crcs, true);
            invalidate_non_frequent_crcs(scan_right_p0_res, 0, P16_SEARCH_STEP_CNT - 1, valid_right_p0_crcs, scan_right_p0_crcs[0].crc);
        }
        if (valid_right_p1_crcs > 0) {
            find_most_frequent_crc(scan_right_p1_crcs, &valid_right_p1_crcs, true);
            invalidate_non_frequent_crcs(scan_right_p1_res, 0, P16_SEARCH_STEP_CNT - 1, valid_right_p1_crcs, scan_right_p1_crcs[0].crc);
        }
        if (valid_right_p2_crcs > 0) {
            find_most_frequent_crc(scan_right_p2_crcs, &valid_right_p2_crcs, true);
            invalidate_non_frequent_crcs(scan_right_p2_res, 0, P16_SEARCH_STEP_CNT - 1, valid_right_p2_crcs, scan_right_p2_crcs[0].crc);
        }
        if (step_max >= P16_SEARCH_STEP_CNT) step_max = P16_SEARCH_STEP_CNT - 1;
        /* the three parts of a coordinate pair as one entry: the middle part has to read (it has no cut-off bits to lean on) */
        for (stat_right_idx = step_min; stat_right_idx <= step_max; stat_right_idx++) {
            orc_crc_handler *r = &scan_right_res[stat_right_idx];
            const orc_crc_handler *p0 = &scan_right_p0_res[stat_right_idx], *p1 = &scan_right_p1_res[stat_right_idx], *p2 = &scan_right_p2_res[stat_right_idx];
            valid_crcs = 0;
            if (p1->result == ORC_REF_CRC_OK) {
                valid_crcs++;
                r->result = ORC_REF_CRC_OK; r->crc = P16_CRC_SILENT;
                r->hyst_dph = p1->hyst_dph; r->shift_stg = p1->shift_stg; r->data_start = p1->data_start; r->data_stop = p1->data_stop;
                if (p2->result == ORC_REF_CRC_OK) { valid_crcs++; r->hyst_dph = (uint8_t)(r->hyst_dph + p2->hyst_dph); if (p2->shift_stg > r->shift_stg) r->shift_stg = p2->shift_stg; }
                else r->hyst_dph = (uint8_t)(r->hyst_dph + ORC_HYST_DEPTH_SAFE);
                if (p0->result == ORC_REF_CRC_OK) { valid_crcs++; r->hyst_dph = (uint8_t)(r->hyst_dph + p0->hyst_dph); if (p0->shift_stg > r->shift_stg) r->shift_stg = p0->shift_stg; }
                else r->hyst_dph = (uint8_t)(r->hyst_dph + ORC_HYST_DEPTH_SAFE);
                if (r->hyst_dph > 0x0F) r->hyst_dph = 0x0F;
                update_crc_stats(scan_right_crcs, *r, &valid_right_crcs);
            } else if (p0->result == ORC_REF_CRC_OK && p2->result == ORC_REF_CRC_OK) {
                valid_crcs = 2;
                r->result = ORC_REF_CRC_OK; r->crc = P16_CRC_SILENT;
                r->hyst_dph = p2->hyst_dph; r->shift_stg = p2->shift_stg; r->data_start = p2->data_start; r->data_stop = p2->data_stop;
                if (p0->hyst_dph > r->hyst_dph) { r->hyst_dph = p0->hyst_dph; r->shift_stg = p0->shift_stg; }
                else if (p0->hyst_dph == r->hyst_dph) { if (p0->shift_stg > r->shift_stg) r->shift_stg = p0->shift_stg; }
                r->hyst_dph = (uint8_t)(r->hyst_dph + ORC_HYST_DEPTH_SAFE);
                if (r->hyst_dph > 0x0F) r->hyst_dph = 0x0F;
                update_crc_stats(scan_right_crcs, *r, &valid_right_crcs);
            } else r->result = ORC_REF_BAD_CRC;
            if (valid_crcs == ORC_P16_SUBLINES) lock_left = true;
        }
        if (valid_right_crcs > 0)
            if (pick_level_by_crc_stats(scan_right_res, &right_ofs, step_min, step_max, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX) != ORC_SPAN_OK) valid_right_crcs = 0;
        if (valid_right_crcs == 0) {
            /* second try: one good outer part is enough */
            reset_crc_stats(scan_right_res, P16_SEARCH_STEP_CNT, NULL);
            reset_crc_stats(scan_right_crcs, ORC_MAX_COLL_CRCS, &valid_right_crcs);
            for (stat_right_idx = step_min; stat_right_idx <= step_max; stat_right_idx++) {
                orc_crc_handler *r = &scan_right_res[stat_right_idx];
                const orc_crc_handler *p0 = &scan_right_p0_res[stat_right_idx], *p2 = &scan_right_p2_res[stat_right_idx];
                valid_crcs = 0;
                if (p2->result == ORC_REF_CRC_OK) {
                    valid_crcs++;
                    r->result = ORC_REF_CRC_OK; r->crc = P16_CRC_SILENT;
                    r->hyst_dph = p2->hyst_dph; r->shift_stg = p2->shift_stg; r->data_start = p2->data_start; r->data_stop = p2->data_stop;
                    r->hyst_dph = (uint8_t)(r->hyst_dph + ORC_HYST_DEPTH_MAX);
                    if (r->hyst_dph > 0x0F) r->hyst_dph = 0x0F;
                    update_crc_stats(scan_right_crcs, *r, &valid_right_crcs);
                } else if (p0->result == ORC_REF_CRC_OK) {
                    valid_crcs++;
                    r->result = ORC_REF_CRC_OK; r->crc = P16_CRC_SILENT;
                    r->hyst_dph = p0->hyst_dph; r->shift_stg = p0->shift_stg; r->data_start = p0->data_start; r->data_stop = p0->data_stop;
                    r->hyst_dph = (uint8_t)(r->hyst_dph + 2 * ORC_HYST_DEPTH_SAFE);
                    if (r->hyst_dph > 0x0F) r->hyst_dph = 0x0F;
                    update_crc_stats(scan_right_crcs, *r, &valid_right_crcs);
                } else r->result = ORC_REF_BAD_CRC;
            }
            if (valid_right_crcs > 0)
                if (pick_level_by_crc_stats(scan_right_res, &right_ofs, step_min, step_max, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX) != ORC_SPAN_OK) valid_right_crcs = 0;
        }
        if (valid_right_crcs > 0) {
            scan_left_res[stat_left_idx].result = ORC_REF_CRC_OK;
            scan_left_res[stat_left_idx].crc = scan_right_res[right_ofs].crc;
            scan_left_res[stat_left_idx].hyst_dph = scan_right_res[right_ofs].hyst_dph;
            scan_left_res[stat_left_idx].shift_stg = scan_right_res[right_ofs].shift_stg;
            scan_left_res[stat_left_idx].data_start = scan_right_res[right_ofs].data_start;
            scan_left_res[stat_left_idx].data_stop = scan_right_res[right_ofs].data_stop;
            update_crc_stats(scan_left_crcs, scan_right_res[right_ofs], &valid_left_crcs);
            if (lock_left) {
                /* all three parts have read at some left coordinate before: the sweep ends where fewer than two still do */
                valid_crcs = 0;
                if (scan_right_p0_res[right_ofs].result == ORC_REF_CRC_OK) valid_crcs++;
                if (scan_right_p1_res[right_ofs].result == ORC_REF_CRC_OK) valid_crcs++;
                if (scan_right_p2_res[right_ofs].result == ORC_REF_CRC_OK) valid_crcs++;
                if (valid_crcs < 2) break;
            }
        }
        stat_left_idx++;
        if (stat_left_idx >= P16_SEARCH_STEP_CNT) break;
    }
    b->force_bit_picker = bitpick_previous;

    if (valid_left_crcs > 0) {
        find_most_frequent_crc(scan_left_crcs, &valid_left_crcs, false);
        invalidate_non_frequent_crcs(scan_left_res, 0, P16_SEARCH_STEP_CNT - 1, valid_left_crcs, scan_left_crcs[0].crc);
    }
    if (valid_left_crcs > 0)
        if (pick_level_by_crc_stats(scan_left_res, &left_ofs, 0, P16_SEARCH_STEP_CNT - 1, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX) != ORC_SPAN_OK) valid_left_crcs = 0;
    if (valid_left_crcs > 0) {
        l->coords.data_start = scan_left_res[left_ofs].data_start;
        l->coords.data_stop = scan_left_res[left_ofs].data_stop;
        l->coords_set = true;
        l->coords_sweeped = true;
        return ORC_LB_RET_OK;
    }
    l->coords = data_loc;       /* CoordinatePair assignment copies every field (frametrimset.cpp:22-36) */
    l->coords_sweeped = false;
    return ORC_LB_RET_NO_COORD;
}

static bool find_pcm16x0_coordinates(orc_binarizer *b, orc_p16_line *l, orc_coords coord_history)   /* :5819-6042 */
{
    bool search_state;
    uint8_t in_line_mode, in_hyst_depth, in_shift_stages;
    orc_coords data_coord;
    uint16_t line_margin;
    orc_video_line *vl = (orc_video_line *)b->video_line;

    if (vl->scan_done) return false;        /* once per video line: the parts behind the first one inherit what it found */
    orc_coords_clear(&data_coord);
    line_margin = (uint16_t)(b->scan_end - b->scan_start);
    line_margin = line_margin / 40;
    if (orc_coords_valid(&coord_history)) data_coord = coord_history;
    else {
        data_coord.data_start = (int16_t)b->scan_start;
        search_state = PIX(b, (uint16_t)data_coord.data_start) > l->ref_level;
        for (uint16_t pixel = b->scan_start; pixel < (b->scan_start + line_margin); pixel++) {
            if (!search_state) { if (PIX(b, pixel) > l->ref_level) { data_coord.data_start = (int16_t)(pixel - 1); break; } }
            else { if (PIX(b, pixel) < l->ref_level) { data_coord.data_start = (int16_t)(pixel - 1); break; } }
        }
        data_coord.data_stop = (int16_t)b->scan_end;
        search_state = PIX(b, (uint16_t)data_coord.data_stop) > l->ref_level;
        for (uint16_t pixel = b->scan_end; pixel > (b->scan_end - line_margin); pixel--) {
            if (!search_state) { if (PIX(b, pixel) > l->ref_level) { data_coord.data_stop = (int16_t)(pixel + 1); break; } }
            else { if (PIX(b, pixel) < l->ref_level) { data_coord.data_stop = (int16_t)(pixel + 1); break; } }
        }
    }
    search_state = false;
    in_line_mode = b->line_part_mode;
    in_hyst_depth = b->hysteresis_depth_lim;
    in_shift_stages = b->shift_stages_lim;
    if (search_pcm16x0_data(b, l, data_coord) == ORC_LB_RET_OK) search_state = true;
    b->line_part_mode = in_line_mode;
    b->hysteresis_depth_lim = in_hyst_depth;
    b->shift_stages_lim = in_shift_stages;
    vl->scan_done = true;
    return search_state;
}

/* ------------------------------------------------------------------ reference level sweep (MODE_INSANE only for this format, :1113-1120) */
/* Binarizer::sweepRefLevel (:3551-3817) with a PCM16X0SubLine as the trial line: every level runs the whole coordinate search over the
 * three parts of the video line (scan_done is reset per level, :3704) and then reads the part the Binarizer is set to.  clear() through
 * the PCMLine pointer is the base clear(): words, picked bits, part and Control Bit of the trial line persist from level to level. */
static void sweep_ref_level_p16(orc_binarizer *b, orc_p16_line *pcm_line, orc_crc_handler *crc_res)
{
    uint8_t low_lvl, high_lvl, read_result;
    uint16_t ref_index;
    orc_p16_line t;
    orc_coords forced_coords;
    orc_video_line *vl = (orc_video_line *)b->video_line;

    orc_p16_clear(&t);                      /* PCM16X0SubLine temp_pcm16x0 (constructor): PART_LEFT, so the early return of :3570 never fires */
    calc_forced_coords(b, &forced_coords);
    low_lvl = (uint8_t)(pcm_line->black_level + 1); high_lvl = (uint8_t)(pcm_line->white_level - 1);
    if (b->digi_set.min_ref_lvl > low_lvl) low_lvl = b->digi_set.min_ref_lvl;
    if (b->digi_set.max_ref_lvl < high_lvl) high_lvl = b->digi_set.max_ref_lvl;
    ref_index = high_lvl;
    while (ref_index >= low_lvl) {
        p16_base_clear(&t);
        p16_set_source_pixels(&t, 0, (uint16_t)(vl->length - 1));
        t.coords.from_doubled = vl->doubled;
        t.black_level = low_lvl; t.white_level = high_lvl;
        t.ref_level = (uint8_t)ref_index;
        if (!orc_p16_crc_valid(&t)) {
            if (!orc_coords_valid(&forced_coords)) { vl->scan_done = false; find_pcm16x0_coordinates(b, &t, b->in_def_coord); }
            else { t.coords = forced_coords; t.coords_set = true; }
            if (t.coords_set) read_pcm_data(b, &t);
        }
        if (t.picked_bits_right != 0) t.hysteresis_depth = (uint8_t)(t.hysteresis_depth + ORC_HYST_DEPTH_MAX + 2);          /* :3753-3765 */
        else if (t.picked_bits_left != 0) t.hysteresis_depth = (uint8_t)(t.hysteresis_depth + ORC_HYST_DEPTH_MAX + 1);
        if (t.hysteresis_depth > 0x0F) t.hysteresis_depth = 0x0F;
        read_result = ORC_REF_NO_PCM;
        if (orc_p16_crc_valid(&t) && orc_coords_valid(&t.coords)) read_result = ORC_REF_CRC_OK;
        else if (t.coords_set) read_result = ORC_REF_BAD_CRC;
        if (read_result != ORC_REF_NO_PCM) {
            crc_res[ref_index].result = read_result;
            crc_res[ref_index].data_start = t.coords.data_start;
            crc_res[ref_index].data_stop = t.coords.data_stop;
            crc_res[ref_index].hyst_dph = t.hysteresis_depth;
            crc_res[ref_index].shift_stg = t.shift_stage;
            crc_res[ref_index].crc = t.calc_crc;
        }
        if (ref_index == 0) break;
        ref_index--;
    }
}

/* Binarizer::calcRefLevelBySweep (:3821-4120), the branches a line without markers takes */
static void calc_ref_level_by_sweep_p16(orc_binarizer *b, orc_p16_line *pcm_line)
{
    uint8_t fast_ref, bin_level, valid_crc_cnt, span_res;
    orc_crc_handler scan_sweep_crcs[256];
    const uint8_t blk1 = (uint8_t)(pcm_line->black_level + 1), wht1 = (uint8_t)(pcm_line->white_level - 1);

    fast_ref = pick_center_ref_level(b, pcm_line->black_level, pcm_line->white_level);
    b->hysteresis_depth_lim = 0;
    b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE;
    reset_crc_stats(scan_sweep_crcs, 256, NULL);
    sweep_ref_level_p16(b, pcm_line, scan_sweep_crcs);
    span_res = ORC_SPAN_NOT_FOUND;
    reset_crc_stats(b->crc_stats, ORC_MAX_COLL_CRCS + 1, &valid_crc_cnt);
    b->crc_stats[0].hyst_dph = 0; b->crc_stats[0].shift_stg = 0;
    for (bin_level = wht1; bin_level > pcm_line->black_level; bin_level--)
        if (scan_sweep_crcs[bin_level].result == ORC_REF_CRC_OK) update_crc_stats(b->crc_stats, scan_sweep_crcs[bin_level], &valid_crc_cnt);
    if (valid_crc_cnt > 0) {
        find_most_frequent_crc(b->crc_stats, &valid_crc_cnt, true);
        invalidate_non_frequent_crcs(scan_sweep_crcs, blk1, wht1, valid_crc_cnt, b->crc_stats[0].crc);
        if (valid_crc_cnt > 0) {
            if (b->crc_stats[0].result < b->digi_set.min_valid_crcs) span_res = ORC_SPAN_TOO_NARROW;
            else span_res = pick_level_by_crc_stats(scan_sweep_crcs, &pcm_line->ref_level, blk1, wht1, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX);
        }
    }
    if (span_res == ORC_SPAN_OK) {
        orc_crc_handler t = scan_sweep_crcs[pcm_line->ref_level];
        pcm_line->ref_level_sweeped = true;
        orc_coords_set(&pcm_line->coords, t.data_start, t.data_stop);
        pcm_line->coords_set = true;
        b->hysteresis_depth_lim = t.hyst_dph;
        if (b->hysteresis_depth_lim > ORC_HYST_DEPTH_MAX) b->hysteresis_depth_lim = ORC_HYST_DEPTH_MAX;
        b->shift_stages_lim = t.shift_stg;
    } else {
        if (span_res == ORC_SPAN_TOO_NARROW) {
            span_res = pick_level_by_crc_stats_opt(b, scan_sweep_crcs, &pcm_line->ref_level, blk1, wht1, ORC_REF_CRC_OK, b->hysteresis_depth_lim, b->shift_stages_lim);
            pcm_line->forced_bad = true;
        } else span_res = pick_level_by_crc_stats(scan_sweep_crcs, &pcm_line->ref_level, blk1, wht1, ORC_REF_NO_PCM, 0xFF, 0xFF);   /* canUseMarkers() == false */
        if (span_res == ORC_SPAN_OK) {
            orc_crc_handler t = scan_sweep_crcs[pcm_line->ref_level];
            orc_coords_set(&pcm_line->coords, t.data_start, t.data_stop);
            pcm_line->coords_set = true;
        } else if (is_ref_level_preset(b)) {
            pcm_line->ref_level = b->in_def_reference;
            if (orc_coords_valid(&b->in_def_coord)) pcm_line->coords = b->in_def_coord;
        } else {
            pcm_line->ref_level = fast_ref;
            if (!orc_coords_valid(&b->in_def_coord)) orc_coords_set(&pcm_line->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
            else pcm_line->coords = b->in_def_coord;
        }
        b->hysteresis_depth_lim = ORC_HYST_DEPTH_MIN;
        b->shift_stages_lim = ORC_SHIFT_STAGES_MIN;
    }
}

/* ------------------------------------------------------------------ Binarizer::processLine, PCM16X0SubLine output */
void orc_binarizer_set_good_parameters_p16(orc_binarizer *b, const orc_p16_line *l)   /* :353-377 */
{
    if (l == NULL) {
        orc_binarizer_set_reference_level(b, 0);
        orc_binarizer_set_data_coordinates2(b, 0, 0);
        orc_binarizer_set_bw_levels(b, 0, 0);
    } else if (orc_p16_crc_valid_ignore_forced(l)) {
        orc_binarizer_set_reference_level(b, l->ref_level);
        orc_binarizer_set_data_coordinates(b, l->coords);
        orc_binarizer_set_bw_levels(b, l->black_level, l->white_level);
    }
}

uint8_t orc_binarizer_process_line_p16(orc_binarizer *b, orc_p16_line *out)   /* :443-1724 */
{
    uint8_t stage_count;
    uint32_t tmp_calc;
    orc_coords forced_coords;
    const orc_video_line *vl = b->video_line;

    if (vl == NULL) return ORC_LB_RET_NULL_VIDEO;
    if (out == NULL) return ORC_LB_RET_NULL_PCM;
    orc_p16_clear(out);
    if (b->line_part_mode == ORC_PART_PCM16X0_LEFT) out->line_part = 0;          /* :496-510; FULL_LINE leaves PART_LEFT of clear() */
    else if (b->line_part_mode == ORC_PART_PCM16X0_MIDDLE) out->line_part = 1;
    else if (b->line_part_mode == ORC_PART_PCM16X0_RIGHT) out->line_part = 2;
    out->frame_number = vl->frame_number;
    out->line_number = vl->line_number;

    if (vl->service_type != ORC_SRV_NO) {
        if (vl->service_type >= ORC_SRV_NEW_FILE && vl->service_type <= ORC_SRV_END_FRAME) p16_set_service(out, vl->service_type);
    } else if (!vl->empty) {
        b->line_length = vl->length;
        out->coords.from_doubled = vl->doubled;
        b->scan_start = 0;
        b->scan_end = (uint16_t)(b->line_length - 1);
        p16_set_source_pixels(out, b->scan_start, b->scan_end);
        if (b->line_length < P16_BITS) return ORC_LB_RET_SHORT_LINE;
        b->mark_start_max = 0; b->mark_end_min = 0xFFFF;         /* no markers in PCM-16x0 (:592-604) */
        tmp_calc = (uint32_t)b->line_length * ORC_INT_CALC_MULT;
        tmp_calc = tmp_calc / P16_BITS;
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
            if (b->proc_state == ORC_STG_INPUT_ALL) {                       /* :774-931 */
                if (!out->blk_wht_set) find_black_white(b, out);
                if (!orc_coords_valid(&forced_coords)) out->coords = b->in_def_coord;
                out->ref_level = b->in_def_reference;
                out->ref_level_sweeped = false;
                if (!out->blk_wht_set) b->proc_state = ORC_STG_NO_GOOD;
                else if ((b->in_def_reference >= out->white_level) || (b->in_def_reference <= out->black_level)) b->proc_state = ORC_STG_REF_FIND;
                else {
                    read_pcm_data(b, out);
                    if (orc_p16_crc_valid(out)) { out->data_by_ext_tune = true; b->proc_state = ORC_STG_DATA_OK; }
                    else b->proc_state = ORC_STG_REF_FIND;
                }
            } else if (b->proc_state == ORC_STG_INPUT_LEVEL) {              /* :932-1072 */
                if (!b->was_BW_scanned) find_black_white(b, out);
                if (!orc_coords_valid(&forced_coords)) orc_coords_set(&out->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
                out->ref_level = b->in_def_reference;
                out->ref_level_sweeped = false;
                b->proc_state = out->blk_wht_set ? ORC_STG_REF_FIND : ORC_STG_NO_GOOD;
            } else if (b->proc_state == ORC_STG_REF_FIND) {                 /* :1073-1390 */
                if (!b->was_BW_scanned) find_black_white(b, out);
                if (!out->blk_wht_set) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    b->do_ref_lvl_sweep = (b->bin_mode == ORC_MODE_INSANE);          /* :1113-1120 */
                    if (b->do_ref_lvl_sweep) b->proc_state = ORC_STG_REF_SWEEP_RUN;
                    else {
                        b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE;
                        b->shift_stages_lim = ORC_SHIFT_STAGES_MIN;
                        b->proc_state = ORC_STG_READ_PCM;
                        out->ref_level = pick_center_ref_level(b, out->black_level, out->white_level);
                        if (orc_coords_valid(&forced_coords)) { out->coords = forced_coords; out->coords_set = true; }
                        else {
                            if (!orc_coords_valid(&b->in_def_coord)) orc_coords_set(&out->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
                            else out->coords = b->in_def_coord;
                            if (b->digi_set.en_coord_search && b->do_coord_search) find_pcm16x0_coordinates(b, out, b->in_def_coord);
                        }
                        if (!out->coords_set) {                                  /* :1301-1320 */
                            if (b->bin_mode == ORC_MODE_DRAFT) { b->hysteresis_depth_lim = 2; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                            else { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE; }
                        } else { b->hysteresis_depth_lim = b->in_max_hysteresis_depth; b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE; }
                    }
                }
            } else if (b->proc_state == ORC_STG_REF_SWEEP_RUN) {        /* :1391-1400 */
                calc_ref_level_by_sweep_p16(b, out);
                b->proc_state = ORC_STG_READ_PCM;
            } else if (b->proc_state == ORC_STG_READ_PCM) {                 /* :1401-1533 */
                if (orc_coords_valid(&forced_coords)) { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                if (out->coords_set) read_pcm_data(b, out);
                if (orc_p16_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                if (b->proc_state != ORC_STG_DATA_OK) {
                    if (orc_coords_valid(&b->in_def_coord) && !orc_coords_valid(&forced_coords) && !b->do_ref_lvl_sweep
                        && !out->forced_bad && !out->coords_set) {
                        if (coords_ne(&out->coords, &b->in_def_coord)) {
                            out->coords = b->in_def_coord;
                            read_pcm_data(b, out);
                            if (orc_p16_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                        }
                    }
                    if (b->proc_state != ORC_STG_DATA_OK) b->proc_state = ORC_STG_NO_GOOD;
                }
            } else if (b->proc_state == ORC_STG_DATA_OK) {                  /* :1534-1621 */
                if (out->forced_bad) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    out->coords_set = true;                                  /* :1568-1574 */
                    out->coords.from_doubled = vl->doubled;
                    break;
                }
            } else if (b->proc_state == ORC_STG_NO_GOOD) {                  /* :1622-1669 */
                if (orc_p16_crc_valid(out)) p16_set_invalid_crc(out);
                out->coords.from_doubled = vl->doubled;
                break;
            } else break;
            if (stage_count > ORC_STG_MAX) break;
        } while (1);
    } else {
        p16_set_silent(out);
        p16_set_invalid_crc(out);
    }
    return ORC_LB_RET_OK;
}
