/* TEST INFRASTRUCTURE ONLY - never part of the product path.
 *
 * CPU restatement of the PCM-1 (Sony Standard B) branch of the reference's line binarizer: SURVEY.md section 8 row a9
 * (front half) with the PCM-1 parts of a2/a3/a5/a6/a7/a10.  One video line -> one PCM1Line:
 *   findPCM1BW (binarizer.cpp:2560-2600) + findBlackWhite (:3116-3473), findPCM1Coordinates (:5601-5813),
 *   searchPCM1Data (:4123-4511, the 25 x 25 coordinate grid with CRC voting), fillPCM1 (:7016-7131),
 *   pickCutBitsUpPCM1 (:6116-6596, the Bit Picker), fillDataWords (:7560-7691), readPCMdata (:7695-8055),
 *   the PCM-1 paths of processLine (:443-1724) and the PCM1Line object (pcm1line.cpp, pcmline.cpp).
 * Not restated: the reference level sweep, which PCM-1 runs in MODE_INSANE only (:1105-1112) - that mode answers
 * ORC_LB_RET_UNSUPPORTED here.
 * Pinned against the reference itself (oracle/_ref, ref_bin_process_pcm1) by tests/test_pcm1_front.py. */
#include "bin_pcm1.h"
#include "bin_internal.h"
#include "pcm1.h"
#include <string.h>

#define PIX(b, x) ((b)->video_line->pixels[(x)])

enum { P1F_BITS = 94, P1F_BITS_PER_WORD = 13, P1F_BITS_PER_CRC = 16, P1F_LEFT_SHIFT = 16, P1F_RIGHT_SHIFT = 52,
       P1F_WORD_CRCC = 6, P1F_CRC_SILENT = 0xECBF };
enum { P1F_SEARCH_STEP_DIV = 4, P1F_SEARCH_MAX_OFS = 12, P1F_SEARCH_STEP_CNT = (P1F_SEARCH_MAX_OFS + 1) * 2 };   /* binarizer.h:254-256 */

/* ------------------------------------------------------------------ PCM1Line : PCMLine */
static void p1_base_clear(orc_p1_line *l)      /* PCMLine::clear, pcmline.cpp:96-116 */
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
static void p1_calc_crc(orc_p1_line *l) { l->calc_crc = orc_pcm1_crc_words(l->words); }          /* pcm1line.cpp:158-171 */
static void p1_set_invalid_crc(orc_p1_line *l) { l->words[P1F_WORD_CRCC] = (uint16_t)~l->calc_crc; }   /* pcmline.cpp:194-198 */
static void p1_set_silent(orc_p1_line *l)      /* pcm1line.cpp:94-101 */
{
    for (int i = 0; i < P1F_WORD_CRCC; i++) l->words[i] = 1 << 12;
    p1_calc_crc(l);
}
void orc_p1_clear(orc_p1_line *l)              /* pcm1line.cpp:57-77 */
{
    p1_base_clear(l);
    l->picked_bits_left = l->picked_bits_right = 0;
    memset(l->pixel_coordinates, 0, sizeof(l->pixel_coordinates));
    p1_set_silent(l);
    l->calc_crc = P1F_CRC_SILENT;
    p1_set_invalid_crc(l);
}
static void p1_set_word(orc_p1_line *l, uint8_t index, uint16_t w)   /* pcm1line.cpp:104-117 */
{
    if (index < 7) l->words[index] = (index == P1F_WORD_CRCC) ? w : (uint16_t)(w & 0x1FFF);
}
bool orc_p1_has_header(const orc_p1_line *l)   /* pcm1line.cpp:314-323 */
{
    return l->words[0] == 0x0666 && l->words[1] == 0x0CCC && l->words[2] == 0x1999 && l->words[3] == 0x1333 &&
           l->words[4] == 0x0666 && l->words[5] == 0x0CCC && l->words[P1F_WORD_CRCC] == 0xCCCC;
}
bool orc_p1_crc_valid_ignore_forced(const orc_p1_line *l) { return l->calc_crc == l->words[P1F_WORD_CRCC] || orc_p1_has_header(l); }   /* :326-333 */
bool orc_p1_crc_valid(const orc_p1_line *l) { return !l->forced_bad && orc_p1_crc_valid_ignore_forced(l); }                            /* pcmline.cpp:360-367 */
/* PCMLine::setServiceLine (pcmline.cpp:490-502) calls the base clear(): words, picked bits and the coordinate table stay */
static void p1_set_service(orc_p1_line *l, uint8_t service_type)
{
    uint32_t frame = l->frame_number; uint16_t line = l->line_number;
    p1_base_clear(l);
    l->frame_number = frame; l->line_number = line;
    l->service_type = service_type;
}
static void p1_set_source_pixels(orc_p1_line *l, uint16_t in_start, uint16_t in_stop)   /* pcmline.cpp:207-219 */
{
    if (in_stop > in_start)
        if (P1F_BITS <= (in_stop - in_start)) { l->pixel_start = in_start; l->pixel_stop = in_stop; }
}
static void p1_set_ppb(orc_p1_line *l, orc_coords c)   /* pcmline.cpp:506-519, 94 bits between the data coordinates */
{
    uint8_t bit_count = P1F_BITS;
    l->pixel_size_mult = (uint32_t)(c.data_stop - c.data_start);
    l->pixel_size_mult = (l->pixel_size_mult * ORC_INT_CALC_MULT + bit_count / 2) / bit_count;
    l->pixel_start_offset = c.data_start;
    l->halfpixel_size_mult = (l->pixel_size_mult + 2 / 2) / 2;
}
static uint8_t p1_get_ppb(const orc_p1_line *l) { return (uint8_t)(l->pixel_size_mult / ORC_INT_CALC_MULT); }   /* pcmline.cpp:235-238 */
static const int8_t P1_SH_BG_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:63-66 */
static const int8_t P1_SH_ED_TBL[ORC_PS_STAGES] = { 0, 1, -1, 2, -2 };   /* pcmline.h:68-71 */
static uint16_t p1_pixel_by_calc(const orc_p1_line *l, uint8_t pcm_bit, uint8_t in_shift)   /* pcmline.cpp:249-311, bit_ofs = 0 */
{
    int32_t video_pixel;
    if (pcm_bit >= P1F_BITS) pcm_bit = P1F_BITS - 1;
    video_pixel = (int32_t)((pcm_bit * l->pixel_size_mult) + l->halfpixel_size_mult);
    video_pixel = video_pixel / ORC_INT_CALC_MULT;
    video_pixel = video_pixel + l->pixel_start_offset;
    int8_t bg = P1_SH_BG_TBL[in_shift], ed = P1_SH_ED_TBL[in_shift];
    if (bg == ed) video_pixel += bg;
    else if (pcm_bit < P1F_LEFT_SHIFT) video_pixel += bg;
    else if (pcm_bit > P1F_RIGHT_SHIFT) video_pixel += ed;
    if (video_pixel < l->pixel_start) video_pixel = l->pixel_start;
    else if (video_pixel >= l->pixel_stop) video_pixel = l->pixel_stop - 1;
    return (uint16_t)video_pixel;
}
static void p1_calc_ppb(orc_p1_line *l, orc_coords c)   /* pcmline.cpp:223-232 + pcm1line.cpp:684-690 */
{
    p1_set_ppb(l, c);
    for (uint8_t s = 0; s < ORC_PS_STAGES; s++)
        for (uint8_t bit = 0; bit < P1F_BITS; bit++) l->pixel_coordinates[s][bit] = p1_pixel_by_calc(l, bit, s);
}

/* ------------------------------------------------------------------ AGC: BLACK / WHITE */
static void find_pcm1_bw(const orc_binarizer *b, uint16_t *sprd)   /* :2560-2600 */
{
    uint16_t pixel_limit, search_lim;
    uint32_t temp_calc;
    pixel_limit = (uint16_t)(b->scan_end - b->scan_start);
    temp_calc = pixel_limit / 32;
    search_lim = (uint16_t)(b->scan_end - (uint16_t)temp_calc);
    temp_calc = pixel_limit / 8;
    pixel_limit = (uint16_t)(b->scan_start + (uint16_t)temp_calc);
    for (uint16_t pixel = pixel_limit; pixel < search_lim; pixel++) sprd[PIX(b, pixel)]++;
}

static bool find_black_white(orc_binarizer *b, orc_p1_line *line)   /* :3116-3473 (PCM-1 branch of the type switch) */
{
    uint8_t brt_lev, br_black = 0, br_white = 255, useful_low, useful_high;
    uint8_t low_scan_limit, high_scan_limit, range_limit, bin_low, bin_high;
    uint16_t black_lvl_count, white_lvl_count, search_lim;
    uint32_t temp_calc;
    uint16_t sprd[256];
    bool black_level_detected, white_level_detected;

    memset(sprd, 0, sizeof(sprd));
    find_pcm1_bw(b, sprd);

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
static uint8_t fill_pcm1(orc_binarizer *b, orc_p1_line *l, uint8_t shift_stg)   /* :7016-7131 */
{
    bool prev_high = false;
    uint8_t pcm_bit = 0, pixel_val, low_ref = l->ref_low, high_ref = l->ref_high;
    uint8_t word_bit_pos = P1F_BITS_PER_WORD - 1, word_index = 0;
    uint16_t pcm_word = 0;
    while (pcm_bit <= (P1F_BITS - 1)) {
        pixel_val = PIX(b, l->pixel_coordinates[shift_stg][pcm_bit]);
        if (!prev_high) {
            if (pixel_val > low_ref) { pcm_word |= (uint16_t)(1 << word_bit_pos); prev_high = true; }
        } else {
            if (pixel_val >= high_ref) pcm_word |= (uint16_t)(1 << word_bit_pos);
            else prev_high = false;
        }
        if (word_bit_pos == 0) {
            p1_set_word(l, word_index, pcm_word);
            pcm_word = 0;
            word_index++;
            if (pcm_bit > (P1F_BITS - P1F_BITS_PER_CRC - 1)) break;
            else if (pcm_bit == (P1F_BITS - P1F_BITS_PER_CRC - 1)) word_bit_pos = P1F_BITS_PER_CRC;
            else word_bit_pos = P1F_BITS_PER_WORD;
        }
        word_bit_pos--;
        pcm_bit++;
    }
    p1_calc_crc(l);
    return ORC_STG_DATA_OK;
}

/* Bits whose sampling pixel was clamped to the edge of the picture were cut off by the capture: try every value of them and
 * keep the one and only combination that gives a valid CRC (:6116-6596) */
static uint8_t pick_cut_bits_up_pcm1(orc_binarizer *b, orc_p1_line *l)
{
    bool patch_found = false, coll_lock = false;
    uint8_t max_cut_bits, left_bit_count = 0, right_bit_count = 0;
    uint16_t index, idx_in;
    uint16_t first_pixel_coord, current_pixel_coord, left_rep_limit, right_rep_limit;
    uint16_t left_orig_word, right_orig_word, left_clean_word, right_clean_word;
    uint16_t left_patch_word, right_patch_word, left_fix_word = 0, right_fix_word = 0;

    l->picked_bits_left = 0; l->picked_bits_right = 0;
    max_cut_bits = b->digi_set.left_bit_pick;
    if (b->bin_mode == ORC_MODE_DRAFT) max_cut_bits = max_cut_bits / 2;
    first_pixel_coord = b->scan_start;
    for (index = 0; index < max_cut_bits; index++) {
        current_pixel_coord = l->pixel_coordinates[0][index];
        if ((current_pixel_coord - first_pixel_coord) >= ((p1_get_ppb(l) + 1) / 2)) break;
        if (index == 0) first_pixel_coord = current_pixel_coord;
        left_bit_count = (uint8_t)(index + 1);
    }
    left_rep_limit = (uint16_t)(1 << left_bit_count);
    first_pixel_coord = b->scan_end;
    max_cut_bits = b->digi_set.right_bit_pick;
    if (b->bin_mode == ORC_MODE_DRAFT) max_cut_bits = max_cut_bits / 2;
    for (index = 0; index < max_cut_bits; index++) {
        current_pixel_coord = l->pixel_coordinates[0][P1F_BITS - 1 - index];
        if ((first_pixel_coord - current_pixel_coord) >= ((p1_get_ppb(l) + 1) / 2)) break;
        if (index == 0) first_pixel_coord = current_pixel_coord;
        right_bit_count = (uint8_t)(index + 1);
    }
    right_rep_limit = (uint16_t)(1 << right_bit_count);

    if (b->force_bit_picker && orc_p1_crc_valid(l)) {
        /* forced on a line that reads fine: only report how many bits would have been picked (:6274-6289) */
        l->picked_bits_left = left_bit_count; l->picked_bits_right = right_bit_count;
        return ORC_STG_DATA_OK;
    }
    left_clean_word = right_clean_word = left_orig_word = right_orig_word = 0;
    if (left_bit_count > 0) {
        left_orig_word = l->words[0];
        left_clean_word = (uint16_t)((left_rep_limit - 1) << (P1F_BITS_PER_WORD - left_bit_count));
        left_clean_word = (uint16_t)~left_clean_word;
        left_clean_word = left_orig_word & left_clean_word;
    }
    if (right_bit_count > 0) {
        right_orig_word = l->words[P1F_WORD_CRCC];
        right_clean_word = (uint16_t)(right_rep_limit - 1);
        right_clean_word = (uint16_t)~right_clean_word;
        right_clean_word = right_orig_word & right_clean_word;
    }
    if ((left_bit_count > 0) && (right_bit_count > 0)) {
        for (index = 0; index < left_rep_limit; index++) {
            for (idx_in = 0; idx_in < right_rep_limit; idx_in++) {
                left_patch_word = (uint16_t)(index << (P1F_BITS_PER_WORD - left_bit_count));
                p1_set_word(l, 0, left_clean_word | left_patch_word);
                right_patch_word = idx_in;
                p1_set_word(l, P1F_WORD_CRCC, right_clean_word | right_patch_word);
                p1_calc_crc(l);
                if (orc_p1_crc_valid(l)) {
                    if (patch_found) { coll_lock = true; break; }
                    patch_found = true; left_fix_word = left_patch_word; right_fix_word = right_patch_word;
                }
            }
            if (coll_lock) break;
        }
        if (coll_lock) {
            p1_set_word(l, 0, left_orig_word); p1_set_word(l, P1F_WORD_CRCC, right_orig_word); p1_calc_crc(l);
            l->forced_bad = true;
            return ORC_STG_NO_GOOD;
        } else if (!patch_found) {
            p1_set_word(l, 0, left_orig_word); p1_set_word(l, P1F_WORD_CRCC, right_orig_word); p1_calc_crc(l);
            return ORC_STG_NO_GOOD;
        }
        p1_set_word(l, 0, left_clean_word | left_fix_word); p1_set_word(l, P1F_WORD_CRCC, right_clean_word | right_fix_word); p1_calc_crc(l);
        l->picked_bits_left = left_bit_count; l->picked_bits_right = right_bit_count;
        return ORC_STG_DATA_OK;
    } else if (left_bit_count > 0) {
        for (index = 0; index < left_rep_limit; index++) {
            left_patch_word = (uint16_t)(index << (P1F_BITS_PER_WORD - left_bit_count));
            p1_set_word(l, 0, left_clean_word | left_patch_word);
            p1_calc_crc(l);
            if (orc_p1_crc_valid(l)) {
                if (patch_found) { coll_lock = true; break; }
                patch_found = true; left_fix_word = left_patch_word;
            }
        }
        if (coll_lock) { p1_set_word(l, 0, left_orig_word); p1_calc_crc(l); l->forced_bad = true; return ORC_STG_NO_GOOD; }
        else if (!patch_found) { p1_set_word(l, 0, left_orig_word); p1_calc_crc(l); return ORC_STG_NO_GOOD; }
        p1_set_word(l, 0, left_clean_word | left_fix_word); p1_calc_crc(l);
        l->picked_bits_left = left_bit_count;
        return ORC_STG_DATA_OK;
    } else if (right_bit_count > 0) {
        for (index = 0; index < right_rep_limit; index++) {
            right_patch_word = index;
            p1_set_word(l, P1F_WORD_CRCC, right_clean_word | right_patch_word);
            p1_calc_crc(l);
            if (orc_p1_crc_valid(l)) {
                if (patch_found) { coll_lock = true; break; }
                patch_found = true; right_fix_word = right_patch_word;
            }
        }
        if (coll_lock) { p1_set_word(l, P1F_WORD_CRCC, right_orig_word); p1_calc_crc(l); l->forced_bad = true; return ORC_STG_NO_GOOD; }
        else if (!patch_found) { p1_set_word(l, P1F_WORD_CRCC, right_orig_word); p1_calc_crc(l); return ORC_STG_NO_GOOD; }
        p1_set_word(l, P1F_WORD_CRCC, right_clean_word | right_fix_word); p1_calc_crc(l);
        l->picked_bits_right = right_bit_count;
        return ORC_STG_DATA_OK;
    }
    return ORC_STG_NO_GOOD;
}

static uint8_t fill_data_words(orc_binarizer *b, orc_p1_line *l, uint8_t ref_delta, uint8_t shift_stg)   /* :7560-7650 */
{
    uint8_t low_ref, high_ref, bin_res;
    if (ref_delta > ORC_HYST_DEPTH_MAX) return ORC_STG_NO_GOOD;
    if (shift_stg > ORC_SHIFT_STAGES_MAX) return ORC_STG_NO_GOOD;
    low_ref = get_low_level(l->ref_level, ref_delta);
    high_ref = get_high_level(l->ref_level, ref_delta);
    l->ref_low = low_ref; l->ref_high = high_ref;
    if (low_ref <= l->black_level) { p1_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    if (high_ref >= l->white_level) { p1_set_invalid_crc(l); return ORC_STG_NO_GOOD; }
    l->hysteresis_depth = ref_delta;
    l->shift_stage = shift_stg;
    bin_res = fill_pcm1(b, l, shift_stg);
    if (bin_res == ORC_STG_DATA_OK)
        if ((!orc_p1_crc_valid(l) && (l->ref_level > b->digi_set.min_white_lvl) && ((b->digi_set.left_bit_pick != 0) || (b->digi_set.right_bit_pick != 0)))
            || b->force_bit_picker)
            pick_cut_bits_up_pcm1(b, l);
    return bin_res;
}

static void read_pcm_data(orc_binarizer *b, orc_p1_line *l)   /* :7695-8055 */
{
    bool invalid_hyst;
    uint8_t hyst_cnt, shift_try_cnt, valid_crcs_hyst, valid_crcs_shift, hyst_good_cnt;
    uint8_t valid_delta, valid_shift;

    p1_calc_ppb(l, l->coords);
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
                    if (orc_p1_crc_valid(l)) {
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
static uint8_t search_pcm1_data(orc_binarizer *b, orc_p1_line *l, orc_coords data_loc)   /* :4123-4511 */
{
    bool bitpick_previous;
    uint8_t right_ofs = 0xFF, left_ofs = 0xFF;
    uint8_t stat_left_idx, stat_right_idx, valid_left_crcs, valid_right_crcs;
    uint16_t scan_step = 1, step_span;
    orc_coords left_coord, right_coord;
    orc_crc_handler scan_left_res[P1F_SEARCH_STEP_CNT], scan_right_res[P1F_SEARCH_STEP_CNT];
    orc_crc_handler scan_left_crcs[ORC_MAX_COLL_CRCS], scan_right_crcs[ORC_MAX_COLL_CRCS];

    orc_coords_clear(&left_coord); orc_coords_clear(&right_coord);
    stat_left_idx = 2;
    while (stat_left_idx > 0) {
        p1_calc_ppb(l, data_loc);
        scan_step = p1_get_ppb(l);
        if (scan_step >= P1F_SEARCH_STEP_DIV) scan_step = scan_step / P1F_SEARCH_STEP_DIV; else scan_step = 1;
        step_span = (uint16_t)(scan_step * P1F_SEARCH_MAX_OFS);
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
        stat_left_idx--;
    }
    bitpick_previous = b->force_bit_picker;
    b->force_bit_picker = true;
    b->hysteresis_depth_lim = 0;
    if (b->bin_mode == ORC_MODE_DRAFT || b->bin_mode == ORC_MODE_FAST) b->shift_stages_lim = 0;
    else if (b->bin_mode == ORC_MODE_NORMAL || b->bin_mode == ORC_MODE_INSANE) b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE;

    reset_crc_stats(scan_left_res, P1F_SEARCH_STEP_CNT, NULL);
    reset_crc_stats(scan_left_crcs, ORC_MAX_COLL_CRCS, &valid_left_crcs);
    stat_left_idx = 0;
    for (int16_t start_ofs = left_coord.data_start; start_ofs <= left_coord.data_stop; start_ofs = (int16_t)(start_ofs + scan_step)) {
        reset_crc_stats(scan_right_res, P1F_SEARCH_STEP_CNT, NULL);
        reset_crc_stats(scan_right_crcs, ORC_MAX_COLL_CRCS, &valid_right_crcs);
        stat_right_idx = 0;
        for (int16_t stop_ofs = right_coord.data_stop; stop_ofs >= right_coord.data_start; stop_ofs = (int16_t)(stop_ofs - scan_step)) {
            orc_coords_set(&l->coords, start_ofs, stop_ofs);
            read_pcm_data(b, l);
            scan_right_res[stat_right_idx].crc = l->words[P1F_WORD_CRCC];
            scan_right_res[stat_right_idx].hyst_dph = l->hysteresis_depth;
            scan_right_res[stat_right_idx].shift_stg = l->shift_stage;
            scan_right_res[stat_right_idx].data_start = start_ofs;
            scan_right_res[stat_right_idx].data_stop = stop_ofs;
            scan_right_res[stat_right_idx].result = ORC_REF_BAD_CRC;
            /* picked bits make an entry less desirable in the vote */
            if ((l->picked_bits_left != 0) && (l->picked_bits_right != 0)) scan_right_res[stat_right_idx].hyst_dph = 0x0E;
            else if (l->picked_bits_right != 0) scan_right_res[stat_right_idx].hyst_dph = 0x0D;
            else if (l->picked_bits_left != 0) scan_right_res[stat_right_idx].hyst_dph = 0x0C;
            if (orc_p1_crc_valid(l)) {
                scan_right_res[stat_right_idx].result = ORC_REF_CRC_OK;
                update_crc_stats(scan_right_crcs, scan_right_res[stat_right_idx], &valid_right_crcs);
            }
            stat_right_idx++;
            if (stat_right_idx >= P1F_SEARCH_STEP_CNT) break;
        }
        if (valid_right_crcs > 0) {
            find_most_frequent_crc(scan_right_crcs, &valid_right_crcs, true);
            invalidate_non_frequent_crcs(scan_right_res, 0, P1F_SEARCH_STEP_CNT - 1, valid_right_crcs, scan_right_crcs[0].crc);
            if (valid_right_crcs > 0)
                if (pick_level_by_crc_stats(scan_right_res, &right_ofs, 0, P1F_SEARCH_STEP_CNT - 1, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX) != ORC_SPAN_OK)
                    valid_right_crcs = 0;
        }
        if (valid_right_crcs > 0) {
            scan_left_res[stat_left_idx].result = ORC_REF_CRC_OK;
            scan_left_res[stat_left_idx].crc = scan_right_res[right_ofs].crc;
            scan_left_res[stat_left_idx].hyst_dph = scan_right_res[right_ofs].hyst_dph;
            scan_left_res[stat_left_idx].shift_stg = scan_right_res[right_ofs].shift_stg;
            scan_left_res[stat_left_idx].data_start = scan_right_res[right_ofs].data_start;
            scan_left_res[stat_left_idx].data_stop = scan_right_res[right_ofs].data_stop;
            for (uint8_t index = 0; index < scan_right_crcs[0].result; index++) update_crc_stats(scan_left_crcs, scan_right_crcs[0], &valid_left_crcs);
        } else {
            scan_left_res[stat_left_idx].result = ORC_REF_BAD_CRC;
            scan_left_res[stat_left_idx].crc = 0;
            scan_left_res[stat_left_idx].hyst_dph = ORC_HYST_DEPTH_MAX;
            scan_left_res[stat_left_idx].shift_stg = ORC_SHIFT_STAGES_MAX;
        }
        stat_left_idx++;
        if (stat_left_idx >= P1F_SEARCH_STEP_CNT) break;
    }
    b->force_bit_picker = bitpick_previous;

    if (valid_left_crcs > 0) {
        find_most_frequent_crc(scan_left_crcs, &valid_left_crcs, true);
        invalidate_non_frequent_crcs(scan_left_res, 0, P1F_SEARCH_STEP_CNT - 1, valid_left_crcs, scan_left_crcs[0].crc);
        if (valid_left_crcs > 0)
            if (pick_level_by_crc_stats(scan_left_res, &left_ofs, 0, P1F_SEARCH_STEP_CNT - 1, ORC_REF_CRC_OK, 0x0F, ORC_SHIFT_STAGES_MAX) != ORC_SPAN_OK)
                valid_left_crcs = 0;
    }
    if (valid_left_crcs > 0) {
        l->coords.data_start = scan_left_res[left_ofs].data_start;
        l->coords.data_stop = scan_left_res[left_ofs].data_stop;
        l->coords_set = true;
        l->coords_sweeped = true;
        return ORC_LB_RET_OK;
    }
    /* CoordinatePair assignment copies every field (frametrimset.cpp:22-36) */
    l->coords = data_loc;
    l->coords_sweeped = false;
    return ORC_LB_RET_NO_COORD;
}

static bool find_pcm1_coordinates(orc_binarizer *b, orc_p1_line *l, orc_coords coord_history)   /* :5601-5813 */
{
    bool search_state;
    uint8_t in_hyst_depth, in_shift_stages;
    orc_coords data_coord;
    uint16_t line_margin;

    orc_coords_clear(&data_coord);
    line_margin = (uint16_t)(b->scan_end - b->scan_start);
    line_margin = line_margin / 16;
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
    in_hyst_depth = b->hysteresis_depth_lim;
    in_shift_stages = b->shift_stages_lim;
    if (search_pcm1_data(b, l, data_coord) == ORC_LB_RET_OK) search_state = true;
    b->hysteresis_depth_lim = in_hyst_depth;
    b->shift_stages_lim = in_shift_stages;
    b->p1_scan_done = true;        /* video_line->scan_done (:5810) */
    return search_state;
}

/* ------------------------------------------------------------------ reference level sweep (MODE_INSANE only for this format, :1105-1112) */
/* Binarizer::sweepRefLevel (:3551-3817) with a PCM1Line as the trial line.  The first-try read from the preset coordinates never
 * runs for this format (skip_bin stays false, :3647-3679): every level runs the whole coordinate search.  clear() through the
 * PCMLine pointer is the base clear() (it is not virtual): words and picked bits of the trial line persist from level to level. */
static void sweep_ref_level_p1(orc_binarizer *b, orc_p1_line *pcm_line, orc_crc_handler *crc_res)
{
    uint8_t low_lvl, high_lvl, read_result;
    uint16_t ref_index;
    orc_p1_line t;
    orc_coords forced_coords;

    orc_p1_clear(&t);                       /* PCM1Line temp_pcm1 (constructor) */
    calc_forced_coords(b, &forced_coords);
    low_lvl = (uint8_t)(pcm_line->black_level + 1); high_lvl = (uint8_t)(pcm_line->white_level - 1);
    if (b->digi_set.min_ref_lvl > low_lvl) low_lvl = b->digi_set.min_ref_lvl;
    if (b->digi_set.max_ref_lvl < high_lvl) high_lvl = b->digi_set.max_ref_lvl;
    ref_index = high_lvl;
    while (ref_index >= low_lvl) {
        p1_base_clear(&t);
        p1_set_source_pixels(&t, 0, (uint16_t)(b->video_line->length - 1));
        t.coords.from_doubled = b->video_line->doubled;
        t.black_level = low_lvl; t.white_level = high_lvl;
        t.ref_level = (uint8_t)ref_index;
        if (!orc_p1_crc_valid(&t)) {
            if (!orc_coords_valid(&forced_coords)) find_pcm1_coordinates(b, &t, b->in_def_coord);
            else { t.coords = forced_coords; t.coords_set = true; }
            if (t.coords_set) read_pcm_data(b, &t);
        }
        if (t.picked_bits_left != 0 && t.picked_bits_right != 0) t.hysteresis_depth = (uint8_t)(t.hysteresis_depth + ORC_HYST_DEPTH_MAX + 3);   /* :3735-3752 */
        else if (t.picked_bits_right != 0) t.hysteresis_depth = (uint8_t)(t.hysteresis_depth + ORC_HYST_DEPTH_MAX + 2);
        else if (t.picked_bits_left != 0) t.hysteresis_depth = (uint8_t)(t.hysteresis_depth + ORC_HYST_DEPTH_MAX + 1);
        if (t.hysteresis_depth > 0x0F) t.hysteresis_depth = 0x0F;
        read_result = ORC_REF_NO_PCM;
        if (orc_p1_crc_valid(&t) && orc_coords_valid(&t.coords)) read_result = ORC_REF_CRC_OK;
        else if (t.coords_set) read_result = ORC_REF_BAD_CRC;
        if (read_result != ORC_REF_NO_PCM) {
            crc_res[ref_index].result = read_result;
            crc_res[ref_index].data_start = t.coords.data_start;
            crc_res[ref_index].data_stop = t.coords.data_stop;
            crc_res[ref_index].hyst_dph = t.hysteresis_depth;
            crc_res[ref_index].shift_stg = t.shift_stage;
            crc_res[ref_index].crc = t.calc_crc;
        }
        if (ref_index == 0) break;          /* (the reference's uint16_t counter would wrap; min_ref_lvl > 0 in every preset) */
        ref_index--;
    }
}

/* Binarizer::calcRefLevelBySweep (:3821-4120), the branches a line without markers takes */
static void calc_ref_level_by_sweep_p1(orc_binarizer *b, orc_p1_line *pcm_line)
{
    uint8_t fast_ref, bin_level, valid_crc_cnt, span_res;
    orc_crc_handler scan_sweep_crcs[256];
    const uint8_t blk1 = (uint8_t)(pcm_line->black_level + 1), wht1 = (uint8_t)(pcm_line->white_level - 1);

    fast_ref = pick_center_ref_level(b, pcm_line->black_level, pcm_line->white_level);
    b->hysteresis_depth_lim = 0;
    b->shift_stages_lim = ORC_SHIFT_STAGES_SAFE;
    reset_crc_stats(scan_sweep_crcs, 256, NULL);
    sweep_ref_level_p1(b, pcm_line, scan_sweep_crcs);
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

/* ------------------------------------------------------------------ Binarizer::processLine, PCM1Line output */
void orc_binarizer_set_good_parameters_p1(orc_binarizer *b, const orc_p1_line *l)   /* :353-377 */
{
    if (l == NULL) {
        orc_binarizer_set_reference_level(b, 0);
        orc_binarizer_set_data_coordinates2(b, 0, 0);
        orc_binarizer_set_bw_levels(b, 0, 0);
    } else if (orc_p1_crc_valid_ignore_forced(l)) {
        orc_binarizer_set_reference_level(b, l->ref_level);
        orc_binarizer_set_data_coordinates(b, l->coords);
        orc_binarizer_set_bw_levels(b, l->black_level, l->white_level);
    }
}

uint8_t orc_binarizer_process_line_p1(orc_binarizer *b, orc_p1_line *out)   /* :443-1724 */
{
    uint8_t stage_count;
    uint32_t tmp_calc;
    orc_coords forced_coords;
    const orc_video_line *vl = b->video_line;

    if (vl == NULL) return ORC_LB_RET_NULL_VIDEO;
    if (out == NULL) return ORC_LB_RET_NULL_PCM;
    orc_p1_clear(out);
    out->frame_number = vl->frame_number;
    out->line_number = vl->line_number;
    b->p1_scan_done = false;

    if (vl->service_type != ORC_SRV_NO) {
        if (vl->service_type >= ORC_SRV_NEW_FILE && vl->service_type <= ORC_SRV_END_FRAME) p1_set_service(out, vl->service_type);
    } else if (!vl->empty) {
        b->line_length = vl->length;
        out->coords.from_doubled = vl->doubled;
        b->scan_start = 0;
        b->scan_end = (uint16_t)(b->line_length - 1);
        p1_set_source_pixels(out, b->scan_start, b->scan_end);
        if (b->line_length < P1F_BITS) return ORC_LB_RET_SHORT_LINE;
        b->mark_start_max = 0; b->mark_end_min = 0xFFFF;         /* no markers in PCM-1 (:592-604) */
        tmp_calc = (uint32_t)b->line_length * ORC_INT_CALC_MULT;
        tmp_calc = tmp_calc / P1F_BITS;
        b->estimated_ppb = (uint16_t)((tmp_calc + (ORC_INT_CALC_MULT / 2)) / ORC_INT_CALC_MULT);
        orc_coords_set(&out->coords, (int16_t)b->scan_start, (int16_t)b->scan_end);
        calc_forced_coords(b, &forced_coords);
        if (b->digi_set.en_force_coords && orc_coords_valid(&forced_coords)) {
            /* CoordinatePair assignment: the doubled flag of forced_coords (false after clear()) comes with it */
            out->coords = forced_coords;
            out->coords_set = true;
        }
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
                    if (orc_p1_crc_valid(out)) { out->data_by_ext_tune = true; b->proc_state = ORC_STG_DATA_OK; }
                    else b->proc_state = ORC_STG_REF_FIND;                   /* no coordinates-only retry for PCM-1 (:910-918) */
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
                    b->do_ref_lvl_sweep = (b->bin_mode == ORC_MODE_INSANE);          /* :1104-1112 */
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
                            if (b->digi_set.en_coord_search && b->do_coord_search) find_pcm1_coordinates(b, out, b->in_def_coord);
                        }
                        if (!out->coords_set) { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                        else { b->hysteresis_depth_lim = b->in_max_hysteresis_depth; b->shift_stages_lim = b->in_max_shift_stages; }
                    }
                }
            } else if (b->proc_state == ORC_STG_REF_SWEEP_RUN) {        /* :1391-1400 */
                calc_ref_level_by_sweep_p1(b, out);
                b->proc_state = ORC_STG_READ_PCM;
            } else if (b->proc_state == ORC_STG_READ_PCM) {                 /* :1401-1533 */
                if (orc_coords_valid(&forced_coords)) { b->hysteresis_depth_lim = ORC_HYST_DEPTH_SAFE; b->shift_stages_lim = ORC_SHIFT_STAGES_MIN; }
                if (out->coords_set) read_pcm_data(b, out);
                if (orc_p1_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                if (b->proc_state != ORC_STG_DATA_OK) {
                    if (orc_coords_valid(&b->in_def_coord) && !orc_coords_valid(&forced_coords) && !b->do_ref_lvl_sweep
                        && !out->forced_bad && !out->coords_set) {
                        if (coords_ne(&out->coords, &b->in_def_coord)) {
                            out->coords = b->in_def_coord;
                            read_pcm_data(b, out);
                            if (orc_p1_crc_valid(out)) b->proc_state = ORC_STG_DATA_OK;
                        }
                    }
                    if (b->proc_state != ORC_STG_DATA_OK) b->proc_state = ORC_STG_NO_GOOD;
                }
            } else if (b->proc_state == ORC_STG_DATA_OK) {                  /* :1534-1621 */
                if (out->forced_bad) b->proc_state = ORC_STG_NO_GOOD;
                else {
                    if (orc_p1_has_header(out)) p1_set_service(out, ORC_SRV_HEADER_LINE);      /* setServHeader, pcm1line.cpp:80-86 */
                    out->coords.from_doubled = vl->doubled;
                    break;
                }
            } else if (b->proc_state == ORC_STG_NO_GOOD) {                  /* :1622-1669 */
                if (orc_p1_crc_valid(out)) p1_set_invalid_crc(out);
                out->coords.from_doubled = vl->doubled;
                break;
            } else break;
            if (stage_count > ORC_STG_MAX) break;
        } while (1);
    } else {
        p1_set_silent(out);
        p1_set_invalid_crc(out);
    }
    return ORC_LB_RET_OK;
}
