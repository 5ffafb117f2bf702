/*
 * pcm16.c - CPU restatement of the PCM-16x0 back half: PCM16X0DataBlock (pcm16x0datablock.cpp), PCM16X0Deinterleaver::processBlock
 * (pcm16x0deinterleaver.cpp:128-912) and PCM16X0DataStitcher (pcm16x0datastitcher.cpp:64-5856).
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).  Follows the reference statement by statement, including what looks like slips
 * (isDataFixedByBP over all sub-blocks, the line/sub-line mix-up of the last-resort padding in fillFrameForOutput, odd_res tested
 * in the even branch of findSIDataAlignment): the reference's outputs are the contract.
 */
#include "pcm16.h"
#include "bin_pcm16.h"
#include <stdlib.h>
#include <string.h>

enum { LINES_PF = 245, SUBLINES_PF = 735, SI_OFS = 35, EI_OFS = 490, SI_TRUE = 105, EI_TRUE = 490, IBLK_PF = 7 };
enum { BUF_TRIM = 3 * 640 * 3, MIN_GOOD_SUB = 35 * 6 * 3, MIN_FILL_SI = 105, MIN_FILL_EI = 82 * 3 };
enum { IBLK_DELIM = 45, MAX_PAD_SI = 35, MAX_PAD_EI = 81, MAX_SIL_SI = 34, MAX_SIL_EI = 81 * 3, MAX_BROKEN = 1, MAX_UNCH_SI = 34, MAX_UNCH_EI = 81 * 3,
       MIN_VALID_SI = 17, MIN_VALID_EI = 490 / 3, INVALID_PAD = 0xFF, STATS_DEPTH = 65 };
enum { BIT_EMPH = 0, BIT_RATE = 3, BIT_MODE = 6, BIT_CODE = 9, BIT_MAX = 12 };
enum { DS_NO_DATA, DS_SILENCE, DS_BROKE, DS_NO_PAD, DS_OK };
enum { DI_NULL_LINES, DI_NULL_BLOCK, DI_NO_DATA, DI_OK };
enum { L1 = 0, L2 = 1, L3 = 2, W_L = 0, W_R = 1, W_P = 2 };
enum { AUD_ORIG, AUD_FIX_P, AUD_BROKEN };
enum { ORDER_TFF = 1, ORDER_BFF = 2 };
enum { P16_CRC_SILENT = 0x0E10 };

/* ------------------------------------------------------------------ PCM16X0SubLine as the stitcher sees it */
typedef struct {
    uint32_t frame_number; uint16_t line_number, words[4], calc_crc, queue_order;
    uint8_t ref_level, picked_left, picked_right, line_part, service_type;
    bool control_bit, bw_set, forced_bad, coords_valid;
    const sdv_pcm16x0_bin_rec *src;     /* the record the object was made from (NULL: made by the stitcher) - what the visualiser's feed hands back */
} p16_sub;

static void sub_clear(p16_sub *s)   /* PCM16X0SubLine::clear (pcm16x0subline.cpp:63-86) over PCMLine::clear (pcmline.cpp:95-115) */
{
    memset(s, 0, sizeof(*s));
    s->control_bit = true;
    s->calc_crc = P16_CRC_SILENT; s->words[3] = (uint16_t)~P16_CRC_SILENT;
}
static void sub_from_rec(const sdv_pcm16x0_bin_rec *r, p16_sub *s)
{
    if (r->service_type != SDV_SRV_NO) {
        /* a service line is a fresh PCM16X0SubLine that PCMLine::setServiceLine cleared (pcmline.cpp:490-502; clear() is not virtual):
         * silent words against the inverted silent CRC, calc_crc 0, no levels - whatever else the record holds is ignored */
        sub_clear(s);
        s->calc_crc = 0; s->frame_number = r->frame_number; s->line_number = r->line_number; s->service_type = r->service_type;
        s->src = r;
        return;
    }
    memset(s, 0, sizeof(*s));
    s->frame_number = r->frame_number; s->line_number = r->line_number;
    for (int i = 0; i < 4; i++) s->words[i] = r->words[i];
    s->calc_crc = r->calc_crc; s->queue_order = r->queue_order; s->ref_level = r->ref_level;
    s->picked_left = r->picked_bits_left; s->picked_right = r->picked_bits_right; s->line_part = r->line_part; s->service_type = r->service_type;
    s->control_bit = r->control_bit != 0; s->bw_set = (r->flags & SDV_LF_BW_SET) != 0; s->forced_bad = (r->flags & SDV_LF_FORCED_BAD) != 0;
    s->coords_valid = r->data_start != -32768 && r->data_stop != 32767 && r->data_start < r->data_stop;      /* CoordinatePair::areValid */
    s->src = r;
}
static bool sub_crc_if(const p16_sub *s) { return s->calc_crc == s->words[3]; }          /* isCRCValidIgnoreForced (pcm16x0subline.cpp:284-291) */
static bool sub_crc(const p16_sub *s) { return !s->forced_bad && sub_crc_if(s); }       /* PCMLine::isCRCValid (pcmline.cpp:360-367) */
static bool sub_service(const p16_sub *s) { return s->service_type != SDV_SRV_NO; }

/* ------------------------------------------------------------------ PCM16X0DataBlock */
typedef struct {
    uint32_t frame_number; uint16_t start_line, stop_line, queue_order, sample_rate; uint8_t start_part, stop_part;
    bool emphasis, ei_format, code;
    uint16_t words[3][3]; bool word_crc[3][3], word_valid[3][3], picked_left[3], picked_crc[3]; uint8_t audio_state[3]; bool order_even;
} p16_block;

static void blk_clear(p16_block *b) { memset(b, 0, sizeof(*b)); b->sample_rate = 44056; }     /* pcm16x0datablock.cpp:75-102 */
static void blk_set_word(p16_block *b, int blk, int line, uint16_t w, bool valid, bool pl, bool pc)   /* :105-120 */
{
    b->words[blk][line] = w; b->word_crc[blk][line] = valid; b->word_valid[blk][line] = valid;
    if (blk == 0) b->picked_left[line] = pl;
    b->picked_crc[line] = pc;
}
static int blk_word_to_line(const p16_block *b, int blk, int word)   /* getWordToLine :1029-1155 */
{
    if (word == W_P) return L2;
    bool l_first = ((blk & 1) != 0) != b->order_even;     /* odd order: sub-block 2 has L on LINE_1; even order: sub-blocks 1 and 3 */
    if (word == W_L) return l_first ? L1 : L3;
    return l_first ? L3 : L1;
}
static void blk_fix_word(p16_block *b, int blk, int word, uint16_t w) { int l = blk_word_to_line(b, blk, word); b->words[blk][l] = w; b->word_valid[blk][l] = true; }   /* :150-158 */
static void blk_mark_bad(p16_block *b, int blk, int line) { b->word_crc[blk][line] = false; b->word_valid[blk][line] = false; b->picked_left[line] = false; }     /* :271-279 */
static void blk_mark_broken(p16_block *b, int blk)   /* :231-268 */
{
    for (int i = 0; i < 3; i++)
        if (blk >= 3 || i == blk) { for (int l = 0; l < 3; l++) { b->word_valid[i][l] = false; b->word_crc[i][l] = false; } b->audio_state[i] = AUD_BROKEN; }
}
static int blk_errors_audio(const p16_block *b, int blk)   /* :688-721 */
{
    int n = 0;
    for (int i = 0; i < 3; i++) if (blk >= 3 || i == blk) { if (!b->word_crc[i][L1]) n++; if (!b->word_crc[i][L3]) n++; }
    return n;
}
static int blk_errors_fixed_audio(const p16_block *b, int blk)   /* :724-757 */
{
    int n = 0;
    for (int i = 0; i < 3; i++) if (blk >= 3 || i == blk) { if (!b->word_valid[i][L1]) n++; if (!b->word_valid[i][L3]) n++; }
    return n;
}
static int blk_errors_total(const p16_block *b, int blk)   /* :760-791 */
{
    int n = 0;
    for (int i = 0; i < 3; i++) if (blk >= 3 || i == blk) for (int l = 0; l < 3; l++) if (!b->word_crc[i][l]) n++;
    return n;
}
static bool blk_valid(const p16_block *b, int blk) { return blk_errors_fixed_audio(b, blk) == 0; }   /* isBlockValid :510-517 */
static bool blk_fixed_by_p(const p16_block *b, int blk)   /* :520-539 */
{
    if (blk < 3) return b->audio_state[blk] == AUD_FIX_P;
    return b->audio_state[0] == AUD_FIX_P || b->audio_state[1] == AUD_FIX_P || b->audio_state[2] == AUD_FIX_P;
}
static bool blk_broken(const p16_block *b, int blk)   /* :571-590 */
{
    if (blk < 3) return b->audio_state[blk] == AUD_BROKEN;
    return b->audio_state[0] == AUD_BROKEN || b->audio_state[1] == AUD_BROKEN || b->audio_state[2] == AUD_BROKEN;
}
static bool blk_picked_left_by_sub(const p16_block *b, int blk) { return blk <= 0 && (b->picked_left[L1] || b->picked_left[L3]); }   /* :311-322 */
static bool blk_picked_crc_by_sub(const p16_block *b, int blk)   /* :335-357 */
{
    if (blk >= 3) return false;
    if (b->picked_crc[L1] || b->picked_crc[L3]) return true;
    return b->picked_crc[L2] && blk_fixed_by_p(b, blk);
}
static bool blk_fixed_by_bp(const p16_block *b, int blk)   /* :542-562 - over all sub-blocks the loop asks about `blk` itself */
{
    if (!blk_valid(b, blk)) return false;
    return blk_picked_left_by_sub(b, blk) || blk_picked_crc_by_sub(b, blk);
}
static bool blk_has_picked_sample(const p16_block *b, int blk, int word)   /* hasPickedSample(blk, word < WORD_P) :408-434 */
{
    return blk == 0 && b->picked_left[blk_word_to_line(b, blk, word)];
}
static int blk_picked_audio_samples(const p16_block *b, int blk)   /* :458-475 */
{
    if (blk != 0) return 0;
    return (blk_has_picked_sample(b, 0, W_L) ? 1 : 0) + (blk_has_picked_sample(b, 0, W_R) ? 1 : 0);
}
static bool blk_has_picked_parity(const p16_block *b, int blk) { return (blk == 0 && b->picked_left[L2]) || b->picked_crc[L2]; }   /* :437-454 */
static bool blk_has_picked_word_any(const p16_block *b, int blk)   /* hasPickedWord(blk) :360-405, word = WORD_CNT */
{
    if (blk == 0) return b->picked_left[L1] || b->picked_crc[L1] || b->picked_left[L2] || b->picked_crc[L2] || b->picked_left[L3] || b->picked_crc[L3];
    return b->picked_crc[L1] || b->picked_crc[L3];
}
static bool blk_crc_ok(const p16_block *b, int blk, int word) { return b->word_crc[blk][blk_word_to_line(b, blk, word)]; }
static bool blk_word_valid(const p16_block *b, int blk, int word) { return b->word_valid[blk][blk_word_to_line(b, blk, word)]; }
static uint16_t blk_word(const p16_block *b, int blk, int word) { return b->words[blk][blk_word_to_line(b, blk, word)]; }
static int16_t blk_sample(const p16_block *b, int blk, int word) { return (int16_t)blk_word(b, blk, word); }
static bool blk_can_force_check(const p16_block *b) { return !blk_broken(b, 3) && blk_errors_total(b, 3) == 0; }   /* :282-298 */
static bool blk_silent(const p16_block *b)   /* :626-637 */
{
    for (int i = 0; i < 3; i++) if (blk_sample(b, i, W_L) != 0 || blk_sample(b, i, W_R) != 0) return false;
    return true;
}
static void blk_mark_unsafe(p16_block *b)   /* :186-228 */
{
    for (int i = 0; i < 3; i++) {
        bool full_bad = !b->word_crc[i][L2] && blk_errors_audio(b, i) > 0;
        if (b->audio_state[i] != AUD_BROKEN) {
            if (!full_bad) { b->word_valid[i][L1] = b->word_crc[i][L1]; b->word_valid[i][L3] = b->word_crc[i][L3]; }
            else { b->word_valid[i][L1] = false; b->word_valid[i][L3] = false; }
            b->audio_state[i] = AUD_ORIG;
        }
    }
}

/* ------------------------------------------------------------------ PCM16X0Deinterleaver */
typedef struct { bool force_ecc_check, en_p_code, ignore_crc, ei_format; } p16_di;
enum { STG_CRC_CHECK, STG_P_CORR, STG_BAD_BLOCK, STG_NO_CHECK, STG_DATA_OK, STG_CONVERT_MAX };
enum { FIX_NOT_NEED, FIX_BROKEN, FIX_DONE };
enum { NO_ERR_INDEX = 64 };

static void di_set_word_data(const p16_di *d, const p16_sub *l1, const p16_sub *l2, const p16_sub *l3, p16_block *b)   /* setWordData :711-787 */
{
    const p16_sub *ls[3] = { l1, l2, l3 };
    for (int line = 0; line < 3; line++) {
        const p16_sub *l = ls[line];
        bool ok = d->ignore_crc ? (l->coords_valid && l->bw_set) : sub_crc(l);
        blk_set_word(b, 0, line, l->words[0], ok, l->picked_left != 0, l->picked_right != 0);
        blk_set_word(b, 1, line, l->words[1], ok, false, l->picked_right != 0);
        blk_set_word(b, 2, line, l->words[2], ok, false, l->picked_right != 0);
    }
    b->frame_number = l1->frame_number; b->start_line = l1->line_number; b->start_part = l1->line_part;
    b->stop_line = l3->line_number; b->stop_part = l3->line_part; b->queue_order = l3->queue_order;
}
static uint16_t di_syndrome_p(const p16_block *b, int blk) { return (uint16_t)(blk_word(b, blk, W_L) ^ blk_word(b, blk, W_R) ^ blk_word(b, blk, W_P)); }   /* :790-803 */
static int di_fix_by_p(p16_block *b, int blk, int bad_ptr, uint16_t synd_mask)   /* fixByP :806-912 */
{
    uint16_t check = di_syndrome_p(b, blk);
    if (check == 0) { if (bad_ptr != NO_ERR_INDEX) blk_fix_word(b, blk, bad_ptr, blk_word(b, blk, bad_ptr)); return FIX_NOT_NEED; }
    if (bad_ptr == NO_ERR_INDEX) return FIX_BROKEN;
    if ((synd_mask & check) == 0) { blk_fix_word(b, blk, bad_ptr, (uint16_t)(check ^ blk_word(b, blk, bad_ptr))); return FIX_DONE; }
    return FIX_BROKEN;
}
/* processBlock :128-708 over the queue q[0..qn) */
static int di_process_block(const p16_di *d, const p16_sub *q, size_t qn, uint16_t line_sh, bool even_order, p16_block *b)
{
    uint16_t min_data = (uint16_t)((d->ei_format ? 2 * EI_OFS : 2 * SI_OFS) + line_sh);
    if (qn <= (size_t)min_data) return DI_NO_DATA;
    b->order_even = even_order;
    int step = d->ei_format ? EI_OFS : SI_OFS;
    const p16_sub *l1 = &q[line_sh], *l2 = &q[line_sh + step], *l3 = &q[line_sh + 2 * step];
    di_set_word_data(d, l1, l2, l3, b);
    uint8_t pick_cnt = (uint8_t)(l1->picked_left + l2->picked_left + l3->picked_left);
    for (int i = 0; i < 3; i++) b->audio_state[i] = AUD_ORIG;
    for (int blk = 0; blk < 3; blk++) {
        int state = STG_CRC_CHECK, stage_count = 0;
        int err_total = blk_errors_total(b, blk), err_audio = blk_errors_audio(b, blk);
        uint16_t pick_mask = 0;
        for (;;) {
            stage_count++;
            if (state == STG_CRC_CHECK) {
                if (err_total > 1) state = STG_BAD_BLOCK;
                else if (d->en_p_code) {
                    if (d->force_ecc_check) state = STG_P_CORR;
                    else if (err_total > 0) state = err_audio > 0 ? STG_P_CORR : STG_DATA_OK;
                    else state = STG_DATA_OK;
                } else {
                    if (err_audio > 0) state = STG_BAD_BLOCK;
                    else if (d->force_ecc_check) state = STG_NO_CHECK;
                    else state = STG_DATA_OK;
                }
            } else if (state == STG_P_CORR) {
                int bad_ptr = NO_ERR_INDEX;
                if (!blk_crc_ok(b, blk, W_L)) bad_ptr = W_L;
                else if (!blk_crc_ok(b, blk, W_R)) bad_ptr = W_R;
                else if (!blk_crc_ok(b, blk, W_P)) bad_ptr = W_P;
                if (bad_ptr != W_P) {
                    int fix = di_fix_by_p(b, blk, bad_ptr, pick_mask);
                    if (fix == FIX_BROKEN) {
                        if (blk_picked_audio_samples(b, blk) > 1) { blk_mark_bad(b, blk, L1); blk_mark_bad(b, blk, L3); state = STG_BAD_BLOCK; }
                        else if (blk_picked_audio_samples(b, blk) == 1) {
                            if (blk_has_picked_parity(b, blk)) { blk_mark_bad(b, blk, L1); blk_mark_bad(b, blk, L3); state = STG_BAD_BLOCK; }
                            else {
                                if (b->picked_left[L1]) { blk_mark_bad(b, blk, L1); state = STG_P_CORR; }
                                else if (b->picked_left[L3]) { blk_mark_bad(b, blk, L3); state = STG_P_CORR; }
                                else { state = STG_BAD_BLOCK; blk_mark_broken(b, 3); }
                                if (pick_cnt > 0) {
                                    uint16_t m = (uint16_t)(16 - pick_cnt);
                                    m = (uint16_t)(1u << (m & 31));         /* a shift count past the word (x86 takes it modulo 32) */
                                    pick_mask = (uint16_t)(m - 1);
                                }
                            }
                        } else {
                            if (blk_has_picked_parity(b, blk)) { blk_mark_bad(b, blk, L2); state = STG_NO_CHECK; }
                            else { state = STG_BAD_BLOCK; blk_mark_broken(b, blk); }
                        }
                    } else if (fix == FIX_NOT_NEED) state = STG_DATA_OK;
                    else { state = STG_DATA_OK; b->audio_state[blk] = AUD_FIX_P; }
                } else state = STG_NO_CHECK;
            } else break;
            if (stage_count > STG_CONVERT_MAX) break;
        }
    }
    return DI_OK;
}

void orc_pcm16x0_deint_blocks(const sdv_pcm16x0_bin_rec *lines, size_t n_lines, int ei_format, int force_check, int p_code, int ignore_crc,
                              int first_shift, int first_even, orc_p16_block_rec *out, size_t n_blocks)
{
    p16_sub *q = (p16_sub *)malloc((n_lines ? n_lines : 1) * sizeof(p16_sub));
    for (size_t i = 0; i < n_lines; i++) sub_from_rec(&lines[i], &q[i]);
    p16_di d = { force_check != 0, p_code != 0, ignore_crc != 0, ei_format != 0 };
    bool even = first_even != 0;
    for (size_t k = 0; k < n_blocks; k++) {
        p16_block b; blk_clear(&b);
        int ret = di_process_block(&d, q, n_lines, (uint16_t)(first_shift + (int)k), even, &b);
        orc_p16_block_rec *o = &out[k];
        memset(o, 0, sizeof(*o));
        o->frame_number = b.frame_number; o->start_line = b.start_line; o->stop_line = b.stop_line; o->queue_order = b.queue_order;
        o->start_part = b.start_part; o->stop_part = b.stop_part;
        for (int i = 0; i < 3; i++) {
            for (int w = 0; w < 3; w++) { o->words[i][w] = blk_word(&b, i, w); o->word_crc[i][w] = blk_crc_ok(&b, i, w); o->word_valid[i][w] = blk_word_valid(&b, i, w); }
            o->picked_left[i] = b.picked_left[i]; o->picked_crc[i] = b.picked_crc[i]; o->audio_state[i] = b.audio_state[i];
        }
        o->order_even = b.order_even; o->ret = (uint8_t)ret;
        even = !even;
    }
    free(q);
}

/* ------------------------------------------------------------------ PCM16X0DataStitcher */
typedef struct { uint16_t index, valid, silent, unchecked, broken; } p16_stats;
static void stats_clear(p16_stats *s) { s->index = s->valid = 0; s->silent = s->unchecked = s->broken = 0xFF; }   /* frametrimset.cpp:374-378 */
static bool stats_less(const p16_stats *a, const p16_stats *b)   /* FieldStitchStats::operator< (frametrimset.cpp:312-370) */
{
    if (a->broken != b->broken) return a->broken < b->broken;
    if (a->valid != b->valid) return a->valid > b->valid;
    if (a->unchecked != b->unchecked) return a->unchecked < b->unchecked;
    if (a->silent != b->silent) return a->silent < b->silent;
    return a->index < b->index;
}
/* std::sort(...)[0]: the comparator is a strict total order for distinct indices, so the front of the sorted range is the minimum */
static p16_stats stats_min(const p16_stats *v, int n)
{
    p16_stats m = v[0];
    for (int i = 1; i < n; i++) if (stats_less(&v[i], &m)) m = v[i];
    return m;
}

typedef struct {   /* FrameAsmPCM16x0 (frametrimset.h:116-249) */
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop;
    uint16_t odd_top_padding, odd_bottom_padding, even_top_padding, even_bottom_padding, blocks_broken, blocks_fix_bp, blocks_fix_p, blocks_fix_cwd;
    uint8_t field_order, odd_ref, even_ref, service_type;
    bool order_preset, order_guessed, odd_emphasis, even_emphasis, silence, padding_ok, ei_format;
} p16_frasm;
static void frasm_clear_misc(p16_frasm *f)   /* FrameAsmPCM16x0::clearMisc (frametrimset.cpp:812-822) over FrameAsmDescriptor::clearMisc (:466-478) */
{
    uint32_t fn = f->frame_number; uint16_t a = f->odd_top_data, b = f->odd_bottom_data, c = f->even_top_data, d = f->even_bottom_data;
    memset(f, 0, sizeof(*f));
    f->frame_number = fn; f->odd_top_data = a; f->odd_bottom_data = b; f->even_top_data = c; f->even_bottom_data = d;
    f->silence = true;
}
static void frasm_clear(p16_frasm *f)   /* FrameAsmPCM16x0::clear (:803-809) */
{
    memset(f, 0, sizeof(*f));
    f->odd_bottom_data = f->even_bottom_data = 0xFFFF;
    f->silence = true;
}
static void frasm_to_pod(const p16_frasm *f, sdv_frame_asm_pcm16x0 *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f->frame_number;
    o->odd_std_lines = f->odd_std_lines; o->even_std_lines = f->even_std_lines; o->odd_data_lines = f->odd_data_lines; o->even_data_lines = f->even_data_lines;
    o->odd_valid_lines = f->odd_valid_lines; o->even_valid_lines = f->even_valid_lines;
    o->odd_top_data = f->odd_top_data; o->odd_bottom_data = f->odd_bottom_data; o->even_top_data = f->even_top_data; o->even_bottom_data = f->even_bottom_data;
    o->odd_sample_rate = f->odd_sample_rate; o->even_sample_rate = f->even_sample_rate;
    o->blocks_total = f->blocks_total; o->blocks_drop = f->blocks_drop; o->samples_drop = f->samples_drop;
    o->odd_top_padding = f->odd_top_padding; o->odd_bottom_padding = f->odd_bottom_padding; o->even_top_padding = f->even_top_padding; o->even_bottom_padding = f->even_bottom_padding;
    o->blocks_broken = f->blocks_broken; o->blocks_fix_bp = f->blocks_fix_bp; o->blocks_fix_p = f->blocks_fix_p; o->blocks_fix_cwd = f->blocks_fix_cwd;
    o->field_order = f->field_order; o->odd_ref = f->odd_ref; o->even_ref = f->even_ref; o->service_type = f->service_type;
    o->flags = (uint8_t)((f->order_preset ? SDV_FA_ORDER_PRESET : 0) | (f->order_guessed ? SDV_FA_ORDER_GUESSED : 0) |
                         (f->odd_emphasis ? SDV_FA1_ODD_EMPHASIS : 0) | (f->even_emphasis ? SDV_FA1_EVEN_EMPHASIS : 0) |
                         (f->silence ? SDV_FA16_SILENCE : 0) | (f->padding_ok ? SDV_FA16_PADDING_OK : 0) | (f->ei_format ? SDV_FA16_EI_FORMAT : 0));
}

/* a deque<PCM16X0SubLine> with room on both sides */
enum { DQ_CAP = 4096, DQ_HEAD = 1024 };
typedef struct { p16_sub *buf; int lo, hi; } p16_deque;
static void dq_clear(p16_deque *d) { d->lo = d->hi = DQ_HEAD; }
static size_t dq_size(const p16_deque *d) { return (size_t)(d->hi - d->lo); }
static void dq_push_back(p16_deque *d, const p16_sub *s) { if (d->hi < DQ_CAP) d->buf[d->hi++] = *s; }
static void dq_pop_back(p16_deque *d) { if (d->hi > d->lo) d->hi--; }
static void dq_push_front(p16_deque *d, const p16_sub *s) { if (d->lo > 0) d->buf[--d->lo] = *s; }
static p16_sub *dq_at(p16_deque *d, size_t i) { return &d->buf[d->lo + (int)i]; }

/* circarray<T, 65> as the stitcher uses it: fill(), push(), and order-free counting over [0..64] (circbuffer.h:32-175) */
typedef struct { uint16_t v[STATS_DEPTH]; int head, tail; bool full; } p16_ring;
static void ring_fill(p16_ring *r, uint16_t x) { for (int i = 0; i < STATS_DEPTH; i++) r->v[i] = x; r->head = r->tail = 0; r->full = true; }
static void ring_push(p16_ring *r, uint16_t x)
{
    r->v[r->head] = x;
    if (r->full) { r->tail = (r->tail + 1) % STATS_DEPTH; r->head = r->tail; }
    else { r->head = (r->head + 1) % STATS_DEPTH; r->full = r->head == r->tail; }
}

typedef struct {
    sdv_pcm16x0_stitch_settings st;
    bool ignore_crc;
    p16_frasm f1;
    p16_sub *trim; uint16_t trim_fill;
    p16_sub odd[SUBLINES_PF], even[SUBLINES_PF];
    p16_deque pq;                       /* padding_queue */
    p16_sub *conv; size_t conv_lo, conv_hi, conv_cap;     /* conv_queue */
    p16_block padding_block; p16_di pad_checker;
    p16_ring stats_emph, stats_code, stats_srate, stats_padding;
    uint16_t f1_srate; bool f1_emph, f1_code, file_start, file_end;
    sdv_sample_pair *out; size_t out_n, out_cap; sdv_frame_asm_pcm16x0 *frames; size_t frames_n, frames_cap;
    sdv_pcm16x0_bin_rec *vl; size_t vl_n, vl_cap;           /* ... and its assembled sub-lines (newLineProcessed) */
    sdv_pcm16x0_block_rec *vb; size_t vb_n, vb_cap;         /* the visualiser's feed (newBlockProcessed), when asked for */
} p16_stitcher;
enum { EMPH_UNKNOWN, EMPH_OFF, EMPH_ON, CONTENT_UNKNOWN = 0, CONTENT_AUDIO, CONTENT_CODE };

static void conv_push(p16_stitcher *s, const p16_sub *l)
{
    if (s->conv_hi == s->conv_cap) {
        if (s->conv_lo > 0) { memmove(s->conv, s->conv + s->conv_lo, (s->conv_hi - s->conv_lo) * sizeof(p16_sub)); s->conv_hi -= s->conv_lo; s->conv_lo = 0; }
        if (s->conv_hi == s->conv_cap) { s->conv_cap = s->conv_cap ? s->conv_cap * 2 : 4096; s->conv = (p16_sub *)realloc(s->conv, s->conv_cap * sizeof(p16_sub)); }
    }
    s->conv[s->conv_hi++] = *l;
}
static size_t conv_size(const p16_stitcher *s) { return s->conv_hi - s->conv_lo; }
static void out_pair(p16_stitcher *s, const sdv_sample_pair *p) { if (s->out_n < s->out_cap) s->out[s->out_n] = *p; s->out_n++; }
static void out_frasm(p16_stitcher *s, const p16_frasm *f) { if (s->frames_n < s->frames_cap) frasm_to_pod(f, &s->frames[s->frames_n]); s->frames_n++; }
static void out_service(p16_stitcher *s, uint8_t srv)   /* outputFileStart :4931-4970 / outputFileStop :5120-5162 */
{
    p16_frasm d; frasm_clear(&d); d.service_type = srv; out_frasm(s, &d);
    sdv_sample_pair p; memset(&p, 0, sizeof(p)); p.sample_rate = 44056; p.service_type = srv; out_pair(s, &p);
}
static void update_pad_stats(p16_stitcher *s, uint8_t pad) { ring_push(&s->stats_padding, pad); }   /* :4355-4365, always valid */
static uint8_t probable_padding(const p16_stitcher *s)   /* getProbablePadding :4368-4423 */
{
    uint8_t cnt[MAX_PAD_EI]; memset(cnt, 0, sizeof(cnt));
    int n = 0;
    for (int i = 0; i < STATS_DEPTH; i++) if (s->stats_padding.v[i] != INVALID_PAD) { if (s->stats_padding.v[i] < MAX_PAD_EI) cnt[s->stats_padding.v[i]]++; n++; }
    uint8_t max_idx = INVALID_PAD, max_cnt = 0;
    if (n > 0) for (int i = 0; i < MAX_PAD_EI; i++) if (cnt[i] > max_cnt) { max_cnt = cnt[i]; max_idx = (uint8_t)i; }
    return max_idx;
}
static void reset_state(p16_stitcher *s)   /* resetState :64-85 */
{
    ring_fill(&s->stats_emph, EMPH_UNKNOWN); ring_fill(&s->stats_code, CONTENT_UNKNOWN); ring_fill(&s->stats_srate, 0);
    ring_fill(&s->stats_padding, INVALID_PAD);
    dq_clear(&s->pq); s->conv_lo = s->conv_hi = 0;
    s->f1_srate = 44056; s->f1_emph = false; s->f1_code = false;
    frasm_clear_misc(&s->f1);
}

/* findFrameTrim :213-563 */
static bool three_from_left(const p16_stitcher *s, uint16_t i) { return (i + 3) <= s->trim_fill && s->trim[i].line_part == 0; }
static void find_frame_trim(p16_stitcher *s)
{
    p16_frasm *f = &s->f1; const p16_sub *t = s->trim;
    uint16_t i, o_good = 0, e_good = 0;
    bool e_top = false, o_top = false, o_skip = false, e_skip = false;
    s->file_start = s->file_end = false;
    f->even_top_data = f->even_bottom_data = f->odd_top_data = f->odd_bottom_data = 0;
    i = 0;
    while (i < s->trim_fill) {
        bool has_valid = false;
        if (t[i].frame_number == f->frame_number) {
            if (!sub_service(&t[i])) {
                if (three_from_left(s, i)) for (int k = 0; k < 3; k++) has_valid = has_valid || sub_crc(&t[i + k]);
                if (has_valid) {
                    if ((t[i].line_number % 2) == 0) { e_good = (uint16_t)(e_good + 3); if (e_good > MIN_GOOD_SUB) e_skip = true; }
                    else { o_good = (uint16_t)(o_good + 3); if (o_good > MIN_GOOD_SUB) o_skip = true; }
                }
            } else if (t[i].service_type == SDV_SRV_NEW_FILE) s->file_start = true;
            else if (t[i].service_type == SDV_SRV_END_FILE) s->file_end = true;
        }
        i = (uint16_t)(i + (has_valid ? 3 : 1));
    }
    bool subline_skip = false;
    i = 0;
    while (i < s->trim_fill) {
        if (sub_service(&t[i]) && t[i].service_type != SDV_SRV_FILLER) { i++; continue; }
        bool has_valid = false;
        if (t[i].frame_number == f->frame_number) {
            bool even = (t[i].line_number % 2) == 0;
            bool skip = even ? e_skip : o_skip;
            bool *top = even ? &e_top : &o_top;
            bool avail = three_from_left(s, i);
            if (avail) for (int k = 0; k < 3; k++) has_valid = has_valid || (skip ? sub_crc_if(&t[i + k]) : t[i + k].bw_set);
            if (!*top) {
                if (has_valid) { if (even) f->even_top_data = t[i].line_number; else f->odd_top_data = t[i].line_number; subline_skip = *top = true; }
            } else {
                if (!avail) subline_skip = false;
                if (has_valid) { if (even) f->even_bottom_data = t[i].line_number; else f->odd_bottom_data = t[i].line_number; }
            }
        }
        i = (uint16_t)(i + (subline_skip ? 3 : 1));
    }
}

/* splitFrameToFields :566-750 */
static void split_frame_to_fields(p16_stitcher *s)
{
    p16_frasm *f = &s->f1; p16_sub *t = s->trim;
    uint32_t ref_o = 0, ref_e = 0, ref_ob = 0, ref_eb = 0;
    for (uint16_t i = 0; i < s->trim_fill; i++) {
        uint16_t ln = t[i].line_number;
        if (sub_service(&t[i]) && t[i].service_type != SDV_SRV_FILLER) continue;
        if (t[i].frame_number != f->frame_number) continue;
        if ((ln % 2) == 0) {
            if (((f->even_top_data != f->even_bottom_data) || (f->even_top_data != 0)) && ln >= f->even_top_data && ln <= f->even_bottom_data && f->even_data_lines < SUBLINES_PF) {
                t[i].queue_order = f->even_data_lines;
                s->even[f->even_data_lines] = t[i]; f->even_data_lines++;
                ref_eb += t[i].ref_level;
                if (sub_crc(&t[i])) { f->even_valid_lines++; ref_e += t[i].ref_level; }
            }
        } else if (ln >= f->odd_top_data && ln <= f->odd_bottom_data && f->odd_data_lines < SUBLINES_PF) {
            t[i].queue_order = f->odd_data_lines;
            s->odd[f->odd_data_lines] = t[i]; f->odd_data_lines++;
            ref_ob += t[i].ref_level;
            if (sub_crc(&t[i])) { f->odd_valid_lines++; ref_o += t[i].ref_level; }
        }
    }
    f->odd_ref = f->odd_valid_lines > 0 ? (uint8_t)(ref_o / f->odd_valid_lines) : (f->odd_data_lines > 0 ? (uint8_t)(ref_ob / f->odd_data_lines) : 0);
    f->even_ref = f->even_valid_lines > 0 ? (uint8_t)(ref_e / f->even_valid_lines) : (f->even_data_lines > 0 ? (uint8_t)(ref_eb / f->even_data_lines) : 0);
}

/* prescanForFalsePosCRCs :753-833 */
static void prescan_false_pos(p16_sub *field, uint16_t f_size)
{
    int part_no = 0; uint16_t i0 = 0, i1 = 0;
    for (uint16_t i = 0; i < f_size; i++) {
        if (part_no == 0) { i0 = i; part_no = 1; }
        else if (part_no == 1) { i1 = i; part_no = 2; }
        else {
            part_no = 0;
            p16_sub *p0 = &field[i0], *p1 = &field[i1], *p2 = &field[i];
            if (p0->frame_number == p1->frame_number && p1->frame_number == p2->frame_number && p0->line_number == p1->line_number && p1->line_number == p2->line_number) {
                bool c0 = sub_crc(p0), c1 = sub_crc(p1), c2 = sub_crc(p2);
                if ((c0 && !c1 && !c2 && p0->picked_left != 0) || (!c0 && !c1 && c2 && p2->picked_right != 0)) p0->forced_bad = p1->forced_bad = p2->forced_bad = true;
            } else break;
        }
    }
}

/* cutFieldTop :836-865 */
static void cut_field_top(p16_sub *field, uint16_t *f_size, uint16_t cut_cnt)
{
    cut_cnt = (uint16_t)(cut_cnt * 3);
    if (cut_cnt > 0) {
        size_t lim = (size_t)((int)(*f_size) - (int)cut_cnt);
        for (size_t i = 0; i < lim; i++) {
            if ((i + cut_cnt) >= SUBLINES_PF) break;
            if ((i + cut_cnt) < (*f_size)) field[i] = field[i + cut_cnt]; else sub_clear(&field[i]);
        }
        *f_size = (uint16_t)(*f_size - cut_cnt);
    }
}

/* findZeroControlBitOffset :868-1055 */
static int16_t find_zero_ctrl(const p16_sub *field, uint16_t f_size, bool from_top)
{
    uint8_t cnt_stat[64]; int16_t ofs_stat[64]; int n_stat = 0;
    uint8_t zero_cnt = 0, run_cnt = 0;
    int16_t start;
    if (!from_top) {
        start = (int16_t)f_size; start++;
        while (start >= 3) {
            start = (int16_t)(start - 3);
            for (int iblk = 0; iblk < IBLK_PF; iblk++) {
                int16_t so = (int16_t)(start - iblk * SI_TRUE);
                if (so < 0) break;
                if (field[so].line_part != 1) { zero_cnt = 0; break; }
                if (sub_crc(&field[so]) && !field[so].control_bit) zero_cnt++;
            }
            if (n_stat < 64) { ofs_stat[n_stat] = (int16_t)(start - 1); cnt_stat[n_stat] = zero_cnt; n_stat++; }
            zero_cnt = 0; run_cnt++;
            if (run_cnt > (SI_OFS * 3 / 2)) break;
        }
    } else {
        start = 0; start++;
        while (start < ((int)f_size - 3)) {
            start = (int16_t)(start + 3);
            for (int iblk = 0; iblk < IBLK_PF; iblk++) {
                int16_t so = (int16_t)(start + iblk * SI_TRUE);
                if (so >= (int)f_size) break;
                if (field[so].line_part != 1) { zero_cnt = 0; break; }
                if (sub_crc(&field[so]) && !field[so].control_bit) zero_cnt++;
            }
            if (n_stat < 64) { ofs_stat[n_stat] = (int16_t)(start - 1); cnt_stat[n_stat] = zero_cnt; n_stat++; }
            zero_cnt = 0; run_cnt++;
            if (run_cnt > (SI_OFS * 3 / 2)) break;
        }
    }
    zero_cnt = 0; start = 0;
    for (int i = 0; i < n_stat; i++) if (cnt_stat[i] > zero_cnt) { zero_cnt = cnt_stat[i]; start = ofs_stat[i]; }
    return zero_cnt > 0 ? start : (int16_t)-1;
}

/* estimateBlockNumber :1058-1126 */
static uint8_t estimate_block_number(const p16_sub *field, uint16_t f_size, int16_t zero_ofs)
{
    uint8_t out = IBLK_PF - 1;
    if (zero_ofs < (int)f_size) {
        if (zero_ofs < 0) out = 0;
        else {
            uint16_t ln = field[zero_ofs].line_number;
            for (int k = 0; k <= 5; k++) if (ln < IBLK_DELIM + k * (2 * SI_OFS)) { out = (uint8_t)k; break; }
        }
    }
    return out;
}

/* the burst bookkeeping shared by trySIPadding (:1173-1388) and tryEIPadding (:2426-2564) */
typedef struct { uint16_t valid_c, sil_c, unch_c, brk_c, valid_m, sil_m, unch_m, brk_m; } p16_bursts;
static void bursts_block(p16_bursts *u, const p16_block *b, uint16_t max_sil, uint16_t max_unch)
{
    if (blk_valid(b, 3) && !blk_silent(b) && blk_can_force_check(b)) u->valid_c++;
    else if (u->valid_c > u->valid_m) u->valid_m = u->valid_c;
    if (blk_silent(b)) { u->sil_c++; if (u->sil_c >= max_sil) u->valid_c = 0; }
    else { if (u->sil_c > u->sil_m) u->sil_m = u->sil_c; u->sil_c = 0; }
    if (!blk_can_force_check(b) || blk_fixed_by_p(b, 3)) { u->unch_c++; if (u->unch_c > max_unch) u->valid_c = 0; }
    else { if (u->unch_c > u->unch_m) u->unch_m = u->unch_c; u->unch_c = 0; }
    if (blk_broken(b, 3)) { u->brk_c++; if (u->brk_c >= MAX_BROKEN) u->valid_c = 0; }
    else { if (u->brk_c > u->brk_m) u->brk_m = u->brk_c; u->brk_c = 0; }
}
static void bursts_end(p16_bursts *u)
{
    if (u->valid_c > u->valid_m) u->valid_m = u->valid_c;
    if (u->sil_c > u->sil_m) u->sil_m = u->sil_c;
    if (u->unch_c > u->unch_m) u->unch_m = u->unch_c;
    if (u->brk_c > u->brk_m) u->brk_m = u->brk_c;
}

/* trySIPadding :1129-1553 over padding_queue */
static uint8_t try_si_padding(p16_stitcher *s, uint8_t padding, p16_stats *stats)
{
    p16_stats ib[IBLK_PF];
    for (int i = 0; i < IBLK_PF; i++) stats_clear(&ib[i]);
    for (int iblk = 0; iblk < IBLK_PF; iblk++) {
        p16_bursts u; memset(&u, 0, sizeof(u));
        bool run_lock = false, even_block = false;
        for (uint16_t li = 0; li < SI_OFS; li++) {
            uint16_t start = (uint16_t)(li + iblk * SI_TRUE);
            if (di_process_block(&s->pad_checker, dq_at(&s->pq, 0), dq_size(&s->pq), start, even_block, &s->padding_block) != DI_OK) break;
            run_lock = true;
            s->padding_block.queue_order = li;
            bursts_block(&u, &s->padding_block, MAX_SIL_SI, MAX_UNCH_SI);
            even_block = !even_block;
        }
        bursts_end(&u);
        if (run_lock) { ib[iblk].index = (uint16_t)iblk; ib[iblk].valid = u.valid_m; ib[iblk].silent = u.sil_m; ib[iblk].unchecked = u.unch_m; ib[iblk].broken = u.brk_m; }
    }
    /* the first and the last interleave block are left out (a deque of 7: front().index == 0 also when it did not run) */
    int lo = 0, hi = IBLK_PF;
    if (ib[lo].index == 0) lo++;
    if (ib[hi - 1].index == 6) hi--;
    uint16_t top_broken = 0;
    for (int i = lo; i < hi; i++) if (ib[i].broken > top_broken) top_broken = ib[i].broken;
    for (int i = lo; i < hi; i++) ib[i].broken = top_broken;
    p16_stats m = stats_min(&ib[lo], hi - lo);
    if (stats) { stats->index = padding; stats->valid = m.valid; stats->silent = m.silent; stats->unchecked = m.unchecked; stats->broken = m.broken; }
    if (m.unchecked > MAX_UNCH_SI) return DS_NO_PAD;
    if (m.valid == 0) return DS_NO_PAD;
    if (m.silent > MAX_SIL_SI) return DS_SILENCE;
    if (m.broken >= MAX_BROKEN) return DS_BROKE;
    return DS_OK;
}

static void pq_fill_from_field(p16_stitcher *s, const p16_sub *field, uint16_t count, uint16_t *pad_bottom)   /* :1613-1646 / :1757-1786 */
{
    dq_clear(&s->pq);
    for (uint16_t l = 0; l < count; l++) dq_push_back(&s->pq, &field[l]);
    p16_sub e; sub_clear(&e);
    e.frame_number = field[0].frame_number; e.line_number = field[count - 1].line_number; e.queue_order = field[count - 1].queue_order; e.line_part = 0;
    while (dq_size(&s->pq) < SUBLINES_PF) {
        if (e.line_part == 0) e.line_number = (uint16_t)(e.line_number + 2);
        e.queue_order++;
        dq_push_back(&s->pq, &e);
        e.line_part++;
        if (e.line_part >= 3) e.line_part = 0;
        (*pad_bottom)++;
    }
}
static void pq_shift_line(p16_stitcher *s, p16_sub *e)   /* :1711-1719 / :1825-1833 */
{
    for (uint8_t i = 3; i > 0; i--) { dq_pop_back(&s->pq); e->line_part = (uint8_t)(i - 1); dq_push_front(&s->pq, e); }
}

/* findSIPadding :1557-2243 */
static uint8_t find_si_padding(p16_stitcher *s, p16_sub *field, uint16_t *f_size, uint16_t *top_padding, uint16_t *bottom_padding)
{
    uint8_t res = DS_NO_PAD, iblk_num, pad;
    uint16_t count, pad_top = 0, pad_bottom = 0, min_broken;
    int16_t zero_ofs, last_ofs;
    bool lock = false;
    *bottom_padding = 0;
    *top_padding = (uint16_t)((SUBLINES_PF - *f_size) / 3);
    if (*f_size < MIN_FILL_SI) return DS_NO_DATA;
    count = *f_size;
    pq_fill_from_field(s, field, count, &pad_bottom);
    p16_sub e; sub_clear(&e);
    e.frame_number = field[0].frame_number;       /* line number and queue order stay 0 */
    zero_ofs = find_zero_ctrl(field, count, true);
    if (zero_ofs >= 0 && (zero_ofs + 3 + 1) < (int)count)
        if (sub_crc(&field[zero_ofs + 4]) && !field[zero_ofs + 4].control_bit) zero_ofs = (int16_t)(zero_ofs + 3);
    iblk_num = estimate_block_number(field, count, zero_ofs);
    if (s->st.p_correction) {
        s->pad_checker.force_ecc_check = true; s->pad_checker.en_p_code = true; s->pad_checker.ei_format = false;
        pad = probable_padding(s);
        if (pad != INVALID_PAD) {
            for (uint8_t ins = 0; ins < pad; ins++) pq_shift_line(s, &e);
            if (try_si_padding(s, pad, NULL) == DS_OK) {
                lock = true;
                update_pad_stats(s, pad);
                pad_top = pad;
                pad_bottom = pad_bottom >= pad_top ? (uint16_t)(pad_bottom - pad_top) : 0;
                res = DS_OK;
            } else {
                pad_bottom = 0;
                uint32_t fn = e.frame_number;
                pq_fill_from_field(s, field, count, &pad_bottom);
                sub_clear(&e); e.frame_number = fn;
            }
        }
        if (!lock) {
            p16_stats sd[MAX_PAD_SI], mb[MAX_PAD_SI]; int n_mb = 0;
            for (int i = 0; i < MAX_PAD_SI; i++) stats_clear(&sd[i]);
            for (pad = 0; pad < MAX_PAD_SI; pad++) { try_si_padding(s, pad, &sd[pad]); pq_shift_line(s, &e); }
            min_broken = sd[0].broken;
            for (pad = 0; pad < MAX_PAD_SI; pad++) if (sd[pad].broken < min_broken) min_broken = sd[pad].broken;
            for (pad = 0; pad < MAX_PAD_SI; pad++) if (sd[pad].broken == min_broken && sd[pad].valid > 0) mb[n_mb++] = sd[pad];
            if (n_mb > 0) {
                p16_stats m = stats_min(mb, n_mb);
                if (m.unchecked <= MAX_UNCH_SI) {
                    if (m.silent < MAX_SIL_SI) {
                        if (min_broken == 0) res = m.valid > MIN_VALID_SI ? DS_OK : DS_NO_PAD;
                        else res = DS_BROKE;
                        lock = true;
                        pad_top = m.index;
                        pad_bottom = pad_bottom >= pad_top ? (uint16_t)(pad_bottom - pad_top) : 0;
                        update_pad_stats(s, (uint8_t)pad_top);
                    } else res = DS_SILENCE;
                }
            }
        }
    }
    dq_clear(&s->pq);
    if (lock) {
        last_ofs = (int16_t)(iblk_num * SI_OFS);
        if (last_ofs < (int)pad_top) {
            last_ofs = (int16_t)((iblk_num + 1) * SI_OFS);
            last_ofs = (int16_t)(last_ofs - pad_top);
            pad_top = 0;
            cut_field_top(field, f_size, (uint16_t)last_ofs);
            count = *f_size;
        } else if (last_ofs > (int)pad_top) {
            last_ofs = (int16_t)((iblk_num - 1) * SI_OFS);
            pad_top = (uint16_t)(pad_top + last_ofs);
        }
        pad_top = (uint16_t)(pad_top * 3);
        pad_bottom = (uint16_t)(SUBLINES_PF - pad_top);
        if (pad_bottom >= count) pad_bottom = (uint16_t)(pad_bottom - count);
        else { pad_bottom = (uint16_t)(count - pad_bottom); count = (uint16_t)(count - pad_bottom); pad_bottom = 0; }
    } else if (zero_ofs >= 0) {
        pad_top = pad_bottom = 0;
        last_ofs = (int16_t)(3 + iblk_num * SI_TRUE);
        last_ofs = (int16_t)(last_ofs - zero_ofs);
        if (last_ofs > 0) pad_top = (uint16_t)last_ofs;
        else if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); cut_field_top(field, f_size, (uint16_t)(last_ofs / 3)); count = *f_size; }
        last_ofs = (int16_t)pad_top;
        last_ofs = (int16_t)(last_ofs + count);
        last_ofs = (int16_t)(SUBLINES_PF - last_ofs);
        if (last_ofs > 0) pad_bottom = (uint16_t)last_ofs;
        else if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); count = (uint16_t)(count - last_ofs); }
    } else { pad_bottom = 0; pad_top = (uint16_t)(SUBLINES_PF - count); }
    *top_padding = (uint16_t)(pad_top / 3);
    *bottom_padding = (uint16_t)(pad_bottom / 3);
    *f_size = count;
    return res;
}

/* findSIDataAlignment :2246-2377 */
static void find_si_alignment(p16_stitcher *s)
{
    p16_frasm *f = &s->f1;
    uint16_t top = 0, bottom = 0;
    f->order_preset = true; f->order_guessed = false; f->field_order = s->st.field_order == ORDER_BFF ? ORDER_BFF : ORDER_TFF;
    uint8_t odd_res = find_si_padding(s, s->odd, &f->odd_data_lines, &top, &bottom);
    if (odd_res == DS_OK) { f->padding_ok = true; f->silence = false; }
    else { f->padding_ok = false; f->silence = odd_res == DS_SILENCE; }
    f->odd_top_padding = top; f->odd_bottom_padding = bottom;
    uint8_t even_res = find_si_padding(s, s->even, &f->even_data_lines, &top, &bottom);
    if (even_res == DS_OK) { /* padding_ok and silence keep the odd field's verdict */ }
    else { f->padding_ok = false; if (odd_res == DS_SILENCE) f->silence = true; }
    f->even_top_padding = top; f->even_bottom_padding = bottom;
}

/* tryEIPadding :2380-2646 over padding_queue */
static uint8_t try_ei_padding(p16_stitcher *s, uint16_t padding, p16_stats *stats)
{
    if (dq_size(&s->pq) < EI_TRUE) return DS_NO_DATA;
    size_t n = 0; bool even_block = false, run_lock = false;
    p16_bursts u; memset(&u, 0, sizeof(u));
    while (((size_t)(LINES_PF * 2 * 2) + n + 1) < dq_size(&s->pq)) {
        if (di_process_block(&s->pad_checker, dq_at(&s->pq, 0), dq_size(&s->pq), (uint16_t)n, even_block, &s->padding_block) != DI_OK) break;
        run_lock = true;
        bursts_block(&u, &s->padding_block, MAX_SIL_EI, MAX_UNCH_EI);
        n++;
        even_block = !even_block;
    }
    bursts_end(&u);
    if (stats && run_lock) { stats->index = padding; stats->valid = u.valid_m; stats->silent = u.sil_m; stats->unchecked = u.unch_m; stats->broken = u.brk_m; }
    if (u.unch_m > MAX_UNCH_EI) return DS_NO_PAD;
    if (u.valid_m == 0) return DS_NO_PAD;
    if (u.sil_m > MAX_SIL_EI) return DS_SILENCE;
    if (u.brk_m >= MAX_BROKEN) return DS_BROKE;
    return DS_OK;
}

/* findEIPadding :2649-2994 */
static uint8_t find_ei_padding(p16_stitcher *s, uint8_t field_order)
{
    p16_frasm *f = &s->f1;
    uint8_t res = DS_NO_PAD, field_padding = 0;
    bool lock = false;
    const p16_sub *field1, *field2; uint16_t f1_size, f2_size;
    if (field_order == ORDER_TFF) { field1 = s->odd; field2 = s->even; f1_size = f->odd_data_lines; f2_size = f->even_data_lines; }
    else { field1 = s->even; field2 = s->odd; f1_size = f->even_data_lines; f2_size = f->odd_data_lines; }
    f->odd_bottom_padding = 0; f->even_bottom_padding = 0;
    f->odd_top_padding = (uint16_t)((SUBLINES_PF - f->odd_data_lines) / 3 - f->odd_bottom_padding);
    f->even_top_padding = (uint16_t)((SUBLINES_PF - f->even_data_lines) / 3 - f->even_bottom_padding);
    if (s->st.p_correction) {
        dq_clear(&s->pq);
        for (uint16_t i = 0; i < f1_size; i++) dq_push_back(&s->pq, &field1[i]);
        p16_sub e; sub_clear(&e);
        e.line_number = field1[f1_size - 1].line_number; e.frame_number = field1[f1_size - 1].frame_number;
        size_t last_pad_line = dq_size(&s->pq);
        s->pad_checker.ignore_crc = s->ignore_crc; s->pad_checker.force_ecc_check = true; s->pad_checker.en_p_code = true; s->pad_checker.ei_format = true;
        p16_stats sd[MAX_PAD_EI], mb[MAX_PAD_EI]; int n_mb = 0;
        for (int i = 0; i < MAX_PAD_EI; i++) stats_clear(&sd[i]);
        for (uint16_t pad = 0; pad < MAX_PAD_EI; pad++) {
            for (uint16_t i = 0; i < f2_size; i++) dq_push_back(&s->pq, &field2[i]);
            try_ei_padding(s, pad, &sd[pad]);
            while (dq_size(&s->pq) > last_pad_line) dq_pop_back(&s->pq);
            e.line_number = (uint16_t)(e.line_number + 2);
            for (uint8_t part = 0; part < 3; part++) { e.line_part = part; dq_push_back(&s->pq, &e); }
            last_pad_line = dq_size(&s->pq);
        }
        uint16_t min_broken = sd[0].broken;
        for (int pad = 0; pad < MAX_PAD_EI; pad++) if (sd[pad].broken < min_broken) min_broken = sd[pad].broken;
        for (int pad = 0; pad < MAX_PAD_EI; pad++) if (sd[pad].broken == min_broken && sd[pad].valid > 0) mb[n_mb++] = sd[pad];
        if (n_mb > 0) {
            p16_stats m = stats_min(mb, n_mb);
            if (m.unchecked <= MAX_UNCH_EI) {
                if (m.silent < MAX_SIL_EI) {
                    if (min_broken == 0) res = m.valid > MIN_VALID_EI ? DS_OK : DS_NO_PAD;
                    else res = DS_BROKE;
                    lock = true;
                    field_padding = (uint8_t)m.index;
                    update_pad_stats(s, field_padding);
                } else res = DS_SILENCE;
            }
        }
    }
    dq_clear(&s->pq);
    if (lock) {
        if (field_order == ORDER_TFF) { f->odd_bottom_padding = field_padding; f->even_top_padding = 0; }
        else { f->even_bottom_padding = field_padding; f->odd_top_padding = 0; }
    }
    return res;
}

/* conditionEIFramePadding :2997-3464 */
static void condition_ei_frame_padding(p16_sub *field1, p16_sub *field2, uint16_t *f1_size, uint16_t *f2_size,
                                       uint16_t *f1_top, uint16_t *f1_bottom, uint16_t *f2_top, uint16_t *f2_bottom)
{
    uint8_t iblk_num;
    uint16_t inter = *f1_bottom;
    int16_t zero_ofs, last_ofs;
    bool pos_lock = false;
    zero_ofs = find_zero_ctrl(field2, *f2_size, false);
    if (zero_ofs >= 0) {
        pos_lock = true;
        iblk_num = estimate_block_number(field2, *f2_size, zero_ofs);
        zero_ofs = (int16_t)(*f2_size - zero_ofs);
        last_ofs = (int16_t)((SI_OFS - 2) * 3 - zero_ofs);
        if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); *f2_size = (uint16_t)(*f2_size - last_ofs); }
        else if (last_ofs > 0) *f2_bottom = (uint16_t)(*f2_bottom + last_ofs / 3);
        last_ofs = (int16_t)((IBLK_PF - iblk_num - 1) * SI_TRUE);
        *f2_bottom = (uint16_t)(*f2_bottom + last_ofs / 3);
        last_ofs = (int16_t)(LINES_PF - (*f2_size) / 3);
        last_ofs = (int16_t)(last_ofs - *f2_bottom);
        if (last_ofs < 0) {
            last_ofs = (int16_t)(0 - last_ofs);
            zero_ofs = (int16_t)((last_ofs / SI_OFS) + 1);
            zero_ofs = (int16_t)(zero_ofs * SI_OFS);
            last_ofs = (int16_t)(*f2_bottom - zero_ofs);
            if (last_ofs < 0) { *f2_top = *f2_bottom = 0; pos_lock = false; }
            else {
                *f2_bottom = (uint16_t)last_ofs;
                last_ofs = (int16_t)(LINES_PF - (*f2_size) / 3);
                last_ofs = (int16_t)(last_ofs - *f2_bottom);
            }
        }
        if (last_ofs > (int)inter) {
            if ((last_ofs - (int)inter) < 2) { *f2_top = inter; *f2_bottom = (uint16_t)(*f2_bottom + (last_ofs - inter)); }
            else { *f2_top = *f2_bottom = 0; pos_lock = false; }
        } else if (pos_lock) *f2_top = (uint16_t)last_ofs;
    }
    if (pos_lock) {
        zero_ofs = (int16_t)(inter - *f2_top);
        *f1_bottom = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)((*f1_size + *f2_size) / 3);
        zero_ofs = (int16_t)(zero_ofs + *f1_bottom + *f2_top);
        zero_ofs = (int16_t)(zero_ofs + *f2_bottom);
        zero_ofs = (int16_t)((2 * LINES_PF) - zero_ofs);
        if (zero_ofs < 0) { *f1_top = *f1_bottom = *f2_top = *f2_bottom = 0; pos_lock = false; }
        else *f1_top = (uint16_t)zero_ofs;
    }
    if (!pos_lock) {
        zero_ofs = find_zero_ctrl(field1, *f1_size, false);
        if (zero_ofs >= 0) {
            pos_lock = true;
            uint8_t iblk_cnt = (uint8_t)(zero_ofs / SI_TRUE);
            zero_ofs = (int16_t)(zero_ofs - iblk_cnt * SI_TRUE);
            zero_ofs = (int16_t)((SUBLINES_PF + 2 * 3) - zero_ofs);
            zero_ofs = (int16_t)(zero_ofs / 3);
            *f1_top = (uint16_t)zero_ofs;
            zero_ofs = (int16_t)(LINES_PF - *f1_top);
            zero_ofs = (int16_t)(zero_ofs - (*f1_size) / 3);
            if (zero_ofs < 0) pos_lock = false;
            else {
                *f1_bottom = (uint16_t)zero_ofs;
                zero_ofs = (int16_t)(inter - *f1_bottom);
                if (zero_ofs < 0) pos_lock = false;
                else {
                    *f2_top = (uint16_t)zero_ofs;
                    zero_ofs = (int16_t)((*f2_size) / 3);
                    zero_ofs = (int16_t)(zero_ofs + *f2_top);
                    zero_ofs = (int16_t)(LINES_PF - zero_ofs);
                    if (zero_ofs < 0) { *f2_bottom = 0; *f2_size = (uint16_t)(*f2_size - (0 - zero_ofs) * 3); }
                    else *f2_bottom = (uint16_t)zero_ofs;
                }
            }
        }
    }
    if (!pos_lock) {
        zero_ofs = (int16_t)(inter / 2);
        *f2_top = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)(inter * 3);
        zero_ofs = (int16_t)(zero_ofs - (*f2_top) * 3);
        *f1_bottom = (uint16_t)(zero_ofs / 3);
        zero_ofs = (int16_t)((*f1_size) / 3);
        zero_ofs = (int16_t)(zero_ofs + *f1_bottom);
        zero_ofs = (int16_t)(LINES_PF - zero_ofs);
        if (zero_ofs < 0) {
            *f1_top = 0;
            zero_ofs = (int16_t)((*f1_size) / 3);
            zero_ofs = (int16_t)(LINES_PF - zero_ofs);
            *f1_bottom = (uint16_t)zero_ofs;
            zero_ofs = (int16_t)(inter - *f1_bottom);
            *f2_top = (uint16_t)zero_ofs;
        } else *f1_top = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)((*f2_size) / 3);
        zero_ofs = (int16_t)(zero_ofs + *f2_top);
        zero_ofs = (int16_t)(LINES_PF - zero_ofs);
        if (zero_ofs < 0) { *f2_bottom = 0; *f2_size = (uint16_t)(*f2_size - (0 - zero_ofs) * 3); }
        else *f2_bottom = (uint16_t)zero_ofs;
    }
}

/* findEIDataAlignment :3467-3585 */
static uint8_t find_ei_alignment(p16_sub *field, uint16_t *f_size, uint16_t *top_pad, uint16_t *bottom_pad)
{
    int16_t zero_ofs = find_zero_ctrl(field, *f_size, false), last_ofs;
    if (zero_ofs < 0) return DS_NO_PAD;
    *top_pad = *bottom_pad = 0;
    uint8_t iblk_num = estimate_block_number(field, *f_size, zero_ofs);
    zero_ofs = (int16_t)(*f_size - zero_ofs);
    last_ofs = (int16_t)((SI_OFS - 2) * 3 - zero_ofs);
    if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); *f_size = (uint16_t)(*f_size - last_ofs); }
    else if (last_ofs > 0) *bottom_pad = (uint16_t)(*bottom_pad + last_ofs / 3);
    last_ofs = (int16_t)((IBLK_PF - iblk_num - 1) * SI_TRUE);
    *bottom_pad = (uint16_t)(*bottom_pad + last_ofs / 3);
    last_ofs = (int16_t)(LINES_PF - (*f_size) / 3);
    last_ofs = (int16_t)(last_ofs - *bottom_pad);
    if (last_ofs < 0) {
        last_ofs = (int16_t)(0 - last_ofs);
        if (last_ofs < SI_OFS && last_ofs < (int)(*f_size)) { cut_field_top(field, f_size, (uint16_t)last_ofs); return DS_OK; }
        return DS_NO_PAD;
    }
    *top_pad = (uint16_t)(*top_pad + last_ofs);
    return DS_OK;
}

/* findEIFrameStitching :3588-4115 */
enum { STG_TRY_PREVIOUS, STG_TRY_TFF, STG_TRY_BFF, STG_FULL_PREPARE, STG_INTERPAD_TFF, STG_INTERPAD_BFF, STG_ALIGN_TFF, STG_ALIGN_BFF, STG_FB_CTRL_EST,
       STG_PAD_NO_GOOD, STG_PAD_OK, STG_PAD_SILENCE, STG_PAD_MAX };
static uint8_t find_ei_frame_stitching(p16_stitcher *s)
{
    p16_frasm *f = &s->f1;
    f->order_preset = true; f->order_guessed = false; f->field_order = s->st.field_order == ORDER_BFF ? ORDER_BFF : ORDER_TFF;
    bool tff = f->field_order == ORDER_TFF;
    int state = STG_TRY_PREVIOUS, stage_count = 0;
    for (;;) {
        stage_count++;
        if (state == STG_TRY_PREVIOUS) {
            uint8_t r = probable_padding(s);
            if (r != INVALID_PAD) {
                uint8_t inter_pad = r;
                dq_clear(&s->pq);
                uint16_t c1 = tff ? f->odd_data_lines : f->even_data_lines, c2 = tff ? f->even_data_lines : f->odd_data_lines;
                const p16_sub *p1 = tff ? s->odd : s->even, *p2 = tff ? s->even : s->odd;
                for (uint16_t i = 0; i < c1; i++) dq_push_back(&s->pq, &p1[i]);
                p16_sub e; sub_clear(&e);
                if (c1 > 0) { e.line_number = p1[c1 - 1].line_number; e.frame_number = p1[c1 - 1].frame_number; }     /* (an empty first field: the reference reads before its buffer) */
                for (uint8_t k = 0; k < inter_pad; k++) { e.line_number = (uint16_t)(e.line_number + 2); for (uint8_t part = 0; part < 3; part++) { e.line_part = part; dq_push_back(&s->pq, &e); } }
                for (uint16_t i = 0; i < c2; i++) dq_push_back(&s->pq, &p2[i]);
                r = try_ei_padding(s, inter_pad, NULL);
                if (r == DS_OK) {
                    update_pad_stats(s, inter_pad);
                    if (tff) { f->odd_bottom_padding = inter_pad; f->even_top_padding = 0; f->silence = false; state = STG_ALIGN_TFF; }
                    else { f->even_bottom_padding = inter_pad; f->odd_top_padding = 0; f->silence = false; state = STG_ALIGN_BFF; }
                } else state = STG_FULL_PREPARE;
            } else state = STG_FULL_PREPARE;
        } else if (state == STG_FULL_PREPARE) {
            f->odd_top_padding = f->odd_bottom_padding = f->even_top_padding = f->even_bottom_padding = 0;
            if ((f->odd_data_lines < MIN_FILL_EI && f->even_data_lines < MIN_FILL_EI) || (f->odd_data_lines + f->even_data_lines) < (2 * MIN_FILL_EI)) state = STG_FB_CTRL_EST;
            else state = tff ? STG_INTERPAD_TFF : STG_INTERPAD_BFF;
        } else if (state == STG_INTERPAD_TFF || state == STG_INTERPAD_BFF) {
            bool t = state == STG_INTERPAD_TFF;
            if ((t ? f->odd_data_lines : f->even_data_lines) < MIN_FILL_EI) state = STG_FB_CTRL_EST;
            else {
                uint8_t r = find_ei_padding(s, t ? ORDER_TFF : ORDER_BFF);
                f->silence = false; f->padding_ok = false;
                if (r == DS_OK) state = t ? STG_ALIGN_TFF : STG_ALIGN_BFF;
                else { if (r == DS_SILENCE) f->silence = true; if (t) f->odd_bottom_padding = 0; else f->even_bottom_padding = 0; state = STG_FB_CTRL_EST; }
            }
        } else if (state == STG_ALIGN_TFF) {
            condition_ei_frame_padding(s->odd, s->even, &f->odd_data_lines, &f->even_data_lines, &f->odd_top_padding, &f->odd_bottom_padding, &f->even_top_padding, &f->even_bottom_padding);
            f->padding_ok = true; state = STG_PAD_OK;
        } else if (state == STG_ALIGN_BFF) {
            condition_ei_frame_padding(s->even, s->odd, &f->even_data_lines, &f->odd_data_lines, &f->even_top_padding, &f->even_bottom_padding, &f->odd_top_padding, &f->odd_bottom_padding);
            f->padding_ok = true; state = STG_PAD_OK;
        } else if (state == STG_FB_CTRL_EST) {
            state = STG_PAD_OK;
            if (find_ei_alignment(s->odd, &f->odd_data_lines, &f->odd_top_padding, &f->odd_bottom_padding) != DS_OK) {
                f->odd_bottom_padding = 0; f->odd_top_padding = (uint16_t)((SUBLINES_PF - f->odd_data_lines) / 3); state = STG_PAD_NO_GOOD;
            }
            if (find_ei_alignment(s->even, &f->even_data_lines, &f->even_top_padding, &f->even_bottom_padding) != DS_OK) {
                f->even_bottom_padding = 0; f->even_top_padding = (uint16_t)((SUBLINES_PF - f->even_data_lines) / 3); state = STG_PAD_NO_GOOD;
            }
        } else break;
        if (stage_count > STG_PAD_MAX) return DS_NO_PAD;
    }
    return state == STG_PAD_OK ? DS_OK : (state == STG_PAD_SILENCE ? DS_SILENCE : DS_NO_PAD);
}

/* addLinesFromField :4452-4528 / addFieldPadding :4531-4591 */
static uint16_t add_lines(p16_stitcher *s, const p16_sub *field, uint16_t count, uint16_t *q_order, uint16_t *last_line)
{
    if (!(SUBLINES_PF >= (int)count)) return 0;
    for (uint16_t i = 0; i < count; i++) {
        p16_sub c = field[i];
        c.queue_order = *q_order;
        conv_push(s, &c);
        *q_order = (uint16_t)(*q_order + 1);
        if (field[i].line_part == 2) *last_line = (uint16_t)(field[i].line_number + 2);
    }
    return count;
}
static uint16_t add_padding(p16_stitcher *s, uint32_t frame, uint16_t line_cnt, uint16_t *q_order, uint16_t *last_line)
{
    uint16_t n = 0;
    p16_sub e; sub_clear(&e);
    for (uint16_t i = 0; i < line_cnt; i++) {
        e.frame_number = frame; e.line_number = *last_line; *last_line = (uint16_t)(*last_line + 2);
        for (uint8_t part = 0; part < 3; part++) { e.line_part = part; e.queue_order = *q_order; *q_order = (uint16_t)(*q_order + 1); conv_push(s, &e); n++; }
    }
    return n;
}

/* collectCtrlBitStats :4745-4912: the return value is `order even` of the flags block */
static bool collect_ctrl_bits(p16_stitcher *s, bool *emph_o, uint16_t *rate_o, bool *code_o)
{
    uint8_t emph = 0, rate = 0, code = 0, emph_cnt = 0, rate_cnt = 0, code_cnt = 0;
    if (conv_size(s) < SUBLINES_PF) return false;
    const p16_sub *q = s->conv + s->conv_lo; size_t qn = conv_size(s);
    for (int iblk = 0; iblk < IBLK_PF * 2; iblk++) {
        size_t sc = (size_t)iblk * SI_TRUE + 1;         /* start of the interleave block, on its PART_MIDDLE */
        if (sc + BIT_CODE >= qn) break;                 /* (the reference reads on; a frame always leaves 1470 sub-lines here) */
        const p16_sub *le = &q[sc + BIT_EMPH], *lr = &q[sc + BIT_RATE], *lc = &q[sc + BIT_CODE];
        if (sub_crc(le)) { emph_cnt++; if (!le->control_bit) emph++; }
        if (sub_crc(lr)) { rate_cnt++; if (!lr->control_bit) rate++; }
        if (sub_crc(lc)) { code_cnt++; if (!lc->control_bit) code++; }
    }
    *emph_o = emph > emph_cnt / 2;
    *rate_o = rate > rate_cnt / 2 ? 44100 : 44056;
    *code_o = code > code_cnt / 2;
    return emph_cnt >= 2 && rate_cnt >= 2 && code_cnt >= 2;
}

/* fillFrameForOutput :4594-4742 */
static void fill_frame_for_output(p16_stitcher *s)
{
    p16_frasm *f = &s->f1;
    bool tff = f->field_order == ORDER_TFF;
    for (int field = 0; field < 2; field++) {
        bool odd = tff == (field == 0);
        uint16_t added = 0, q_ord = 1, last_line = (uint16_t)(tff == (field == 0) ? 1 : 2);
        added = (uint16_t)(added + add_padding(s, f->frame_number, odd ? f->odd_top_padding : f->even_top_padding, &q_ord, &last_line));
        added = (uint16_t)(added + add_lines(s, odd ? s->odd : s->even, odd ? f->odd_data_lines : f->even_data_lines, &q_ord, &last_line));
        added = (uint16_t)(added + add_padding(s, f->frame_number, odd ? f->odd_bottom_padding : f->even_bottom_padding, &q_ord, &last_line));
        /* the shortfall is counted in sub-lines and handed over as a count of lines */
        if (added < SUBLINES_PF) added = (uint16_t)(added + add_padding(s, f->frame_number, (uint16_t)(SUBLINES_PF - added), &q_ord, &last_line));
    }
    bool emph, code; uint16_t rate;
    bool even_order = collect_ctrl_bits(s, &emph, &rate, &code);
    /* updateCtrlBitStats :4126-4166 */
    if (!even_order) { ring_push(&s->stats_emph, EMPH_UNKNOWN); ring_push(&s->stats_code, CONTENT_UNKNOWN); ring_push(&s->stats_srate, 0); }
    else { ring_push(&s->stats_emph, emph ? EMPH_ON : EMPH_OFF); ring_push(&s->stats_code, code ? CONTENT_CODE : CONTENT_AUDIO); ring_push(&s->stats_srate, rate == 44100 ? 44100 : 44056); }
    if (even_order) { s->f1_srate = rate; s->f1_emph = emph; s->f1_code = code; }
    else {
        int a = 0, b = 0;      /* getProbableSampleRate :4289-4346 */
        for (int i = 0; i < STATS_DEPTH; i++) { if (s->stats_srate.v[i] == 44056) a++; else if (s->stats_srate.v[i] == 44100) b++; }
        s->f1_srate = (a > 0 || b > 0) ? (a < b ? 44100 : 44056) : 44056;
        a = b = 0;             /* getProbableEmphasesBit :4169-4226: the BIT (true = emphasis off) lands in f1_emph */
        for (int i = 0; i < STATS_DEPTH; i++) { if (s->stats_emph.v[i] == EMPH_OFF) a++; else if (s->stats_emph.v[i] == EMPH_ON) b++; }
        s->f1_emph = (a > 0 || b > 0) ? !(a < b) : true;
        a = b = 0;             /* getProbableCodeBit :4229-4286: the BIT (true = audio) lands in f1_code */
        for (int i = 0; i < STATS_DEPTH; i++) { if (s->stats_code.v[i] == CONTENT_CODE) a++; else if (s->stats_code.v[i] == CONTENT_AUDIO) b++; }
        s->f1_code = (a > 0 || b > 0) ? (a < b) : true;
    }
}

/* a PCM16X0DataBlock as the visualiser's feed carries it (include/sdvpcm.h): by line, as the class stores it */
static void blk_to_vis_rec(const p16_block *b, sdv_pcm16x0_block_rec *o)
{
    memset(o, 0, sizeof(*o));
    for (int i = 0; i < 3; i++) {
        for (int l = 0; l < 3; l++) {
            o->words[i][l] = b->words[i][l];
            if (b->word_crc[i][l]) o->word_crc |= (uint16_t)(1u << (3 * i + l));
            if (b->word_valid[i][l]) o->word_valid |= (uint16_t)(1u << (3 * i + l));
        }
        if (b->picked_left[i]) o->picked_left |= (uint8_t)(1u << i);
        if (b->picked_crc[i]) o->picked_crc |= (uint8_t)(1u << i);
        o->audio_state[i] = b->audio_state[i];
    }
    o->flags = (uint8_t)((b->order_even ? SDV_P16B_EVEN_ORDER : 0) | (b->ei_format ? SDV_P16B_EI_FORMAT : 0) | (b->emphasis ? SDV_P16B_EMPHASIS : 0) | (b->code ? SDV_P16B_CODE : 0));
    o->sample_rate = b->sample_rate;
}
/* outputDataBlock :4973-5117 */
static void output_data_block(p16_stitcher *s, const p16_block *b)
{
    if (s->vb) { if (s->vb_n < s->vb_cap) blk_to_vis_rec(b, &s->vb[s->vb_n]); s->vb_n++; }      /* emit newBlockProcessed(*in_block) :5116 (the pairs do not change it) */
    for (int blk = 0; blk < 3; blk++) {
        bool state, lv, rv, lf, rf;
        if (!blk_broken(b, blk)) {
            state = blk_valid(b, 3);
            lf = state && blk_crc_ok(b, blk, W_L); rf = state && blk_crc_ok(b, blk, W_R);
            lv = blk_word_valid(b, blk, W_L); rv = blk_word_valid(b, blk, W_R);
        } else state = lv = rv = lf = rf = false;
        sdv_sample_pair p; memset(&p, 0, sizeof(p));
        p.emphasis = b->emphasis; p.sample_rate = b->sample_rate;
        p.audio_word[0] = blk_sample(b, blk, W_L); p.audio_word[1] = blk_sample(b, blk, W_R);
        p.sample_flags[0] = (uint8_t)((state ? SDV_SF_BLOCK_OK : 0) | (lv ? SDV_SF_WORD_VALID : 0) | (lf ? SDV_SF_WORD_FIXED : 0));
        p.sample_flags[1] = (uint8_t)((state ? SDV_SF_BLOCK_OK : 0) | (rv ? SDV_SF_WORD_VALID : 0) | (rf ? SDV_SF_WORD_FIXED : 0));
        out_pair(s, &p);
    }
}

/* performDeinterleave :5165-5447 */
/* a queued sub-line as a record of the binarizer's type: the object is a copy of what the record was made into, with a new queue_order (:4470) and,
 * maybe, forced bad since (:800-820); a line the stitcher made itself (addFieldPadding :4537-4571) is a cleared PCM16X0SubLine with numbers and part */
static void sub_to_vis_rec(const p16_sub *l, sdv_pcm16x0_bin_rec *o)
{
    if (l->src) {
        *o = *l->src;
        o->queue_order = l->queue_order;
        if (l->forced_bad) o->flags |= SDV_LF_FORCED_BAD;
        o->flags = (uint8_t)((o->flags & ~SDV_LF_CRC_VALID) | ((!sub_service(l) && sub_crc(l)) ? SDV_LF_CRC_VALID : 0));      /* isCRCValid() as it stands now */
        return;
    }
    memset(o, 0, sizeof(*o));
    o->frame_number = l->frame_number; o->line_number = l->line_number;
    for (int i = 0; i < 4; i++) o->words[i] = l->words[i];
    o->calc_crc = l->calc_crc; o->data_start = -32768; o->data_stop = 32767; o->queue_order = l->queue_order;
    o->line_part = l->line_part; o->control_bit = l->control_bit ? 1 : 0; o->service_type = l->service_type;
}
static void vis_end_frame(p16_stitcher *s, uint32_t frame)       /* where MainWindow emits newFrameAssembled (mainwindow.cpp:3956) */
{
    if (!s->vl) return;
    if (s->vl_n < s->vl_cap) {
        p16_sub e; sub_clear(&e);
        e.frame_number = frame; e.calc_crc = 0; e.service_type = SDV_SRV_END_FRAME;
        sub_to_vis_rec(&e, &s->vl[s->vl_n]);
    }
    s->vl_n++;
}
static void perform_deinterleave(p16_stitcher *s, uint8_t format)
{
    if (s->vl)          /* "dump the whole line buffer out (for visualization)" :5196-5213 */
        for (size_t i = s->conv_lo; i < s->conv_hi; i++)
            if (s->conv[i].frame_number == s->f1.frame_number) { if (s->vl_n < s->vl_cap) sub_to_vis_rec(&s->conv[i], &s->vl[s->vl_n]); s->vl_n++; }
    p16_frasm *f = &s->f1;
    p16_di d = { !s->ignore_crc, s->st.p_correction != 0, s->ignore_crc, format == SDV_P16_FORMAT_EI };
    uint16_t frame_lim = d.ei_format ? EI_TRUE * 3 : SI_TRUE, interleave_lim = d.ei_format ? EI_OFS : SI_OFS, valid_cnt = 0;
    uint8_t broken_countdown = 0;
    f->ei_format = d.ei_format;
    while (conv_size(s) >= frame_lim) {
        bool even_order = false; uint16_t line_in_block = 0;
        for (uint16_t i = 0; i < interleave_lim; i++) {
            p16_block b; blk_clear(&b);
            di_process_block(&d, s->conv + s->conv_lo, conv_size(s), i, even_order, &b);
            f->blocks_total = (uint16_t)(f->blocks_total + 3);
            b.queue_order = line_in_block;
            b.sample_rate = (s->st.sample_rate_preset == 44100 || s->st.sample_rate_preset == 44056) ? s->st.sample_rate_preset : s->f1_srate;   /* setBlockSampleRate :4915-4928 */
            b.emphasis = s->f1_emph; b.code = s->f1_code; b.ei_format = d.ei_format;
            f->odd_sample_rate = f->even_sample_rate = b.sample_rate; f->odd_emphasis = f->even_emphasis = b.emphasis; f->ei_format = b.ei_format;
            if (!blk_silent(&b)) {
                if (blk_valid(&b, 3) && !blk_has_picked_word_any(&b, 0)) valid_cnt++;
                if (s->st.mask_seams && !f->padding_ok && !f->silence && valid_cnt < 3) blk_mark_unsafe(&b);
                if (s->st.broke_mask > 0 && blk_broken(&b, 3)) broken_countdown = s->st.broke_mask;
                if (broken_countdown != 0) blk_mark_unsafe(&b);
            }
            for (int k = 0; k < 3; k++) if (!blk_valid(&b, k)) f->blocks_drop++;
            for (int k = 0; k < 3; k++) if (blk_broken(&b, k)) f->blocks_broken++;
            for (int k = 0; k < 3; k++) if (blk_fixed_by_p(&b, k)) f->blocks_fix_p++;
            for (int k = 0; k < 3; k++) if (blk_fixed_by_bp(&b, k)) f->blocks_fix_bp++;
            if (!blk_valid(&b, 3)) f->samples_drop = (uint16_t)(f->samples_drop + blk_errors_fixed_audio(&b, 3));
            if (broken_countdown > 0) broken_countdown--;
            output_data_block(s, &b);
            even_order = !even_order; line_in_block++;
        }
        s->conv_lo += frame_lim;
    }
    if (s->conv_lo == s->conv_hi) s->conv_lo = s->conv_hi = 0;
}

/* one turn of doFrameReassemble (:5695-5834) for the frame recs[lo..hi) (its END_FRAME excluded) */
static void stitch_frame(p16_stitcher *s, const sdv_pcm16x0_bin_rec *recs, size_t lo, size_t hi, uint32_t frame)
{
    p16_frasm *f = &s->f1;
    f->frame_number = frame;
    /* fillUntilFullFrame :141-210 */
    s->trim_fill = 0;
    for (size_t i = lo; i < hi; i++)
        if (recs[i].frame_number == frame && recs[i].service_type != SDV_SRV_END_FRAME && s->trim_fill < BUF_TRIM) sub_from_rec(&recs[i], &s->trim[s->trim_fill++]);
    find_frame_trim(s);
    if (s->file_start) reset_state(s);
    if (!s->file_end) {
        split_frame_to_fields(s);
        prescan_false_pos(s->odd, f->odd_data_lines);
        prescan_false_pos(s->even, f->even_data_lines);
        if (s->st.format != SDV_P16_FORMAT_EI) find_si_alignment(s); else find_ei_frame_stitching(s);
        if (s->file_start) out_service(s, SDV_PAIR_SRV_NEW_FILE);
        fill_frame_for_output(s);
        perform_deinterleave(s, s->st.format);
        f->odd_data_lines /= 3; f->even_data_lines /= 3; f->odd_valid_lines /= 3; f->even_valid_lines /= 3;
        f->odd_std_lines = f->even_std_lines = LINES_PF;
        out_frasm(s, f);
        vis_end_frame(s, frame);
    } else {
        out_service(s, SDV_PAIR_SRV_END_FILE);
        reset_state(s);
    }
    frasm_clear(f);
    s->file_start = s->file_end = false;
}

void orc_default_pcm16x0_stitch_settings(sdv_pcm16x0_stitch_settings *st)
{
    memset(st, 0, sizeof(*st));
    st->format = SDV_P16_FORMAT_SI; st->field_order = ORDER_TFF; st->p_correction = 1; st->use_ecc = 1; st->mask_seams = 1; st->broke_mask = MAX_PAD_EI;
    st->sample_rate_preset = 1;
}

long orc_pcm16x0_stitch_run(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                            sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames)
{
    return orc_pcm16x0_stitch_run_vis(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL, 0, NULL);
}
long orc_pcm16x0_stitch_run_vis(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    return orc_pcm16x0_stitch_run_feeds(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, blocks, blocks_cap, n_blocks, NULL, 0, NULL);
}
long orc_pcm16x0_stitch_run_feeds(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                  sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                  sdv_pcm16x0_bin_rec *lines, size_t lines_cap, size_t *n_lines)
{
    p16_stitcher *s = (p16_stitcher *)calloc(1, sizeof(p16_stitcher));
    s->vb = blocks; s->vb_cap = blocks_cap; s->vl = lines; s->vl_cap = lines_cap;
    s->st = *st; s->ignore_crc = !st->use_ecc;
    s->out = out; s->out_cap = out_cap; s->frames = frames; s->frames_cap = frames_cap;
    s->trim = (p16_sub *)malloc(BUF_TRIM * sizeof(p16_sub));
    s->pq.buf = (p16_sub *)malloc(DQ_CAP * sizeof(p16_sub));
    s->pad_checker.force_ecc_check = true; s->pad_checker.en_p_code = true;
    frasm_clear(&s->f1);
    reset_state(s);
    frasm_clear(&s->f1);
    size_t lo = 0;
    for (size_t i = 0; i < n_recs; i++)
        if (recs[i].service_type == SDV_SRV_END_FRAME) { stitch_frame(s, recs, lo, i, recs[i].frame_number); lo = i + 1; }
    long n = s->out_n > out_cap ? -1 : (long)s->out_n;
    if (n_frames) *n_frames = s->frames_n;
    if (n_blocks) *n_blocks = s->vb_n;
    if (n_lines) *n_lines = s->vl_n;
    free(s->trim); free(s->pq.buf); free(s->conv); free(s);
    return n;
}
