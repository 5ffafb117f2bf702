/*
 * deint.c - CPU restatement of STC007DataBlock (stc007datablock.cpp:53-730) and
 * STC007Deinterleaver::processBlock with the P/Q error correction (stc007deinterleaver.cpp:286-2088).
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).
 *
 * The b-adjacent code matrices of the reference (row tables stc007deinterleaver.cpp:4-75) are not
 * copied: T is "multiply by x" in GF(2)[x]/(x^14 + x^8 + 1) (what the T^1 row table encodes), its powers
 * and inverses are applied as shifts, and (T^k + I)^-1 is obtained by Gaussian elimination at first use.
 * tests/ check the result against the reference's own vectors and against the real reference.
 */
#include "deint.h"
#include <string.h>

/* ---- GF(2)^14 helpers ---- */
static uint16_t t_mul(uint16_t v)           /* T v  (row table :8-11) */
{
    uint16_t hi = (v >> 13) & 1;
    return (uint16_t)(((v << 1) & 0x3FFF) ^ (hi ? 0x0101 : 0));
}
static uint16_t t_inv(uint16_t w)           /* T^-1 w (row table :32-35) */
{
    uint16_t b0 = w & 1;
    return (uint16_t)((((w ^ (b0 ? 0x0101 : 0)) >> 1) | (b0 << 13)) & 0x3FFF);
}
uint16_t orc_tmat_pow(uint16_t v, int k)
{
    v &= 0x3FFF;
    for (; k > 0; k--) v = t_mul(v);
    for (; k < 0; k++) v = t_inv(v);
    return v;
}
/* (T^k + I)^-1 as 14 row masks, built once by Gauss-Jordan on [T^k + I | I] */
static uint16_t g_tki_inv[6][14];
static bool g_tki_ready = false;
static void build_tki(void)
{
    for (int k = 1; k <= 5; k++) {
        uint16_t a[14], inv[14];
        /* row i of (T^k + I): bit j set when output bit i depends on input bit j */
        for (int i = 0; i < 14; i++) { a[i] = 0; inv[i] = (uint16_t)(1u << i); }
        for (int j = 0; j < 14; j++) {
            uint16_t col = (uint16_t)(orc_tmat_pow((uint16_t)(1u << j), k) ^ (1u << j));
            for (int i = 0; i < 14; i++) if (col & (1u << i)) a[i] |= (uint16_t)(1u << j);
        }
        for (int c = 0; c < 14; c++) {
            int p = -1;
            for (int r = c; r < 14; r++) if (a[r] & (1u << c)) { p = r; break; }
            if (p < 0) continue;                  /* singular: cannot happen for k = 1..5 */
            uint16_t t = a[p]; a[p] = a[c]; a[c] = t; t = inv[p]; inv[p] = inv[c]; inv[c] = t;
            for (int r = 0; r < 14; r++) if (r != c && (a[r] & (1u << c))) { a[r] ^= a[c]; inv[r] ^= inv[c]; }
        }
        for (int i = 0; i < 14; i++) g_tki_inv[k][i] = inv[i];
    }
    g_tki_ready = true;
}
uint16_t orc_tki_inv(uint16_t v, int k)
{
    if (!g_tki_ready) build_tki();
    uint16_t r = 0;
    for (int i = 0; i < 14; i++) if (__builtin_parity((unsigned)(g_tki_inv[k][i] & v))) r |= (uint16_t)(1u << i);
    return r;
}

uint16_t orc_calc_p(const uint16_t *w) { return (uint16_t)(w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4] ^ w[5]); }   /* :1297-1304 */
uint16_t orc_calc_q(const uint16_t *w)                                                                     /* :1307-1317 */
{
    /* multMatrix uses 14-bit row masks, so bits 14/15 of the (16-bit mode) words never contribute */
    return (uint16_t)(orc_tmat_pow(w[0], 6) ^ orc_tmat_pow(w[1], 5) ^ orc_tmat_pow(w[2], 4) ^ orc_tmat_pow(w[3], 3) ^ orc_tmat_pow(w[4], 2) ^ orc_tmat_pow(w[5], 1));
}

/* ---- STC007DataBlock ---- */
void orc_block_clear(orc_stc_block *b)      /* stc007datablock.cpp:53-69 */
{
    memset(b, 0, sizeof(*b));
    b->sample_rate = 44056;
    b->resolution = ORC_RES_14BIT; b->audio_state = ORC_AUD_ORIG;
}
static void blk_set_word(orc_stc_block *b, uint8_t i, uint16_t w, bool line_valid, bool cwd_fixed)   /* :81-89 */
{
    if (i < ORC_BLK_WORD_CNT) { b->words[i] = w; b->line_crc[i] = b->word_valid[i] = line_valid; b->cwd_fixed[i] = cwd_fixed; }
}
static void blk_set_fixed(orc_stc_block *b, uint8_t i) { if (i < ORC_BLK_WORD_CNT) b->word_valid[i] = true; }                          /* :102-108 */
static void blk_set_valid(orc_stc_block *b, uint8_t i) { if (i < ORC_BLK_WORD_CNT) { b->word_valid[i] = true; b->cwd_fixed[i] = false; } } /* :111-119 */
static void blk_clear_cwd(orc_stc_block *b, uint8_t i) { if (i < ORC_BLK_WORD_CNT) b->cwd_fixed[i] = false; }                           /* :231-237 */
static uint8_t ind_limit(const orc_stc_block *b) { return b->resolution == ORC_RES_16BIT ? ORC_BLK_WORD_P0 : ORC_BLK_WORD_Q0; }
void orc_block_mark_unsafe(orc_stc_block *b)   /* :168-201 */
{
    if (b->audio_state == ORC_AUD_BROKEN) return;
    for (uint8_t i = 0; i <= ind_limit(b); i++) { b->word_valid[i] = b->line_crc[i]; b->line_crc[i] = b->cwd_fixed[i] = false; }
    b->audio_state = ORC_AUD_ORIG; b->cwd_applied = false;
}
void orc_block_mark_broken(orc_stc_block *b)   /* :204-228 */
{
    for (uint8_t i = 0; i <= ind_limit(b); i++) b->word_valid[i] = b->line_crc[i] = b->cwd_fixed[i] = false;
    b->audio_state = ORC_AUD_BROKEN; b->cwd_applied = false;
}
uint8_t orc_block_errors_audio_source(const orc_stc_block *b) { uint8_t n = 0; for (int i = 0; i <= 5; i++) if (!b->line_crc[i]) n++; return n; }      /* :577-590 */
uint8_t orc_block_errors_audio_fixed(const orc_stc_block *b) { uint8_t n = 0; for (int i = 0; i <= 5; i++) if (!b->word_valid[i]) n++; return n; }     /* :609-622 */
uint8_t orc_block_errors_total_source(const orc_stc_block *b) { uint8_t n = 0; for (int i = 0; i <= ind_limit(b); i++) if (!b->line_crc[i]) n++; return n; }   /* :625-650 */
uint8_t orc_block_errors_total_cwd(const orc_stc_block *b) { uint8_t n = 0; for (int i = 0; i <= ind_limit(b); i++) if (!b->line_crc[i] && !b->cwd_fixed[i]) n++; return n; }   /* :653-678 */
bool orc_block_is_valid(const orc_stc_block *b) { return orc_block_errors_audio_fixed(b) == 0; }     /* :305-312 */
bool orc_block_can_force_check(const orc_stc_block *b)   /* :246-272 */
{
    if (b->audio_state != ORC_AUD_BROKEN) {
        if (b->resolution == ORC_RES_14BIT) { if (orc_block_errors_total_cwd(b) <= 1) return true; }
        else if (orc_block_errors_total_cwd(b) == 0) return true;
    }
    return false;
}
static bool blk_data_altered_by_cwd(const orc_stc_block *b) { for (int i = 0; i <= 7; i++) if (b->cwd_fixed[i]) return true; return false; }   /* :328-338 */
static bool blk_data_fixed_by_cwd(const orc_stc_block *b) { return b->cwd_applied ? blk_data_altered_by_cwd(b) : false; }                        /* :341-348 */
int16_t orc_block_get_sample(const orc_stc_block *b, uint8_t index)   /* :507-562 */
{
    if (index > 5) return 0;
    if (!b->m2_format) return (b->resolution == ORC_RES_16BIT) ? (int16_t)b->words[index] : (int16_t)(b->words[index] << 2);
    uint16_t w = b->words[index];
    if ((w & (1 << 13)) == 0) w = (uint16_t)(w << 3);
    else {
        bool pos = (w & (1 << 12)) == 0;
        w = (uint16_t)(w & ~(1 << 13));
        if (!pos) w |= (1 << 15) | (1 << 14) | (1 << 13);
    }
    return (int16_t)w;
}
static bool blk_near_silence(const orc_stc_block *b, uint8_t index)   /* :417-446 */
{
    int16_t s = orc_block_get_sample(b, index);
    int lim = ((b->resolution == ORC_RES_16BIT) || b->m2_format) ? (1 << 2) : (1 << 4);
    if (s >= (int16_t)lim) return false;
    if (s < (0 - (int16_t)lim)) return false;
    return true;
}
bool orc_block_is_almost_silent(const orc_stc_block *b)   /* :450-461 */
{
    return (blk_near_silence(b, 0) || blk_near_silence(b, 2) || blk_near_silence(b, 4)) && (blk_near_silence(b, 1) || blk_near_silence(b, 3) || blk_near_silence(b, 5));
}
bool orc_block_is_silent(const orc_stc_block *b) { for (uint8_t i = 0; i <= 5; i++) if (orc_block_get_sample(b, i) != 0) return false; return true; }   /* :465-478 */

/* ---- STC007Line accessors used by setWordData ---- */
static bool line_word_crc_ok(const orc_stc_line *l, int i) { return l->forced_bad ? false : l->word_crc[i]; }   /* stc007line.cpp:656-667 */
static bool line_fixed_by_cwd(const orc_stc_line *l)                                                              /* stc007line.cpp:628-641 */
{
    if (orc_stc_crc_valid(l)) for (int i = 0; i <= ORC_STC_WORD_Q; i++) if (!l->word_crc[i] && l->word_valid[i]) return true;
    return false;
}

/* ---- STC007Deinterleaver ---- */
void orc_deint_init(orc_deint *d)   /* :83-96 */
{
    d->data_res_mode = ORC_RES_MODE_14BIT_AUTO; d->ignore_crc = false; d->force_ecc_check = true;
    d->en_p_code = true; d->en_q_code = true; d->en_cwd = false;
}
void orc_deint_set_q(orc_deint *d, bool flag);
void orc_deint_set_p(orc_deint *d, bool flag) { d->en_p_code = flag; if (!flag) { orc_deint_set_q(d, false); d->en_cwd = false; } }   /* :210-234 */
void orc_deint_set_q(orc_deint *d, bool flag) { d->en_q_code = flag; if (flag) orc_deint_set_p(d, true); }                             /* :237-260 */

enum { STG_DATA_FILL = 0, STG_ERROR_CHECK, STG_TASK_SELECTION, STG_CWD_CORR, STG_P_CORR, STG_Q_CORR, STG_BAD_BLOCK, STG_NO_CHECK, STG_DATA_OK, STG_CONVERT_MAX };
enum { FIX_NOT_NEED = 0, FIX_SWITCH_P, FIX_BROKEN, FIX_NA, FIX_DONE };
enum { NO_ERR_INDEX = 64, MAX_PASSES = 3 };

static void set_word_data(const orc_deint *d, const orc_stc_line *const *ln, orc_stc_block *b, uint8_t res)   /* :1126-1294 */
{
    bool ok[8];
    for (int k = 0; k < 8; k++)
        ok[k] = !d->ignore_crc ? line_word_crc_ok(ln[k], k) : (orc_coords_valid(&ln[k]->coords) && ln[k]->blk_wht_set);
    if (res == ORC_RES_14BIT) {
        for (int k = 0; k < 8; k++) {
            blk_set_word(b, (uint8_t)k, ln[k]->words[k], ok[k], line_fixed_by_cwd(ln[k]));
            b->w_frame[k] = ln[k]->frame_number; b->w_line[k] = ln[k]->line_number;
        }
    } else {
        static const int s_ofs[7] = { 12, 10, 8, 6, 4, 2, 0 };
        for (int k = 0; k < 7; k++) {
            bool sok = !d->ignore_crc ? line_word_crc_ok(ln[k], ORC_STC_WORD_Q) : (orc_coords_valid(&ln[k]->coords) && ln[k]->blk_wht_set);
            uint16_t f1 = (uint16_t)(ln[k]->words[k] << 2);
            uint16_t s = (uint16_t)((ln[k]->words[ORC_STC_WORD_Q] >> s_ofs[k]) & 0x0003);
            blk_set_word(b, (uint8_t)k, (uint16_t)(f1 + s), ok[k] & sok, line_fixed_by_cwd(ln[k]));
            b->w_frame[k] = ln[k]->frame_number; b->w_line[k] = ln[k]->line_number;
        }
        blk_set_word(b, ORC_BLK_WORD_Q0, 0, true, false);
        b->w_frame[7] = ln[7]->frame_number; b->w_line[7] = ln[7]->line_number;
    }
    if (res < 3) b->resolution = res;
}

static void recalc_p(orc_stc_block *b)   /* :1336-1373 */
{
    uint16_t old_p = b->words[ORC_BLK_WORD_P0], p = orc_calc_p(b->words);
    if (old_p != p) { blk_set_word(b, ORC_BLK_WORD_P0, p, b->line_crc[ORC_BLK_WORD_P0], false); blk_set_fixed(b, ORC_BLK_WORD_P0); }
    else blk_set_valid(b, ORC_BLK_WORD_P0);
}

static uint8_t fix_by_p(orc_stc_block *b, uint8_t first_bad)   /* :1376-1464 */
{
    b->audio_state = ORC_AUD_ORIG;
    uint16_t check = (uint16_t)(orc_calc_p(b->words) ^ b->words[ORC_BLK_WORD_P0]);
    if (check == 0) { if (first_bad != NO_ERR_INDEX) blk_set_valid(b, first_bad); return FIX_NOT_NEED; }
    if (first_bad == NO_ERR_INDEX) return FIX_BROKEN;
    uint16_t fix_word = (uint16_t)(check ^ b->words[first_bad]);
    blk_set_word(b, first_bad, fix_word, false, b->word_valid[first_bad]);
    blk_set_fixed(b, first_bad);
    return FIX_DONE;
}

static uint8_t fix_by_q(orc_stc_block *b, uint8_t first_bad, uint8_t second_bad)   /* :1468-2048 */
{
    uint16_t synd_p = 0, synd_q, error_first = 0, error_second = 0;
    bool fix_found = false;
    b->audio_state = ORC_AUD_ORIG;
    if (second_bad == NO_ERR_INDEX) if (!b->word_valid[ORC_BLK_WORD_P0]) second_bad = ORC_BLK_WORD_P0;
    synd_q = (uint16_t)(orc_calc_q(b->words) ^ b->words[ORC_BLK_WORD_Q0]);
    if (second_bad == ORC_BLK_WORD_P0) {
        if (synd_q == 0) { if (first_bad != NO_ERR_INDEX) blk_set_valid(b, first_bad); recalc_p(b); return FIX_NOT_NEED; }
    } else {
        synd_p = (uint16_t)(orc_calc_p(b->words) ^ b->words[ORC_BLK_WORD_P0]);
        if (synd_p == 0 && synd_q == 0) { blk_set_valid(b, first_bad); blk_set_valid(b, second_bad); return FIX_NOT_NEED; }
    }
    if (second_bad != ORC_BLK_WORD_P0 && !b->word_valid[ORC_BLK_WORD_P0]) return FIX_NA;
    if (first_bad == NO_ERR_INDEX) return FIX_BROKEN;
    if (second_bad == NO_ERR_INDEX) return FIX_SWITCH_P;
    /* the 21 (first, second) branches of :1627-1977 in closed form:
     *   second == P : e1 = T^-(6-first) Sq
     *   else        : e1 = (T^(second-first) + I)^-1 (T^-(6-second) Sq + Sp),  e2 = e1 + Sp          */
    if (first_bad <= 5 && second_bad == ORC_BLK_WORD_P0) { error_first = orc_tmat_pow(synd_q, -(6 - (int)first_bad)); fix_found = true; }
    else if (first_bad <= 4 && second_bad <= 5 && second_bad > first_bad) {
        error_first = orc_tmat_pow(synd_q, -(6 - (int)second_bad));
        error_first ^= synd_p;
        error_first = orc_tki_inv(error_first, (int)second_bad - (int)first_bad);
        error_second = (uint16_t)(error_first ^ synd_p);
        fix_found = true;
    }
    if (!fix_found) return FIX_BROKEN;
    uint16_t old1 = b->words[first_bad], fix1 = (uint16_t)(old1 ^ error_first);
    if (error_first != 0) { blk_set_word(b, first_bad, fix1, false, b->cwd_fixed[first_bad]); blk_set_fixed(b, first_bad); }
    else blk_set_valid(b, first_bad);
    uint16_t old2 = b->words[second_bad];
    if (second_bad == ORC_BLK_WORD_P0) error_second = (uint16_t)(old2 ^ orc_calc_p(b->words));
    uint16_t fix2 = (uint16_t)(old2 ^ error_second);
    if (error_second != 0) { blk_set_word(b, second_bad, fix2, false, b->cwd_fixed[second_bad]); blk_set_fixed(b, second_bad); }
    else blk_set_valid(b, second_bad);
    return (error_first == 0 && error_second == 0) ? FIX_NOT_NEED : FIX_DONE;
}

uint8_t orc_deint_process_block(const orc_deint *d, const orc_stc_line *lines, size_t n_lines, uint16_t line_shift, orc_stc_block *out)   /* :286-1123 */
{
    uint8_t run_audio_res, stage_count, fill_passes, all_crc_errs, aud_crc_errs, first_bad, second_bad, fix_result, proc_state;
    if (out == NULL) return ORC_DI_RET_NULL_BLOCK;
    if (lines == NULL) return ORC_DI_RET_NULL_LINES;
    if (n_lines <= (size_t)(ORC_MIN_DEINT_DATA + line_shift)) return ORC_DI_RET_NO_DATA;
    if (d->data_res_mode == ORC_RES_MODE_14BIT) { run_audio_res = ORC_RES_14BIT; fill_passes = MAX_PASSES; }
    else if (d->data_res_mode == ORC_RES_MODE_14BIT_AUTO) { run_audio_res = ORC_RES_14BIT; fill_passes = 0; }
    else if (d->data_res_mode == ORC_RES_MODE_16BIT_AUTO) { run_audio_res = ORC_RES_16BIT; fill_passes = 0; }
    else { run_audio_res = ORC_RES_16BIT; fill_passes = MAX_PASSES; }
    first_bad = second_bad = NO_ERR_INDEX;
    all_crc_errs = aud_crc_errs = 0;
    proc_state = STG_DATA_FILL;
    stage_count = 0;
    do {
        stage_count++;
        if (proc_state == STG_DATA_FILL) {
            const orc_stc_line *ln[8];
            for (int k = 0; k < 8; k++) ln[k] = &lines[ORC_INTERLEAVE_OFS * k + line_shift];
            orc_block_clear(out);
            set_word_data(d, ln, out, run_audio_res);
            out->audio_state = ORC_AUD_ORIG;
            fill_passes++;
            proc_state = STG_ERROR_CHECK;
        } else if (proc_state == STG_ERROR_CHECK) {
            first_bad = second_bad = NO_ERR_INDEX;
            for (uint8_t i = 0; i <= 5; i++)
                if (!out->line_crc[i]) {
                    if (first_bad == NO_ERR_INDEX) first_bad = i;
                    else if (second_bad == NO_ERR_INDEX) { second_bad = i; break; }
                }
            aud_crc_errs = orc_block_errors_audio_source(out);
            all_crc_errs = orc_block_errors_total_source(out);
            proc_state = STG_TASK_SELECTION;
        } else if (proc_state == STG_TASK_SELECTION) {
            proc_state = STG_BAD_BLOCK;
            if (all_crc_errs <= 2) {
                if (aud_crc_errs == 0) {
                    if (!d->force_ecc_check) proc_state = STG_DATA_OK;
                    else if (d->en_p_code) proc_state = STG_P_CORR;
                    else proc_state = STG_NO_CHECK;
                } else if (aud_crc_errs == 1) {
                    if (d->en_p_code) proc_state = STG_P_CORR;
                } else if (aud_crc_errs == 2) {
                    if (run_audio_res == ORC_RES_14BIT) { if (d->en_q_code) proc_state = STG_Q_CORR; }
                    else if (d->en_cwd && !blk_data_fixed_by_cwd(out)) proc_state = STG_CWD_CORR;
                }
            } else if (d->en_cwd && !out->cwd_applied) proc_state = STG_CWD_CORR;
        } else if (proc_state == STG_CWD_CORR) {
            uint8_t fix_count = 0;
            proc_state = STG_BAD_BLOCK;
            for (uint8_t i = 0; i <= 7; i++) if (out->cwd_fixed[i]) { blk_set_fixed(out, i); out->cwd_applied = true; fix_count++; }
            if (fix_count != 0) {
                first_bad = second_bad = NO_ERR_INDEX;
                all_crc_errs = aud_crc_errs = 0;
                for (uint8_t i = 0; i <= 7; i++) {
                    if (!out->word_valid[i]) all_crc_errs++;
                    if (i <= 5 && !out->word_valid[i]) {
                        aud_crc_errs++;
                        if (first_bad == NO_ERR_INDEX) first_bad = i;
                        else if (second_bad == NO_ERR_INDEX) second_bad = i;
                    }
                }
                proc_state = STG_TASK_SELECTION;
            }
        } else if (proc_state == STG_P_CORR) {
            proc_state = STG_BAD_BLOCK;
            if (out->word_valid[ORC_BLK_WORD_P0]) {
                fix_result = fix_by_p(out, first_bad);
                if (fix_result == FIX_BROKEN) orc_block_mark_broken(out);
                else {
                    proc_state = STG_DATA_OK;
                    blk_clear_cwd(out, first_bad);
                    if (fix_result == FIX_DONE) out->audio_state = ORC_AUD_FIX_P;
                    else if (fix_result == FIX_NOT_NEED) { if (first_bad < ORC_BLK_WORD_P0) out->audio_state = ORC_AUD_FIX_P; }
                    if (run_audio_res == ORC_RES_14BIT && d->en_q_code) {
                        if (out->word_valid[ORC_BLK_WORD_Q0]) {
                            if (d->force_ecc_check) {
                                uint16_t synd_q = (uint16_t)(orc_calc_q(out->words) ^ out->words[ORC_BLK_WORD_Q0]);
                                if (synd_q != 0) { proc_state = STG_BAD_BLOCK; orc_block_mark_broken(out); }
                            }
                        } else {
                            uint16_t q = orc_calc_q(out->words);
                            if (out->words[ORC_BLK_WORD_Q0] != q) { blk_set_word(out, ORC_BLK_WORD_Q0, q, out->line_crc[ORC_BLK_WORD_Q0], false); blk_set_fixed(out, ORC_BLK_WORD_Q0); }
                            else blk_set_valid(out, ORC_BLK_WORD_Q0);
                        }
                    }
                }
            } else {
                if (run_audio_res == ORC_RES_14BIT) {
                    if (d->en_q_code) proc_state = STG_Q_CORR;
                    else if (aud_crc_errs == 0) proc_state = STG_NO_CHECK;
                } else if (aud_crc_errs == 0) proc_state = STG_NO_CHECK;
            }
        } else if (proc_state == STG_Q_CORR) {
            proc_state = STG_BAD_BLOCK;
            if (out->word_valid[ORC_BLK_WORD_Q0]) {
                fix_result = fix_by_q(out, first_bad, second_bad);
                if (!out->line_crc[ORC_BLK_WORD_P0]) second_bad = ORC_BLK_WORD_P0;
                if (fix_result == FIX_DONE) { proc_state = STG_DATA_OK; blk_clear_cwd(out, first_bad); blk_clear_cwd(out, second_bad); out->audio_state = ORC_AUD_FIX_Q; }
                else if (fix_result == FIX_NOT_NEED) {
                    proc_state = STG_DATA_OK; blk_clear_cwd(out, first_bad); blk_clear_cwd(out, second_bad);
                    if (first_bad < ORC_BLK_WORD_P0) out->audio_state = ORC_AUD_FIX_Q;
                } else if (fix_result == FIX_SWITCH_P) proc_state = STG_P_CORR;
                else if (fix_result == FIX_BROKEN) orc_block_mark_broken(out);
            } else if (first_bad == NO_ERR_INDEX) {
                proc_state = STG_NO_CHECK;
                uint16_t ecc = orc_calc_p(out->words);
                blk_set_word(out, ORC_BLK_WORD_P0, ecc, false, out->cwd_fixed[ORC_BLK_WORD_P0]); blk_set_fixed(out, ORC_BLK_WORD_P0);
                ecc = orc_calc_q(out->words);
                blk_set_word(out, ORC_BLK_WORD_Q0, ecc, false, out->cwd_fixed[ORC_BLK_WORD_Q0]); blk_set_fixed(out, ORC_BLK_WORD_Q0);
            }
        } else if (proc_state == STG_BAD_BLOCK) {
            out->cwd_applied = false;
            if (fill_passes >= MAX_PASSES) break;
            run_audio_res = (run_audio_res == ORC_RES_16BIT) ? ORC_RES_14BIT : ORC_RES_16BIT;
            proc_state = STG_DATA_FILL;
        } else break;     /* STG_NO_CHECK, STG_DATA_OK */
        if (stage_count > (STG_CONVERT_MAX * MAX_PASSES)) break;
    } while (1);
    return ORC_DI_RET_OK;
}
