/* deint.h - STC007DataBlock + STC007Deinterleaver restatement (oracle/deint.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_DEINT_H
#define ORC_DEINT_H
#include "sdv_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_BLK_WORD_P0 = 6, ORC_BLK_WORD_Q0 = 7, ORC_BLK_WORD_CNT = 8, ORC_INTERLEAVE_OFS = 16, ORC_MIN_DEINT_DATA = 112 };
enum { ORC_RES_14BIT = 0, ORC_RES_16BIT, ORC_RES_AUTO };
enum { ORC_AUD_ORIG = 0, ORC_AUD_FIX_P, ORC_AUD_FIX_Q, ORC_AUD_BROKEN };
enum { ORC_DI_RET_NULL_LINES = 0, ORC_DI_RET_NULL_BLOCK, ORC_DI_RET_NO_DATA, ORC_DI_RET_OK };
enum { ORC_RES_MODE_14BIT = 0, ORC_RES_MODE_14BIT_AUTO, ORC_RES_MODE_16BIT_AUTO, ORC_RES_MODE_16BIT };

/* STC007DataBlock (stc007datablock.h:144-160) */
typedef struct {
    uint32_t w_frame[ORC_BLK_WORD_CNT];
    uint16_t w_line[ORC_BLK_WORD_CNT];
    uint16_t sample_rate;
    bool emphasis, cwd_applied;
    uint16_t words[ORC_BLK_WORD_CNT];
    bool line_crc[ORC_BLK_WORD_CNT], cwd_fixed[ORC_BLK_WORD_CNT], word_valid[ORC_BLK_WORD_CNT];
    bool m2_format;
    uint8_t resolution, audio_state;
} orc_stc_block;

/* STC007Deinterleaver settings (stc007deinterleaver.h:151-161) */
typedef struct {
    uint8_t data_res_mode;
    bool ignore_crc, force_ecc_check, en_p_code, en_q_code, en_cwd;
} orc_deint;

void orc_deint_init(orc_deint *d);                          /* STC007Deinterleaver::clear */
void orc_deint_set_p(orc_deint *d, bool flag);              /* setPCorrection */
void orc_deint_set_q(orc_deint *d, bool flag);              /* setQCorrection */
uint8_t orc_deint_process_block(const orc_deint *d, const orc_stc_line *lines, size_t n_lines, uint16_t line_shift, orc_stc_block *out);

void orc_block_clear(orc_stc_block *b);
int16_t orc_block_get_sample(const orc_stc_block *b, uint8_t index);
bool orc_block_is_valid(const orc_stc_block *b);
void orc_block_mark_unsafe(orc_stc_block *b);
void orc_block_mark_broken(orc_stc_block *b);
uint8_t orc_block_errors_audio_source(const orc_stc_block *b);
uint8_t orc_block_errors_audio_fixed(const orc_stc_block *b);
uint8_t orc_block_errors_total_source(const orc_stc_block *b);
uint8_t orc_block_errors_total_cwd(const orc_stc_block *b);
bool orc_block_is_silent(const orc_stc_block *b);
bool orc_block_is_almost_silent(const orc_stc_block *b);
bool orc_block_can_force_check(const orc_stc_block *b);
uint16_t orc_calc_p(const uint16_t *w);
uint16_t orc_calc_q(const uint16_t *w);
uint16_t orc_tmat_pow(uint16_t v, int k);                   /* T^k v, k in [-6, 6] */
uint16_t orc_tki_inv(uint16_t v, int k);                    /* (T^k + I)^-1 v, k in [1, 5] */
#ifdef __cplusplus
}
#endif
#endif
