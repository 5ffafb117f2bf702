/* pcm16.h - PCM-16x0 back half restatement (oracle/pcm16.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_PCM16_H
#define ORC_PCM16_H
#include "sdv_oracle.h"
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif
/* One PCM16X0DataBlock as PCM16X0Deinterleaver::processBlock leaves it (pcm16x0datablock.h:120-140), for block-level checks. */
typedef struct orc_p16_block_rec {
    uint32_t frame_number;
    uint16_t start_line, stop_line, queue_order;
    uint8_t start_part, stop_part;
    uint16_t words[3][3];           /* [sub-block][WORD_L, WORD_R, WORD_P] (getWord) */
    uint8_t word_crc[3][3], word_valid[3][3];     /* isWordCRCOk / isWordValid, same order */
    uint8_t picked_left[3], picked_crc[3];        /* hasPickedLeft(line) / hasPickedCRC(line) */
    uint8_t audio_state[3];
    uint8_t order_even, ret;        /* ret = DI_RET_* */
} orc_p16_block_rec;
/* processBlock(line_sh, even_order) for line_sh = first_shift .. first_shift + n_blocks - 1 over one queue of sub-lines; even_order
 * alternates from `first_even` */
void orc_pcm16x0_deint_blocks(const sdv_pcm16x0_bin_rec *lines, size_t n_lines, int ei_format, int force_check, int p_code, int ignore_crc,
                              int first_shift, int first_even, orc_p16_block_rec *out, size_t n_blocks);
void orc_default_pcm16x0_stitch_settings(sdv_pcm16x0_stitch_settings *st);
long orc_pcm16x0_stitch_run(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                            sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames);
/* ... with the visualiser's feed: the blocks outputDataBlock hands to newBlockProcessed, next to the pairs */
long orc_pcm16x0_stitch_run_vis(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks);
/* ... and with the assembled sub-lines performDeinterleave hands to newLineProcessed (:5196-5213), as records of the binarizer's type, an END_FRAME
 * record behind every frame's (sdv_set_pcm16x0_stitch_line_output in include/sdvpcm.h) */
long orc_pcm16x0_stitch_run_feeds(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                  sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                  sdv_pcm16x0_bin_rec *lines, size_t lines_cap, size_t *n_lines);
#ifdef __cplusplus
}
#endif
#endif
