/* pcm1.h - PCM-1 back half restatement (oracle/pcm1.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_PCM1_H
#define ORC_PCM1_H
#include "sdv_oracle.h"
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif
uint16_t orc_pcm1_crc_words(const uint16_t *w6);
long orc_pcm1_stitch_run(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                         sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames);
long orc_pcm1_stitch_run_vis(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                             sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm1_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                             sdv_pcm1_asm_line_rec *lines, size_t lines_cap, size_t *n_lines);
void orc_default_pcm1_stitch_settings(sdv_pcm1_stitch_settings *st);
#ifdef __cplusplus
}
#endif
#endif
