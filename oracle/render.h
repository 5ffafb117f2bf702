/* render.h - RenderPCM's canvases of binarized lines, restated (render.c).  TEST INFRASTRUCTURE ONLY. */
#ifndef SDV_ORACLE_RENDER_H
#define SDV_ORACLE_RENDER_H
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif
enum { ORC_VIS_STC007 = 0, ORC_VIS_PCM1 = 1, ORC_VIS_PCM16X0 = 2, ORC_VIS_STC007_BLOCKS_NTSC = 3, ORC_VIS_STC007_BLOCKS_PAL = 4, ORC_VIS_STC007_ASM_NTSC = 5, ORC_VIS_STC007_ASM_PAL = 6,
       ORC_VIS_PCM1_BLOCKS = 7, ORC_VIS_PCM1_ASM = 8, ORC_VIS_PCM16X0_BLOCKS = 9 };
void orc_vis_canvas_size(int kind, uint32_t *w, uint32_t *h);
/* The records of whole frames (sdv_line_rec / sdv_pcm1_bin_rec / sdv_pcm16x0_bin_rec by kind; ORC_VIS_PCM1_ASM: sdv_pcm1_asm_line_rec, 1470 to a frame) drawn on `canvas` (h x w pixels, kept between
 * calls like RenderPCM::img_data); the canvas as it is at every END_FRAME record goes to out[frame] (out_cap canvases).  Returns the frames. */
long orc_vis_render_lines(int kind, const void *recs, size_t n_recs, uint32_t *canvas, uint32_t *out, size_t out_cap);
/* The data blocks window: frame f's frame_blocks[f] blocks (consecutive in `blocks`: sdv_block_rec, sdv_pcm1_block_rec or sdv_pcm16x0_block_rec by kind)
 * drawn from row 0 of the kept canvas - one row per block, 23 rows per PCM-1 block. */
long orc_vis_render_blocks(int kind, const void *blocks, size_t n_blocks, const uint32_t *frame_blocks, size_t n_frames, uint32_t *canvas,
                           uint32_t *out, size_t out_cap);
/* The assembled-lines window: frame f's frame_lines[f] lines (consecutive in `lines`) drawn from row 0 of the kept canvas, one row per line. */
long orc_vis_render_asm_lines(int kind, const sdv_asm_line_rec *lines, size_t n_lines, const uint32_t *frame_lines, size_t n_frames, uint32_t *canvas,
                              uint32_t *out, size_t out_cap);
#ifdef __cplusplus
}
#endif
#endif
