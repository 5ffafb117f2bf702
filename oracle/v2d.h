/* v2d.h - types of the VideoToDigital restatement (oracle/v2d.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_V2D_H
#define ORC_V2D_H
#include "sdv_oracle.h"
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { orc_coords *v; int n, cap; } orc_coord_list;

/* FrameBinDescriptor (frametrimset.h:69-97) */
typedef struct {
    uint32_t frame_id; uint16_t line_length;
    uint16_t lines_odd, lines_even, lines_pcm_odd, lines_pcm_even, lines_bad_odd, lines_bad_even, lines_dup_odd, lines_dup_even;
    orc_coords data_coord;
} orc_frame_stats;

/* VideoToDigital members (videotodigital.h:112-135) + the locals of doBinarize that live across
 * frames (videotodigital.cpp:700-720) */
typedef struct {
    orc_bin_preset fine_bin_preset;
    orc_binarizer line_converter;
    uint8_t binarization_mode;
    bool check_line_copy, coordinate_damper, reset_stats, m2_format;
    orc_frame_stats signal_quality;
    uint8_t field_state;
    uint16_t line_in_field_cnt, good_coords_in_field, pcm_lines_in_field;
    orc_coords frame_avg, target_coord;
    orc_coord_list last_valid_coord_list, frame_valid_coord_list, frame_invalid_coord_list, long_valid_coords;
    orc_stc_line stc007_line, last_stc007_line;
} orc_v2d;

void orc_v2d_init(orc_v2d *v);
void orc_v2d_free(orc_v2d *v);
void orc_v2d_set_fine_settings(orc_v2d *v, const orc_bin_preset *p);
void orc_v2d_begin_frame(orc_v2d *v);
bool orc_v2d_line(orc_v2d *v, const orc_video_line *src, sdv_line_rec *out_rec, orc_frame_stats *out_stats);
int orc_v2d_end_file_frame(orc_v2d *v, int height, uint32_t frame_no, sdv_line_rec *out, orc_frame_stats *out_stats);
int orc_v2d_frame(orc_v2d *v, const uint8_t *luma, size_t stride, int width, int height, uint32_t frame_no,
                  bool new_file, bool doubled, sdv_line_rec *out, orc_frame_stats *out_stats);
#ifdef __cplusplus
}
#endif
#endif
